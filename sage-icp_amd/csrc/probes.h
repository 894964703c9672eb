// Instrumentation of the search body (kernels.hip, icp_body) for the probe builds under profiles/ — none of it is in the
// product: every macro below is empty unless its switch is defined when the library is built
// (python sage-icp_amd/build.py <out.so> -DSAGE_NN_TIMING ...; tests/test_probe_variants.py compiles each so they do not rot).
//
//   SAGE_NN_TIMING        k_icp: shader cycles per phase of a wave, its lifetime, the span of one chosen launch per wave
//                         (profiles/icp_tail.py, phase_probe.py); sageicp_debug_nn_phases / _nn_raw / _nn_spans
//   SAGE_LOOP_TIMING      k_loop: cycles per phase of a pass, per-workgroup / per-wave / solver stamps (profiles/loop_tail.py,
//                         loop_times.py, solve_split.py) — the stamps of the loop itself stay in kernels.hip beside the loop
//   SAGE_ICP_DELAY_PROBE  k_icp: the pose arrives `dbg_delay` ticks after a wave's start / the pass repeated inside the launch
//
// The macros name the locals of icp_body they read (that is what keeps the body one line per probe).
#pragma once

namespace sageicp {

// ---------------------------------------------------------------------------------------------- SAGE_NN_TIMING
#ifdef SAGE_NN_TIMING
constexpr unsigned kNnTimingSlots = 1u << 17;
__device__ unsigned long long g_nn_phase[8ull * kNnTimingSlots];   // per wave: 5 phases, lifetime, realtime, count
__device__ unsigned long long g_nn_span[4ull * kNnTimingSlots];    // per wave, iteration g_nn_span_iter: start, end (100-MHz ticks), HW_ID, pairs
__device__ int g_nn_span_iter;
#define NN_T(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); tph[i] += _t - tprev; tprev = _t; } while (0)
#define PROBE_NN_BEGIN                                                       \
    unsigned long long tph[5] = {0, 0, 0, 0, 0};                              \
    unsigned long long tprev = __builtin_amdgcn_s_memtime();                  \
    const unsigned long long tstart = tprev;                                  \
    const unsigned long long rstart = __builtin_amdgcn_s_memrealtime()
#define PROBE_NN_COUNTERS unsigned n_consume = 0u, n_exact = 0u, n_exact_lanes = 0u
#define PROBE_NN_CONSUME ++n_consume
#define PROBE_NN_EXACT(pa, pb) do { ++n_exact; n_exact_lanes += static_cast<unsigned>(__popcll(__ballot(pa)) + __popcll(__ballot(pb))); } while (0)
#define PROBE_NN_WORK(P, q, first_lane, npairs) do { if ((first_lane) && (P).work) (P).work[q] = (npairs); } while (0)
// private slot per wave (no contended atomics: they would stall the very loads being timed)
#define PROBE_NN_END(P, valid, ci, npairs, lane, wave_id)                                                                  \
    do {                                                                                                                   \
        NN_T(4);                                                                                                           \
        unsigned long long np_packed, xp_packed;                                                                           \
        {   /* points handed to the queries of this wave: max over the queries | sum */                                    \
            unsigned mx = (valid) ? (npairs) : 0u, sm = ((valid) && (ci) == 0u) ? (npairs) : 0u;                           \
            for (int d = 1; d < 64; d <<= 1) {                                                                             \
                mx = max(mx, static_cast<unsigned>(__shfl_xor(mx, d, 64)));                                                \
                sm += __shfl_xor(sm, d, 64);                                                                               \
            }                                                                                                              \
            np_packed = (static_cast<unsigned long long>(mx) << 32) | sm;                                                  \
            /* pair steps of this wave | of them with a fetch of full records | lanes that fetched */                      \
            xp_packed = (static_cast<unsigned long long>(n_consume) << 40) | (static_cast<unsigned long long>(n_exact) << 20) | n_exact_lanes; \
        }                                                                                                                  \
        if ((lane) == 0 && (wave_id) < kNnTimingSlots && (wave_id) < (P).nwaves) {                                         \
            unsigned long long *tt = g_nn_phase + 8ull * (wave_id);                                                        \
            for (int k = 0; k < 5; ++k) tt[k] += tph[k];                                                                   \
            tt[5] += __builtin_amdgcn_s_memtime() - tstart;                                                                \
            tt[6] += __builtin_amdgcn_s_memrealtime() - rstart;                                                            \
            tt[7] += 1ull;                                                                                                 \
            unsigned long long *sp = g_nn_span + 4ull * (wave_id);                                                         \
            if ((P).st->iter == g_nn_span_iter) {                                                                          \
                sp[0] = rstart;                                                                                            \
                sp[1] = __builtin_amdgcn_s_memrealtime();                                                                  \
                sp[2] = xp_packed;                                                                                         \
                sp[3] = np_packed;                                                                                         \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
#else
#define NN_T(i) do { } while (0)
#define PROBE_NN_BEGIN do { } while (0)
#define PROBE_NN_COUNTERS do { } while (0)
#define PROBE_NN_CONSUME do { } while (0)
#define PROBE_NN_EXACT(pa, pb) do { } while (0)
#define PROBE_NN_WORK(P, q, first_lane, npairs) do { } while (0)
#define PROBE_NN_END(P, valid, ci, npairs, lane, wave_id) do { } while (0)
#endif

// -------------------------------------------------------------------------------------------- SAGE_LOOP_TIMING
#ifdef SAGE_LOOP_TIMING
// (icp_body<PERSIST>: cycles per phase into the pass's LoopGroup, summed by k_loop)
#define LP_T(i) do { if constexpr (PERSIST) { const unsigned long long _t = __builtin_amdgcn_s_memtime(); G->ph[i] += _t - G->tprev; G->tprev = _t; } } while (0)
// per workgroup and iteration: max points of a query | stale queries | points (LDS words kLpDbg ..)
#define PROBE_LOOP_WAVE_STATS(smem, valid, ci, npairs, stale, lane)                                                        \
    do {                                                                                                                   \
        unsigned mx = (valid) ? (npairs) : 0u, sm = ((valid) && (ci) == 0u) ? (npairs) : 0u, stl = ((stale) && (ci) == 0u) ? 1u : 0u; \
        for (int d = 1; d < 64; d <<= 1) {                                                                                 \
            mx = max(mx, static_cast<unsigned>(__shfl_xor(mx, d, 64)));                                                    \
            sm += __shfl_xor(sm, d, 64);                                                                                   \
            stl += __shfl_xor(stl, d, 64);                                                                                 \
        }                                                                                                                  \
        if ((lane) == 0) {                                                                                                 \
            atomicMax(&(smem)[kLpDbg], mx);                                                                                \
            atomicAdd(&(smem)[kLpDbg + 1], stl);                                                                           \
            atomicAdd(&(smem)[kLpDbg + 2], sm);                                                                            \
        }                                                                                                                  \
    } while (0)
#else
#define LP_T(i) do { } while (0)
#define PROBE_LOOP_WAVE_STATS(smem, valid, ci, npairs, stale, lane) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------- SAGE_ICP_DELAY_PROBE
#ifdef SAGE_ICP_DELAY_PROBE
#define PROBE_DELAY_BEGIN const unsigned long long probe_t0 = __builtin_amdgcn_s_memrealtime()
// the pose becomes available `dbg_delay` ticks (100 MHz) after this wave started, with the prologue loads already in
// flight — what hiding k_fin under the prologue would cost
#define PROBE_DELAY_WAIT(P)                                                                                                \
    do {                                                                                                                   \
        if ((P).dbg_delay) {                                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            while (__builtin_amdgcn_s_memrealtime() - probe_t0 < (P).dbg_delay) __builtin_amdgcn_s_sleep(8);               \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
        }                                                                                                                  \
    } while (0)
// the same pass again inside the launch — what an iteration costs on L2s that were not emptied by a kernel boundary (its
// sums are added a second time: the solve does not care)
#define PROBE_DELAY_REPEAT(pass)                                                                                           \
    do {                                                                                                                   \
        for (unsigned r = 0; r < P.dbg_repeat; ++r) {                                                                      \
            __syncthreads();                                                                                               \
            pass;                                                                                                          \
        }                                                                                                                  \
    } while (0)
#else
#define PROBE_DELAY_BEGIN do { } while (0)
#define PROBE_DELAY_WAIT(P) do { } while (0)
#define PROBE_DELAY_REPEAT(pass) do { } while (0)
#endif

}  // namespace sageicp
