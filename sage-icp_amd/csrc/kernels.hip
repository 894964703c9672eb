// Hand-written HIP kernels for gfx950 (CDNA4, wave64) — SAGE-ICP registration hot path.
// One ICP iteration = k_nn -> k_gn, both on one stream, no host round trip.
//
//   k_nn     one wavefront per chunk of 4 consecutive queries of the spatially sorted frame.
//            Prologue (three lanes per query): apply the cumulative pose to the pristine frame
//            (TransformPoints, reference core/Registration.cpp:103-111,133 — `source` is never
//            rewritten in place), home voxel by the reference's exact fp64 divide + truncation
//            (core/VoxelHashMap.cpp:52-54), and cut the chunk into GROUPS: runs of queries that
//            share a home voxel, which see the same 27-voxel candidate list.  A group's
//            probe-table row (27 x map_.find, core/VoxelHashMap.cpp:66-78) is cached across
//            iterations and re-probed by 27 lanes only when the home voxel changed.
//            Per group the occupied voxels' points are enumerated once in reference order (x
//            outer, y, z inner, then insertion order) into an LDS list of byte offsets (start-mark
//            bitmap + v_mbcnt, no search, no scalar loop), and the (query x candidate) pairs are
//            spread over the 64 lanes — W = 64 / pow2(group size) lanes per query, each lane
//            striding the list through a raw buffer resource — followed by a W-lane two-phase
//            argmin (v_min_f64 over DPP, then the smallest enumeration index among the lanes
//            that hold the minimum).  Replaces VoxelHashMap::GetCorrespondences' per-point
//            lambda (core/VoxelHashMap.cpp:51-96).
//   k_gn     acceptance test (core/VoxelHashMap.cpp:109-115) + robust-weighted point-to-point
//            Gauss-Newton accumulation (Registration.cpp:62-90) as 16 closed-form fp64 sums +
//            count, LDS-transposed block reduction -> one partial per workgroup; the
//            last-arriving workgroup then finishes the iteration: fixed-order reduction of the
//            partials, 6x6 LDL^T solve and SE3 exp with their divisions / sincos spread over
//            lanes, pose composition and the convergence test (Registration.cpp:92-93,135-137).
//   k_fin    the same finish as its own launch (multi-GPU: after the RCCL all-reduce).
//   k_tf     TransformPoints for the stand-alone API entry (Registration.cpp:103-111).
//   k_scatter_points / k_scatter_slots   refresh of the HBM mirror of the host map.
//
// Roofline: HBM / cache-gather bound integer/byte + fp64 compare work (~0.1 flop/B) — no MFMA (a
// 6x6 outer product sum is not a dense contraction).  What matters here is coalescing (a voxel
// block is one contiguous run of 32-B records; identical addresses across the lanes of a group
// collapse into one request), LDS staging of the candidate enumeration, wave-uniform
// branch-free inner loops with as few VALU instructions per pair as the exact fp64 semantics
// allow (19), no device-scope atomics on hot words, and load balance by hardware dispatch of
// many small workgroups.
//
// Built with -ffp-contract=off: distances are the plain IEEE sequence
// dx*dx + (dy*dy + dz*dz) the CPU evaluates, so the argmin is index-exact against the oracle.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <vector>

#ifndef SAGE_NN_WAVES
#define SAGE_NN_WAVES 1
#endif
// k_nn tuning (measured on MI355X, c2): candidate loads in flight per lane when a query has 32-64
// lanes / 2-16 lanes, and the occupancy the register allocation aims for
#ifndef SAGE_NN_U_BIG
#define SAGE_NN_U_BIG 2
#endif
#ifndef SAGE_NN_U_SMALL
#define SAGE_NN_U_SMALL 4
#endif
#ifndef SAGE_NN_OCC
#define SAGE_NN_OCC 8
#endif
#ifndef SAGE_NN_STRIPE
#define SAGE_NN_STRIPE 64      // chunks of the sorted frame per XCD stripe
#endif

#include "kernels.h"
#include "se3_math.h"
#include "sageicp_types.h"

namespace sageicp {

__device__ __forceinline__ uint32_t rl_u32(uint32_t v, int lane) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), lane));
}

// -------------------------------------------------------------------------------- probe_row
// Probe-table row of one home voxel (hx, hy, hz), built by one wave:
//   row[v] = {exclusive candidate offset, index of the voxel block's first point}
// for neighbour voxel v = 0..26 (x outer, y, z inner — the reference's enumeration order,
// core/VoxelHashMap.cpp:66-78: 27 x map_.find).  27 lanes probe the GPU-resident open-addressed
// hash (linear probing, 16-B slots, load factor <= 0.25, one 16-B load per step), a 32-lane
// prefix sum turns the counts into offsets; entry 27 carries the total and 28..31 a sentinel, so
// any flat candidate index is located by a fixed 5-step binary search over 32 entries.  The row
// is also stored in `blks` (with the voxel it describes in `tabkey`) for the next iterations.
__device__ __forceinline__ uint2 probe_row(const NnParams &P, int lane, const uint32_t *lds,
                                           unsigned skey_word, int h, unsigned slot) {
    // rare path (stale rows only): keep its lane-derived constants and LDS addresses from being
    // hoisted into the registers of the caller's hot loop
    asm volatile("" : "+v"(lane));
    skey_word = __builtin_amdgcn_readfirstlane(skey_word);
    asm volatile("" : "+s"(skey_word));
    const int *skey = reinterpret_cast<const int *>(lds + skey_word);
    const unsigned v = static_cast<unsigned>(lane);
    // home voxel of query h of the chunk: skey[comp * chunk + h], one component per lane 0..2
    const unsigned kv = static_cast<unsigned>(skey[min(v, 2u) * P.chunk + static_cast<unsigned>(h)]);
    const int hx = static_cast<int>(rl_u32(kv, 0)), hy = static_cast<int>(rl_u32(kv, 1)),
              hz = static_cast<int>(rl_u32(kv, 2));
    uint32_t blk = kEmptySlot;
    if (v < 27u) {
        const int vx = hx + static_cast<int>(v / 9u) - 1;
        const int vy = hy + static_cast<int>((v / 3u) % 3u) - 1;
        const int vz = hz + static_cast<int>(v % 3u) - 1;
        uint32_t sl = voxel_hash(vx, vy, vz) & P.mask;
        for (;;) {
            int4 e = reinterpret_cast<const int4 *>(P.table)[sl];
            asm volatile("" : "+v"(e.x), "+v"(e.y), "+v"(e.z), "+v"(e.w));   // one 16-B load
            if (static_cast<uint32_t>(e.w) == kEmptySlot) break;
            if (e.x == vx && e.y == vy && e.z == vz) { blk = static_cast<uint32_t>(e.w); break; }
            sl = (sl + 1) & P.mask;
        }
    }
    const uint32_t cnt = (blk == kEmptySlot) ? 0u : (blk & 255u);
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 32);
        if ((v & 31u) >= static_cast<unsigned>(d)) incl += t;
    }
    uint2 row;
    row.x = (v > 27u) ? 0xFFFFFFFFu : incl - cnt;
    row.y = (blk == kEmptySlot) ? 0u : (blk >> 8) * static_cast<uint32_t>(P.cap);   // first point
    if (v < 32u) P.blks[slot * 32u + v] = row;
    if (v == 0u) {
        int4 k;
        k.x = 0; k.y = hx; k.z = hy; k.w = hz;
        P.tabkey[slot] = k;                   // the row now describes this home voxel
    }
    return row;
}

// ------------------------------------------------------------------------------------- k_nn
// Cross-lane helpers.  DPP moves run on the VALU (no LDS traffic, no scalar instructions); the
// patterns used are involutions (quad swaps, half-row and row mirrors), so each step is an
// exchange and every lane of a segment ends with the segment's result.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return static_cast<unsigned>(
        __builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // lane i <-> 7 - i   within each 8
constexpr int kDppMirror = 0x140;       // lane i <-> 15 - i  within each 16

// v_min_f64 without the quieting v_max_f64 x, x pairs the compiler puts around fmin(): the
// operands here are never NaN.
__device__ __forceinline__ double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// min(v, v of the partner lane) in one VOP2-DPP instruction.  The s_nop covers the two wait
// states a DPP read needs after a VALU write of the same register (the compiler's hazard
// recogniser does not look into inline assembly).
#define SAGE_MIN_U32_DPP(name, ctrl)                                                          \
    __device__ __forceinline__ unsigned name(unsigned v) {                                    \
        unsigned r;                                                                           \
        asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf"         \
            : "=v"(r)                                                                         \
            : "v"(v));                                                                        \
        return r;                                                                             \
    }
SAGE_MIN_U32_DPP(min_u32_xor1, "quad_perm:[1,0,3,2]")
SAGE_MIN_U32_DPP(min_u32_xor2, "quad_perm:[2,3,0,1]")
SAGE_MIN_U32_DPP(min_u32_half_mirror, "row_half_mirror")
SAGE_MIN_U32_DPP(min_u32_mirror, "row_mirror")
#undef SAGE_MIN_U32_DPP

// One group: `len` (1..16) consecutive queries that share a home voxel.  W = 2^LW lanes serve each
// query.  The 27-voxel neighbourhood is walked VOXEL BY VOXEL, the home voxel first: a voxel's
// points are one contiguous run of 32-B records, lane ci of a query takes points ci, ci + W, ...
// (scalar base offset + per-lane index through the raw buffer resource: no address arithmetic, no
// candidate list).  After the home voxel every other occupied voxel is kept only if its CELL can
// still hold a better point for at least one query of the group — an exact test:
//   a point stored in voxel v lies in v's cell (it was inserted by the same fp64 divide +
//   truncation, VoxelHashMap.cpp:165), so its squared distance to the query is at least the
//   squared distance to the cell, and its semantically scaled distance (VoxelHashMap.cpp:87-88)
//   at least min(th, 1) times that.  If that lower bound (taken with a relative slack of 1e-9
//   and an absolute slack on every face, far above fp64 rounding) is strictly above the best
//   scaled distance the query already holds, no point of v can win or tie.
// The surviving voxels are visited in ascending order and every lane keeps the lexicographic
// minimum of (scaled distance, enumeration index) — enumeration index = the reference's order, x
// outer, y, z inner, then insertion order (VoxelHashMap.cpp:57-63,73-75) — so the result is the
// sequential strict-< scan's, index for index.
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// One map point through the buffer path: byte offset = scalar `soff` (the voxel block's first
// point) + per-lane `voff`; the resource carries the 64-bit base, so a load costs no VALU address
// arithmetic.
__device__ __forceinline__ Point4 load_point(__amdgpu_buffer_rsrc_t pts, uint32_t voff, uint32_t soff) {
    const v4u a = __builtin_amdgcn_raw_buffer_load_b128(pts, voff, soff, 0);
    const v4u b = __builtin_amdgcn_raw_buffer_load_b128(pts, voff + 16u, soff, 0);
    Point4 q;
    q.x = __hiloint2double(static_cast<int>(a.y), static_cast<int>(a.x));
    q.y = __hiloint2double(static_cast<int>(a.w), static_cast<int>(a.z));
    q.z = __hiloint2double(static_cast<int>(b.y), static_cast<int>(b.x));
    q.l = __hiloint2double(static_cast<int>(b.w), static_cast<int>(b.z));
    return q;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// lanes 0..26 <-> neighbour voxel (ox, oy, oz) + 1 = (lane / 9, lane / 3 % 3, lane % 3)
constexpr unsigned long long kMaskXNeg = 0x00001FFull, kMaskXPos = 0x7FC0000ull;
constexpr unsigned long long kMaskYNeg = 0x01C0E07ull, kMaskYPos = 0x70381C0ull;
constexpr unsigned long long kMaskZNeg = 0x1249249ull, kMaskZPos = 0x4924924ull;
constexpr unsigned kHomeVoxel = 13u;

template <int LW>      // W = 2^LW lanes per query
__device__ __forceinline__ void nn_group(const NnParams &P, __amdgpu_buffer_rsrc_t pts, int lane,
                                         int start, int len, const uint2 ob, const Point4 &p,
                                         const double *gap, unsigned &pairs_eval) {
    constexpr int W = 1 << LW;
    const int qi = lane >> LW;                  // query of this lane within the group
    const unsigned ci = lane & (W - 1);
    const bool active = qi < len;
    const int pli = static_cast<int>(p.l);
    const double th = P.sem_th;

    // closest_distance2 starts at numeric_limits<double>::max() (VoxelHashMap.cpp:80); the value
    // travels as a kernel argument so that it sits in scalar registers
    double best = P.dist_init;                  // scaled squared distance
    unsigned best_f = 0xFFFFFFFFu;              // its enumeration index (the tie-break key)
    unsigned best_off = 0u;                     // its byte offset in the point array

    // lane v < 27 holds voxel v's row: ob.x = candidates before it, ob.y = its first point
    const unsigned cnt = dpp_u32<0x130>(ob.x) - ob.x;             // wave_shl:1 -> points in voxel v
    const unsigned occ = static_cast<unsigned>(__ballot(lane < 27 && cnt != 0u));

    auto visit = [&](int v) {                   // v is wave-uniform
        const unsigned off_v = rl_u32(ob.x, v), cnt_v = rl_u32(cnt, v);
        const unsigned base = rl_u32(ob.y, v) << 5;              // byte offset of 32-B points
        pairs_eval += cnt_v;
        for (unsigned i0 = 0; i0 < cnt_v; i0 += W) {             // uniform trip count
            const unsigned i = i0 + ci;
            if (i < cnt_v) {
                const Point4 nb = load_point(pts, i << 5, base);
                const double dx = nb.x - p.x, dy = nb.y - p.y, dz = nb.z - p.z;
                double d = dx * dx + (dy * dy + dz * dz);
                // same label, or either side unlabelled (VoxelHashMap.cpp:87-88)
                // ((int)(a * b) == 0  <=>  |a * b| < 1 under truncation toward zero)
                const bool same = static_cast<int>(nb.l) == pli || fabs(nb.l * p.l) < 1.0;
                const double ds = d * th;
                d = same ? ds : d;
                const unsigned f = off_v + i;
                // lexicographic (d, f): voxels are not visited in enumeration order (home first);
                // a NaN distance never wins
                const bool lt = d < best, eq = d == best, fl = f < best_f;
                const bool take = lt | (eq & fl);               // no short-circuit branches
                best = min_f64(best, d);
                best_f = take ? f : best_f;
                best_off = take ? base + (i << 5) : best_off;
            }
        }
    };

    if ((occ >> kHomeVoxel) & 1u) visit(static_cast<int>(kHomeVoxel));

    // what every query holds after its home voxel bounds the rest of its search
    double m = best;
    if (W >= 2) m = min_f64(m, dpp_f64<kDppXor1>(m));
    if (W >= 4) m = min_f64(m, dpp_f64<kDppXor2>(m));
    if (W >= 8) m = min_f64(m, dpp_f64<kDppHalfMirror>(m));
    if (W >= 16) m = min_f64(m, dpp_f64<kDppMirror>(m));
    if (W >= 32) m = min_f64(m, __shfl_xor(m, 16, 64));
    if (W >= 64) m = min_f64(m, __shfl_xor(m, 32, 64));

    unsigned need = P.keep_all;                 // 0, or all 27 voxels when pruning is off
    {
        const bool xn = (kMaskXNeg >> lane) & 1ull, xp = (kMaskXPos >> lane) & 1ull;
        const bool yn = (kMaskYNeg >> lane) & 1ull, yp = (kMaskYPos >> lane) & 1ull;
        const bool zn = (kMaskZNeg >> lane) & 1ull, zp = (kMaskZPos >> lane) & 1ull;
        for (int q = 0; q < len; ++q) {         // lane v < 27: lower bound of voxel v for query q
            const double bq = readlane_f64(m, q << LW);
            // uniform addresses: three broadcast reads, then per-lane selects
            const double2 g01 = *reinterpret_cast<const double2 *>(gap + 6 * q);
            const double2 g23 = *reinterpret_cast<const double2 *>(gap + 6 * q + 2);
            const double2 g45 = *reinterpret_cast<const double2 *>(gap + 6 * q + 4);
            double gx = xp ? g01.y : 0.0, gy = yp ? g23.y : 0.0, gz = zp ? g45.y : 0.0;
            gx = xn ? g01.x : gx;
            gy = yn ? g23.x : gy;
            gz = zn ? g45.x : gz;
            const double lb = gx + (gy + gz);
            need |= static_cast<unsigned>(__ballot(lb <= bq));
        }
    }
    need &= occ & ~(1u << kHomeVoxel);
    while (need) {
        const int v = __builtin_ctz(need);
        need &= need - 1u;
        visit(v);
    }

    // argmin over the W lanes of each query, lexicographic in (distance, enumeration index) like
    // the sequential strict-< scan it replaces: first the minimum distance (never NaN: a NaN
    // distance fails every comparison), then the smallest index among the lanes that hold it.
    m = best;
    if (W >= 2) m = min_f64(m, dpp_f64<kDppXor1>(m));
    if (W >= 4) m = min_f64(m, dpp_f64<kDppXor2>(m));
    if (W >= 8) m = min_f64(m, dpp_f64<kDppHalfMirror>(m));
    if (W >= 16) m = min_f64(m, dpp_f64<kDppMirror>(m));
    if (W >= 32) m = min_f64(m, __shfl_xor(m, 16, 64));
    if (W >= 64) m = min_f64(m, __shfl_xor(m, 32, 64));
    const unsigned mine = (best == m) ? best_f : 0xFFFFFFFFu;
    unsigned minf = mine;
    if (W >= 2) minf = min_u32_xor1(minf);
    if (W >= 4) minf = min_u32_xor2(minf);
    if (W >= 8) minf = min_u32_half_mirror(minf);
    if (W >= 16) minf = min_u32_mirror(minf);
    if (W >= 32) minf = min(minf, static_cast<unsigned>(__shfl_xor(static_cast<int>(minf), 16, 64)));
    if (W >= 64) minf = min(minf, static_cast<unsigned>(__shfl_xor(static_cast<int>(minf), 32, 64)));

    // The argmin is stored unconditionally; the acceptance test on the unscaled distance
    // (VoxelHashMap.cpp:111) is applied where the pair is consumed (k_gn / the host join).
    // Enumeration indices are unique, so exactly one lane of a query holds the winner.
    if (active) {
        if (minf == 0xFFFFFFFFu) {
            if (ci == 0) P.nn_idx[start + qi] = -1;
        } else if (mine == minf) {
            P.nn_idx[start + qi] = static_cast<int>(best_off >> 5);
        }
    }
}

#ifdef SAGE_NN_TIMING
constexpr unsigned kNnTimingSlots = 1u << 17;
__device__ unsigned long long g_nn_phase[4ull * kNnTimingSlots];   // per chunk: {head, group, lifetime, count}
#define NN_T(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); tph[i] += _t - tprev; tprev = _t; } while (0)
#else
#define NN_T(i) do { } while (0)
#endif

constexpr int kNnWaves = SAGE_NN_WAVES;     // waves per k_nn workgroup

__global__ __launch_bounds__(64 * kNnWaves, SAGE_NN_OCC) void k_nn(NnParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    if (P.check_done && P.st->done) return;
#ifdef SAGE_NN_TIMING
    unsigned long long tph[4] = {0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
#endif

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));   // wave-uniform
    // per-wave LDS (words): the chunk's transformed queries {x, y, z, label} | their scaled
    // squared gaps to the six faces of the home cell | their home voxels [comp][query]
    const NnLds L = nn_lds_layout(P.chunk);
    uint32_t *wl = smem + wv * L.wave_words;
    // raw buffer resource over the point array (bounds-checked, 32-bit byte offsets)
    const __amdgpu_buffer_rsrc_t pts = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<Point4 *>(P.pts), 0, static_cast<int>(P.pts_bytes), 0x00020000);
    double *spt = reinterpret_cast<double *>(wl + L.spt);
    double *gap = reinterpret_cast<double *>(wl + L.gap);
    int *skey = reinterpret_cast<int *>(wl + L.skey);

    // One wave per chunk of `chunk` consecutive queries (a group never crosses a chunk) and many
    // more workgroups than the chip holds at once: the hardware dispatcher hands the next
    // workgroup to whichever CU frees a slot, which balances the load.  (Work per query varies
    // several-fold across the scene; persistent waves with a static share of the queries left
    // the kernel waiting on its heaviest wave, and device-scope ticket counters were 20x slower.)
    // Workgroup b is dispatched to XCD b % 8 (observed; speed only): XCD x serves the stripes
    // x, x+8, x+16, ... of kStripe consecutive workgroups' worth of the spatially sorted frame,
    // so each private L2 sees a few compact regions of the map and every XCD gets the same mix
    // of dense and sparse regions.
    unsigned long long wave_candidates = 0;   // wave-uniform: sum over this wave's queries of C_q
    unsigned long long wave_pairs = 0;        // (query, candidate) pairs actually evaluated
    constexpr unsigned kStripe = SAGE_NN_STRIPE / kNnWaves;   // workgroups per stripe
    unsigned cand_slot = 0;
    {
      const unsigned chunk = P.chunk;
      const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
      const unsigned quad = ((j / kStripe) * 8u + xcd) * kStripe + (j % kStripe);
      uint2 *blks = P.blks;
      auto load_table = [&](unsigned g) -> uint2 {
          uint2 ob;
          ob.x = 0xFFFFFFFFu; ob.y = 0u;
          if (lane < 32) ob = blks[g * 32u + lane];
          return ob;
      };
      const unsigned c = __builtin_amdgcn_readfirstlane(quad * kNnWaves + wv);
      cand_slot = c;
      if (c < P.nchunks) {
        const unsigned q0 = c * chunk;
        // ---- chunk prologue.  Lane (comp, i) = (lane >> log2 chunk, lane & (chunk-1)) owns
        // coordinate `comp` of query q0 + i: it applies the cumulative pose to the pristine frame
        // point (TransformPoints, Registration.cpp:103-111,133 — `source` is never rewritten in
        // place), takes the home voxel index with the reference's fp64 divide + truncation
        // (VoxelHashMap.cpp:52-54), measures the distance to the two faces of the home cell on
        // its axis (the pruning bounds of nn_group) and parks everything in LDS for the pair lanes.
        uint2 ob = load_table(q0);                   // the first query of a chunk is always a head
        const unsigned qi_own = static_cast<unsigned>(lane) & (chunk - 1u);
        const unsigned comp = static_cast<unsigned>(lane) >> P.chunk_log2;
        const unsigned q_own = q0 + qi_own;
        const bool lv = comp < 3u && q_own < static_cast<unsigned>(P.n);
        int key = 0, ckey = 0x7F7F7F7F;
        if (lv) {
            const double *f = reinterpret_cast<const double *>(P.frame + q_own);
            ckey = reinterpret_cast<const int *>(P.tabkey + q_own)[1 + comp];
            const double fx = f[0], fy = f[1], fz = f[2], fl = f[3];
            double sv;
            if (P.apply_pose) {
                const double *R = P.st->R + 3u * comp;
                sv = R[0] * fx + R[1] * fy + R[2] * fz + P.st->T[4u + comp];
            } else {
                sv = comp == 0u ? fx : (comp == 1u ? fy : fz);
            }
            // static_cast<int>(p / voxel_size): exact fp64 divide, truncation toward zero
            const double vs = P.voxel_size;
            key = static_cast<int>(sv / vs);
            // The cell of voxel index k on one axis (truncation toward zero: cell 0 is two voxels
            // wide): [k vs, (k+1) vs) for k > 0, (-vs, vs) for k = 0, ((k-1) vs, k vs] for k < 0.
            // Points stored in the voxel below / above the home voxel therefore lie at or beyond
            // `below_hi` / `above_lo`; the gaps are shortened by an absolute slack that dwarfs the
            // rounding of the divide, the product and the subtraction (~1e-16 relative).
            const double below_hi = static_cast<double>(key <= 0 ? key - 1 : key) * vs;
            const double above_lo = static_cast<double>(key >= 0 ? key + 1 : key) * vs;
            const double slack = 1e-9 * vs + 1e-13 * fabs(sv);
            const double glo = fmax((sv - below_hi) - slack, 0.0);
            const double ghi = fmax((above_lo - sv) - slack, 0.0);
            spt[4u * qi_own + comp] = sv;
            skey[lane] = key;
            gap[6u * qi_own + 2u * comp] = (glo * glo) * P.prune_scale;
            gap[6u * qi_own + 2u * comp + 1u] = (ghi * ghi) * P.prune_scale;
            if (comp == 0u) spt[4u * qi_own + 3u] = fl;
            if (P.src) {                              // the queries as searched, for k_gn
                double *o = reinterpret_cast<double *>(P.src + q_own);
                o[comp] = sv;
                if (comp == 0u) o[3] = fl;
            }
        }
        // Group heads: a query whose home voxel differs from its predecessor's (chunks are
        // aligned to rows of 16 lanes, so the predecessor is one DPP row-shift away), and the
        // first query of the chunk.  Stale rows: heads whose cached probe-table row was built
        // for another voxel (all of them in the first iteration, a handful afterwards: the pose
        // moves by millimetres per iteration and the map does not change during a call).
        const int prev = static_cast<int>(dpp_u32<0x111>(static_cast<unsigned>(key)));   // row_shr:1
        const unsigned long long ne = __ballot(lv && qi_own > 0u && key != prev);
        const unsigned long long mm = __ballot(lv && key != ckey);
        const unsigned nvalid = min(chunk, static_cast<unsigned>(P.n) - q0);
        const unsigned vmask = (nvalid >= 32u) ? 0xFFFFFFFFu : ((1u << nvalid) - 1u);
        unsigned heads = (static_cast<unsigned>(ne | (ne >> chunk) | (ne >> (2u * chunk))) | P.cap_heads) & vmask;
        const unsigned stale = static_cast<unsigned>(mm | (mm >> chunk) | (mm >> (2u * chunk))) & heads;
        while (heads) {
        const int h = __builtin_ctz(heads);
        heads &= heads - 1u;
        const int hn = heads ? __builtin_ctz(heads) : static_cast<int>(nvalid);   // end of the group
        const uint2 ob_next = load_table(q0 + static_cast<unsigned>(heads ? hn : h));   // prefetch
        const int start = static_cast<int>(q0) + h;
        const int len = hn - h;
        if ((stale >> h) & 1u) {
            ob = probe_row(P, lane, smem, wv * L.wave_words + L.skey, h, static_cast<unsigned>(start));
        }
        // (query x candidate) pairs over the lanes: W = 64 / pow2ceil(len) lanes per query
        const int lgq = (len <= 1) ? 0 : (32 - __builtin_clz(static_cast<unsigned>(len - 1)));
        const int lw = 6 - lgq;
        Point4 p;                                     // this lane's query
        {
            const double4 t = *reinterpret_cast<const double4 *>(spt + 4 * (h + min(lane >> lw, len - 1)));
            p.x = t.x; p.y = t.y; p.z = t.z; p.l = t.w;
        }
        NN_T(0);
        wave_candidates += static_cast<unsigned long long>(rl_u32(ob.x, 27)) * static_cast<unsigned>(len);
        unsigned pe = 0;
        switch (lw) {
            case 6: nn_group<6>(P, pts, lane, start, len, ob, p, gap + 6 * h, pe); break;
            case 5: nn_group<5>(P, pts, lane, start, len, ob, p, gap + 6 * h, pe); break;
            case 4: nn_group<4>(P, pts, lane, start, len, ob, p, gap + 6 * h, pe); break;
            case 3: nn_group<3>(P, pts, lane, start, len, ob, p, gap + 6 * h, pe); break;
            default: nn_group<2>(P, pts, lane, start, len, ob, p, gap + 6 * h, pe); break;
        }
        wave_pairs += static_cast<unsigned long long>(pe) * static_cast<unsigned>(len);
        NN_T(1);
        ob = ob_next;
        }
      }
    }
#ifdef SAGE_NN_TIMING
    // private slot per chunk (no contended atomics: they would stall the very loads being timed)
    if (lane == 0 && cand_slot < kNnTimingSlots) {
        unsigned long long *t = g_nn_phase + 4ull * cand_slot;
        t[0] += tph[0];
        t[1] += tph[1];
        t[2] += __builtin_amdgcn_s_memtime() - tstart;
        t[3] += 1ull;
    }
#endif
    // sum_q C_q for the roofline accounting: one private slot per chunk, summed by the host.  (A
    // single device-scope atomic per wave serialised 8192 updates on one address and set a
    // ~100 us floor under this kernel.)
    if (P.cand_counter && lane == 0 && wave_candidates) {
        atomicAdd(P.cand_counter + 2u * cand_slot, wave_candidates);   // fire-and-forget, private address
        atomicAdd(P.cand_counter + 2u * cand_slot + 1u, wave_pairs);
    }
}

// ------------------------------------------------------------------------------------ WaveLanes
// Lane policy (se3_math.h) for the wave that finishes an iteration: its 64 lanes all hold the
// same (uniform) values, so independent fp64 divisions / sincos arguments are moved to separate
// lanes, evaluated by ONE vector instruction sequence, and read back with v_readlane.  A serial
// lane spent ~2 us of every iteration in the 21 divisions of the 6x6 LDL^T alone.
// Must be called with lanes 0..5 active and uniform operands.
struct WaveLanes {
    static __device__ __forceinline__ void divide6(const double (&n)[6], const double (&d)[6],
                                                   double (&q)[6]) {
        const int lane = static_cast<int>(threadIdx.x & 63u);
        double nn = n[0], dd = d[0];
#pragma unroll
        for (int i = 1; i < 6; ++i) {
            nn = (lane == i) ? n[i] : nn;
            dd = (lane == i) ? d[i] : dd;
        }
        const double qq = nn / dd;
#pragma unroll
        for (int i = 0; i < 6; ++i) q[i] = readlane_f64(qq, i);
    }
    static __device__ __forceinline__ void sincos2(double a0, double a1, double &s0, double &c0,
                                                   double &s1, double &c1) {
        const int lane = static_cast<int>(threadIdx.x & 63u);
        double sv, cv;
        sincos(lane == 1 ? a1 : a0, &sv, &cv);
        s0 = readlane_f64(sv, 0); c0 = readlane_f64(cv, 0);
        s1 = readlane_f64(sv, 1); c1 = readlane_f64(cv, 1);
    }
};

// ------------------------------------------------------------------------------ finish_iteration
// Executed by ONE workgroup of 256 threads once per ICP iteration: fixed-order reduction of the
// k_gn workgroup partials (bit-reproducible), then one lane assembles the 6x6 normal equations
// from the 16 closed-form sums, solves them (register-resident pivoted LDL^T), applies SE3 exp,
// composes the pose and tests convergence (Registration.cpp:92-93,135-137).
//   mode 0: reduce partials + solve      (single GPU)
//   mode 1: reduce partials -> st->sums  (multi GPU, before the RCCL all-reduce)
//   mode 2: solve from st->sums          (multi GPU, after the all-reduce)
#ifdef SAGE_GN_TIMING
__device__ unsigned long long g_gn_phase[16];
#define FIN_STAMP(i) do { if (threadIdx.x == 0) fin_t[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FIN_STAMP(i) do { } while (0)
#endif
__device__ __forceinline__ void finish_iteration(IcpState *st, const double *partials, int nparts,
                                                 int mode) {
    __shared__ double slice[8][32];
    __shared__ double S[kNumSums];
#ifdef SAGE_GN_TIMING
    unsigned long long fin_t[8];
#endif
    FIN_STAMP(0);

    if (mode != 2) {
        const int comp = threadIdx.x & 31, sl = threadIdx.x >> 5;
        double v = 0.0;
        if (comp < kNumSums) {
            // fixed summation order; loads are independent, 8 in flight per thread
            int b = sl;
            for (; b + 56 < nparts; b += 64) {
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = partials[(b + 8 * u) * kNumSums + comp];
#pragma unroll
                for (int u = 0; u < 8; ++u) v += t[u];
            }
            for (; b < nparts; b += 8) v += partials[b * kNumSums + comp];
        }
        slice[sl][comp] = v;
        __syncthreads();
        if (threadIdx.x < kNumSums) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += slice[i][threadIdx.x];   // fixed order
            S[threadIdx.x] = s;
            st->sums[threadIdx.x] = s;
        }
        __syncthreads();
        if (mode == 1) return;
    } else {
        if (threadIdx.x < kNumSums) S[threadIdx.x] = st->sums[threadIdx.x];
        __syncthreads();
    }

    if (threadIdx.x >= 64) return;
    // wave 0, all 64 lanes, uniform data (see WaveLanes); lane 0 / lane 1 publish the state
    FIN_STAMP(1);
    const int lane = static_cast<int>(threadIdx.x);
    double JTJ[36], JTr[6], neg[6], x[6], est[7];
    assemble_normal_equations(S, JTJ, JTr);
#pragma unroll
    for (int i = 0; i < 6; ++i) neg[i] = -JTr[i];
    ldlt_solve6_t<WaveLanes>(JTJ, neg, x);
    FIN_STAMP(2);
    se3_exp_t<WaveLanes>(x, est);
    FIN_STAMP(3);

    // the two compositions (Registration.cpp:135 and the cumulative pose) on lanes 0 and 1
    double rhs[7], Tn[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) rhs[i] = (lane == 1) ? st->T_icp[i] : st->T[i];
    se3_mul(est, rhs, Tn);
    if (lane < 2) {
        double *dst = (lane == 1) ? st->T_icp : st->T;
#pragma unroll
        for (int i = 0; i < 7; ++i) dst[i] = Tn[i];
    }
    if (lane != 0) return;
    quat_to_mat(Tn, st->R);

    // ||log(exp(x))|| == ||x|| (principal branch, |omega| < pi, which a Gauss-Newton step of a
    // converging registration always satisfies): the reference's estimation.log().norm()
    // (Registration.cpp:137) without the atan2/sincos round trip on one serial lane.
    double nrm = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) nrm += x[i] * x[i];
    nrm = sqrt(nrm);
    if (!(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] < 9.0)) {   // |omega| >= 3: take the exact path
        double lg[6];
        se3_log(est, lg);
        nrm = 0.0;
        for (int i = 0; i < 6; ++i) nrm += lg[i] * lg[i];
        nrm = sqrt(nrm);
    }
    st->last_step_norm = nrm;
    const int it = st->iter;
    if (it < kHistory) st->n_corr[it] = static_cast<uint32_t>(S[kCount]);
    st->iter = it + 1;
    unsigned long long done = 0;
    if (nrm < kEstimationThreshold) {
        st->converged = 1;
        st->done = 1;
        done = 1;
    } else if (it + 1 >= kMaxIterations) {
        st->done = 1;
        done = 1;
    }
    if (st->done) done = 1;                    // e.g. stopped by a failed multi-GPU exchange
    if (IcpProgress *pg = st->progress) {
        // host-mapped: the pose, then the progress word, as relaxed system-scope (write-through)
        // stores — a release here would write back this XCD's whole L2 every iteration; the host
        // only steers its look-ahead and its re-sort heuristic by these values and reads the
        // final state through an ordinary copy after the loop
#pragma unroll
        for (int i = 0; i < 7; ++i)
            __hip_atomic_store(&pg->T[i], Tn[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pg->word, (done << 32) | static_cast<unsigned long long>(it + 1),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#ifdef SAGE_GN_TIMING
    fin_t[4] = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 4; ++i) atomicAdd(&g_gn_phase[8 + i], fin_t[i + 1] - fin_t[i]);
#endif
}

// --------------------------------------------------------------------------------- exchange_sums
// One workgroup per rank (the last arriver of k_gn), see P2pBlock.  st->sums holds this rank's
// sums on entry and the sums over all ranks on exit.  Stores to the peers are system-scope
// write-through atomics, completed (s_waitcnt) and fenced before the tag goes out; the tags are
// polled with system-scope loads and an acquire fence precedes the reads of the rows.
__device__ __forceinline__ void exchange_sums(IcpState *st, const P2pParams &X) {
    const int t = static_cast<int>(threadIdx.x);
    const unsigned long long g = *X.exchanges;
    const int slot = static_cast<int>(g & 1ull);
    const unsigned long long tag = g + 1ull;
    if (t < kNumSums) {
        const double v = st->sums[t];
        for (int r = 0; r < X.nranks; ++r)
            __hip_atomic_store(&X.block[r]->sums[slot][X.rank][t], v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        for (int r = 0; r < X.nranks; ++r)
            __hip_atomic_store(&X.block[r]->flag[X.rank], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    P2pBlock *mine = X.block[X.rank];
    __shared__ int s_late;
    if (t == 0) s_late = 0;
    __syncthreads();
    if (t < X.nranks) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(&mine->flag[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < tag) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t0 > X.timeout_ticks) {
                s_late = 1;
                break;
            }
        }
    }
    __syncthreads();
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    if (t < kNumSums) {
        double s = 0.0;
        for (int r = 0; r < X.nranks; ++r)       // rank order: the same sum on every rank
            s += __hip_atomic_load(&mine->sums[slot][r][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        st->sums[t] = s;
    }
    if (t == 0) {
        *X.exchanges = tag;
        if (s_late) {                          // stop the loop; the host reports the failure
            st->exchange_failed = 1;
            st->done = 1;
        }
    }
}

// ------------------------------------------------------------------------------------ k_gn
#ifdef SAGE_GN_TIMING
#define GN_STAMP(i) do { if (threadIdx.x == 0) gn_t[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GN_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(256) void k_gn(GnParams P) {
    if (P.check_done && P.st->done) return;
    // block reduction scratch: component c of thread t at red[c][(t >> 4) * 17 + (t & 15)]
    // (rows of 16 values padded to 17 doubles: conflict-free for both the write and the read)
    __shared__ double red[kCount][16 * 17];
    __shared__ unsigned wave_pairs[4];
#ifdef SAGE_GN_TIMING
    unsigned long long gn_t[8];
    GN_STAMP(0);
#endif

    double acc[kCount];
#pragma unroll
    for (int i = 0; i < kCount; ++i) acc[i] = 0.0;
    unsigned pairs = 0;   // accepted pairs of this wave (wave-uniform, exact in any order)

    const double k = P.kernel;
    const double k2 = k * k;

    // per-pair accumulation; `use` masks rejected / out-of-range pairs (w = 0 adds exact zeros)
    auto accumulate = [&](const Point4 &s, const Point4 &g, bool use) {
        const double sx = s.x, sy = s.y, sz = s.z;
        const double rx = sx - g.x, ry = sy - g.y, rz = sz - g.z;
        const double r2 = rx * rx + (ry * ry + rz * rz);
        // acceptance: (closest_neighboor - point).norm() < max_correspondance_distance
        // (VoxelHashMap.cpp:111); explicit pairs (align_clouds entry) are all taken
        if (!P.tgt_pairs && !(sqrt(r2) < P.max_dist)) use = false;
        pairs += static_cast<unsigned>(__popcll(__ballot(use)));
        if (!use) return;
        const double den = k + r2;
        const double w = k2 / (den * den);   // square(th) / square(th + residual2)
        const double wsx = w * sx, wsy = w * sy, wsz = w * sz;
        acc[kW] += w;
        acc[kWsx] += wsx; acc[kWsy] += wsy; acc[kWsz] += wsz;
        acc[kWxx] += wsx * sx; acc[kWxy] += wsx * sy; acc[kWxz] += wsx * sz;
        acc[kWyy] += wsy * sy; acc[kWyz] += wsy * sz; acc[kWzz] += wsz * sz;
        acc[kWrx] += w * rx; acc[kWry] += w * ry; acc[kWrz] += w * rz;
        acc[kWcx] += w * (sy * rz - sz * ry);
        acc[kWcy] += w * (sz * rx - sx * rz);
        acc[kWcz] += w * (sx * ry - sy * rx);
    };

    // grid-stride over the queries, two per step so that the dependent gathers
    // (nn_idx -> target point) of both are in flight together
    const int stride = gridDim.x * 256;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < P.n; q += 2 * stride) {
        const int q2 = q + stride;
        const bool has2 = q2 < P.n;
        int i1 = 0, i2 = 0;
        if (!P.tgt_pairs) {
            i1 = P.nn_idx[q];
            i2 = has2 ? P.nn_idx[q2] : -1;
        }
        const Point4 s1 = P.src[q];
        const Point4 s2 = P.src[has2 ? q2 : q];
        Point4 g1, g2;
        if (P.tgt_pairs) {
            g1 = P.tgt_pairs[q];
            g2 = P.tgt_pairs[has2 ? q2 : q];
        } else {
            g1 = P.pts[i1 < 0 ? 0 : i1];
            g2 = P.pts[i2 < 0 ? 0 : i2];
        }
        accumulate(s1, g1, i1 >= 0);
        accumulate(s2, g2, has2 && i2 >= 0);
    }

    GN_STAMP(1);
    // Block reduction in a fixed order (bit-reproducible): every thread parks its 16 sums in LDS,
    // then 16 threads per component each add 16 parked values and finish with four DPP exchange
    // steps inside their row of 16 lanes.  (A shuffle tree per component and wave cost ~3.9 us
    // of LDS-permute traffic per launch.)
    {
        const int t = static_cast<int>(threadIdx.x);
        const int slot = (t >> 4) * 17 + (t & 15);
#pragma unroll
        for (int c = 0; c < kCount; ++c) red[c][slot] = acc[c];
        if ((t & 63) == 0) wave_pairs[t >> 6] = pairs;
        __syncthreads();
        const double *row = &red[t >> 4][(t & 15) * 17];
        double v = row[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) v += row[i];
        v += dpp_f64<kDppXor1>(v);
        v += dpp_f64<kDppXor2>(v);
        v += dpp_f64<kDppHalfMirror>(v);
        v += dpp_f64<kDppMirror>(v);
        double *out = P.partials + static_cast<size_t>(blockIdx.x) * kNumSums;
        if ((t & 15) == 0) out[t >> 4] = v;
        if (t == 0)
            out[kCount] = static_cast<double>((wave_pairs[0] + wave_pairs[1]) +
                                              (wave_pairs[2] + wave_pairs[3]));
        if (t > kCount && t < kNumSums) out[t] = 0.0;
    }
    if (P.fuse_mode < 0) return;
    GN_STAMP(2);

    // Last-arriver hand-off (placement independent): partials are published with an agent-scope
    // release before the ticket, the workgroup that draws the last ticket acquires (drops its
    // CU's stale L1 lines of `partials`, which other CUs rewrite every iteration) and finishes the
    // iteration.  One returning atomic per workgroup (<= 512 per launch) on a private word.
    __shared__ unsigned s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(P.ticket, 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(P.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
    }
    __syncthreads();
    GN_STAMP(3);
    if (P.fuse_mode == 3) {
        // multi-GPU without a collective launch: local sums -> peers -> global sums -> solve
        finish_iteration(P.st_rw, P.partials, static_cast<int>(gridDim.x), 1);
        __syncthreads();
        exchange_sums(P.st_rw, P.p2p);
        __syncthreads();
        finish_iteration(P.st_rw, P.partials, 0, 2);
        return;
    }
    finish_iteration(P.st_rw, P.partials, static_cast<int>(gridDim.x), P.fuse_mode);
#ifdef SAGE_GN_TIMING
    if (threadIdx.x == 0) {
        gn_t[4] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 4; ++i) atomicAdd(&g_gn_phase[i], gn_t[i + 1] - gn_t[i]);
        atomicAdd(&g_gn_phase[4], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------ k_fin
// Stand-alone launch of finish_iteration: the post-all-reduce solve of the multi-GPU path
// (mode 2).  On a single GPU the last workgroup of k_gn runs it instead (no extra launch).
__global__ __launch_bounds__(256) void k_fin(IcpState *st, const double *partials, int nparts,
                                             int mode, int standalone) {
    if (!standalone && st->done) return;
    finish_iteration(st, partials, nparts, mode);
}

// ------------------------------------------------------------------------------------ k_tf
__global__ __launch_bounds__(256) void k_tf(Point4 *pts, int n, const IcpState *st) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Point4 p = pts[i];
    const double *R = st->R;
    const double x = R[0] * p.x + R[1] * p.y + R[2] * p.z + st->T[4];
    const double y = R[3] * p.x + R[4] * p.y + R[5] * p.z + st->T[5];
    const double z = R[6] * p.x + R[7] * p.y + R[8] * p.z + st->T[6];
    p.x = x; p.y = y; p.z = z;
    pts[i] = p;
}

#ifdef SAGE_GN_TIMING
extern "C" void sageicp_debug_gn_phases(unsigned long long out[16], int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gn_phase), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gn_phase), z, sizeof(z));
    }
}
#endif
#ifdef SAGE_NN_TIMING
extern "C" void sageicp_debug_nn_phases(unsigned long long out[8], int reset) {
    // out: {sum head cycles, sum group cycles, sum wave lifetime, waves, max mean lifetime of a
    //       chunk slot, slots used, 0, 0}
    std::vector<unsigned long long> h(4ull * kNnTimingSlots);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_nn_phase), h.size() * sizeof(unsigned long long));
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (unsigned s = 0; s < kNnTimingSlots; ++s) {
        const unsigned long long *t = &h[4ull * s];
        if (!t[3]) continue;
        out[0] += t[0]; out[1] += t[1]; out[2] += t[2]; out[3] += t[3];
        if (t[2] / t[3] > out[4]) out[4] = t[2] / t[3];
        ++out[5];
    }
    if (reset) {
        std::fill(h.begin(), h.end(), 0ull);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nn_phase), h.data(), h.size() * sizeof(unsigned long long));
    }
}
#endif

// ------------------------------------------------------------------------------- mirror refresh
// Scatter the records the host changed since the last sync into the HBM mirror: one staged copy
// + one kernel per array instead of thousands of small hipMemcpy calls.
__global__ __launch_bounds__(256) void k_scatter_points(const uint32_t *idx, const Point4 *vals,
                                                        uint32_t n, Point4 *pts) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) pts[idx[i]] = vals[i];
}
__global__ __launch_bounds__(256) void k_scatter_slots(const uint32_t *idx, const Slot *vals,
                                                       uint32_t n, Slot *table) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) table[idx[i]] = vals[i];
}

// sums of k_nn's per-chunk counters {candidates in the neighbourhood, pairs evaluated} into the
// loop state (one workgroup; it rides on the state copy the host makes anyway instead of a
// read-back of the counters)
__global__ __launch_bounds__(1024) void k_sum_candidates(const unsigned long long *c, int n,
                                                         IcpState *st) {
    __shared__ unsigned long long part[2][16];
    unsigned long long v = 0, w = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        v += c[2 * i];
        w += c[2 * i + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_down(v, off, 64);
        w += __shfl_down(w, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = v;
        part[1][threadIdx.x >> 6] = w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0, u = 0;
        for (int i = 0; i < 16; ++i) {
            t += part[0][i];
            u += part[1][i];
        }
        st->sum_candidates = t;
        st->sum_pairs = u;
    }
}

// ------------------------------------------------------------------------------------ launchers
void launch_sum_candidates(const unsigned long long *c, int n, IcpState *st, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_candidates, dim3(1), dim3(1024), 0, s, c, n, st);
}
void launch_scatter_points(const uint32_t *idx, const Point4 *vals, uint32_t n, Point4 *pts,
                           hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_scatter_points, dim3((n + 255) / 256), dim3(256), 0, s, idx, vals, n, pts);
}
void launch_scatter_slots(const uint32_t *idx, const Slot *vals, uint32_t n, Slot *table,
                          hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_scatter_slots, dim3((n + 255) / 256), dim3(256), 0, s, idx, vals, n, table);
}

int nn_grid_for(int n, int chunk) {
    // one wave per chunk, 4 waves per workgroup, rounded up to whole stripes on all 8 XCDs
    const long nchunks = (static_cast<long>(n) + chunk - 1) / chunk;
    const long quads = (nchunks + kNnWaves - 1) / kNnWaves;
    const long per_round = 8L * (SAGE_NN_STRIPE / SAGE_NN_WAVES);   // 8 XCDs x kStripe
    const long blocks = ((quads + per_round - 1) / per_round) * per_round;
    return static_cast<int>(blocks < per_round ? per_round : blocks);
}

int gn_grid_for(int n) {
    static const int cap = [] {
        const char *v = std::getenv("SAGEICP_GN_BLOCKS");
        int c = v ? std::atoi(v) : 128;   // measured best with the fused finish (ticket per block)
        return c < 1 ? 1 : (c > kMaxGnBlocks ? kMaxGnBlocks : c);
    }();
    long blocks = (static_cast<long>(n) + 255) / 256;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return static_cast<int>(blocks);
}


void launch_nn(const NnParams &p, hipStream_t s) {
    if (p.n <= 0) return;
    const int grid = nn_grid_for(p.n, static_cast<int>(p.chunk));
    const size_t lds = kNnWaves * nn_lds_layout(p.chunk).wave_words * sizeof(uint32_t);
    hipLaunchKernelGGL(k_nn, dim3(grid), dim3(64 * kNnWaves), lds, s, p);
}

int launch_gn(const GnParams &p, hipStream_t s) {
    const int grid = gn_grid_for(p.n);
    hipLaunchKernelGGL(k_gn, dim3(grid), dim3(256), 0, s, p);
    return grid;
}

void launch_fin(IcpState *st, const double *partials, int nparts, int mode, int standalone,
                hipStream_t s) {
    hipLaunchKernelGGL(k_fin, dim3(1), dim3(256), 0, s, st, partials, nparts, mode, standalone);
}

void launch_tf(Point4 *pts, int n, const IcpState *st, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_tf, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, st);
}

}  // namespace sageicp
