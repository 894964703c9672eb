// Hand-written HIP kernels for gfx950 (CDNA4, wave64) — SAGE-ICP registration hot path.
// One ICP iteration = k_icp -> k_fin, both on one stream, no host round trip.
//
//   k_icp    VoxelHashMap::GetCorrespondences (core/VoxelHashMap.cpp:48-130) and the accumulation
//            of AlignClouds (core/Registration.cpp:59-90) in ONE launch.  W = 1..16 lanes own a
//            query (few lanes for frames that fill the chip: fewest instructions per query; many
//            for small frames / shards: shortest dependent chain).  Per query: apply the
//            cumulative pose to the pristine frame point (TransformPoints, Registration.cpp:
//            103-111,133 — `source` is never rewritten in place); home voxel by the reference's
//            exact fp64 divide + truncation (VoxelHashMap.cpp:52-54); the query's cached
//            neighbourhood row (27 x map_.find, VoxelHashMap.cpp:66-78, redone only when the home
//            voxel changed) staged in LDS; scan of the home voxel, then ONLY of the neighbour
//            voxels whose cell can still hold a better point (exact lower bound, see below);
//            lexicographic (distance, enumeration order) argmin = the reference's sequential
//            strict-< scan, index for index.  Fused epilogue: acceptance test on the unscaled
//            distance (VoxelHashMap.cpp:111), robust weight and the 16 closed-form fp64 sums of
//            JtJ / Jtr per accepted pair, reduced in a fixed order to one partial per workgroup.
//   k_rows   builds every query's neighbourhood row after a (re-)sort of the frame (27 lanes probe
//            the GPU-resident open-addressed hash per query).
//   k_fin    one workgroup: fixed-order reduction of the partials (bit-reproducible), [exchange of
//            the sums with the peer GPUs,] 6x6 LDL^T solve and SE3 exp with their divisions /
//            sincos spread over lanes, pose composition and the convergence test
//            (Registration.cpp:92-93,135-137).
//   k_gn     the accumulation alone on explicit pairs (the stand-alone AlignClouds entry).
//   k_tf     TransformPoints for the stand-alone API entry (Registration.cpp:103-111).
//   k_scatter_points / k_scatter_slots   refresh of the HBM mirror of the host map.
//
// Pruning (exact).  A point stored in voxel v lies in v's cell — it was inserted by the same fp64
// divide + truncation (VoxelHashMap.cpp:165) — so its squared distance to the query is at least
// the squared distance from the query to the cell, and its semantically scaled distance
// (VoxelHashMap.cpp:87-88: d2 * th for matching labels, d2 otherwise) at least min(th, 1) times
// that.  If this lower bound — taken with a relative slack of 1e-9 and an absolute slack on
// every face, both far above fp64 rounding — is strictly above the scaled distance the query
// already holds after its home voxel, no point of v can win or tie, and v is skipped.  On the
// synthetic street scenes 75-85 % of the (query, map point) pairs the reference evaluates are
// never loaded.
//
// Compact candidates (exact).  What the kernel is made of is bytes through the eight L2s, so the
// scan does not read the 32-B fp64 records: k_derive_cand keeps a 16-B copy of every map point
// (fp32 x, y, z, label) and the scan evaluates THAT — fp32 distance, label class — against a
// threshold that no point able to win or tie can exceed (the fp32 error is bounded per query; see
// `set_thresholds`).  Only a candidate under the threshold is fetched as fp64 and goes through
// the reference's comparison.  With the previous answer as seed almost nothing passes, so a
// scanned point costs 16 bytes and ~11 fp32 instructions instead of 32 bytes and 22 fp64 ones,
// and the decision — every comparison that can change the result is the fp64 one — is the
// sequential scan's, index for index.
//
// Roofline: cache-gather bound integer/byte + fp64 compare work (~0.1 flop/B) — no MFMA (a 6x6
// outer product sum is not a dense contraction).  What matters here: one 32-B record per lane
// and step through a raw buffer resource (no address arithmetic), rows staged once in LDS with a
// conflict-free stride, a branch-light per-lane state machine with the next point's load in
// flight while the current one is evaluated, no device-scope atomics, and load balance by
// hardware dispatch of many small workgroups in XCD-aware stripes of the spatially sorted frame.
//
// Built with -ffp-contract=off: distances are the plain IEEE sequence
// SAGE_SQNORM3_*(dx*dx, dy*dy, dz*dz) (sageicp_types.h) the CPU evaluates, so the argmin is
// index-exact against the oracle.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <type_traits>
#include <vector>

#ifndef SAGE_ICP_STRIPE
#define SAGE_ICP_STRIPE 8      // workgroups of the sorted frame per XCD stripe
#endif
#ifndef SAGE_ICP_OCC
#define SAGE_ICP_OCC 6         // waves per SIMD k_icp's register allocation must allow at least (<= 80 registers); the
                               // variants that matter sit at 72 and run 7 (tools/resource_usage.sh)
#endif
// pairs of candidates a lane of k_loop keeps in flight while it scans: of full records / of compact ones
#ifndef SAGE_LOOP_DEPTH_FULL
#define SAGE_LOOP_DEPTH_FULL 2
#endif
#ifndef SAGE_LOOP_DEPTH_COMPACT
#define SAGE_LOOP_DEPTH_COMPACT 2
#endif
#define SAGE_LOOP_DEPTH_OF(filt) ((filt) ? SAGE_LOOP_DEPTH_COMPACT : SAGE_LOOP_DEPTH_FULL)

#include "kernels.h"
#include "se3_math.h"
#include "sageicp_types.h"
#include "probes.h"

namespace sageicp {


// ---------------------------------------------------------------------------- cross-lane helpers
// DPP moves run on the VALU (no LDS traffic, no scalar instructions); the patterns used are
// involutions (quad swaps, half-row and row mirrors), so each step is an exchange and every lane
// of a segment ends with the segment's result.  (bound_ctrl with no `old` operand: a lane without a
// source — only the row shifts of the block sums have such lanes, and nobody reads them — gets zero, and
// the move is ONE instruction per dword; with old = the value itself the compiler first copied the value
// into the destination: 4 instructions per fp64 exchange instead of 2, 64 of them in every pass's sums.)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // lane i <-> 7 - i   within each 8
constexpr int kDppMirror = 0x140;       // lane i <-> 15 - i  within each 16

// v_min_f64 without the quieting v_max_f64 x, x pairs the compiler puts around fmin(): one
// operand (the running best) is never NaN, and a NaN candidate leaves it unchanged.
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

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return static_cast<unsigned>(
        __builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, true));
}
template <int W>
__device__ __forceinline__ unsigned seg_or_u32(unsigned m) {   // over segments of W <= 16 lanes
    if (W >= 2) m |= dpp_u32<kDppXor1>(m);
    if (W >= 4) m |= dpp_u32<kDppXor2>(m);
    if (W >= 8) m |= dpp_u32<kDppHalfMirror>(m);
    if (W >= 16) m |= dpp_u32<kDppMirror>(m);
    return m;
}
template <int W>
__device__ __forceinline__ unsigned seg_add_u32(unsigned m) {
    if (W >= 2) m += dpp_u32<kDppXor1>(m);
    if (W >= 4) m += dpp_u32<kDppXor2>(m);
    if (W >= 8) m += dpp_u32<kDppHalfMirror>(m);
    if (W >= 16) m += dpp_u32<kDppMirror>(m);
    return m;
}
template <int W>
__device__ __forceinline__ double seg_min_f64(double m) {      // over segments of W <= 16 lanes
    if (W >= 2) m = min_f64(m, dpp_f64<kDppXor1>(m));
    if (W >= 4) m = min_f64(m, dpp_f64<kDppXor2>(m));
    if (W >= 8) m = min_f64(m, dpp_f64<kDppHalfMirror>(m));
    if (W >= 16) m = min_f64(m, dpp_f64<kDppMirror>(m));
    return m;
}
template <int W>
__device__ __forceinline__ unsigned seg_min_u32(unsigned m) {
    if (W >= 2) m = min_u32_xor1(m);
    if (W >= 4) m = min_u32_xor2(m);
    if (W >= 8) m = min_u32_half_mirror(m);
    if (W >= 16) m = min_u32_mirror(m);
    return m;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

typedef unsigned v4u __attribute__((ext_vector_type(4)));

// (what the workgroups of a launch share with each other and with the solving wave: agent-scope atomics only —
// the eight L2s are not coherent with each other)
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// One map point through the buffer path: `off` is the point's byte offset, the resource carries the
// 64-bit base, so a load costs no 64-bit address arithmetic.  (The point array stays under 4 GiB:
// 2^24 storage units of 128 B, capi.hip.)
__device__ __forceinline__ Point4 load_point(__amdgpu_buffer_rsrc_t pts, uint32_t off) {
    const v4u a = __builtin_amdgcn_raw_buffer_load_b128(pts, off, 0, 0);
    const v4u b = __builtin_amdgcn_raw_buffer_load_b128(pts, off + 16u, 0, 0);
    Point4 q;
    q.x = __hiloint2double(static_cast<int>(a.y), static_cast<int>(a.x));
    q.y = __hiloint2double(static_cast<int>(a.w), static_cast<int>(a.z));
    q.z = __hiloint2double(static_cast<int>(b.y), static_cast<int>(b.x));
    q.l = __hiloint2double(static_cast<int>(b.w), static_cast<int>(b.z));
    return q;
}

// One compact candidate record (16 B): fp32 x, y, z, label.  `off` is its byte offset; `imm` (a constant) travels in the
// instruction's scalar offset: a second record at a fixed distance from the first costs no address arithmetic.
__device__ __forceinline__ uint4 load_cand(__amdgpu_buffer_rsrc_t cands, uint32_t off, int imm = 0) {
    const v4u a = __builtin_amdgcn_raw_buffer_load_b128(cands, off, imm, 0);
    return make_uint4(a.x, a.y, a.z, a.w);
}

// ------------------------------------------------------------------------------- k_derive_cand
// The compact copy of the map: one thread per hash slot converts the points of its voxel.  A label
// that is not an integer of magnitude < 2^24 cannot be classified in fp32: it is stored as a NaN
// and raises bit 0 of *flags, which makes k_icp use the looser of its two thresholds everywhere.
__global__ __launch_bounds__(256) void k_derive_cand(const Slot *table, uint32_t nslots, const Point4 *pts,
                                                     uint4 *cand, uint64_t nslots_pts, uint32_t *flags) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nslots) return;
    const uint32_t w = table[i].blk;
    if (w == kEmptySlot || w == kTombstone) return;
    const uint32_t cnt = w & 255u;
    const uint64_t first = static_cast<uint64_t>(w >> 8) * kUnitPoints;      // the voxel's region
    bool inexact = false;
    for (uint32_t j = 0; j < cnt && first + j < nslots_pts; ++j) {
        const Point4 q = pts[first + j];
        float lf = static_cast<float>(q.l);
        if (!(static_cast<double>(lf) == q.l && q.l == trunc(q.l) && fabs(q.l) < 16777216.0)) {
            lf = __uint_as_float(0x7FC00000u);
            inexact = true;
        }
        cand[first + j] = make_uint4(__float_as_uint(static_cast<float>(q.x)), __float_as_uint(static_cast<float>(q.y)),
                                     __float_as_uint(static_cast<float>(q.z)), __float_as_uint(lf));
    }
    if (inexact) atomicOr(flags, 1u);
}
void launch_derive_cand(const Slot *table, uint32_t nslots, const Point4 *pts, uint4 *cand, uint64_t nslots_pts,
                        uint32_t *flags, hipStream_t s) {
    if (nslots) hipLaunchKernelGGL(k_derive_cand, dim3((nslots + 255u) / 256u), dim3(256), 0, s, table, nslots,
                                   pts, cand, nslots_pts, flags);
}

// --------------------------------------------------------------------------------- hash probing
// The GPU-resident open-addressed hash: linear probing, 16-B slots, load factor <= 0.25, one 16-B
// load per step.  Returns the slot's packed word (first unit of the voxel's points << 8) | count —
// what a neighbourhood row keeps of a found voxel — or kEmptySlot.
__device__ __forceinline__ uint32_t probe_resolve(const Slot *table, uint32_t mask, uint32_t sl,
                                                  int4 e, int vx, int vy, int vz) {
    for (;;) {
        if (static_cast<uint32_t>(e.w) == kEmptySlot) return kEmptySlot;
        if (e.x == vx && e.y == vy && e.z == vz) return static_cast<uint32_t>(e.w);
        sl = (sl + 1u) & mask;                    // tombstones (map_update.hip) never match
        e = reinterpret_cast<const int4 *>(table)[sl];
    }
}
__device__ __forceinline__ uint32_t probe_voxel(const Slot *table, uint32_t mask, int vx, int vy,
                                                int vz) {
    const uint32_t sl = voxel_hash(vx, vy, vz) & mask;
    const int4 e = reinterpret_cast<const int4 *>(table)[sl];
    return probe_resolve(table, mask, sl, e, vx, vy, vz);
}

// The query as searched: cumulative pose applied to the pristine frame point
// (Registration.cpp:103-111), home voxel by exact fp64 divide + truncation toward zero
// (VoxelHashMap.cpp:52-54).
struct Query {
    double x, y, z, l;
    int kx, ky, kz;
};
// QUAD: the lanes of a query come in aligned groups of four (k_icp with 4+ lanes per query) that
// all hold the same point: lane a < 3 of a quad then divides axis a only (an fp64 divide is ~15
// instructions) and the three voxel indices are handed round by quad broadcasts — the same
// divisions, a third of them per lane.
// The voxel index is trunc(fl(x / voxel_size)) — the reference's fp64 divide, to the bit.  An fp64 division is ~35
// instructions; x * fl(1 / voxel_size) is one, and differs from the quotient by a few ulps at most: wherever it lies
// farther than 1e-9 (1 + |q|) from an integer its truncation IS the quotient's, and only a wave that holds a lane
// closer than that to a cell face (one pass in thousands) pays for the division (`inv_vs` = 0: always divide).
__device__ __forceinline__ int voxel_index(double x, double voxel_size, double inv_vs) {
    const double q = x * inv_vs;
    const bool unsure = !(fabs(q - __builtin_rint(q)) > 1e-9 * (1.0 + fabs(q))) || inv_vs == 0.0;
    int k = static_cast<int>(q);
    if (__ballot(unsure)) {
        asm volatile("" ::: "memory");         // (keeps this a branch: if-converted, the division ran in every pass)
        k = static_cast<int>(x / voxel_size);
    }
    return k;
}
template <bool QUAD = false>
__device__ __forceinline__ Query make_query(const Point4 &f, const double *R, const double *t, int apply_pose,
                                            double voxel_size, double inv_vs = 0.0) {
    Query q;
    q.x = f.x; q.y = f.y; q.z = f.z; q.l = f.l;
    if (apply_pose) {
        q.x = R[0] * f.x + R[1] * f.y + R[2] * f.z + t[0];
        q.y = R[3] * f.x + R[4] * f.y + R[5] * f.z + t[1];
        q.z = R[6] * f.x + R[7] * f.y + R[8] * f.z + t[2];
    }
    if constexpr (QUAD) {
        const unsigned a = threadIdx.x & 3u;
        const double num = a == 0u ? q.x : (a == 1u ? q.y : q.z);
        const unsigned k = static_cast<unsigned>(voxel_index(num, voxel_size, inv_vs));
        q.kx = static_cast<int>(dpp_u32<0x00>(k));       // quad_perm [0,0,0,0]
        q.ky = static_cast<int>(dpp_u32<0x55>(k));       // quad_perm [1,1,1,1]
        q.kz = static_cast<int>(dpp_u32<0xAA>(k));       // quad_perm [2,2,2,2]
    } else {
        q.kx = voxel_index(q.x, voxel_size, inv_vs);
        q.ky = voxel_index(q.y, voxel_size, inv_vs);
        q.kz = voxel_index(q.z, voxel_size, inv_vs);
    }
    return q;
}

// ------------------------------------------------------------------------------------ k_rows
// Builds the neighbourhood row of every query: 32 lanes per query, lane v < 27 probes voxel v.
__global__ __launch_bounds__(256) void k_rows(IcpParams P) {
    const int lane = static_cast<int>(threadIdx.x & 63u);
    const unsigned v = threadIdx.x & 31u;
    const unsigned q = blockIdx.x * 8u + (threadIdx.x >> 5);
    const bool valid = q < static_cast<unsigned>(P.n);
    const Point4 f = P.frame[valid ? q : 0u];
    const Query s = make_query(f, P.st->R, P.st->T + 4, P.apply_pose, P.voxel_size);
    uint32_t w = kEmptySlot;
    if (valid && v < 27u)
        w = probe_voxel(P.table, P.mask, s.kx + static_cast<int>(v / 9u) - 1,
                        s.ky + static_cast<int>((v / 3u) % 3u) - 1, s.kz + static_cast<int>(v % 3u) - 1);
    const unsigned c = (w == kEmptySlot) ? 0u : (w & 255u);
    const unsigned long long b = __ballot(c != 0u);
    const unsigned occ = static_cast<unsigned>(b >> (lane & 32));
    unsigned cq = c;                          // sum over the 32 lanes of the query
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) cq += __shfl_xor(cq, d, 32);
    if (!valid) return;
    uint32_t out = w;
    if (v == kRowCq) out = cq;
    if (v == kRowKey) out = static_cast<uint32_t>(s.kx);
    if (v == kRowKey + 1) out = static_cast<uint32_t>(s.ky);
    if (v == kRowKey + 2) out = static_cast<uint32_t>(s.kz);
    if (v == kRowOcc) out = occ;
    P.rows[static_cast<size_t>(q) * kRowWords + v] = out;
}

// ------------------------------------------------------------------------------------- k_icp

// per-workgroup LDS header of k_icp (words): arrival counter | the workgroup's fixed-point accumulators
// (kWgAccWords 64-bit words)
constexpr unsigned kWgAccWords = 52;         // (16 sums + pair count) x 3 digits = 51 | [51] overflow flag
constexpr unsigned kWgArrive = 0, kWgGo = 1, kWgAcc = 2;      // kWgGo: chained launches: 0 go on, else leave
constexpr unsigned kWgPose = kWgAcc + 2u * kWgAccWords;       // chained launches: R[9], t[3] of this iteration (16-B aligned)
constexpr unsigned kWgHeaderWords = (kWgPose + 24u + 15u) & ~15u;
__host__ __device__ constexpr unsigned icp_wave_words(int lw) {
    // the rows of the wave's queries (kRowLdsStride words each); reused by the epilogue's
    // transposed reduction, 16 components x (queries + 2) fp64
    return static_cast<unsigned>(kRowLdsStride * (64 >> lw) + 64);
}

// ---- the Gauss-Newton sums as exact fixed-point numbers ---------------------------------------------
// The unit that is rounded is a BLOCK of four consecutive queries of the (sorted) frame: the 16 pair terms
// of its queries are added as (t0 + t1) + (t2 + t3) in fp64, and that block sum is split exactly into three
// signed digits of 40 bits (weights 2^0, 2^-40, 2^-80; what lies below 2^-80 is dropped).  From there on
// everything is integer addition — over the blocks of a wave (DPP), the waves and groups of a workgroup
// (LDS atomics), the workgroups (global atomics into the shared accumulators, kernels.h) — which is
// associative: the accumulated bits depend on the order of the frame and on NOTHING else.  Not on the lanes
// per query (round 4 rounded once per wave: a frame registered with 4 and with 8 lanes per query differed
// in the last bits), not on how queries are cut into waves, groups and workgroups, not on which loop ran,
// not on the order of arrival.  (Blocks never straddle waves: a wave holds 4, 8, 16, 32 or 64 queries.)
//
// to_digits: v = a + b 2^-40 + c 2^-80 exactly (three integers of at most 40 bits and a sign, each exactly
// representable), each as int64 through the 2^52 + 2^51 trick (both numbers lie in [2^52, 2^53): their
// bit patterns differ by exactly the integer).  `ok` is cleared when |a| leaves the range the accumulators
// have room for: the overflow flag then sends the frame through the fp64 partials (capi.hip).
__device__ __forceinline__ void to_digits(double v, double limit, long long &d0, long long &d1, long long &d2, bool &ok) {
    const double a = __builtin_rint(v);
    const double r1 = (v - a) * 1099511627776.0;             // 2^40, exact
    const double b = __builtin_rint(r1);
    const double c2 = __builtin_rint((r1 - b) * 1099511627776.0);
    ok &= fabs(a) < limit;
    d0 = __double_as_longlong(a + 6755399441055744.0) - 0x4338000000000000ll;
    d1 = __double_as_longlong(b + 6755399441055744.0) - 0x4338000000000000ll;
    d2 = __double_as_longlong(c2 + 6755399441055744.0) - 0x4338000000000000ll;
}
// |a| of one block stays below a limit.  k_icp + k_fin: IcpParams::digit_limit — k_fin adds the 32 copies of the
// accumulators in plain 64-bit integers, so what has to stay inside 63 bits is the sum over ALL the blocks of the
// frame: the host passes 2^62 / blocks rounded down to a power of two, at most 2^46 (capi_run.hip, icp_params; a
// frame of 500k points: 2^45 — coordinates of 3 10^6 m; ADVICE r05: a fixed 2^46 let a frame of 1M+ points at UTM
// coordinates wrap silently).  k_loop: 2^40, its words also carry a count in their low byte (coordinates of 10^5 m:
// 4 x (10^5)^2 = 4 10^10 < 2^40 = 1.1 10^12; plan_loop checks the number of blocks per copy).  The lower digits
// are below 2^39 in magnitude: 2^24 - 1 blocks at most (kMaxQueries).
constexpr double kDigitLimitCounted = 1099511627776.0;

template <int CTRL>
__device__ __forceinline__ long long dpp_i64(long long v) {
    int lo = static_cast<int>(v), hi = static_cast<int>(v >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo);
}
// lane i <- lane i + N (the lanes this is used on — the first lanes of queries — always have their partner
// inside the wave; shifts below 16 stay inside a row of 16 lanes: DPP row_shl, no LDS traffic)
template <int N>
__device__ __forceinline__ double lane_shl_f64(double v) {
    if constexpr (N < 16) return dpp_f64<0x100 + N>(v);
    else return __shfl_down(v, N, 64);
}

// A wave's 16 pair terms per query (t[], on the first lane of every query; zeros for a query without a pair)
// -> block sums -> digits -> added into the workgroup's accumulators `wgacc` (LDS, kWgAccWords 64-bit words).
// `red` = LDS scratch of this wave, 16 fp64 per block.
template <int LW>
__device__ __forceinline__ void wave_terms_to_wgacc(const double (&t)[kCount], unsigned pairs, int lane, double *red,
                                                    unsigned long long *wgacc, double limit, double scale) {
    constexpr int W = 1 << LW, QW = 64 >> LW, NBLK = QW / 4;
    // A. block sums, (t0 + t1) + (t2 + t3), on the first lane of every block
    double y[kCount];
#pragma unroll
    for (int c = 0; c < kCount; ++c) {
        const double x = t[c] + lane_shl_f64<W>(t[c]);
        y[c] = x + lane_shl_f64<2 * W>(x);
    }
    // B. four blocks at a time (512 B of scratch whatever the lanes per query): the blocks' first lanes park
    // their 16 sums, C. lane (c, j) = (lane >> 2, lane & 3) converts component c of block j
    const int c = lane >> 2, j = lane & 3;
    long long d0 = 0, d1 = 0, d2 = 0;
    bool ok = true;
    const int blk = lane / (4 * W);
    const bool first = (lane & (4 * W - 1)) == 0;
#pragma unroll
    for (int r = 0; r < NBLK; r += 4) {
        if (first && blk >= r && blk < r + 4) {
            double *dst = red + (blk - r) * kCount;
#pragma unroll
            for (int cc = 0; cc < kCount; ++cc) dst[cc] = y[cc];
        }
        // (values pass from lane to lane through LDS here: the hardware serves a wave's LDS instructions in
        // order, but the COMPILER has to be told that the loads below see other lanes' stores)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const double v = r + j < NBLK ? red[j * kCount + c] : 0.0;
        long long e0, e1, e2;
        to_digits(v * scale, limit, e0, e1, e2, ok);       // (scale: a power of two, 1 normally)
        d0 += e0; d1 += e1; d2 += e2;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // D. every lane adds its digits into the workgroup's accumulators (integers: any order; three
    // fire-and-forget LDS atomics — reducing the four lanes of a component first costs 24 more instructions)
    (void)__hip_atomic_fetch_add(wgacc + 3 * c, static_cast<unsigned long long>(d0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    (void)__hip_atomic_fetch_add(wgacc + 3 * c + 1, static_cast<unsigned long long>(d1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    (void)__hip_atomic_fetch_add(wgacc + 3 * c + 2, static_cast<unsigned long long>(d2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!ok) (void)__hip_atomic_fetch_or(wgacc + kWgAccWords - 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == 0)
        (void)__hip_atomic_fetch_add(wgacc + 3 * kCount, static_cast<unsigned long long>(pairs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The workgroup's accumulators (LDS) -> the shared ones: lane l sends word l (digit l % 3 of value l / 3) as ONE
// fire-and-forget 64-bit integer atomic into the copy `dst`, and clears the word for the next iteration.
// COUNTED (k_loop): every word also counts its contributions — a workgroup adds (digit << 8) + 1, the low
// byte of a word says how many workgroups are in it — so that whoever reads the accumulators inside the
// launch knows, word by word, when they are complete, and the workgroup neither waits for its atomics to
// be acknowledged nor keeps a separate arrival counter.  (The form that did both — s_waitcnt vmcnt(0), then
// one of eight counters — ran an iteration in the same time, profiles/r04/loop_times_counted_words.txt against
// loop_times_xcd_stripes_rowshift.txt: this one is kept for having one protocol less and no ordering
// assumption at all.)
template <bool COUNTED = false>
__device__ __forceinline__ void wgacc_flush(unsigned long long *wgacc, long long *dst, long long *overflow) {
    const int lane = static_cast<int>(threadIdx.x & 63u);
    const int l = min(lane, static_cast<int>(kWgAccWords) - 1);
    const long long x = static_cast<long long>(wgacc[l]);
    const bool ok = wgacc[kWgAccWords - 1] == 0ull;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < static_cast<int>(kWgAccWords)) wgacc[lane] = 0ull;
    if constexpr (COUNTED) {
        // (the overflow travels IN the counted words: word 3 kAccValues of the copy counts its workgroups like the
        // others and carries, as its "digit", how many of them overflowed — whoever finds the counts complete has
        // the flags of exactly those workgroups; a flag at another address could land after the counts)
        (void)overflow;
        if (lane <= 3 * kAccValues)
            (void)__hip_atomic_fetch_add(dst + lane, lane == 3 * kAccValues ? (ok ? 1ll : 257ll) : (ok ? x * 256 + 1 : 1ll),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (lane < 3 * kAccValues) {
        {
            if (ok) (void)__hip_atomic_fetch_add(dst + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else (void)__hip_atomic_fetch_or(overflow, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- k_loop (the whole ICP loop in one launch, below) shares the body --------------------------
// A GROUP = the 64 >> LW consecutive queries one wave of k_icp would hold.  A workgroup of k_loop owns
// a few groups for the whole call and keeps per query, in LDS: the neighbourhood row (kRowLdsStride
// words) and a state record of kLoopStateWords words —
//   [0..7]   f   the pristine frame point (fp64 x, y, z, label)
//   [8..15]  pp  the previous answer's fp64 record (the seed of the next search and, while the
//                answer stays, the target of the pair: no load)
//   [16,17]  the previous answer {key = (voxel << 8) | slot, byte offset of its record}
//   [18..21] the home voxel the row was built for | the row's occupancy mask
// One copy per QUERY (its lanes read the same address: a broadcast), where the first k_loop kept a
// copy per LANE in 22 registers: the search now compiles at k_icp's budget (7 waves per SIMD) and a
// wave is no longer tied to one group — the waves of a workgroup take its groups from an LDS counter.
struct LoopGroup {
    uint32_t *rows;            // LDS: the rows of ALL the workgroup's queries, [slot][kRowLdsStride]
    uint32_t *state;           // LDS: ... and their state records, [slot][kLoopStateWords]
    const uint32_t *perm;      // LDS: the workgroup's blocks of four queries in the order of this iteration (heaviest first)
    uint32_t *work;            // LDS: per block, the most points one of its queries was handed this iteration
    unsigned unit;             // which (64 >> LW) / 4 blocks of `perm` this pass takes
    double *red;               // LDS: the running wave's scratch for the epilogue's transposed block sums
    unsigned long long *wgacc; // LDS: the workgroup's fixed-point accumulators (wave_terms_to_wgacc)
    unsigned q_first;          // the workgroup's first query (slot 0; a multiple of four)
    unsigned slot;             // this pass's slot in the per-wave counters (IcpParams::counters)
#ifdef SAGE_LOOP_TIMING
    unsigned long long ph[8], tprev;           // probe builds: cycles per phase of the body, summed over the iterations
#endif
};
constexpr int kLoopMaxWaves = kLoopMaxWavesHost;   // waves per workgroup of k_loop (<= 512 threads)
constexpr unsigned kLoopStripe = kLoopStripeHost;  // workgroups of k_loop per XCD stripe (its grid: a multiple of 32)
constexpr int kNoVoxel = 0x7FFFFFFF;        // a home voxel no point has (|index| < 2^20): row not built yet
constexpr unsigned kStPrev = 16, kStKey = 18;
// LDS of a k_loop workgroup (words): header { arrival counter | next group | the pose of this iteration
// (R[9], t[3]) | done | the workgroup's fixed-point accumulators } | the groups { rows, state } |
// one scratch for the transposed block sums per wave
constexpr unsigned kLpArrive = 0, kLpNext = 1, kLpDone = 2;
constexpr unsigned kLpPose = 4;                                    // 12 doubles, 16-B aligned
constexpr unsigned kLpDbg = kLpPose + 24u;                         // probe builds: max points of a query | stale queries | points
constexpr unsigned kLpFirst = kLpDbg + 4u;                         // per wave: the unit it takes first in every iteration (LoopParams::deal)
constexpr unsigned kLpAcc = kLpFirst + 8u;                         // kWgAccWords 64-bit words
constexpr unsigned kLpHeaderWords = (kLpAcc + 2u * kWgAccWords + 15u) & ~15u;
__host__ __device__ constexpr unsigned loop_group_words(int lw) {
    return static_cast<unsigned>((kRowLdsStride + kLoopStateWords) * (64 >> lw));
}
__host__ __device__ constexpr unsigned loop_red_words() { return 2u * kCount * 4u; }    // 16 fp64 for each of four blocks of queries
__host__ __device__ constexpr unsigned loop_perm_words(unsigned nblk) { return 2u * ((nblk + 7u) & ~7u); }   // perm | work, 32-B aligned

// PERSIST (k_loop): the body runs on group `G` — rows and per-query state in LDS — with the pose from
// `pose` (LDS: R[9], t[3]); nothing is read from or written to the global rows / nn_prev arrays;
// the body ends with the group's sums parked at G->ws.
#ifndef SAGE_LOOP_FLAT_MINW
#define SAGE_LOOP_FLAT_MINW 8      // k_loop: flat-order scan from this many lanes per query (icp_body)
#endif
template <int LW, bool FUSED, bool FILT, bool PERSIST = false, bool FLATQ = false>
__device__ __forceinline__ void icp_body(const IcpParams &P, uint32_t *smem, LoopGroup *G = nullptr,
                                         const double *pose = nullptr);

// a pair of scanned points in flight: compact records (FILT) or full ones
struct PairCompact {
    uint4 a, b;
    unsigned ka, oa;                // key / compact offset of a; b: key + W, offset + W records
    bool ha, hb;
};
struct PairFull {
    Point4 a, b;
    unsigned ka;
    bool ha, hb;
};
// ... of a scan in flat order (below): b may lie in another voxel than a
struct PairCompactFlat {
    uint4 a, b;
    unsigned ka, oa, kb, ob;
    bool ha, hb;
};
struct PairFullFlat {
    Point4 a, b;
    unsigned ka, kb;
    bool ha, hb;
};

// Chained launches (IcpParams::chain): wave 0 of a workgroup of the launch of iteration `it` waits for the pose the solving
// wave publishes after iteration it - 1 (25 self-tagged granules, tag = it; kernels.h, LoopShared) and hands it to the
// workgroup through LDS.  smem[kWgGo] != 0: the workgroup leaves — the loop ended before this iteration (the done granule
// carries an older tag, or this tag with its done word set), or a wait timed out somewhere.
__device__ __forceinline__ void chain_wait_pose(const IcpParams &P, uint32_t *smem, int lane) {
    LoopShared *sh = P.chain;
    const unsigned long long tag = static_cast<unsigned long long>(P.chain_iter);
    const unsigned long long *src = lane < kLoopPoseGranules ? &sh->pose[lane] : &sh->abort_word[0];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned go = 0u;
    unsigned long long g;
    for (;;) {
        g = ld_agent(src);
        const unsigned long long g24 = static_cast<unsigned long long>(static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(g), 24))) |
                                       (static_cast<unsigned long long>(static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(g >> 32), 24))) << 32);
        const bool ok = lane >= kLoopPoseGranules || (g >> 32) == tag;
        if (__all(ok)) {
            if (static_cast<uint32_t>(g24) != 0u) go = 1u;             // the pose after the LAST iteration: nothing left to do
            break;
        }
        if ((g24 >> 32) < tag && static_cast<uint32_t>(g24) != 0u) {   // the loop ended before this iteration
            go = 1u;
            break;
        }
        const bool late = __builtin_amdgcn_s_memrealtime() - t0 > P.chain_timeout;
        if (__any(lane == kLoopPoseGranules && g != 0ull) || late) {
            if (lane == 0 && late) {
                st_agent(&sh->abort_word[0], 1ull);
                const_cast<IcpState *>(P.st)->loop_aborted = 1;
            }
            go = 2u;
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    if (!go && lane < 24) smem[kWgPose + static_cast<unsigned>(lane)] = static_cast<uint32_t>(g);
    if (lane == 0) smem[kWgGo] = go;
}

template <int LW, bool FUSED, bool FILT, bool FLATQ>
__global__ __launch_bounds__(64 * kIcpWavesPerBlock) __attribute__((amdgpu_waves_per_eu(SAGE_ICP_OCC, 8)))
void k_icp(IcpParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    if (P.chain) {
        // chained launches: whether the loop has ended travels with the pose (chain_wait_pose); launch 0 tells the solving
        // wave that the shared block has been zeroed and the loop starts — or that it never will (a non-finite frame point)
        if (P.chain_iter == 0) {
            const bool bad = P.st->done != 0;
            if (blockIdx.x == 0 && threadIdx.x == 0) st_agent(&P.chain->go[0], P.chain_epoch | (bad ? 0x8000000000000000ull : 0ull));
            if (bad) return;
        }
    } else if (P.check_done && P.st->done) {
        return;
    }
    icp_body<LW, FUSED, FILT, false, FLATQ>(P, smem);
    PROBE_DELAY_REPEAT((icp_body<LW, FUSED, FILT, false, FLATQ>(P, smem)));
}

template <int LW, bool FUSED, bool FILT, bool PERSIST, bool FLATQ>
__device__ __forceinline__ void icp_body(const IcpParams &P, uint32_t *smem, LoopGroup *G, const double *pose) {
    static_assert(!PERSIST || FUSED, "the persistent loop always accumulates");
    constexpr int W = 1 << LW;                 // lanes per query
    constexpr int QW = 64 >> LW;               // queries per wave
    constexpr int SH = 5;                      // points are addressed by byte offset
    PROBE_NN_BEGIN;
    PROBE_DELAY_BEGIN;
    int lane;
    if constexpr (PERSIST) {
        // (k_loop: re-derived in every pass — what the compiler knows to be invariant across the iteration
        // loop it hoists out of it and keeps, with everything computed from it, in registers the scan needs)
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        lane = static_cast<int>(l);
    } else {
        lane = static_cast<int>(threadIdx.x & 63u);
    }
    const int wv = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    if (FUSED && !PERSIST) {
        if (threadIdx.x == 0) {
            smem[kWgArrive] = 0u;
            smem[kWgGo] = 0u;
        }
        if (threadIdx.x < 2u * kWgAccWords) smem[kWgAcc + threadIdx.x] = 0u;
        // the pose of this iteration, for the workgroup (a chained launch after the first gets it from the solving wave, below)
        if (threadIdx.x < 12u && !(P.chain && P.chain_iter > 0))
            reinterpret_cast<double *>(smem + kWgPose)[threadIdx.x] = threadIdx.x < 9u ? P.st->R[threadIdx.x] : P.st->T[4u + threadIdx.x - 9u];
        __syncthreads();
    }
    uint32_t *wl;
    if constexpr (PERSIST) wl = G->rows;
    else wl = smem + kWgHeaderWords + static_cast<unsigned>(wv) * icp_wave_words(LW);

    // Workgroup b is dispatched to XCD b % 8 (observed; speed only): XCD x serves the stripes
    // x, x+8, x+16, ... of kStripe consecutive workgroups' worth of the spatially sorted frame,
    // so each private L2 sees a few compact regions of the map and every XCD gets the same mix
    // of dense and sparse regions.  (k_loop: stripes of kLoopStripe workgroups; one contiguous eighth
    // of the frame per XCD left the XCDs with 57k to 97k points to look at per iteration on a c2 shard,
    // and the iteration ends with the slowest, profiles/r04/loop_times.txt.)
    constexpr unsigned kStripe = SAGE_ICP_STRIPE;
    unsigned wave_id, stripe_id = 0u;                                               // wave-uniform
    if constexpr (PERSIST) {
        wave_id = G->slot;                     // (k_loop maps its workgroups to groups of queries itself)
    } else {
        const unsigned xcd = blockIdx.x & 7u, jb = blockIdx.x >> 3;
        unsigned stripe = (jb / kStripe) * 8u + xcd;            // the stripe dispatched at this position ...
        if (P.stripe_order) {
            // ... in the order of the work an earlier iteration measured, heaviest first (kernels.h): what runs last is
            // light.  Within a SIMD the waves of the heavier stripes go first as well (s_setprio by the quarter of the
            // order this stripe lies in: the long chains run while there is other work to cover their stalls).
            const unsigned nstripes = gridDim.x / kStripe;
            switch (stripe * 4u / nstripes) {
                case 0: __builtin_amdgcn_s_setprio(3); break;
                case 1: __builtin_amdgcn_s_setprio(2); break;
                case 2: __builtin_amdgcn_s_setprio(1); break;
                default: break;
            }
            stripe = P.stripe_order[stripe];
        }
        stripe_id = stripe;
        const unsigned wg = stripe * kStripe + (jb % kStripe);
        wave_id = wg * static_cast<unsigned>(kIcpWavesPerBlock) + static_cast<unsigned>(wv);
    }

    const int qw = lane >> LW;                 // this lane's query within the wave
    const unsigned ci = static_cast<unsigned>(lane) & (W - 1u);
    // k_loop: a pass takes QW / 4 BLOCKS of four consecutive queries — not necessarily neighbours: the workgroup's
    // blocks are re-ordered every iteration by the work they were (a wave's pass lasts as long as its heaviest
    // query: blocks of like work share a wave).  The four queries of a block stay together, in order, on one
    // aligned group of lanes, which is all the exact block sums ask for (wave_terms_to_wgacc).
    unsigned q, qslot = 0u, bslot = 0u;
    if constexpr (PERSIST) {
        bslot = G->perm[G->unit * (QW / 4) + static_cast<unsigned>(qw >> 2)];
        qslot = bslot * 4u + (static_cast<unsigned>(qw) & 3u);
        q = G->q_first + qslot;
    } else {
        q = wave_id * QW + static_cast<unsigned>(qw);
    }
    const bool valid = q < static_cast<unsigned>(P.n);
    const unsigned qc = valid ? q : 0u;        // keeps the loads of idle lanes legal
    uint32_t *lrow = wl + (PERSIST ? qslot : static_cast<unsigned>(qw)) * kRowLdsStride;
    const uint32_t *grow = P.rows + static_cast<size_t>(qc) * kRowWords;

    // raw buffer resource over the point array (bounds-checked, 32-bit byte offsets)
    const __amdgpu_buffer_rsrc_t pts = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<Point4 *>(P.pts), 0, static_cast<int>(P.pts_bytes), 0x00020000);

    const __amdgpu_buffer_rsrc_t cands = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(P.cand), 0, static_cast<int>(P.cand_bytes), 0x00020000);
    constexpr int SHC = FILT ? 4 : 5;          // the scan's records (compact or full), by byte offset

    // ---- prologue: the query, its home voxel, its neighbourhood row ------------------------------
    // Everything the prologue needs is requested at once (one memory round trip): the row key, the
    // frame point, the previous iteration's record and — speculatively, before the key has been
    // checked — this lane's share of the cached row (words 0..27 in seven 16-B pieces).
    constexpr int NP = (7 + W - 1) / W;        // pieces per lane
    uint4 rk;                                  // key x, y, z | occupancy
    uint2 prev = make_uint2(0xFFFFFFFFu, 0u);     // the previous iteration's record of this query
    Point4 f;
    // (named registers, not an array: the compiler leaves a uint4 array in scratch memory)
    uint4 pc0 = make_uint4(0u, 0u, 0u, 0u), pc1 = pc0, pc2 = pc0, pc3 = pc0, pc4 = pc0, pc5 = pc0, pc6 = pc0;
    uint32_t *lst = nullptr;                   // k_loop: this query's state record (LDS)
    if constexpr (PERSIST) {
        // k_loop: everything is already here, in LDS — the state record and the row
        lst = G->state + qslot * kLoopStateWords;
        const uint4 fa = *reinterpret_cast<const uint4 *>(lst), fb = *reinterpret_cast<const uint4 *>(lst + 4);
        const uint4 pk = *reinterpret_cast<const uint4 *>(lst + kStPrev);
        const uint2 ko = *reinterpret_cast<const uint2 *>(lst + kStPrev + 4);
        f.x = __hiloint2double(static_cast<int>(fa.y), static_cast<int>(fa.x));
        f.y = __hiloint2double(static_cast<int>(fa.w), static_cast<int>(fa.z));
        f.z = __hiloint2double(static_cast<int>(fb.y), static_cast<int>(fb.x));
        f.l = __hiloint2double(static_cast<int>(fb.w), static_cast<int>(fb.z));
        prev = make_uint2(pk.x, pk.y);
        rk = make_uint4(pk.z, pk.w, ko.x, ko.y);
    } else {
        rk = *reinterpret_cast<const uint4 *>(grow + kRowKey);
        if (FUSED) prev = P.nn_prev[qc];
        f = P.frame[qc];
        auto row_piece = [&](int k) {
            const unsigned p = min(ci + static_cast<unsigned>(W * k), 6u);
            return *reinterpret_cast<const uint4 *>(grow + 4u * p);
        };
        pc0 = row_piece(0); pc1 = pc0; pc2 = pc0; pc3 = pc0; pc4 = pc0; pc5 = pc0; pc6 = pc0;
        if (NP > 1) pc1 = row_piece(1);
        if (NP > 2) pc2 = row_piece(2);
        if (NP > 3) pc3 = row_piece(3);
        if (NP > 4) pc4 = row_piece(4);
        if (NP > 5) pc5 = row_piece(5);
        if (NP > 6) pc6 = row_piece(6);
    }
    PROBE_DELAY_WAIT(P);
    if constexpr (FUSED && !PERSIST) {
        if (P.chain && P.chain_iter > 0) {
            // (the loads above are in flight while wave 0 waits for the pose: the launch started under the solve)
            if (wv == 0) chain_wait_pose(P, smem, lane);
            __syncthreads();
            if (smem[kWgGo]) return;
        }
    }
    const Query s = [&]() {
        if constexpr (PERSIST) {
            return make_query<(W >= 4)>(f, pose, pose + 9, 1, P.voxel_size, P.inv_voxel_size);
        } else if constexpr (FUSED) {
            const double *lpose = reinterpret_cast<const double *>(smem + kWgPose);       // the workgroup's copy
            return make_query<(W >= 4)>(f, lpose, lpose + 9, P.apply_pose, P.voxel_size, P.inv_voxel_size);
        } else {
            return make_query<(W >= 4)>(f, P.st->R, P.st->T + 4, P.apply_pose, P.voxel_size, P.inv_voxel_size);
        }
    }();
    const bool stale = valid && (static_cast<uint32_t>(s.kx) != rk.x || static_cast<uint32_t>(s.ky) != rk.y ||
                                 static_cast<uint32_t>(s.kz) != rk.z);
    unsigned occ = rk.w;
    NN_T(0);
    LP_T(0);
    if constexpr (!PERSIST) {
        // stage the row in LDS (a stale one is overwritten below)
        auto stage = [&](int k, const uint4 &v) {
            const unsigned p = ci + static_cast<unsigned>(W * k);
            if (p < 7u) *reinterpret_cast<uint4 *>(lrow + 4u * p) = v;
        };
        stage(0, pc0);
        if (NP > 1) stage(1, pc1);
        if (NP > 2) stage(2, pc2);
        if (NP > 3) stage(3, pc3);
        if (NP > 4) stage(4, pc4);
        if (NP > 5) stage(5, pc5);
        if (NP > 6) stage(6, pc6);
    }
    if (__ballot(stale)) {
        // A query crossed a voxel face since its row was built (a few % of the queries per iteration
        // at the start of a cold registration, almost none near convergence; every query in the first
        // pass of k_loop): its lanes rebuild the row in LDS (and in the cache).  With eight or more
        // lanes per query (k_loop: four) a step into a NEIGHBOURING voxel keeps what the two neighbourhoods share —
        // 18 of the 27 voxels after a step through a face, 12 through an edge, 8 through a corner: the
        // words move inside the row, only the new layer is probed (one batch of loads instead of two or
        // three), and the previous answer, if it lies in the shared part, stays the seed under its new
        // enumeration key.  Otherwise all 27 voxels are probed, up to three / four in flight per lane.
        if (stale) {
            unsigned o = 0u, cq = 0u;
            constexpr int NV = (27 + W - 1) / W;          // voxels per lane
            constexpr bool kShift = NV <= (PERSIST ? 7 : 4);     // (k_icp with four lanes per query sits on its register edge)
            auto tally = [&](int v, uint32_t w) {
                const unsigned c = (w == kEmptySlot) ? 0u : (w & 255u);
                lrow[v] = w;
                if constexpr (!PERSIST) P.rows[static_cast<size_t>(q) * kRowWords + static_cast<unsigned>(v)] = w;
                o |= (c != 0u ? 1u : 0u) << v;
                cq += c;
            };
            // one probe: first slot load issued by `start`, resolved (and the row word stored) by `finish`
            auto start = [&](int v, uint32_t &sl, int4 &e) {
                const int vc = v < 27 ? v : 26;
                sl = voxel_hash(s.kx + vc / 9 - 1, s.ky + (vc / 3) % 3 - 1, s.kz + vc % 3 - 1) & P.mask;
                e = reinterpret_cast<const int4 *>(P.table)[sl];
            };
            auto finish = [&](int v, uint32_t sl, const int4 &e) {
                if (v >= 27) return;
                tally(v, probe_resolve(P.table, P.mask, sl, e, s.kx + v / 9 - 1, s.ky + (v / 3) % 3 - 1,
                                       s.kz + v % 3 - 1));
            };
            if constexpr (kShift) {
                // (unsigned differences: a row not built yet carries kNoVoxel = 0x7FFFFFFF, and a signed difference from
                // a negative index would overflow)
                const unsigned dxu = static_cast<unsigned>(s.kx) - rk.x, dyu = static_cast<unsigned>(s.ky) - rk.y,
                               dzu = static_cast<unsigned>(s.kz) - rk.z;
                const bool nearv = dxu + 1u <= 2u && dyu + 1u <= 2u && dzu + 1u <= 2u;
                const int dx = static_cast<int>(dxu), dy = static_cast<int>(dyu), dz = static_cast<int>(dzu);
                // the old words of this lane's voxels (LDS is in order within a wave: every read here
                // precedes the writes below, also those of the query's other lanes)
                uint32_t ow[NV];
                bool reuse[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int v = static_cast<int>(ci) + W * j;
                    const unsigned a = static_cast<unsigned>(v / 9) + dxu, b = static_cast<unsigned>((v / 3) % 3) + dyu,
                                   c = static_cast<unsigned>(v % 3) + dzu;
                    reuse[j] = nearv && v < 27 && a <= 2u && b <= 2u && c <= 2u;
                    ow[j] = lrow[reuse[j] ? a * 9u + b * 3u + c : 0u];
                }
                uint32_t sl[NV];
                int4 e[NV];
                constexpr int NB = PERSIST ? 4 : 3;       // probes in flight per lane (k_icp: its register budget)
#pragma unroll
                for (int j0 = 0; j0 < NV; j0 += NB) {
#pragma unroll
                    for (int j = j0; j < j0 + NB && j < NV; ++j) {
                        sl[j] = 0u;
                        e[j] = make_int4(0, 0, 0, 0);
                        if (!reuse[j]) start(static_cast<int>(ci) + W * j, sl[j], e[j]);
                    }
#pragma unroll
                    for (int j = j0; j < j0 + NB && j < NV; ++j) {
                        const int v = static_cast<int>(ci) + W * j;
                        if (reuse[j]) tally(v, ow[j]);
                        else finish(v, sl[j], e[j]);
                    }
                }
                // the previous answer under the new enumeration, if its voxel is still one of the 27
                if (nearv && prev.x != 0xFFFFFFFFu) {
                    const int vo = static_cast<int>(prev.x >> 8);
                    const int a = vo / 9 - dx, b = (vo / 3) % 3 - dy, c = vo % 3 - dz;
                    const bool in = static_cast<unsigned>(a) <= 2u && static_cast<unsigned>(b) <= 2u &&
                                    static_cast<unsigned>(c) <= 2u;
                    prev.x = in ? (static_cast<unsigned>(a * 9 + b * 3 + c) << 8) | (prev.x & 255u) : 0xFFFFFFFFu;
                } else {
                    prev.x = 0xFFFFFFFFu;
                }
            } else {
                prev.x = 0xFFFFFFFFu;          // (the key's meaning went with the old row)
#pragma unroll 1
                for (int k0 = 0; k0 < NV; k0 += 3) {
                    const int v0 = static_cast<int>(ci) + W * k0;
                    const int v1 = k0 + 1 < NV ? v0 + W : 27, v2 = k0 + 2 < NV ? v0 + 2 * W : 27;
                    uint32_t s0, s1, s2;
                    int4 e0, e1, e2;
                    start(v0, s0, e0);
                    start(v1, s1, e1);
                    start(v2, s2, e2);
                    finish(v0, s0, e0);
                    finish(v1, s1, e1);
                    finish(v2, s2, e2);
                }
            }
            o = seg_or_u32<W>(o);              // the lanes of a query are stale together
            cq = seg_add_u32<W>(cq);
            occ = o;
            if (ci == 0u) {
                lrow[kRowCq] = cq;
                if constexpr (!PERSIST) {
                    uint4 t;
                    t.x = static_cast<uint32_t>(s.kx); t.y = static_cast<uint32_t>(s.ky);
                    t.z = static_cast<uint32_t>(s.kz); t.w = o;
                    *reinterpret_cast<uint4 *>(P.rows + static_cast<size_t>(q) * kRowWords + kRowKey) = t;
                    P.rows[static_cast<size_t>(q) * kRowWords + kRowCq] = cq;
                }
            }
            if constexpr (PERSIST) {
                if (ci == 0u) {
                    *reinterpret_cast<uint2 *>(lst + kStKey) = make_uint2(static_cast<uint32_t>(s.kx), static_cast<uint32_t>(s.ky));
                    *reinterpret_cast<uint2 *>(lst + kStKey + 2) = make_uint2(static_cast<uint32_t>(s.kz), o);
                }
            }
        }
    }
    if (!valid) occ = 0u;

    // Squared gaps to the six faces of the home cell, pre-scaled by prune_scale.  The cell of voxel
    // index k on one axis (truncation toward zero: cell 0 is two voxels wide): [k vs, (k+1) vs)
    // for k > 0, (-vs, vs) for k = 0, ((k-1) vs, k vs] for k < 0.  Points stored in the voxel
    // below / above the home voxel therefore lie at or beyond `below_hi` / `above_lo`; the gaps are
    // shortened by an absolute slack that dwarfs the rounding of the divide, the product and the
    // subtraction (~1e-16 relative).
    double gx[3], gy[3], gz[3];
    {
        const double vs = P.voxel_size, sc = P.prune_scale;
        auto gaps = [&](double v, int k, double (&g)[3]) {
            const double below_hi = static_cast<double>(k <= 0 ? k - 1 : k) * vs;
            const double above_lo = static_cast<double>(k >= 0 ? k + 1 : k) * vs;
            const double slack = 1e-9 * vs + 1e-13 * fabs(v);
            const double lo = fmax((v - below_hi) - slack, 0.0);
            const double hi = fmax((above_lo - v) - slack, 0.0);
            g[0] = (lo * lo) * sc;
            g[1] = 0.0;
            g[2] = (hi * hi) * sc;
        };
        gaps(s.x, s.kx, gx);
        gaps(s.y, s.ky, gy);
        gaps(s.z, s.kz, gz);
    }
    NN_T(1);
    LP_T(1);

    // ---- search -----------------------------------------------------------------------------------
    // closest_distance2 starts at numeric_limits<double>::max() (VoxelHashMap.cpp:80); the value
    // travels as a kernel argument so that it sits in scalar registers
    double best = P.dist_init;                 // scaled squared distance
    unsigned bkey = 0xFFFFFFFFu;               // (voxel << 8) | slot: the enumeration order
    const int pli = static_cast<int>(s.l);
    const double th = P.sem_th;
    unsigned npairs = 0u;                      // points this query's lanes were handed

    // voxel cursor of this lane: it takes points ci, ci + W, ... of the open voxel.  k = (voxel <<
    // 8) | slot of its next point, kend = (voxel << 8) | points in the voxel, off = where the
    // compact record of point k lives (bytes).  Only the key of the winner is
    // tracked; its offset is rebuilt from the row once per query.
    unsigned k = ci, kend = 0u, off = 0u;
    // The reference's comparison, fp64, on a full record.  Branch-free: a lane that holds no
    // candidate here (`on` false) turns its distance into a NaN, which loses every comparison.
    auto evaluate = [&](const Point4 &nb, bool on, unsigned key) {
        const double dx = nb.x - s.x, dy = nb.y - s.y, dz = nb.z - s.z;
        double d = SAGE_SQNORM3_NN(dx * dx, dy * dy, dz * dz);
        // same label, or either side unlabelled (VoxelHashMap.cpp:87-88)
        // ((int)(a * b) == 0  <=>  |a * b| < 1 under truncation toward zero)
        const bool same = static_cast<int>(nb.l) == pli || fabs(nb.l * s.l) < 1.0;
        const double ds = d * th;
        d = same ? ds : d;
        d = __hiloint2double(on ? __double2hiint(d) : 0x7FF80000, __double2loint(d));
        // lexicographic (d, key): the home voxel is visited first, out of enumeration order;
        // a NaN distance never wins
        const bool lt = d < best, eq = d == best, kl = key < bkey;
        const bool take = lt | (eq & kl);
        best = min_f64(best, d);
        bkey = take ? key : bkey;
    };

    // ---- the fp32 filter in front of it ------------------------------------------------------
    // p, q: map point and query (fp64), p32, q32 their fp32 roundings, u = 2^-24.  Per axis
    // |fl(p32 - q32) - (p - q)| <= u(|p| + |q| + |p - q|) <= 2u(|q| + |p - q|) <= E_a with
    // E_a = 2^-22 (|q_a| + 4 voxel_size): a candidate lies within three voxels of the query.  The
    // fp32 sum of squares D32 is within (1 + u)^4 of the exact one of the rounded differences, so
    // |p - q| >= sqrt(D32 (1 - 5u)) - |E|.  A candidate whose scaled distance scale * |p - q|^2
    // can be <= b (no worse than what is held: it could win or tie) therefore has
    //     D32 <= (sqrt(b / scale) + |E|)^2 / (1 - 5u) <= (b / scale)(1 + 2^-10) k + |E|^2 (1 + 2^10) k
    // (2xy <= e x^2 + y^2 / e), k = 1 + 1e-6; the right-hand side, rounded up to fp32, is the
    // threshold: Ts for candidates of the query's label class (scale = sem_th), Td for the others
    // (scale = 1).  Anything at or under it is fetched as fp64 and compared by `evaluate`.  A label
    // that fp32 cannot classify (k_derive_cand's flag, or a query label that is no small integer)
    // gets the looser of the two.  Pruning off (sem_th negative or NaN): both infinite.
    // (FILT is off for small frames and sparse voxels, where a scan is a handful of points and the
    // filter's set-up and its occasional extra round trip cost more than the bytes it saves: the
    // scan then reads the full records and every point goes through `evaluate`.)
    const float qx = static_cast<float>(s.x), qy = static_cast<float>(s.y), qz = static_cast<float>(s.z);
    const float plab = static_cast<float>(pli);
    const bool q_zero = pli == 0;              // an unlabelled query: every candidate is of its class
    bool unknown = false;
    double slack = 0.0;
    if constexpr (FILT) {
        const bool q_exact = s.l == trunc(s.l) && fabs(s.l) < 16777216.0;
        unknown = !q_exact || (P.cand_flags[0] & 1u);
        const double ex = fabs(s.x) + 4.0 * P.voxel_size, ey = fabs(s.y) + 4.0 * P.voxel_size,
                     ez = fabs(s.z) + 4.0 * P.voxel_size;
        slack = (ex * ex + (ey * ey + ez * ez)) * P.filt_slack;     // 2^-44 (1 + 2^10) k
    }
    PROBE_NN_COUNTERS;
    double fb = best;                          // what the thresholds were derived from (>= the query's final best)
    float Ts = 0.0f, Td = 0.0f, Tmax = 0.0f;   // Tmax: the looser of the two
    auto round_up = [](double x) {             // the next fp32 above x (an infinity becomes a NaN:
        return __uint_as_float(__float_as_uint(static_cast<float>(x)) + 1u);   // `D32 > NaN` is false, nothing is dropped)
    };
    auto set_thresholds = [&]() {
        if constexpr (!FILT) return;
        float a = round_up(fb * P.filt_inv_same + slack), b = round_up(fb * P.filt_inv_diff + slack);
        float m = a > b ? a : b;               // the looser one; a NaN stands for an infinity
        if (a != a || b != b) m = __uint_as_float(0x7FC00000u);
        if (unknown) a = b = m;
        Ts = a; Td = b; Tmax = m;
    };
    // The filter in two steps.  Every scanned point pays for the fp32 distance and ONE comparison with the looser
    // threshold (packed arithmetic: x and y in one instruction, z and the label difference in another — six vector
    // instructions per point); only a step in which some lane holds a point under it looks at the label classes
    // (`tight`), and only a point under the threshold of ITS class is fetched.  (Any association, and fused: the
    // filter's bound assumes four roundings of relative size u — this sum rounds three times; the library is built
    // with -ffp-contract=off for the fp64 comparisons that decide.)
    typedef float f2 __attribute__((ext_vector_type(2)));
    auto dist32 = [&](const uint4 &c, float &dl) {
        const f2 u = f2{__uint_as_float(c.x), __uint_as_float(c.y)} - f2{qx, qy};
        const f2 v = f2{__uint_as_float(c.z), __uint_as_float(c.w)} - f2{qz, plab};
        const f2 sq = u * u;
        dl = v.y;                              // label - query label (exact: small integers; a NaN label stays a NaN)
        return __builtin_fmaf(v.x, v.x, sq.x) + sq.y;
    };
    auto tight = [&](const uint4 &c, float d, float dl) {
        const bool same = (dl == 0.0f) | (__uint_as_float(c.w) == 0.0f) | q_zero;
        return !(d > (same ? Ts : Td));
    };

    // Per-lane state machine over the voxels in `need` (and the one already open).  A step handles
    // two points of the open voxel (k and k + W); two register sets alternate, so while one pair
    // is filtered the loads of the next pair are in flight (no register copies across the loop
    // edge: the wait before a pair leaves the younger loads outstanding).
    // FLAT: the lanes of a query stride through the points of its open voxels as ONE sequence — lane c
    // takes points c, c + W, c + 2W, ... of the concatenation — instead of starting again at slot c in
    // every voxel.  With 16 lanes and ~10 points per voxel the restart leaves lanes 10..15 idle in every
    // voxel and gives lanes 0..9 one point per voxel: a query that must look at all 27 voxels is a chain
    // of 27 points per lane; in flat order it is 270 / 16 = 17.  (Which lane looks at a point changes
    // nothing: the answer is the lexicographic minimum over all of them, reduced across the lanes
    // afterwards.)  A launch ends with its slowest wave — the one holding the heaviest query
    // (profiles/r04/loop_times.txt) — and few points per voxel relative to the lanes per query make
    // the restart's chain the longer one: always with 8 and 16 lanes (k_loop c1 17.8 -> 13.4 us per
    // iteration, 15k-query shards 19.0 -> 15.8 with 16 lanes, the streamed sources' ICP 1.52 -> 1.44 ms),
    // with 2 and 4 lanes where the host finds fewer than 2 W points per voxel on average (P.flat: c5
    // 213 -> 223 frames/s; against c2's and c4's ~12 points per voxel the per-voxel restart is the faster
    // one by 1.5 and 5 %, profiles/r04/flat_where.txt).
    constexpr bool FLAT = PERSIST ? (W >= SAGE_LOOP_FLAT_MINW) : FLATQ;
    using Pair = std::conditional_t<FLAT, std::conditional_t<FILT, PairCompactFlat, PairFullFlat>,
                                    std::conditional_t<FILT, PairCompact, PairFull>>;
    auto scan = [&](unsigned need, const Point4 *seed, bool seeded, unsigned seed_key) {
        // flat order: the next point of this lane — its key, where its record lives, whether there is one
        auto next = [&](bool &h, unsigned &key, unsigned &o) {
            while (k >= kend && need) {        // past the end of the open voxel by k - kend points: on into the next
                const unsigned e = k - kend;
                const unsigned v = static_cast<unsigned>(__builtin_ctz(need));
                need &= need - 1u;
                const uint32_t w = lrow[v];
                kend = (v << 8) | (w & 255u);
                k = (v << 8) + e;
                off = (((w >> 8) * kUnitPoints) + e) << SHC;
                npairs += w & 255u;
            }
            h = k < kend;
            key = k;
            o = off;
            k += h ? static_cast<unsigned>(W) : 0u;
            off += h ? static_cast<unsigned>(W) << SHC : 0u;
        };
        auto issue = [&](Pair &n, bool &more) {
            if constexpr (FLAT) {
                unsigned oa, ob;
                next(n.ha, n.ka, oa);
                next(n.hb, n.kb, ob);
                if constexpr (FILT) {
                    n.oa = oa;
                    n.ob = ob;
                    n.a = load_cand(cands, n.ha ? oa : 0u);
                    n.b = load_cand(cands, n.hb ? ob : 0u);
                } else {
                    n.a = load_point(pts, n.ha ? oa : 0u);
                    n.b = load_point(pts, n.hb ? ob : 0u);
                }
                __builtin_amdgcn_sched_barrier(0);
                more = (k < kend) | (need != 0u);
                return;
            } else {
            while (k >= kend && need) {        // open this lane's next voxel
                const unsigned v = static_cast<unsigned>(__builtin_ctz(need));
                need &= need - 1u;
                const uint32_t w = lrow[v];
                kend = (v << 8) | (w & 255u);
                k = (v << 8) | ci;
                off = (((w >> 8) * kUnitPoints) + ci) << SHC;
                npairs += w & 255u;
            }
            n.ha = k < kend;
            n.hb = k + W < kend;
            n.ka = k;
            if constexpr (FILT) n.oa = off;
            // issued by every lane (idle lanes re-read record 0): a load behind a branch would make
            // the compiler drain the whole queue before the other set is looked at
            const unsigned oa = n.ha ? off : 0u;
            if constexpr (FILT) {
                // (b lies W records behind a — a constant in the instruction; a lane without a b reads whatever
                // lies there, under the buffer's bounds check, and its `hb` drops it)
                n.a = load_cand(cands, oa);
                n.b = load_cand(cands, oa, W << SHC);
            } else {
                const unsigned ob = n.hb ? off + (static_cast<unsigned>(W) << SHC) : 0u;
                n.a = load_point(pts, oa);
                n.b = load_point(pts, ob);
            }
            // the filtering of the other set stays below these loads (the scheduler would
            // otherwise sink them under the arithmetic it believes is ready)
            __builtin_amdgcn_sched_barrier(0);
            // (a lane past the end of its voxel keeps counting: the next voxel it opens sets k and off afresh)
            k += 2u * W;
            off += (2u * W) << SHC;
            more = (k < kend) | (need != 0u);
            }
        };
        auto consume = [&](const Pair &n) {
            unsigned kb;                        // b's key: W points on in a's voxel, or its own (flat order)
            if constexpr (FLAT) kb = n.kb; else kb = n.ka + W;
            if constexpr (!FILT) {
                evaluate(n.a, n.ha, n.ka);
                evaluate(n.b, n.hb, kb);
            } else {
            // (the candidate already held — the seed met again in its voxel — needs no second look)
            float dla, dlb;
            const float da = dist32(n.a, dla), db = dist32(n.b, dlb);
            const bool la = n.ha & !(da > Tmax) & (n.ka != bkey), lb = n.hb & !(db > Tmax) & (kb != bkey);
            PROBE_NN_CONSUME;
            if (__ballot(la | lb)) {
            const bool pa = la & tight(n.a, da, dla), pb = lb & tight(n.b, db, dlb);
            if (__ballot(pa | pb)) {
                // rarer and rarer as the registration settles (a fifth of the pair steps at the
                // start of a cold one, 2 % near convergence): the full records of the candidates
                // that passed (a compact offset is half the byte offset of the full record).
                // (Parking them and fetching in batches was tried: more registers, no fewer trips.)
                PROBE_NN_EXACT(pa, pb);
                unsigned ob;
                if constexpr (FLAT) ob = n.ob; else ob = n.oa + (static_cast<unsigned>(W) << SHC);
                const Point4 ea = load_point(pts, pa ? n.oa << 1 : 0u);
                const Point4 eb = load_point(pts, pb ? ob << 1 : 0u);
                evaluate(ea, pa, n.ka);
                evaluate(eb, pb, kb);
                fb = min_f64(fb, best);
                set_thresholds();
            }
            }
            }
        };
        // (k_loop has registers to spare — a few waves per SIMD, 128 registers each — and an iteration
        // of it ends with its SLOWEST wave, so three / four sets in flight were tried there
        // (SAGE_LOOP_DEPTH_FULL / _COMPACT): 15k queries 18.8 -> 19.9 us per iteration with three sets
        // of full records, 20.6 -> 22.0 with four of compact ones (profiles/r04/loop_depth.txt) — the
        // slow waves are not waiting for their loads.  Two sets everywhere.)
        constexpr int DEPTH = !PERSIST ? 2 : SAGE_LOOP_DEPTH_OF(FILT);
        Pair A;
        bool more = false;
        issue(A, more);
        if constexpr (DEPTH == 1) {
            // (probe: one set in flight — 11 registers fewer; the other waves of the SIMD cover the round trips)
            if (seed) evaluate(*seed, seeded, seed_key);
            fb = min_f64(fb, best);
            set_thresholds();
            for (;;) {
                consume(A);
                if (!__ballot(more)) break;
                issue(A, more);
            }
            return;
        }
        Pair B;
        if constexpr (DEPTH == 2) {
            // the seed's load is older than A's: waiting for it leaves A's loads in flight
            if (seed) evaluate(*seed, seeded, seed_key);
            fb = min_f64(fb, best);
            set_thresholds();
            // One exit per double step: an exit between the two halves gives the loop header a
            // predecessor with B's loads pending, and the compiler then drains the queue (vmcnt(0))
            // before every issue(B) — the overlap this loop exists for.  A scan that ends after the
            // first half pays one idle half step instead.
            // (Peeling short scans out of the loop — one pair step, or none — was tried: the extra
            // paths cost 10 registers and spills, every workload lost 5-10 %.)
            for (;;) {
                issue(B, more);
                consume(A);
                issue(A, more);
                consume(B);
                if (!__ballot(A.ha | more)) break;
            }
        } else if constexpr (DEPTH == 3) {
            Pair C;
            issue(B, more);
            if (seed) evaluate(*seed, seeded, seed_key);
            fb = min_f64(fb, best);
            set_thresholds();
            for (;;) {
                issue(C, more);
                consume(A);
                issue(A, more);
                consume(B);
                issue(B, more);
                consume(C);
                if (!__ballot(A.ha | B.ha | more)) break;
            }
        } else {
            Pair C, D;
            issue(B, more);
            issue(C, more);
            if (seed) evaluate(*seed, seeded, seed_key);
            fb = min_f64(fb, best);
            set_thresholds();
            for (;;) {
                issue(D, more);
                consume(A);
                issue(A, more);
                consume(B);
                issue(B, more);
                consume(C);
                issue(C, more);
                consume(D);
                if (!__ballot(A.ha | B.ha | C.ha | more)) break;
            }
        }
    };

    // The previous iteration's nearest neighbour is still a point of this neighbourhood as long as
    // the home voxel has not changed (the row, and with it the meaning of `key`, is the same; the
    // map is constant during a call): evaluated first, it gives every query — also one whose
    // home voxel is empty — a tight bound before anything is scanned.  It is an ordinary
    // candidate: meeting it again in the scan changes nothing.
    constexpr unsigned kHome = 13u;
    bool merged = false;                       // this query's home voxel is scanned with its neighbours
    if constexpr (PERSIST) {
        // k_loop holds the seed's record in registers: a seeded query takes its bound from the seed
        // alone — no memory — and scans its home voxel together with the neighbours that survive
        // that bound, in ONE pass (any bound at or above the final best prunes exactly; after the
        // first iterations the seed IS the answer for most queries and the bound is the final
        // one).  Only queries without a seed (the first pass of a call, a rebuilt row) scan their
        // home voxel first; a wave without such a query skips that pass altogether.
        const bool seeded = valid && prev.x != 0xFFFFFFFFu;          // (a rebuilt row re-keyed or dropped it)
        const Point4 pp = *reinterpret_cast<const Point4 *>(lst + 8);
        evaluate(pp, seeded, prev.x);
        merged = seeded;
        const unsigned first = seeded ? 0u : (occ & (1u << kHome));
        if (__ballot(first != 0u)) scan(first, nullptr, false, 0u);
    } else if (FUSED) {
        const bool seeded = valid && prev.x != 0xFFFFFFFFu;          // (a rebuilt row re-keyed or dropped it)
        const Point4 pp = load_point(pts, seeded ? prev.y : 0u);      // the full record
        scan(occ & (1u << kHome), &pp, seeded, prev.x);
    } else {
        scan(occ & (1u << kHome), nullptr, false, 0u);
    }
    NN_T(2);
    LP_T(2);
    // what the query holds after its home voxel (or its seed) bounds the rest of its search
    const double bound = seg_min_f64<W>(best);
    fb = bound;                                // (set_thresholds runs at the start of the scan)
    unsigned need = P.keep_all;
    if constexpr (W >= 4) {
        // The 26 bound tests are the same for every lane of the query: lane a < 3 takes the x-layer
        // a (nine voxels, one add and one compare each on top of the shared gy + gz sums), the
        // layers meet in two DPP exchanges.  Same operands, same association: the same mask as
        // the loop below, in 40 instructions instead of 100.
        const unsigned a = ci & 3u;
        const double ga = a == 0u ? gx[0] : (a == 2u ? gx[2] : 0.0);
        unsigned layer = 0u;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double lb = ga + (gy[b] + gz[c]);
                layer |= (lb <= bound) ? (1u << (3 * b + c)) : 0u;
            }
        }
        layer = (a < 3u && ci < 4u) ? layer << (9u * a) : 0u;
        layer |= dpp_u32<kDppXor1>(layer);
        layer |= dpp_u32<kDppXor2>(layer);          // lanes 0..3 of the query now hold all three layers
        if (W >= 8) layer |= dpp_u32<kDppHalfMirror>(layer);
        if (W >= 16) layer |= dpp_u32<kDppMirror>(layer);
        need |= layer & ~(1u << kHome);
    } else {
#pragma unroll
        for (int v = 0; v < 27; ++v) {
            if (v == static_cast<int>(kHome)) continue;
            const double lb = gx[v / 9] + (gy[(v / 3) % 3] + gz[v % 3]);
            need |= (lb <= bound) ? (1u << v) : 0u;
        }
    }
    LP_T(3);
    scan((merged ? (need | (1u << kHome)) : (need & ~(1u << kHome))) & occ, nullptr, false, 0u);
    LP_T(4);
    if constexpr (PERSIST) {
        // what this block cost: the most points one of its queries was handed (a rebuilt row counts as a few more) —
        // next iteration's order of the workgroup's blocks (k_loop)
        if (valid && ci == 0u)
            (void)__hip_atomic_fetch_max(G->work + bslot, npairs + (stale ? 48u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    // argmin over the W lanes of the query: first the minimum distance (never NaN: a NaN distance
    // fails every comparison), then the smallest key among the lanes that hold it; the winner's
    // offset follows from its key and the row
    const double m = seg_min_f64<W>(best);
    const unsigned mine = (best == m) ? bkey : 0xFFFFFFFFu;
    const unsigned mkey = seg_min_u32<W>(mine);
    const bool found = valid && mkey != 0xFFFFFFFFu;       // else: empty neighbourhood (hazard H1)
    const unsigned woff = (lrow[found ? mkey >> 8 : 0u] >> 8) * (kUnitPoints * 32u) +
                          ((mkey & 255u) << SH);
    NN_T(3);
    LP_T(5);

    if (P.counters) {                          // C_q and pairs handed out, summed over the wave
        // (both fit 16 bits per query: one packed value goes through the four DPP exchanges inside
        // the rows of 16 lanes, four v_readlane collect the rows; the wave's private slot takes the
        // sums as fire-and-forget atomics — a read-modify-write would hold the wave for a round trip)
        const unsigned long long ab = (valid && ci == 0u)
                                          ? (static_cast<unsigned long long>(lrow[kRowCq]) << 32) | npairs : 0ull;
        unsigned lo = static_cast<unsigned>(ab), hi = static_cast<unsigned>(ab >> 32);
        lo += dpp_u32<kDppXor1>(lo);        hi += dpp_u32<kDppXor1>(hi);
        lo += dpp_u32<kDppXor2>(lo);        hi += dpp_u32<kDppXor2>(hi);
        lo += dpp_u32<kDppHalfMirror>(lo);  hi += dpp_u32<kDppHalfMirror>(hi);
        lo += dpp_u32<kDppMirror>(lo);      hi += dpp_u32<kDppMirror>(hi);
        unsigned b = 0u, a = 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            b += static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(lo), 16 * r));
            a += static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(hi), 16 * r));
        }
        if (lane == 0 && wave_id < P.nwaves) {
            (void)__hip_atomic_fetch_add(&P.counters[2u * wave_id], static_cast<unsigned long long>(a),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(&P.counters[2u * wave_id + 1u], static_cast<unsigned long long>(b),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    if (!FUSED) {
        if (valid && ci == 0u) P.nn_idx[q] = found ? static_cast<int>(woff >> SH) : -1;
    } else {
        // ---- fused epilogue: acceptance + Gauss-Newton terms of this query's pair -----------------
        if constexpr (!PERSIST) {
            if (valid && ci == 0u) P.nn_prev[q] = make_uint2(found ? mkey : 0xFFFFFFFFu, woff);
        }
        Point4 g;
        if constexpr (PERSIST) {
            // The answer of most queries is the previous iteration's (and then so is its record: the
            // map is constant during a call, and the key was carried over if the row was rebuilt):
            // only a query whose answer changed fetches a record, on all of its lanes (one request:
            // the same address), since every lane evaluates the seed.
            const bool changed = found && mkey != prev.x;
            g = *reinterpret_cast<const Point4 *>(lst + 8);
            if (__ballot(changed)) {
                const Point4 t = load_point(pts, changed ? woff : 0u);
                if (changed) {
                    g = t;
                    if (ci == 0u) *reinterpret_cast<Point4 *>(lst + 8) = t;
                }
            }
            if (ci == 0u) *reinterpret_cast<uint2 *>(lst + kStPrev) = make_uint2(found ? mkey : 0xFFFFFFFFu, woff);
            LP_T(6);
        }
        PROBE_NN_WORK(P, q, valid && ci == 0u, npairs);
        // Branch-free: every lane computes the terms of "its" pair from operands that are zeroed unless it is the
        // first lane of a query with an accepted answer (the products are then exact zeros of either sign, which
        // the block sums and their digits do not tell apart) — clearing sixteen fp64 registers twice around two
        // nested branches cost more than the selects.
        double t[kCount];
        const bool cand = found && ci == 0u;
        if constexpr (!PERSIST) g = load_point(pts, cand ? woff : 0u);      // (the other lanes re-read record 0: no branch)
        const double rx0 = s.x - g.x, ry0 = s.y - g.y, rz0 = s.z - g.z;
        // (closest_neighboor - point).head<3>().norm() < max_correspondance_distance (VoxelHashMap.cpp:111)
        const bool use = cand && SAGE_SQNORM3_ACCEPT(rx0 * rx0, ry0 * ry0, rz0 * rz0) <= P.accept_r2;
        {
            // residual.squaredNorm() (Registration.cpp:79): its own reduction (sageicp_types.h)
            const double r2 = SAGE_SQNORM3_RESID(rx0 * rx0, ry0 * ry0, rz0 * rz0);
            const double k = P.kernel;
            const double den = k + r2;
            const double wq = (k * k) / (den * den);   // square(th) / square(th + residual2)
            const double w = use ? wq : 0.0;
            const double sx = use ? s.x : 0.0, sy = use ? s.y : 0.0, sz = use ? s.z : 0.0;
            const double rx = use ? rx0 : 0.0, ry = use ? ry0 : 0.0, rz = use ? rz0 : 0.0;
            const double wsx = w * sx, wsy = w * sy, wsz = w * sz;
            t[kW] = w;
            t[kWsx] = wsx; t[kWsy] = wsy; t[kWsz] = wsz;
            t[kWxx] = wsx * sx; t[kWxy] = wsx * sy; t[kWxz] = wsx * sz;
            t[kWyy] = wsy * sy; t[kWyz] = wsy * sz; t[kWzz] = wsz * sz;
            t[kWrx] = w * rx; t[kWry] = w * ry; t[kWrz] = w * rz;
            t[kWcx] = w * (sy * rz - sz * ry);
            t[kWcy] = w * (sz * rx - sx * rz);
            t[kWcz] = w * (sx * ry - sy * rx);
        }
        const unsigned pairs = static_cast<unsigned>(__popcll(__ballot(use)));
        if constexpr (PERSIST) {
            // k_loop: block sums -> exact digits -> the workgroup's accumulators (the rows stay: the scratch
            // is the running wave's own)
            wave_terms_to_wgacc<LW>(t, pairs, lane, G->red, G->wgacc, kDigitLimitCounted, P.acc_scale);
            PROBE_LOOP_WAVE_STATS(smem, valid, ci, npairs, stale, lane);
            LP_T(7);
            return;                             // (k_loop closes the workgroup's iteration itself)
        }
        if (P.stripe_work) {
            // (an iteration that measures: the most points one of this wave's queries was handed -> the stripe's heaviest wave)
            unsigned mx = valid ? npairs : 0u;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) mx = max(mx, static_cast<unsigned>(__shfl_xor(mx, d, 64)));
            if (lane == 0) (void)__hip_atomic_fetch_max(&P.stripe_work[stripe_id], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        {
            // k_icp: the same, into this workgroup's accumulators; the last wave to arrive sends them on
            // (what the ticket orders lives in LDS, which serves a CU's waves in order: the ticket is a
            // relaxed LDS atomic between compiler barriers)
            unsigned long long *wgacc = reinterpret_cast<unsigned long long *>(smem + kWgAcc);
            wave_terms_to_wgacc<LW>(t, pairs, lane, reinterpret_cast<double *>(wl), wgacc, P.digit_limit, P.acc_scale);
            unsigned prior = 0u;
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            if (lane == 0)
                prior = __hip_atomic_fetch_add(&smem[kWgArrive], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            prior = __builtin_amdgcn_readfirstlane(prior);
            if (prior == kIcpWavesPerBlock - 1u) {
                if (P.chain)
                    wgacc_flush<true>(wgacc, &P.chain->acc32[P.chain_iter & 1][blockIdx.x & (kChainReplicas - 1)][0], nullptr);
                else
                    wgacc_flush(wgacc, P.acc + static_cast<size_t>(blockIdx.x & (kAccReplicas - 1)) * kAccWords,
                                P.acc + kAccWords - 1);
            }
        }
    }
    PROBE_NN_END(P, valid, ci, npairs, lane, wave_id);
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
// Executed by ONE workgroup of 1024 threads once per ICP iteration (k_fin): fixed-order reduction
// of the workgroup partials of k_icp (bit-reproducible), then the first wave assembles the 6x6
// normal equations from the 16 closed-form sums, solves them (register-resident pivoted LDL^T),
// applies SE3 exp, composes the pose and tests convergence (Registration.cpp:92-93,135-137).
#ifdef SAGE_GN_TIMING
__device__ unsigned long long g_gn_phase[16];
#define FIN_STAMP(i) do { if (threadIdx.x == 0) fin_t[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FIN_STAMP(i) do { } while (0)
#endif

constexpr int kFinThreads = 1024;
constexpr int kFinSlices = 102;             // 10 fp64 pairs per partial x 102 slices = 1020 threads

// Returns false (on every thread) when *done is set: the loop has finished and this launch is a
// no-op.  The flag is fetched together with the partials — one memory round trip, not two.
__device__ __forceinline__ bool reduce_partials(const double *partials, int nparts, double *S /* LDS [kNumSums] */,
                                                const int32_t *done) {
    __shared__ double part[kFinSlices][kNumSums];
    __shared__ double part2[6][kNumSums];
    const int t = static_cast<int>(threadIdx.x);
    const int pr = t % 10, sl = t / 10;
    if (sl < kFinSlices) {
        // fixed summation order; the loads are independent: up to 24 in flight per thread (one
        // memory round trip for up to 2,448 partials, the cold-L2 latency is what this kernel costs)
        double2 v = make_double2(0.0, 0.0);
        const double2 *src = reinterpret_cast<const double2 *>(partials) + pr;
        bool first = true;
        for (int b = sl; b < nparts; b += 24 * kFinSlices) {
            double2 u[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                const int bb = b + k * kFinSlices;
                u[k] = bb < nparts ? src[static_cast<size_t>(bb) * 10] : make_double2(0.0, 0.0);
            }
            if (first && done) {
                // a vector load like the ones above (the index is zero, but formally per lane), so
                // that it travels with them: a scalar load would be waited for before the partials
                // are even requested
                const int32_t d = done[__builtin_amdgcn_mbcnt_lo(0u, 0u)];
                if (__builtin_amdgcn_readfirstlane(d)) return false;
                first = false;
            }
#pragma unroll
            for (int k = 0; k < 24; ++k) { v.x += u[k].x; v.y += u[k].y; }
        }
        if (first && done) {                   // no partials at all (an empty frame)
            const int32_t d = done[__builtin_amdgcn_mbcnt_lo(0u, 0u)];
            if (__builtin_amdgcn_readfirstlane(d)) return false;
        }
        part[sl][2 * pr] = v.x;
        part[sl][2 * pr + 1] = v.y;
    } else if (done) {
        const int32_t d = done[__builtin_amdgcn_mbcnt_lo(0u, 0u)];
        if (__builtin_amdgcn_readfirstlane(d)) return false;
    }
    __syncthreads();
    if (t < 6 * kNumSums) {
        const int c = t % kNumSums, g = t / kNumSums;
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < kFinSlices / 6; ++k) v += part[g * (kFinSlices / 6) + k][c];
        part2[g][c] = v;
    }
    __syncthreads();
    if (t < kNumSums) {
        double v = part2[0][t];
#pragma unroll
        for (int g = 1; g < 6; ++g) v += part2[g][t];
        S[t] = v;
    }
    __syncthreads();
    return true;
}

// The sums from the fixed-point accumulators k_icp's workgroups added into (kernels.h): one round
// trip for 16 KB, the replicas added exactly (integers), three digits -> one fp64 per sum, and the
// accumulators cleared for the next iteration.  Returns false when *done is set (see above).
__device__ __forceinline__ bool reduce_accumulators(long long *acc, double *S /* LDS [kNumSums] */,
                                                    const int32_t *done, int32_t *overflow, double unscale) {
    __shared__ long long part[kAccReplicas][kAccWords];
    __shared__ long long part2[8][kAccWords];
    const int t = static_cast<int>(threadIdx.x);
    // 2,048 words over 1,024 threads: 16 B each, one coalesced round trip
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    ll2 *src = reinterpret_cast<ll2 *>(acc);
    const ll2 v = src[t];
    if (done) {
        const int32_t d = done[__builtin_amdgcn_mbcnt_lo(0u, 0u)];      // travels with the load above
        if (__builtin_amdgcn_readfirstlane(d)) return false;
    }
    ll2 z;
    z.x = 0; z.y = 0;
    src[t] = z;                                                          // cleared for the next launch of k_icp
    reinterpret_cast<ll2 *>(&part[0][0])[t] = v;
    __syncthreads();
    if (t < 8 * kAccWords) {
        const int w = t % kAccWords, g = t / kAccWords;
        long long s = 0;
#pragma unroll
        for (int k = 0; k < kAccReplicas / 8; ++k) s += part[g * (kAccReplicas / 8) + k][w];
        part2[g][w] = s;
    }
    __syncthreads();
    if (t < kAccWords) {
        long long s = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += part2[g][t];
        part[0][t] = s;
    }
    __syncthreads();
    if (t < kNumSums) {
        double r = 0.0;
        if (t < kAccValues) {
            const double a = static_cast<double>(part[0][3 * t]);
            const double b = static_cast<double>(part[0][3 * t + 1]);
            const double c = static_cast<double>(part[0][3 * t + 2]);
            r = a + (b * 9.094947017729282e-13 + c * 8.271806125530277e-25);      // 2^-40, 2^-80
            if (t < kCount) r *= unscale;      // (a power of two; the pair count is not scaled)
        }
        S[t] = r;
        if (t == 0 && part[0][kAccWords - 1] != 0) *overflow = 1;
    }
    __syncthreads();
    return true;
}

// the solve: wave 0, all 64 lanes, uniform data (see WaveLanes); lane 0 / lane 1 publish the state
// The loop state the finish needs (the two poses the estimate is composed with, the iteration
// count), requested by the first wave BEFORE the reduction so that its cold round trip (~0.7 us)
// runs under it instead of after the solve.
struct FinState {
    double rhs[7];            // lane 1: T_icp, the other lanes: T
    int iter;
};
__device__ __forceinline__ FinState prefetch_state(const IcpState *st) {
    FinState f;
    const int lane = static_cast<int>(threadIdx.x);
    const double *src = (lane == 1) ? st->T_icp : st->T;
#pragma unroll
    for (int i = 0; i < 7; ++i) f.rhs[i] = src[i];
    f.iter = st->iter;
    return f;
}

__device__ __forceinline__ void solve_and_publish(IcpState *st, const double *S, const FinState &pre) {
#ifdef SAGE_GN_TIMING
    unsigned long long fin_t[8];
#endif
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
    double Tn[7];
    se3_mul(est, pre.rhs, Tn);
    if (lane < 2) {
        double *dst = (lane == 1) ? st->T_icp : st->T;
#pragma unroll
        for (int i = 0; i < 7; ++i) dst[i] = Tn[i];
    }
    if (lane != 0) return;
    quat_to_mat(Tn, st->R);

    // ||log(exp(x))|| == ||x|| on the principal branch up to a few ulps, so the reference's
    // estimation.log().norm() (Registration.cpp:137) is taken from x without the atan2 / sincos
    // round trip on one serial lane — except where those ulps could matter: a step within 1e-12
    // (relative 1e-8; the two differ by ~1e-19 there) of the stop threshold, or |omega| >= 3,
    // goes through the exact log so that the stop iteration is the reference's in every case.
    double nrm = sqrt(SAGE_SQNORM6(x));
    if (!(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] < 9.0) ||
        fabs(nrm - kEstimationThreshold) < 1e-12) {
        double lg[6];
        se3_log(est, lg);
        nrm = sqrt(SAGE_SQNORM6(lg));          // the reduction order of a 6-vector's norm(): sageicp_types.h
    }
    st->last_step_norm = nrm;
    const int it = pre.iter;
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
        // host-mapped, a relaxed system-scope (write-through) store: the host only steers its
        // look-ahead by this word and reads the final state through an ordinary copy after the loop
        const unsigned long long seq = static_cast<unsigned long long>(it + 1);
        __hip_atomic_store(&pg->word, (done << 32) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#ifdef SAGE_GN_TIMING
    fin_t[4] = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i < 4; ++i) atomicAdd(&g_gn_phase[8 + i], fin_t[i + 1] - fin_t[i]);
    atomicAdd(&g_gn_phase[4], 1ull);
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
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        // a peer gave up its one-launch loop at this exchange (P2pBlock::abort_tag)
        if (__hip_atomic_load(&mine->abort_tag[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == tag) s_late = 2;
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
        if (s_late == 2) {                     // every rank leaves this exchange and starts the frame again (run_icp)
            st->peer_aborted = 1;
            st->done = 1;
        } else if (s_late) {                   // stop the loop; the host reports the failure
            st->exchange_failed = 1;
            st->done = 1;
        }
    }
}

// ------------------------------------------------------------------------------------ k_fin
__global__ __launch_bounds__(kFinThreads) void k_fin(FinParams P) {
    __shared__ double S[kNumSums];
    IcpState *st = P.st;
    if (P.mode == 2 && !P.standalone && st->done) return;
#ifdef SAGE_GN_TIMING
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    FinState pre{};
    if (threadIdx.x < 64) pre = prefetch_state(st);
    if (P.mode != 2) {
        if (P.acc) {
            if (!reduce_accumulators(P.acc, S, P.standalone ? nullptr : &st->done, &st->acc_overflow, P.acc_unscale)) return;
        } else if (!reduce_partials(P.partials, P.nparts, S, P.standalone ? nullptr : &st->done)) return;
#ifdef SAGE_GN_TIMING
        if (threadIdx.x == 0) atomicAdd(&g_gn_phase[8], __builtin_amdgcn_s_memrealtime() - t_start);
#endif
        if (threadIdx.x < kNumSums) st->sums[threadIdx.x] = S[threadIdx.x];
        if (P.mode == 1) return;
        if (P.mode == 3) {
            __syncthreads();
            exchange_sums(st, P.p2p);
            __syncthreads();
            if (threadIdx.x < kNumSums) S[threadIdx.x] = st->sums[threadIdx.x];
            __syncthreads();
        }
    } else {
        if (threadIdx.x < kNumSums) S[threadIdx.x] = st->sums[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x >= 64) return;
    solve_and_publish(st, S, pre);
}

// ------------------------------------------------------------------------------------ k_loop
// The whole loop of Registration.cpp:127-138 in one launch of the query workgroups (k_loop) beside a
// one-wave solving kernel (k_loop_solve) on a second stream (kernels.h, LoopShared).  Per iteration:
//   every wave        takes the workgroup's groups one after another from an LDS counter and runs
//                     icp_body<PERSIST> on each — pose from LDS, per-query state and rows in LDS —,
//                     which parks the group's sums in the workgroup's LDS;
//   last wave of a    adds the workgroup's sums into the fixed-point accumulators: (digit << 8) + 1 per
//   workgroup         word, fire and forget — the low byte of every word counts who is in it;
//   the solving       reads this iteration's set of accumulators until every word counts all its
//   wave              workgroups (two sets alternate; the one just read is cleared for the iteration
//                     after the next), [exchanges the sums with the peer GPUs,] solves, composes, tests,
//                     and publishes the next pose as 25 self-tagged 8-byte granules (tag = iteration
//                     + 1: the data is the flag, no fence on either side);
//   wave 0 of every   polls the granules (one relaxed agent-scope load per lane and pass), hands the
//   workgroup         pose to its workgroup through LDS;  __syncthreads();  next iteration.
// Every word the workgroups share is accessed with agent-scope atomics only.  Every wait is bounded:
// a timeout raises LoopShared::abort_word and IcpState::loop_aborted, everybody leaves, and the host
// registers the frame through the launch-per-iteration loop instead (a grid that is not fully
// resident — another process or stream holding CUs — ends this way, not in a hang).
#ifdef SAGE_LOOP_TIMING
// probe builds: 100-MHz stamps of the first kLoopTimedIters iterations — per workgroup when it counted
// itself in and when it had the next pose; for the solving wave when all counts were in, the sums
// read, the step solved, the pose published
constexpr int kLoopTimedIters = 32, kLoopTimedWgs = 2048;
__device__ unsigned long long g_loop_wg[kLoopTimedIters][kLoopTimedWgs][4];     // counted in | pose held | a wave took a unit beyond one per wave | ... finished it
__device__ unsigned g_loop_wginfo[kLoopTimedIters][kLoopTimedWgs][4];     // HW_ID | max points of a query | stale queries | points
__device__ unsigned long long g_loop_solver[kLoopTimedIters][4];
__device__ unsigned long long g_loop_solver2[kLoopTimedIters][4];      // inside the solve: after the LDL^T | the exponential | the composition | the norm
__device__ unsigned long long g_loop_wave[kLoopTimedIters][kLoopTimedWgs][8][2];      // per wave: its FIRST unit of the iteration: end stamp | start stamp (low 32) << 32 ... see LOOP_STAMP_WAVE
__device__ unsigned long long g_loop_phase[16];     // [0..7] cycles per body phase, [8] wait for the pose, [9] closing a workgroup, [10] group passes
#define LOOP_STAMP_SOLVER(it, k) do { if ((it) < kLoopTimedIters && (threadIdx.x & 63u) == 0u) g_loop_solver[it][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define LOOP_STAMP_SOLVER2(it, k) do { if ((it) < kLoopTimedIters && (threadIdx.x & 63u) == 0u) g_loop_solver2[it][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" void sageicp_debug_loop_solver2(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_loop_solver2), sizeof(unsigned long long) * kLoopTimedIters * 4);
}
#define LOOP_STAMP_WG(it, k) do { if ((it) < kLoopTimedIters && blockIdx.x < kLoopTimedWgs && (threadIdx.x & 63u) == 0u) g_loop_wg[it][blockIdx.x][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" void sageicp_debug_loop_phases(unsigned long long *out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_loop_phase), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_loop_phase), z, sizeof(z));
    }
}
extern "C" void sageicp_debug_loop_waves(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_loop_wave), sizeof(unsigned long long) * kLoopTimedIters * kLoopTimedWgs * 8 * 2);
}
extern "C" void sageicp_debug_loop_info(unsigned *info) {
    (void)hipMemcpyFromSymbol(info, HIP_SYMBOL(g_loop_wginfo), sizeof(unsigned) * kLoopTimedIters * kLoopTimedWgs * 4);
}
extern "C" void sageicp_debug_loop_times(unsigned long long *wg, unsigned long long *solver) {
    (void)hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_loop_wg), sizeof(unsigned long long) * kLoopTimedIters * kLoopTimedWgs * 4);
    (void)hipMemcpyFromSymbol(solver, HIP_SYMBOL(g_loop_solver), sizeof(unsigned long long) * kLoopTimedIters * 4);
}
#else
#define LOOP_STAMP_SOLVER(it, k) do { } while (0)
#define LOOP_STAMP_SOLVER2(it, k) do { } while (0)
#define LOOP_STAMP_WG(it, k) do { } while (0)
#endif


// exchange_sums for ONE wave (the solving wave of k_loop_solve): S (LDS) holds this rank's sums on entry
// and the sums over all ranks, added in rank order, on exit; `g` is the exchange counter (the same on
// every rank).  Returns 0, 1 when a peer's sums did not arrive in time, 2 when a peer gave up its one-launch loop at
// this exchange (P2pBlock::abort_tag).
__device__ __forceinline__ int exchange_sums_wave(double *S, const P2pParams &X, unsigned long long g) {
    const int lane = static_cast<int>(threadIdx.x & 63u);
    const int slot = static_cast<int>(g & 1ull);
    const unsigned long long tag = g + 1ull;
    if (lane < kNumSums) {
        const double v = S[lane];
        for (int r = 0; r < X.nranks; ++r)
            __hip_atomic_store(&X.block[r]->sums[slot][X.rank][lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
        for (int r = 0; r < X.nranks; ++r)
            __hip_atomic_store(&X.block[r]->flag[X.rank], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    P2pBlock *mine = X.block[X.rank];
    bool late = false, gone = false;
    if (lane < X.nranks) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(&mine->flag[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < tag) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > X.timeout_ticks) {
                late = true;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        gone = __hip_atomic_load(&mine->abort_tag[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == tag;
    }
    late = __any(late);
    gone = __any(gone);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (lane < kNumSums) {
        double s = 0.0;
        for (int r = 0; r < X.nranks; ++r)       // rank order: the same sum on every rank
            s += __hip_atomic_load(&mine->sums[slot][r][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        S[lane] = s;
    }
    __builtin_amdgcn_wave_barrier();
    return gone ? 2 : (late ? 1 : 0);
}
// This rank leaves its one-launch loop at exchange `g` (a wait inside the launch timed out): the peers are told through
// the flag of that exchange, so that everybody leaves it together.
__device__ __forceinline__ void exchange_abort_wave(const P2pParams &X, unsigned long long g) {
    const int lane = static_cast<int>(threadIdx.x & 63u);
    if (lane == 0)
        for (int r = 0; r < X.nranks; ++r)
            __hip_atomic_store(&X.block[r]->abort_tag[X.rank], g + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)          // ... and the flag of the exchange, so that nobody waits for this rank's sums
        for (int r = 0; r < X.nranks; ++r)
            __hip_atomic_store(&X.block[r]->flag[X.rank], g + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One iteration's finish by the solving wave (all 64 lanes, uniform data).  Returns the value of the
// done granule it published: 0 go on, 1 finished, 2 aborted.
struct SolveLds {
    double T[14];              // T[7] | T_icp[7]
    double S[kNumSums];
    double pub[12];
    long long digits[kAccWords];
};
template <int COPIES>
__device__ __forceinline__ unsigned loop_finish_iteration(const LoopParams &L, const P2pParams &X, SolveLds &m, int it,
                                                          unsigned long long &xg) {
    LoopShared *sh = L.sh;
    IcpState *st = L.st;
    const int lane = static_cast<int>(threadIdx.x & 63u);          // (one wave)
    double *sT = m.T, *S = m.S, *pub = m.pub;
    long long *digits = m.digits;
    const unsigned long long tag = static_cast<unsigned long long>(it) + 1ull;

    // 1. the sums of this iteration's set of accumulators: read (one round trip per pass) until every word
    // says that all the workgroups adding into it are in (its low byte counts them, wgacc_flush) —
    // the read that finds them complete IS the read of the sums.  The set is then cleared for the
    // iteration after the next (the clears are complete long before that pose is published: the waits
    // of the next iteration's passes cover them).
    {
        const long long per = static_cast<long long>(L.wgs / COPIES);      // workgroups adding into each copy
        long long (*acc)[kAccWords] = COPIES == kLoopReplicas ? sh->acc[it & 1] : sh->acc32[it & 1];
        long long v[COPIES];
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
#pragma unroll
            for (int r = 0; r < COPIES; ++r)
                v[r] = __hip_atomic_load(&acc[r][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
            if (lane <= 3 * kAccValues) {          // (word 3 kAccValues: the overflow count, counted like the sums)
#pragma unroll
                for (int r = 0; r < COPIES; ++r) ok &= (v[r] & 255ll) == per;
            }
            if (__all(ok)) break;
            unsigned long long ab = 0ull;
            if (lane == 0) ab = ld_agent(&sh->abort_word[0]);
            // (its own workgroups' counts are a local matter: the short wait also under a communicator, where the
            // workgroups' patience — timeout_ticks — has to outlast the exchange with the peers)
            const bool late = __builtin_amdgcn_s_memrealtime() - t0 > L.count_timeout_ticks;
            if (__any(ab != 0ull) || late) {
                if (lane == 0) {
                    st_agent(&sh->abort_word[0], 1ull);
                    st->loop_aborted = 1;
                    st_agent(&sh->pose[24], (tag << 32) | 2ull);
                    if (L.progress)        // (chained launches: the host stops enqueuing)
                        __hip_atomic_store(&L.progress->word, (1ull << 32) | static_cast<unsigned long long>(it), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (X.nranks > 1) {
                    // the peers are inside (or on their way to) this very exchange: they leave it with us, and every
                    // rank registers the frame again through the launch-per-iteration form, in step (run_icp)
                    exchange_abort_wave(X, xg);
                    xg += 1ull;
                    if (lane == 0) *X.exchanges = xg;
                }
                return 2u;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        LOOP_STAMP_SOLVER(it, 0);
        long long d = 0;
        if (lane <= 3 * kAccValues) {
#pragma unroll
            for (int r = 0; r < COPIES; ++r) d += (v[r] - per) >> 8;       // (exact: the low byte is the count)
        }
#pragma unroll
        for (int r = 0; r < COPIES; ++r)
            __hip_atomic_store(&acc[r][lane], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        digits[lane] = d;
        __builtin_amdgcn_wave_barrier();
        if (lane < kNumSums) {
            double r = 0.0;
            if (lane < kAccValues) {
                const double a = static_cast<double>(digits[3 * lane]);
                const double b = static_cast<double>(digits[3 * lane + 1]);
                const double c = static_cast<double>(digits[3 * lane + 2]);
                r = a + (b * 9.094947017729282e-13 + c * 8.271806125530277e-25);      // 2^-40, 2^-80
                if (lane < kCount) r *= L.acc_unscale;     // (a power of two; the pair count is not scaled)
            }
            S[lane] = r;
        }
        __builtin_amdgcn_wave_barrier();
    }
    const bool overflow = digits[3 * kAccValues] != 0;       // workgroups whose sums left the range (wgacc_flush)
    LOOP_STAMP_SOLVER(it, 1);

    // 2. multi-GPU: this rank's sums -> the sums over all ranks (direct exchange over xGMI, P2pBlock)
    bool exchange_failed = false;
    if (X.nranks > 1) {
        const int ex = exchange_sums_wave(S, X, xg);
        xg += 1ull;
        if (lane == 0) *X.exchanges = xg;
        exchange_failed = ex == 1;
        if (ex == 2) {
            // a peer gave up its one-launch loop at this exchange: so does this rank (its workgroups see the abort word)
            if (lane == 0) {
                st_agent(&sh->abort_word[0], 1ull);
                st->loop_aborted = 1;
                st->peer_aborted = 1;
                st_agent(&sh->pose[24], (tag << 32) | 2ull);
                if (L.progress)        // (chained launches: the host stops enqueuing)
                    __hip_atomic_store(&L.progress->word, (1ull << 32) | static_cast<unsigned long long>(it), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return 2u;
        }
    }

    // 3. solve, compose, test (Registration.cpp:92-93,135-137) — as k_fin's solve_and_publish
    double JTJ[36], JTr[6], neg[6], x[6], est[7];
    assemble_normal_equations(S, JTJ, JTr);
#pragma unroll
    for (int i = 0; i < 6; ++i) neg[i] = -JTr[i];
    ldlt_solve6_t<WaveLanes>(JTJ, neg, x);
    LOOP_STAMP_SOLVER2(it, 0);
    se3_exp_t<WaveLanes>(x, est);
    LOOP_STAMP_SOLVER2(it, 1);
    double rhs[7], Tn[7];
    {
        const double *src = sT + (lane == 1 ? 7 : 0);      // lane 1: T_icp, the other lanes: T
#pragma unroll
        for (int i = 0; i < 7; ++i) rhs[i] = src[i];
    }
    se3_mul(est, rhs, Tn);
    __builtin_amdgcn_wave_barrier();
    if (lane < 2) {
        double *dst = sT + (lane == 1 ? 7 : 0);
#pragma unroll
        for (int i = 0; i < 7; ++i) dst[i] = Tn[i];
    }
    double Rn[9];
    quat_to_mat(Tn, Rn);
    LOOP_STAMP_SOLVER2(it, 2);
    double nrm = sqrt(SAGE_SQNORM6(x));
    if (!(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] < 9.0) || fabs(nrm - kEstimationThreshold) < 1e-12) {
        double lg[6];                                      // see solve_and_publish
        se3_log(est, lg);
        nrm = sqrt(SAGE_SQNORM6(lg));
    }
    LOOP_STAMP_SOLVER2(it, 3);
    const bool converged = nrm < kEstimationThreshold;
    unsigned done = (converged || it + 1 >= L.max_iterations) ? 1u : 0u;
    // (under a communicator an overflow on this rank alone must not end its loop: the peers would wait
    // for its sums; the flag is raised and the host reports it when the loop has ended everywhere)
    if (overflow && !L.shared_loop) done = 1u;
    if (exchange_failed) done = 1u;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) pub[i] = Rn[i];
        pub[9] = Tn[4]; pub[10] = Tn[5]; pub[11] = Tn[6];
    }
    __builtin_amdgcn_wave_barrier();
    LOOP_STAMP_SOLVER(it, 2);
    // 4. publish: 24 halves of R, t and the done word, each with its tag — before the bookkeeping below: the grid waits
    // for these words, nobody for the history (and the wait that follows would otherwise sit out those stores' round trip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the clears of step 1, issued microseconds ago)
    if (lane < 24) st_agent(&sh->pose[lane], (tag << 32) | reinterpret_cast<const uint32_t *>(pub)[lane]);
    if (lane == 24) st_agent(&sh->pose[24], (tag << 32) | done);
    if (lane == 0) {
        if (it < kHistory) st->n_corr[it] = static_cast<uint32_t>(S[kCount]);
        if (overflow) st->acc_overflow = 1;
        if (exchange_failed) st->exchange_failed = 1;
        if (done) {
            // the final loop state, for the host (ordinary stores: the end of the kernel publishes them)
#pragma unroll
            for (int i = 0; i < 7; ++i) st->T[i] = Tn[i];
#pragma unroll
            for (int i = 0; i < 9; ++i) st->R[i] = Rn[i];
            st->last_step_norm = nrm;
            st->iter = it + 1;
            st->done = 1;
            st->converged = (converged && !exchange_failed) ? 1 : 0;
        }
    }
    if (done && lane == 1) {
#pragma unroll
        for (int i = 0; i < 7; ++i) st->T_icp[i] = Tn[i];
    }
    if (L.progress && lane == 0)       // (chained launches: the host keeps a few launches enqueued ahead and stops at `done`)
        __hip_atomic_store(&L.progress->word, (static_cast<unsigned long long>(done ? 1u : 0u) << 32) | static_cast<unsigned long long>(it + 1),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (done && lane < kNumSums) st->sums[lane] = S[lane];
    LOOP_STAMP_SOLVER(it, 3);
    return done;
}

// The solving wave: one workgroup of one wave, launched on its own stream beside k_loop's grid (the
// solve needs ~120 registers, the search 72: in one kernel every wave would pay for the solver).
struct SolveArgs {
    LoopParams L;
    P2pParams X;
};
template <int COPIES>      // (two kernels: the one beside k_loop keeps its registers — 157, the grid's residency margin was measured with it)
__global__ __launch_bounds__(64) void k_loop_solve(SolveArgs A) {
    __shared__ SolveLds m;
    __builtin_amdgcn_s_setprio(3);             // (the grid waits for this wave: its SIMD's other waves can)
    {
        // Launched before the frame is even sorted, so that this wave holds its registers when the grid
        // of k_loop fills the machine; it waits here until the grid's first workgroup says that the
        // shared block has been zeroed and the loop has started (LoopShared::go == this call's epoch).
        const int lane = static_cast<int>(threadIdx.x & 63u);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned long long g = ld_agent(&A.L.sh->go[0]);
            if ((g & 0x7FFFFFFFFFFFFFFFull) == A.L.epoch) {
                if (g >> 63) return;                        // (sort.hip found a non-finite point: the host reports it)
                break;
            }
            // (the sort, the upload of a frame and a mirror refresh precede the grid: seconds, not the
            // microseconds of the waits inside the loop)
            if (__builtin_amdgcn_s_memrealtime() - t0 > 1000ull * A.L.timeout_ticks + 1000000000ull) {
                if (lane == 0) A.L.st->loop_aborted = 1;
                return;
            }
            __builtin_amdgcn_s_sleep(32);
        }
        if (lane < 14) m.T[lane] = lane < 7 ? A.L.T0[lane] : (lane == 10 ? 1.0 : 0.0);     // T | T_icp = identity (x, y, z, w | t)
    }
    unsigned long long xg = A.X.nranks > 1 ? *A.X.exchanges : 0ull;
    __builtin_amdgcn_wave_barrier();
    for (int it = 0;; ++it) {
        // (the arguments are re-read from the kernel-argument segment every iteration: see k_loop)
        auto ka = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const SolveArgs &K = *(const SolveArgs *)(ka);
        if (loop_finish_iteration<COPIES>(K.L, K.X, m, it, xg)) return;
    }
}

#ifndef SAGE_LOOP_POLL_SLEEP
#define SAGE_LOOP_POLL_SLEEP 8     // x 64 clocks between two looks of a workgroup at the pose granules
#endif
// (Tried and dropped, profiles/r05/mix_ab_*.txt: a workgroup with more units of queries than waves registering
// PAIRS of units at half the lanes per query, one wave per pair, so that every wave makes one pass — bit-identical,
// the sums being exact from the blocks of four queries on, and slower: 37.5 against 34.6 us per iteration on c2.  A
// wave's pass lasts as long as its lanes have points to look at: two units at half the lanes are two passes' worth.)
template <int LW, bool FILT>
__global__ __launch_bounds__(64 * kLoopMaxWaves) __attribute__((amdgpu_waves_per_eu(SAGE_LOOP_OCC, 8)))
void k_loop(LoopArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int QW = 64 >> LW;
    const IcpParams &P = A.P;
    const LoopParams &L = A.L;
    const int wv = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int nw = L.nw;
    const unsigned gpw = static_cast<unsigned>(L.gpw);
    double *s_pose = reinterpret_cast<double *>(smem + kLpPose);

    {
        // (sort.hip found a non-finite point: nobody starts, the host reports it — except under a communicator,
        // where the ranks must keep exchanging in step)
        const bool bad = !L.shared_loop && P.st->bad_input;
        if (blockIdx.x == 0 && threadIdx.x == 0)
            st_agent(&L.sh->go[0], L.epoch | (bad ? 0x8000000000000000ull : 0ull));      // the solving wave may start
        if (bad) return;
    }

#ifdef SAGE_LOOP_INGRID
    // Counter-collection twin (profiles/run_profiles.sh builds it as a variant library): rocprofv3 --pmc runs one
    // kernel at a time, which the grid and its solving wave — two kernels that talk to each other — do not survive.
    // Here the solving wave is one more workgroup of THIS grid (its path spills under the search's register budget:
    // the twin is for bytes and instruction counts, not for time).
    if (blockIdx.x == gridDim.x - 1u) {
        if (threadIdx.x >= 64u) return;
        SolveLds &m = *reinterpret_cast<SolveLds *>(smem);
        const int lane = static_cast<int>(threadIdx.x & 63u);
        if (lane < 14) m.T[lane] = lane < 7 ? L.T0[lane] : (lane == 10 ? 1.0 : 0.0);
        __builtin_amdgcn_wave_barrier();
        P2pParams X{};
        X.nranks = 1;
        unsigned long long xg = 0ull;
        for (int it = 0;; ++it)
            if (loop_finish_iteration<kLoopReplicas>(L, X, m, it, xg)) return;
    }
#endif

    // ---- the units (of QW queries) this workgroup owns for the whole call ----------------------------
    // Workgroup b is dispatched to XCD b % 8 (observed; speed only).  Striped: XCD x serves the stripes
    // x, x + 8, ... of kLoopStripe workgroups' worth of the spatially sorted frame (every XCD gets the
    // same mix of dense and sparse regions, every L2 sees the whole map).  Contiguous: XCD x serves the
    // units [xcd_first[x], xcd_first[x + 1]) — one compact region of the map per L2, the boundaries
    // chosen by the host so that the XCDs hold equal work.
    // Either way the units are dealt out EVENLY over the workgroups that serve them — floor or ceil of
    // units / workgroups each, never more than gpw: a frame of 7,500 units on 1,664 resident workgroups
    // of four waves gives 844 of them a fifth unit instead of leaving 200 with none.
    unsigned g0, gcnt;
    {
        const unsigned xcd = blockIdx.x & 7u, jb = blockIdx.x >> 3;
        const unsigned ngroups = (static_cast<unsigned>(P.n) + QW - 1u) / QW;
        unsigned lo = 0u, cnt = ngroups, idx, nwg;
        if (L.contiguous == 1) {
            lo = L.xcd_first[xcd];
            cnt = L.xcd_first[xcd + 1u] - lo;
            idx = jb;
            nwg = static_cast<unsigned>(L.wgs) >> 3;
        } else {
            idx = ((jb / kLoopStripe) * 8u + xcd) * kLoopStripe + (jb % kLoopStripe);
            nwg = static_cast<unsigned>(L.wgs);
        }
        const unsigned base = cnt / nwg, extra = cnt - base * nwg;      // `extra` workgroups serve base + 1 groups
        g0 = lo + idx * base + min(idx, extra);
        gcnt = min(gpw, base + (idx < extra ? 1u : 0u));
    }
    unsigned long long *wgacc = reinterpret_cast<unsigned long long *>(smem + kLpAcc);
    // LDS after the header: perm[gpw * QW / 4] | work[gpw * QW / 4] | rows of the workgroup's queries | their state
    // records | one scratch for the transposed block sums per wave
    constexpr unsigned BPW = QW / 4;                                  // blocks of four queries per pass
    const unsigned nblk_max = gpw * BPW, nblk = gcnt * BPW;
    uint32_t *perm = smem + kLpHeaderWords;
    uint32_t *work = perm + loop_perm_words(nblk_max) / 2u;
    uint32_t *rows = perm + loop_perm_words(nblk_max);
    uint32_t *state = rows + gpw * QW * kRowLdsStride;
    double *red = reinterpret_cast<double *>(state + gpw * QW * kLoopStateWords + static_cast<unsigned>(wv) * loop_red_words());

    // ---- set-up: the initial pose, the state records of the groups' queries --------------------------
    if (threadIdx.x < 9) s_pose[threadIdx.x] = P.st->R[threadIdx.x];
    else if (threadIdx.x < 12) s_pose[threadIdx.x] = P.st->T[4 + threadIdx.x - 9];
    if (threadIdx.x == 0) {
        smem[kLpArrive] = 0u;
        smem[kLpNext] = 0u;
        smem[kLpDone] = 0u;
#ifdef SAGE_LOOP_TIMING
        smem[kLpDbg] = 0u; smem[kLpDbg + 1] = 0u; smem[kLpDbg + 2] = 0u;
#endif
    }
    if (L.deal && (threadIdx.x & 63u) == 0u) {
        // The heaviest unit of a workgroup (its blocks are ordered by work) should not meet the heaviest units of the
        // other workgroups of its CU on one SIMD: a workgroup's four waves sit on the four SIMDs, one wave of every
        // workgroup of the CU per SIMD, and a SIMD's issue slots are what its waves share.  Workgroup r of the CU
        // (its waves' slot number) hands unit (s + r) mod waves to its wave on SIMD s: every SIMD gets the same mix.
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        smem[kLpFirst + static_cast<unsigned>(wv)] = (((hw >> 4) & 3u) + (hw & 15u)) % static_cast<unsigned>(nw);
    }
    for (unsigned i = threadIdx.x; i < 2u * kWgAccWords; i += blockDim.x) smem[kLpAcc + i] = 0u;
    for (unsigned i = threadIdx.x; i < nblk_max; i += blockDim.x) {
        perm[i] = i;                           // the order of the frame, until the blocks' work is known
        work[i] = 0u;
    }
    for (unsigned sl = threadIdx.x; sl < gcnt * QW; sl += blockDim.x) {
        const unsigned q = g0 * QW + sl;       // slot sl of this workgroup
        const Point4 f = P.frame[q < static_cast<unsigned>(P.n) ? q : 0u];
        {
            uint32_t *lst = state + sl * kLoopStateWords;
            *reinterpret_cast<Point4 *>(lst) = f;
            Point4 z;
            z.x = z.y = z.z = z.l = 0.0;
            *reinterpret_cast<Point4 *>(lst + 8) = z;
            *reinterpret_cast<uint4 *>(lst + kStPrev) = make_uint4(0xFFFFFFFFu, 0u, static_cast<uint32_t>(kNoVoxel),
                                                                   static_cast<uint32_t>(kNoVoxel));   // no answer, no row yet:
            *reinterpret_cast<uint4 *>(lst + kStPrev + 4) = make_uint4(static_cast<uint32_t>(kNoVoxel), 0u, 0u, 0u);   // the first pass builds it
        }
    }
    __syncthreads();
    if (L.deal) {
        // (two waves of a workgroup on one SIMD would ask for the same unit: the first keeps it, the others take what is
        // left — every wave derives the same table, wave 0 stores it)
        unsigned claimed = 0u, kept = 0u, table[kLoopMaxWaves];
        for (int w2 = 0; w2 < nw; ++w2) {
            const unsigned pr = smem[kLpFirst + static_cast<unsigned>(w2)];
            table[w2] = pr;
            if (!((claimed >> pr) & 1u)) { claimed |= 1u << pr; kept |= 1u << w2; }
        }
        for (int w2 = 0; w2 < nw; ++w2)
            if (!((kept >> w2) & 1u)) {
                const unsigned pr = static_cast<unsigned>(__builtin_ctz(~claimed));
                table[w2] = pr;
                claimed |= 1u << pr;
            }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w2 = 0; w2 < nw; ++w2) smem[kLpFirst + static_cast<unsigned>(w2)] = table[w2];
            smem[kLpNext] = static_cast<unsigned>(nw);
        }
        __syncthreads();
    }
#ifdef SAGE_LOOP_TIMING
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_wait = 0, t_close = 0, n_pass = 0;
#endif

    for (int it = 0;; ++it) {
        // Every iteration (and every pass of the body) re-reads its arguments from the kernel-argument
        // segment — scalar loads from the constant cache, as a wave of k_icp does at its start — and
        // re-derives its lane index: values the compiler knows to be invariant across this loop it would
        // hoist out of it and keep alive, ~50 scalars and a dozen vector registers the scan needs.
        auto ka = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const LoopArgs &K = *(const LoopArgs *)(ka);
        const LoopParams &L = K.L;
        LoopShared *sh = L.sh;
        unsigned lane_u;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_u));
        const int lane = static_cast<int>(lane_u);
        // the workgroup's groups, first come first served: a wave held up by a heavy query takes fewer
        bool dealt = L.deal != 0;
        for (;;) {
            unsigned gi = 0u;
            if (dealt) {
                // (this wave's first unit is fixed by where it sits; the units beyond one per wave go first come first served)
                dealt = false;
                gi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(smem[kLpFirst + static_cast<unsigned>(wv)])));
                if (gi >= gcnt) continue;
            } else {
                if (lane == 0)
                    gi = __hip_atomic_fetch_add(&smem[kLpNext], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                gi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(gi)));
                if (gi >= gcnt) break;
            }
            if (L.prio) {
                // A SIMD's issue slots go to its waves by priority: the wave with the heaviest unit of its workgroup (the
                // units are ordered by last iteration's work) runs at the highest, the lightest at the lowest, a unit beyond
                // one per wave — it starts late — at `prio` (3 by default).  The work of a SIMD does not change with the order, but
                // its END does: the long chains run while there is other work to fill their stalls with, and what
                // is left to run alone at the end of an iteration are the short ones (c2: 31.2 -> 28.2 us per iteration,
                // profiles/r06/deal_ab.txt).
                const unsigned rk = gi >= static_cast<unsigned>(nw) ? static_cast<unsigned>(L.prio) : 3u - min(gi, 3u);
                switch (rk) {
                    case 0: __builtin_amdgcn_s_setprio(0); break;
                    case 1: __builtin_amdgcn_s_setprio(1); break;
                    case 2: __builtin_amdgcn_s_setprio(2); break;
                    default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
#ifdef SAGE_LOOP_TIMING
            if (gi >= static_cast<unsigned>(nw)) LOOP_STAMP_WG(it, 2);
            const unsigned long long t_unit = __builtin_amdgcn_s_memrealtime();
#endif
            LoopGroup G;
            G.rows = rows;
            G.state = state;
            G.perm = perm;
            G.work = work;
            G.unit = gi;
            G.red = red;
            G.wgacc = wgacc;
            G.q_first = g0 * QW;
            G.slot = g0 + gi;
#ifdef SAGE_LOOP_TIMING
            for (int i = 0; i < 8; ++i) G.ph[i] = 0;
            G.tprev = __builtin_amdgcn_s_memtime();
            ++n_pass;
#endif
            {
                auto kb = __builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(kb));
                icp_body<LW, true, FILT, true>(((const LoopArgs *)(kb))->P, smem, &G, s_pose);
            }
#ifdef SAGE_LOOP_TIMING
            if (gi >= static_cast<unsigned>(nw)) LOOP_STAMP_WG(it, 3);
            if (it < kLoopTimedIters && blockIdx.x < kLoopTimedWgs && lane == 0 && wv < 8 && gi < static_cast<unsigned>(nw)) {
                unsigned hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                g_loop_wave[it][blockIdx.x][wv][0] = __builtin_amdgcn_s_memrealtime();
                g_loop_wave[it][blockIdx.x][wv][1] = (t_unit << 24) | (static_cast<unsigned long long>(gi & 0xFFu) << 16) | (hw & 0xFFFFu);
            }
#endif
#ifdef SAGE_LOOP_TIMING
            for (int i = 0; i < 8; ++i) ph[i] += G.ph[i];
#endif
        }
#ifdef SAGE_LOOP_TIMING
        const unsigned long long t_a = __builtin_amdgcn_s_memtime();
#endif
        // (what the ticket orders — the groups' sums — lives in LDS, which serves a CU's waves in order)
        unsigned prior = 0u;
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (lane == 0)
            prior = __hip_atomic_fetch_add(&smem[kLpArrive], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        prior = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(prior)));
        const bool last = prior == static_cast<unsigned>(nw) - 1u;
        if (last) {
            // this wave closes the workgroup's iteration
            wgacc_flush<true>(wgacc, &sh->acc[it & 1][blockIdx.x & (kLoopReplicas - 1)][0], &sh->acc[it & 1][0][kAccWords - 1]);
            if (lane == 0) {                   // everybody is in: ready for the next iteration
                smem[kLpArrive] = 0u;
                smem[kLpNext] = L.deal ? static_cast<unsigned>(nw) : 0u;
            }
            // The next iteration's order of the workgroup's blocks: heaviest first, by what they cost in this one (a
            // rank sort on one wave: lane i counts the blocks that go before block i).  Blocks of like work then
            // share a wave — whose pass lasts as long as its heaviest query — and the heaviest waves start first.
            // (Which blocks share a wave does not reach the sums: they are exact from the block on.)
            if (nblk <= 64u && nblk > BPW) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const unsigned i = static_cast<unsigned>(lane);
                const unsigned wi = i < nblk ? work[i] : 0u;
                unsigned rank = 0u;
                for (unsigned j = 0; j < nblk; ++j) {
                    const unsigned wj = work[j];
                    rank += (wj > wi || (wj == wi && j < i)) ? 1u : 0u;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (i < nblk) {
                    perm[rank] = i;
                    work[i] = 0u;
                }
            }
#ifdef SAGE_LOOP_TIMING
            if (lane == 0 && it < kLoopTimedIters && blockIdx.x < kLoopTimedWgs) {
                unsigned hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                unsigned xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                unsigned *o = g_loop_wginfo[it][blockIdx.x];
                o[0] = (xcc << 28) | (hw & 0x0FFFFFFFu);
                o[1] = smem[kLpDbg]; o[2] = smem[kLpDbg + 1]; o[3] = smem[kLpDbg + 2];
                smem[kLpDbg] = 0u; smem[kLpDbg + 1] = 0u; smem[kLpDbg + 2] = 0u;
            }
#endif
            LOOP_STAMP_WG(it, 0);
        }
        if (wv == 0) {
            // the next pose, for this workgroup
            const unsigned long long tag = static_cast<unsigned long long>(it) + 1ull;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned long long g = tag << 32;
            bool aborted = false;
            for (;;) {
                if (lane < kLoopPoseGranules) g = ld_agent(&sh->pose[lane]);
                const bool ok = (g >> 32) == tag;
                if (__all(ok)) break;
                unsigned long long ab = 0ull;
                if (lane == 0) ab = ld_agent(&sh->abort_word[0]);
                const bool late = __builtin_amdgcn_s_memrealtime() - t0 > L.timeout_ticks;
                if (__any(ab != 0ull) || late) {
                    if (lane == 0 && late) {
                        st_agent(&sh->abort_word[0], 1ull);
                        L.st->loop_aborted = 1;
                    }
                    aborted = true;
                    break;
                }
                // (more than a thousand workgroups wait here for most of an iteration, all on the same four
                // cache lines: a pass every ~0.3 us each leaves the L2 channel that serves them — and the
                // accumulators the solving wave is reading — alone)
                __builtin_amdgcn_s_sleep(SAGE_LOOP_POLL_SLEEP);
                // (a big grid backs off twice as long: c2's 1,664 workgroups 30.7 -> 30.3 us per iteration, flat from there
                // to six times as long; c1's 640 prefer the short one — same-box A/B, profiles/r05/poll_sleep.txt)
                if (L.wgs > 1024) __builtin_amdgcn_s_sleep(SAGE_LOOP_POLL_SLEEP);
            }
            if (aborted) {
                if (lane == 0) smem[kLpDone] = 2u;
            } else {
                if (lane < 24) reinterpret_cast<uint32_t *>(s_pose)[lane] = static_cast<uint32_t>(g);
                if (lane == 24) smem[kLpDone] = static_cast<uint32_t>(g);
            }
            LOOP_STAMP_WG(it, 1);
        }
#ifdef SAGE_LOOP_TIMING
        const unsigned long long t_b = __builtin_amdgcn_s_memtime();
        if (last) t_close += t_b - t_a;
#endif
        __syncthreads();
#ifdef SAGE_LOOP_TIMING
        t_wait += __builtin_amdgcn_s_memtime() - t_b;
#endif
        if (smem[kLpDone]) break;
    }
#ifdef SAGE_LOOP_TIMING
    if ((threadIdx.x & 63u) == 0u) {
        for (int i = 0; i < 8; ++i) atomicAdd(&g_loop_phase[i], ph[i]);
        atomicAdd(&g_loop_phase[8], t_wait);
        atomicAdd(&g_loop_phase[9], t_close);
        atomicAdd(&g_loop_phase[10], n_pass);
    }
#endif
}

// ------------------------------------------------------------------------------------ k_gn
// AlignClouds' accumulation (Registration.cpp:62-90) on explicit pairs: every pair is taken.
__global__ __launch_bounds__(256) void k_gn(GnParams P) {
    // block reduction scratch: component c of thread t at red[c][(t >> 4) * 17 + (t & 15)]
    // (rows of 16 values padded to 17 doubles: conflict-free for both the write and the read)
    __shared__ double red[kCount][16 * 17];
    double acc[kCount];
#pragma unroll
    for (int i = 0; i < kCount; ++i) acc[i] = 0.0;
    const double k = P.kernel;
    const double k2 = k * k;
    const int stride = gridDim.x * 256;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < P.n; q += stride) {
        const Point4 s = P.src[q], g = P.tgt[q];
        const double sx = s.x, sy = s.y, sz = s.z;
        const double rx = sx - g.x, ry = sy - g.y, rz = sz - g.z;
        const double r2 = SAGE_SQNORM3_RESID(rx * rx, ry * ry, rz * rz);
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
    }
    // fixed-order block reduction: every thread parks its 16 sums in LDS, then 16 threads per
    // component each add 16 parked values and finish with four DPP exchange steps
    const int t = static_cast<int>(threadIdx.x);
    const int slot = (t >> 4) * 17 + (t & 15);
#pragma unroll
    for (int c = 0; c < kCount; ++c) red[c][slot] = acc[c];
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
    if (t == 0) {
        const int lo = blockIdx.x * 256;
        int cnt = 0;                         // pairs this block owns (grid-stride)
        for (int q = lo; q < P.n; q += stride) cnt += min(256, P.n - q);
        out[kCount] = static_cast<double>(cnt);
    }
    if (t > kCount && t < kNumSums) out[t] = 0.0;
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
extern "C" void sageicp_debug_nn_spans(unsigned long long *out, unsigned nwaves, int next_iter) {
    // raw {start, end, HW_ID, pairs of lane 0} of the first `nwaves` waves of the k_icp launch of
    // the iteration chosen by the previous call; `next_iter` chooses the one the next loop records
    if (nwaves > kNnTimingSlots) nwaves = kNnTimingSlots;
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_span), 4ull * nwaves * sizeof(unsigned long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nn_span_iter), &next_iter, sizeof(int));
}
extern "C" void sageicp_debug_nn_raw(unsigned long long *out, unsigned nwaves) {
    // per wave slot, summed over the launches since the last reset: 5 phases and the lifetime
    // (shader cycles), the lifetime in 100-MHz ticks, launches
    if (nwaves > kNnTimingSlots) nwaves = kNnTimingSlots;
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_phase), 8ull * nwaves * sizeof(unsigned long long));
}
extern "C" void sageicp_debug_nn_phases(unsigned long long out[16], int reset) {
    // out: [0..4] summed cycles of the five phases (loads, row, home scan, rest of the search,
    // epilogue), [5] summed wave lifetime (shader cycles), [6] the same in 100-MHz ticks, [7] waves,
    // [8] the slowest wave slot's mean lifetime (cycles), [9] slots used
    std::vector<unsigned long long> h(8ull * kNnTimingSlots);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_nn_phase), h.size() * sizeof(unsigned long long));
    for (int i = 0; i < 16; ++i) out[i] = 0;
    for (unsigned s = 0; s < kNnTimingSlots; ++s) {
        const unsigned long long *t = &h[8ull * s];
        if (!t[7]) continue;
        for (int k = 0; k < 8; ++k) out[k] += t[k];
        if (t[5] / t[7] > out[8]) out[8] = t[5] / t[7];
        ++out[9];
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

// sums of k_icp's per-wave counters {candidates in the neighbourhood, pairs handed out} into the
// loop state (one workgroup; it rides on the state copy the host makes anyway)
__global__ __launch_bounds__(1024) void k_sum_counters(const unsigned long long *c, int n,
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
void launch_sum_counters(const unsigned long long *c, int n, IcpState *st, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_counters, dim3(1), dim3(1024), 0, s, c, n, st);
}
void launch_scatter_points(const uint32_t *idx, const Point4 *vals, uint32_t n, Point4 *pts,
                           hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_scatter_points, dim3((n + 255) / 256), dim3(256), 0, s, idx, vals, n, pts);
}
__global__ __launch_bounds__(256) void k_scatter_u32(const uint32_t *idx, const uint32_t *vals, uint32_t n,
                                                     uint32_t *dst) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[idx[i]] = vals[i];
}
void launch_scatter_u32(const uint32_t *idx, const uint32_t *vals, uint32_t n, uint32_t *dst, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_scatter_u32, dim3((n + 255) / 256), dim3(256), 0, s, idx, vals, n, dst);
}
void launch_scatter_slots(const uint32_t *idx, const Slot *vals, uint32_t n, Slot *table,
                          hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_scatter_slots, dim3((n + 255) / 256), dim3(256), 0, s, idx, vals, n, table);
}

int icp_stripes_for(int n, int lw) { return icp_blocks_for(n, lw) / SAGE_ICP_STRIPE; }
int icp_blocks_for(int n, int lw) {
    // one wave per 64 >> lw queries, kIcpWavesPerBlock waves per workgroup, rounded up to whole
    // stripes on all 8 XCDs
    const long qw = 64 >> lw;
    const long waves = (static_cast<long>(n) + qw - 1) / qw;
    const long blocks = (waves + kIcpWavesPerBlock - 1) / kIcpWavesPerBlock;
    const long per_round = 8L * SAGE_ICP_STRIPE;
    const long r = ((blocks + per_round - 1) / per_round) * per_round;
    return static_cast<int>(r < per_round ? per_round : r);
}
size_t icp_lds_bytes(int lw) {
    return sizeof(uint32_t) * (kWgHeaderWords + kIcpWavesPerBlock * icp_wave_words(lw));
}

void launch_rows(const IcpParams &p, hipStream_t s) {
    if (p.n <= 0) return;
    hipLaunchKernelGGL(k_rows, dim3((p.n + 7) / 8), dim3(256), 0, s, p);
}

template <int LW>
static void launch_icp_lw(const IcpParams &p, bool fused, hipStream_t s) {
    const int grid = icp_blocks_for(p.n, LW);
    const size_t lds = icp_lds_bytes(LW);
    const dim3 g(grid), b(64 * kIcpWavesPerBlock);
    // flat order: always with 8 and 16 lanes per query, never with one (it is the same thing), on request between
    constexpr bool kBoth = LW == 1 || LW == 2;
    const bool flat = LW >= 3 || (kBoth && p.flat);
    auto go = [&](auto flat_c) {
        constexpr bool F = decltype(flat_c)::value;
        if (p.filter) {
            if (fused) hipLaunchKernelGGL((k_icp<LW, true, true, F>), g, b, lds, s, p);
            else hipLaunchKernelGGL((k_icp<LW, false, true, F>), g, b, lds, s, p);
        } else {
            if (fused) hipLaunchKernelGGL((k_icp<LW, true, false, F>), g, b, lds, s, p);
            else hipLaunchKernelGGL((k_icp<LW, false, false, F>), g, b, lds, s, p);
        }
    };
    if constexpr (kBoth) {
        if (flat) go(std::true_type{});
        else go(std::false_type{});
    } else if constexpr (LW >= 3) {
        go(std::true_type{});
    } else {
        go(std::false_type{});
    }
}
void launch_icp(const IcpParams &p, int lw, bool fused, hipStream_t s) {
    if (p.n <= 0) return;
    switch (lw) {
        case 0: launch_icp_lw<0>(p, fused, s); break;
        case 1: launch_icp_lw<1>(p, fused, s); break;
        case 2: launch_icp_lw<2>(p, fused, s); break;
        case 3: launch_icp_lw<3>(p, fused, s); break;
        default: launch_icp_lw<4>(p, fused, s); break;
    }
}

size_t loop_lds_bytes(int lw, int nw, int gpw) {
    return sizeof(uint32_t) * (kLpHeaderWords + loop_perm_words(static_cast<unsigned>(gpw) * ((64u >> lw) / 4u)) +
                               static_cast<size_t>(gpw) * loop_group_words(lw) + static_cast<size_t>(nw) * loop_red_words());
}
// the kernel of a shape, as an untyped pointer (what the occupancy query and the launch take)
static const void *loop_kernel(int lw, bool filter) {
#define SAGE_LOOP_K(LW_, F_) reinterpret_cast<const void *>(&k_loop<LW_, F_>)
    switch (lw) {
        case 1: return filter ? SAGE_LOOP_K(1, true) : SAGE_LOOP_K(1, false);
        case 2: return filter ? SAGE_LOOP_K(2, true) : SAGE_LOOP_K(2, false);
        case 3: return filter ? SAGE_LOOP_K(3, true) : SAGE_LOOP_K(3, false);
        case 4: return filter ? SAGE_LOOP_K(4, true) : SAGE_LOOP_K(4, false);
        default: return nullptr;
    }
#undef SAGE_LOOP_K
}
int loop_blocks_per_cu(int lw, bool filter, int nw, size_t lds) {
    if (nw < 1 || nw > kLoopMaxWaves || lds > 160 * 1024) return 0;
    const void *k = loop_kernel(lw, filter);
    if (!k) return 0;
    // (a workgroup that owns many units can ask for more than the 64 KB a kernel gets by default)
    if (lds > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 0;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64 * nw, lds) != hipSuccess) return 0;
    return nb;
}
void launch_loop(const IcpParams &p, const LoopParams &l, int lw, hipStream_t s) {
    const size_t lds = loop_lds_bytes(lw, l.nw, l.gpw);
    LoopArgs a;
    a.P = p;
    a.L = l;
    void *args[] = {&a};
#ifdef SAGE_LOOP_INGRID
    (void)hipLaunchKernel(loop_kernel(lw, p.filter != 0), dim3(l.wgs + 1), dim3(64 * l.nw), args, lds, s);
#else
    (void)hipLaunchKernel(loop_kernel(lw, p.filter != 0), dim3(l.wgs), dim3(64 * l.nw), args, lds, s);
#endif
}
void launch_loop_solve(const LoopParams &l, const P2pParams &x, hipStream_t s) {
    SolveArgs a;
    a.L = l;
    a.X = x;
    if (l.copies == kChainReplicas) hipLaunchKernelGGL(k_loop_solve<kChainReplicas>, dim3(1), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_loop_solve<kLoopReplicas>, dim3(1), dim3(64), 0, s, a);
}

int launch_gn(const GnParams &p, hipStream_t s) {
    long blocks = (static_cast<long>(p.n) + 255) / 256;
    if (blocks > 128) blocks = 128;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_gn, dim3(static_cast<int>(blocks)), dim3(256), 0, s, p);
    return static_cast<int>(blocks);
}

void launch_fin(const FinParams &p, hipStream_t s) {
    hipLaunchKernelGGL(k_fin, dim3(1), dim3(kFinThreads), 0, s, p);
}

void launch_tf(Point4 *pts, int n, const IcpState *st, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_tf, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, st);
}

}  // namespace sageicp
