// Internal layout shared by the host map, the device mirror and the HIP kernels.
// gfx950 only.  Not part of the C ABI (see include/sageicp.h for that).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SAGE_HD __host__ __device__
#else
#define SAGE_HD
#endif

// Every squared norm of a 3-vector on the path is a three-term sum whose association Eigen chooses,
// and the nearest-neighbour decision is a strict `<` on such sums: 1 ulp decides exact near-ties.
// Eigen cannot be compiled in this image; the orders below are DERIVED from Eigen 3.4's Redux.h
//  and selected by ONE switch shared in name and meaning with the CPU checker:
//   SAGE_SQNORM3_ORDER = 2 (default)  what Eigen 3.4 evaluates on an SSE2 / NEON build, per call site:
//       NN     (v3neighbor - v3point).squaredNorm()          VoxelHashMap.cpp:87    (x^2 + y^2) + z^2
//       RESID  residual.squaredNorm()                        Registration.cpp:79    (x^2 + y^2) + z^2
//       FAR    (pt - origin).squaredNorm()                   VoxelHashMap.cpp:178   (x^2 + y^2) + z^2
//              — fixed-size-3 double expressions with packet access: linear vectorised reduction,
//              predux(packet(e0, e1)) first, then + e2 (redux_impl<LinearVectorizedTraversal, CompleteUnrolling>);
//       ACCEPT (closest - point).head<3>().norm()            VoxelHashMap.cpp:111   (x^2 + y^2) + z^2
//              — a Block of an expression: no direct access, but evaluator<Block> keeps the packet bit when
//              the block has the storage order of its argument (a column segment of a column expression
//              does), so it reduces like the others (round 4 had x^2 + (y^2 + z^2) here)
//   0  x^2 + (y^2 + z^2) everywhere (the default of rounds 1-3)      1  (x^2 + y^2) + z^2 everywhere
// build.py's build_sqnorm3_variant() builds order 0 as libsageicp_hip.v0.so; SAGE_SQNORM3_ORDER=0 in the
// environment makes the loader and the test-suite use it against the checker built the same way.
#ifndef SAGE_SQNORM3_ORDER
#define SAGE_SQNORM3_ORDER 2
#endif
#define SAGE_SQNORM3_A(xx, yy, zz) ((xx) + ((yy) + (zz)))
#define SAGE_SQNORM3_B(xx, yy, zz) (((xx) + (yy)) + (zz))
#if SAGE_SQNORM3_ORDER == 0
#define SAGE_SQNORM3_NN SAGE_SQNORM3_A
#define SAGE_SQNORM3_RESID SAGE_SQNORM3_A
#define SAGE_SQNORM3_FAR SAGE_SQNORM3_A
#define SAGE_SQNORM3_ACCEPT SAGE_SQNORM3_A
#define SAGE_SQNORM3_CROP SAGE_SQNORM3_A
#elif SAGE_SQNORM3_ORDER == 1
#define SAGE_SQNORM3_NN SAGE_SQNORM3_B
#define SAGE_SQNORM3_RESID SAGE_SQNORM3_B
#define SAGE_SQNORM3_FAR SAGE_SQNORM3_B
#define SAGE_SQNORM3_ACCEPT SAGE_SQNORM3_B
#define SAGE_SQNORM3_CROP SAGE_SQNORM3_B
#else
#define SAGE_SQNORM3_NN SAGE_SQNORM3_B
#define SAGE_SQNORM3_RESID SAGE_SQNORM3_B
#define SAGE_SQNORM3_FAR SAGE_SQNORM3_B
#define SAGE_SQNORM3_ACCEPT SAGE_SQNORM3_B
#define SAGE_SQNORM3_CROP SAGE_SQNORM3_B
#endif
// point.head<3>().norm() of the range crop (Preprocessing.cpp:176): CROP, packet access -> (x^2 + y^2) + z^2.
// estimation.log().norm() (Registration.cpp:137), a 6-vector: three packets p0 + (p1 + p2), then the two
// lanes (order 2); summed left to right in orders 0 / 1.
#if SAGE_SQNORM3_ORDER == 2
#define SAGE_SQNORM6(a) ((((a)[0] * (a)[0]) + (((a)[2] * (a)[2]) + ((a)[4] * (a)[4]))) + \
                         (((a)[1] * (a)[1]) + (((a)[3] * (a)[3]) + ((a)[5] * (a)[5]))))
#else
#define SAGE_SQNORM6(a) ((((((a)[0] * (a)[0] + (a)[1] * (a)[1]) + (a)[2] * (a)[2]) + (a)[3] * (a)[3]) + \
                          (a)[4] * (a)[4]) + (a)[5] * (a)[5])
#endif

namespace sageicp {

// One open-addressed hash slot, 16 B.  `blk` packs (first storage unit of the voxel's points << 8) |
// point_count, so a probe returns where the candidates are and how many without a second load;
// kEmptySlot marks a free slot.  Linear
// probing, power-of-two capacity, load factor <= 0.25 (misses dominate the 27-voxel probe).
struct alignas(16) Slot {
    int32_t x, y, z;
    uint32_t blk;
};
constexpr uint32_t kEmptySlot = 0xFFFFFFFFu;
// device-side map update (map_update.hip): an evicted voxel's slot keeps the probe chain intact
// as a tombstone — a non-empty blk word and a key no stored voxel can have (|index| < 2^20)
constexpr uint32_t kTombstone = 0xFFFFFFFEu;
constexpr int32_t kTombKey = 0x7F7F7F7F;
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr int kMaxBlockBits = 24;   // voxels (blocks) < 2^24 - 2; the slot word keeps 24 bits for the unit and 8
                                    // for the count (the two top unit numbers would collide with the empty /
                                    // tombstone marks: kMaxUnits)
constexpr int kMaxCap = 255;        // basic + critical points per voxel
// Voxel storage: the point array is cut into units of kUnitPoints points (128 B); a voxel's points
// are one region of whole units whose size class follows its count (host_map.hpp).  regions[block] =
// (class << 28) | first unit; the search's rows carry (unit << 8) | count, hence 24 bits of units.
constexpr uint32_t kUnitPoints = 4;
constexpr uint32_t kMaxUnits = (1u << 24) - 2;
constexpr int kMaxClasses = 4;
constexpr uint32_t kNoRegion = 0xFFFFFFFFu;

// Any hash works (the reference's 20-bit hash, VoxelHashMap.hpp:72-77, only shapes bucket
// order, never results).  This one mixes all 96 key bits so linear-probe runs stay short.
SAGE_HD inline uint32_t voxel_hash(int32_t x, int32_t y, int32_t z) {
    uint32_t h = static_cast<uint32_t>(x) * 0x9E3779B1u;
    h ^= static_cast<uint32_t>(y) * 0x85EBCA77u + (h << 6) + (h >> 2);
    h ^= static_cast<uint32_t>(z) * 0xC2B2AE3Du + (h << 6) + (h >> 2);
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

// A map point: exactly Eigen::Vector4d's memory (x, y, z, label), 32 B, fp64 so that the
// device search reproduces the reference's fp64 comparisons index-for-index.
struct alignas(32) Point4 {
    double x, y, z, l;
};

constexpr int kMaxIterations = 500;        // Registration.cpp:96
constexpr double kEstimationThreshold = 1e-4;  // Registration.cpp:97
constexpr int kNumSums = 20;               // 16 closed-form GN sums + count + 3 pad
constexpr int kHistory = 512;

// Loop progress as the host sees it while the loop runs: one 64-bit word (done << 32) | iterations
// in pinned, host-mapped memory, written by the finishing lane after every iteration (a
// write-through store) and polled by run_icp to keep a few iterations enqueued ahead of the GPU
// without a stream synchronisation per check.
struct IcpProgress {
    unsigned long long word;               // (done << 32) | iterations completed
    unsigned long long pad_[7];
};

// Direct exchange of the Gauss-Newton sums between the GPUs of one node (multi-GPU, no RCCL
// launch): every rank owns one of these blocks in fine-grained device memory, exported to its
// peers through HIP IPC.  In exchange g (a counter that only ever grows) the finishing workgroup
// of rank r stores its 20 sums into slot g & 1, row r, of EVERY rank's block over xGMI, then the
// tag g + 1 into flag[r] there; it waits until all flags of its own block carry that tag and adds
// the rows in rank order, so every rank computes bit-identical sums.  A rank cannot run two
// exchanges ahead of a peer (it needs the peer's next tag first), so two slots are enough.
constexpr int kMaxRanks = 8;
struct P2pBlock {
    unsigned long long flag[kMaxRanks];            // written by rank i: the tag of its last exchange
    double sums[2][kMaxRanks][kNumSums];
    // written by rank i: the tag of the exchange at which it last GAVE UP its one-launch loop (a wait inside the launch
    // timed out): nobody's sums of that exchange are used, every rank counts it as made and registers the frame again
    // through the launch-per-iteration form.  Its own word, never overwritten by the tags of later exchanges: a peer that
    // looks late still finds it (a rank cannot give up twice before every peer has been through the first).
    unsigned long long abort_tag[kMaxRanks];
};
struct P2pParams {
    int nranks, rank;
    P2pBlock *block[kMaxRanks];                    // block[rank] is this rank's own
    unsigned long long *exchanges;                 // device counter g (the same on every rank)
    unsigned long long timeout_ticks;              // s_memrealtime ticks (100 MHz) before giving up
};

// Device-resident loop state, written by k_fin, read by every kernel of the next iteration.
struct IcpState {
    double T[7];        // cumulative pose applied to the pristine frame: T_icp * initial_guess
    double R[9];        // rotation matrix of T (row-major), refreshed with T
    double T_icp[7];    // product of the per-iteration estimates (Registration.cpp:135)
    double last_step_norm;
    int32_t iter;       // iterations completed
    int32_t done;       // 1: converged or hit kMaxIterations -> later launches are no-ops
    int32_t converged;
    int32_t peer_aborted;           // multi-GPU: a peer left its one-launch loop (a wait timed out there) and said so through the
                                    // exchange: every rank leaves the same exchange and registers the frame again, in step
    double sums[kNumSums];          // last reduced GN sums (diagnostics / multi-GPU exchange)
    unsigned long long sum_candidates;  // sum over launches and queries of C_q (roofline bytes)
    unsigned long long sum_pairs;       // (query, candidate) pairs k_icp actually scanned
    uint32_t n_corr[kHistory];      // accepted correspondences per iteration (all ranks)
    IcpProgress *progress;          // host-mapped progress block (nullptr: not published)
    int32_t exchange_failed;        // a peer's sums did not arrive in time (multi-GPU direct exchange)
    int32_t acc_overflow;           // a workgroup's sum did not fit the fixed-point accumulators (|value| >= 2^50)
    int32_t loop_aborted;           // k_loop: a wait inside the launch timed out (the host falls back to the launch-per-iteration loop)
    int32_t bad_input;              // a coordinate or label of the frame is not finite (sort.hip): the call fails with SAGEICP_ERR_INVALID
};

// Index into the 16 closed-form sums of AlignClouds (Registration.cpp:59-94):
//   JTJ = [[Sw I, -hat(Sws)], [hat(Sws), Sw(|s|^2 I - s s^T)]],  JTr = [Swr ; Sw(s x r)]
enum Sum : int {
    kW = 0,
    kWsx, kWsy, kWsz,
    kWxx, kWxy, kWxz, kWyy, kWyz, kWzz,
    kWrx, kWry, kWrz,
    kWcx, kWcy, kWcz,   // w * (s x r)
    kCount,
};

}  // namespace sageicp
