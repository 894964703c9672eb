// Launch interface between capi.hip (host side) and kernels.hip / sort.hip (device side).
#pragma once

#include <hip/hip_runtime.h>

#include "sageicp_types.h"

namespace sageicp {

// Neighbourhood row of one query: the 27 voxels around its home voxel as found in the map's hash,
// cached across the iterations of one RegisterFrame call (the map is constant during a call and
// the pose moves by millimetres per iteration, so a query's home voxel rarely changes).
//   [0..26]  (first unit of the voxel's region << 8) | count (a unit = 4 points of the point array;
//            kernels.hip row_word) of voxel (ox, oy, oz) + 1 =
//            (v / 9, v / 3 % 3, v % 3) — x outer, y, z inner, the reference's enumeration order
//            (VoxelHashMap.cpp:57-63) — or kEmptySlot
//   [27]     C_q: points stored in the neighbourhood (the algorithmic-bytes accounting)
//   [28..30] the home voxel the row was built for (kRowNeverBuilt after a re-sort)
//   [31]     occupancy mask: bit v set when voxel v holds points
constexpr int kRowWords = 32;
constexpr int kRowCq = 27, kRowKey = 28, kRowOcc = 31;
constexpr int kRowLdsStride = 36;          // words; 16-B pieces of 8 consecutive rows hit distinct banks

struct LoopShared;
struct IcpParams {
    const Point4 *frame;      // pristine (sorted) frame, or ready-made queries (apply_pose == 0)
    int n;
    const IcpState *st;       // pose to apply and the done flag
    int check_done;           // 1 inside the ICP loop: later launches of a finished loop are no-ops
    int apply_pose;
    double voxel_size;
    double inv_voxel_size;    // fl(1 / voxel_size): the fast path of the voxel index (kernels.hip, voxel_index); 0: always divide
    uint32_t *rows;           // [n][kRowWords] cached neighbourhood rows
    const Slot *table;        // the open-addressed voxel hash; a slot word — and a row word — is
                              // (first unit (4 points) of the voxel's region << 8) | count (host_map.hpp)
    uint32_t mask;
    const Point4 *pts;
    uint32_t pts_bytes;       // size of the point array: under 4 GiB (2^24 units of 128 B), read with 32-bit
                              // byte offsets through a buffer resource
    int filter;               // 1: the scan reads the compact copy and filters in fp32 (big frames, dense voxels)
    int flat;                 // 1: with 2 or 4 lanes per query the lanes stride through a query's voxels as one
                              // sequence (sparse voxels; 8 and 16 lanes always do; kernels.hip "flat order")
    const uint4 *cand;        // compact copy of pts (fp32 x, y, z, label; k_derive_cand), same indexing
    uint32_t cand_bytes;      // its size
    const uint32_t *cand_flags;   // bit 0: some label of the map cannot be classified in fp32
    double filt_inv_same;     // (1 / sem_th)(1 + 2^-10)(1 + 1e-6): fp32 threshold of the label class (inf: off)
    double filt_inv_diff;     // (1 + 2^-10)(1 + 1e-6): of the other candidates (inf: off)
    double filt_slack;        // 2^-44 (1 + 2^10)(1 + 1e-6): times sum (|q_a| + 4 voxel)^2
    double sem_th;
    double dist_init;         // DBL_MAX
    double prune_scale;       // min(sem_th, 1) * (1 - 1e-9): scaled distance >= this x squared
                              // distance to the voxel's cell (0 when pruning is off)
    unsigned keep_all;        // 0x7FFFFFF: visit every occupied voxel (pruning off: sem_th < 0 or
                              // not a number); 0: prune by the cell lower bound
    // search-only mode (GetCorrespondences): the semantic nearest neighbour of every query
    int32_t *nn_idx;          // out: the point's index in the point array (unit * 4 + slot) or -1
    // fused mode (the ICP loop): acceptance + Gauss-Newton accumulation in the same launch
    double kernel;            // robust kernel k: w = k^2 / (k + |r|^2)^2 (Registration.cpp:79)
    double accept_r2;         // largest r2 with sqrt(r2) < max_correspondence_distance (exact form
                              // of the acceptance test VoxelHashMap.cpp:111; -1: accept nothing)
    uint2 *nn_prev;           // [n] in/out: every query's record of the previous iteration {key =
                              // (voxel << 8) | slot of its nearest neighbour, its byte offset};
                              // key 0xFFFFFFFF: none.  Seeds the next search with a tight bound.
    uint32_t *work;           // instrumented builds only: [n] map points handed to each query
    long long *acc;           // out: the Gauss-Newton sums as fixed-point accumulators (see kAcc* below):
                              // every workgroup ADDS its 16 sums and its pair count with integer atomics
    double digit_limit;       // k_icp + k_fin: the leading digit of one block of four queries stays below this — a power of two
                              // chosen by the host so that the sum over ALL the frame's blocks stays inside 63 bits (icp_params)
    double acc_scale;         // a power of two (1 normally): the sums are accumulated as fixed-point numbers of
                              // sum x acc_scale — a frame whose sums leave the range (coordinates of 10^7 m) is
                              // registered again at 2^-24, 2^-48 (capi.hip); FinParams / LoopParams::acc_unscale undo it
    // k_icp, heaviest first: the launch-per-iteration form ends with the waves that started last, and with the sorted frame
    // dealt out in its own order those are as heavy as any — half of a c4 launch is its tail (profiles/r06/icp_tail_c4.txt).
    // The stripes of kStripe consecutive workgroups (what one XCD takes at a time) are dispatched in the order of the work
    // an earlier iteration measured, heaviest first, so that what runs last is light and short.  A stripe's work is that of
    // its heaviest wave: what a launch waits for at its end is single long waves, not busy stripes (the sum of the waves
    // instead: c4 -3 % against -7 %, profiles/r06/lpt_max_ab.txt; stripes of 4, 2, 1 workgroups: within 1.5 %, lpt_stripe_ab.txt).
    uint32_t *stripe_work;    // optional, out: [stripes] max over the stripe's waves of the most points one of a wave's queries was handed
    const uint32_t *stripe_order;  // optional: [stripes] the stripe dispatched at each position (null: the order of the frame)
    unsigned long long *counters;  // optional: [2 x waves] running sums of {C_q, pairs evaluated}
    unsigned nwaves;          // waves that own queries: ceil(n / (64 >> lw))
#ifdef SAGE_ICP_DELAY_PROBE
    unsigned dbg_delay;       // probe builds: ticks (100 MHz) a wave waits for "the pose" after its start
    unsigned dbg_repeat;      // probe builds: extra passes of the whole body inside one launch (warm L2s)
#endif
    // k_icp, chained (frames beyond the LDS, one GPU): the launches of the iterations follow each other without a k_fin in
    // between — the solving wave of the one-launch loop (k_loop_solve, resident beside them on its own stream) collects the
    // sums as the last workgroup of a launch sends them, solves while the next launch starts, and the first wave of every
    // workgroup of that launch waits for the pose it publishes: one kernel boundary per iteration instead of two, and the
    // solve under it.  Same sums, same solve: the same bits.
    LoopShared *chain;        // non-null: this launch is iteration `chain_iter` of a chained loop
    int chain_iter;
    unsigned long long chain_timeout;   // 100-MHz ticks a workgroup waits for its pose
    unsigned long long chain_epoch;     // launch 0 tells the solving wave to start (LoopShared::go)
};

#ifndef SAGE_ICP_WAVES
#define SAGE_ICP_WAVES 4
#endif
constexpr int kIcpWavesPerBlock = SAGE_ICP_WAVES;
// lanes per query: a power of two 1..16 (lw = log2); few for frames that fill the chip (fewer
// instructions per query), many for small frames / shards (shorter dependent chains per wave)
int icp_blocks_for(int n, int lw);
size_t icp_lds_bytes(int lw);
void launch_rows(const IcpParams &p, hipStream_t s);                     // (re)build every row
// the compact copy of the map's points the scan reads (see kernels.hip)
void launch_derive_cand(const Slot *table, uint32_t nslots, const Point4 *pts, uint4 *cand, uint64_t nslots_pts,
                        uint32_t *flags, hipStream_t s);
void launch_icp(const IcpParams &p, int lw, bool fused, hipStream_t s);

struct GnParams {             // stand-alone AlignClouds on explicit pairs
    const Point4 *src;
    const Point4 *tgt;
    int n;
    double kernel;
    double *partials;         // [gridDim.x][kNumSums]
};
int launch_gn(const GnParams &p, hipStream_t s);   // returns the number of partials written

// Finish of an iteration, one workgroup: fixed-order reduction of the partials, [exchange of the
// sums with the peer GPUs,] 6x6 solve, SE3 exp, pose composition, convergence test.
//   mode 0: reduce + solve            (single GPU)
//   mode 1: reduce -> st->sums        (multi GPU, before the RCCL all-reduce)
//   mode 2: solve from st->sums       (multi GPU, after the all-reduce)
//   mode 3: reduce + direct exchange over xGMI + solve
// Fixed-point accumulators of the Gauss-Newton sums.  A workgroup of k_icp used to leave one fp64
// partial (160 B) for k_fin, whose single workgroup then pulled 1,920 (c2) to 7,813 (c4) of them
// through one CU: 3.6 to 13 us of every iteration.  Integer addition is associative, so the
// workgroups can add into shared accumulators with fire-and-forget atomics in any order and the
// result is still bit-reproducible — and exact: the pair terms of every BLOCK of four consecutive
// queries are added in fp64 in a fixed order, and each block sum becomes a 120-bit fixed-point number
// held as three signed digits of 40 bits (weights 2^0, 2^-40, 2^-80) in 64-bit words (kernels.hip,
// to_digits); from there on everything is integer addition, so the accumulated bits depend on the
// order of the frame and on nothing else (lanes per query, waves, workgroups, which loop ran, order
// of arrival).  Nothing is rounded again until k_fin converts the totals to fp64 once.  kAccReplicas
// copies (chosen by workgroup index) keep the atomics of a launch off any single address; k_fin adds
// the copies (exactly) and clears them.
constexpr int kAccReplicas = 32;
constexpr int kAccWords = 64;              // per replica: (16 sums + pair count) x 3 digits = 51 used
constexpr int kAccValues = 17;
struct FinParams {
    IcpState *st;
    const double *partials;
    long long *acc;           // non-null: the sums come from the fixed-point accumulators (k_icp), not from partials (k_gn)
    double acc_unscale;       // 1 / IcpParams::acc_scale
    int nparts;
    int mode;
    int standalone;           // 1: run even when st->done (AlignClouds entry)
    P2pParams p2p;            // mode 3 only
};
void launch_fin(const FinParams &p, hipStream_t s);

// ---- k_loop: the whole ICP loop of a frame that fits the machine's LDS in ONE launch -------------
// (Registration.cpp:127-138.)  The frame is cut into GROUPS of 64 >> lw consecutive queries — what one
// wave of k_icp holds.  A workgroup owns `gpw` groups for the whole call and keeps, per query, in LDS:
// the 128-B neighbourhood row and a 96-B state record (pristine frame point, the previous answer's key
// and fp64 record, the home voxel the row was built for).  Its waves take the groups one after another
// from an LDS counter, so what has to be resident is the LDS, not one wave per group: the c2 frame
// (7,500 groups at four lanes per query) runs on 28 waves per CU.  Nothing per-query is read from or
// written to HBM after the first pass; the L2s stay warm across iterations.
// An iteration ends with the workgroups adding their sums into fixed-point accumulators whose words
// count their contributors, ONE solving wave — its own kernel (k_loop_solve, ~120 registers against
// the search's 72) on a second stream, resident beside the grid — reading them when they are
// complete, [exchanging the sums with the peer GPUs,] solving and publishing the next pose, and one
// wave per workgroup waiting for it: no kernel boundary, no k_fin.  Everything the workgroups share
// inside the launch is accessed with agent-scope atomics only (per-XCD L2s are not coherent with each
// other); the block is zeroed before every launch; every wait is bounded.
#ifndef SAGE_LOOP_REPLICAS
#define SAGE_LOOP_REPLICAS 8
#endif
constexpr int kLoopReplicas = SAGE_LOOP_REPLICAS;      // accumulator copies (workgroup b adds into copy b & 7: one copy per XCD)
// (One copy per XCD is what matters, not their number: workgroup b runs on XCD b % 8, so a copy's words are only ever
// touched by one XCD's atomics.  Four copies — two XCDs per copy — cost a 15 k-query grid 12 % and a 30 k one 7 %, sixteen are
// no different from eight: profiles/r06/replicas_ab.txt, copies_ab.txt.  The chained launches' 32 and k_fin's 32 keep the
// same property: b % 32 determines b % 8.)
constexpr int kLoopPoseGranules = 25;      // R[9], t[3] as 24 x {tag, 32 bits} + {tag, done}
constexpr int kChainReplicas = 32;         // accumulator copies of the chained launches (k_icp: thousands of workgroups, <= 255 per copy)
struct LoopShared {
    long long acc[2][kLoopReplicas][kAccWords];         // as FinParams::acc, one set per iteration parity, every word
                                                        // (digit << 8) | workgroups in it; word 51: (workgroups whose sums overflowed << 8) | workgroups
    unsigned long long pose[32];                        // granules (tag << 32) | payload, tag = iteration + 1
    unsigned long long abort_word[16];                  // [0] != 0: a wait timed out somewhere — everybody leaves
    unsigned long long go[16];                          // [0]: the call's epoch, stored by k_loop's first workgroup when the
                                                        // grid has started (bit 63: the frame holds a non-finite point)
    long long acc32[2][kChainReplicas][kAccWords];      // the chained launches' accumulators (as acc above, 32 copies; last:
                                                        // what k_loop touches keeps its small offsets)
};
struct LoopParams {
    LoopShared *sh;
    IcpState *st;                  // in: the initial pose; out: the final loop state (written by the solving wave)
    int nw;                        // waves per workgroup (<= kLoopMaxWaves)
    int gpw;                       // units of (64 >> lw) queries a workgroup owns at most (LDS for that many)
    int wgs;                       // query workgroups (a multiple of 8 x kLoopStripe); each accumulator copy counts wgs / 8
    int contiguous;                // 0: XCD x serves stripes x, x + 8, ... of kLoopStripe workgroups; 1: XCD x serves the
                                   // groups [xcd_first[x], xcd_first[x + 1]) of the (spatially sorted) frame
    uint32_t xcd_first[9];
    unsigned long long timeout_ticks;      // s_memrealtime ticks (100 MHz) any wait may take
    unsigned long long count_timeout_ticks;    // ... the solving wave's wait for its own workgroups' sums (the same, except
                                           // under a communicator: timeout_ticks then also covers the peers' exchange)
    int max_iterations;            // kMaxIterations (tests: fewer)
    unsigned long long epoch;      // identifies this call: the solving wave is launched FIRST (it must be resident when the
                                   // grid fills the machine) and waits for LoopShared::go to carry this number
    double T0[7];                  // the initial pose (the solving wave starts before the loop state is uploaded)
    double acc_unscale;            // 1 / IcpParams::acc_scale
    int shared_loop;               // 1: under a communicator — an overflowing sum or a bad frame point does not end
                                   // this rank's loop on its own (the ranks must keep exchanging in step)
    int prio;                      // wave priorities by the work of a wave's unit (kernels.hip, k_loop): 0 off | 1..3: on, the
                                   // priority of a unit beyond one per wave
    int deal;                      // 1: a wave's FIRST unit of an iteration is fixed by where the wave sits — unit (SIMD +
                                   // the workgroup's slot on the CU) mod waves — instead of first come first served (k_loop)
    int copies;                    // accumulator copies the workgroups add into: kLoopReplicas (k_loop: LoopShared::acc) or
                                   // kChainReplicas (chained k_icp launches: LoopShared::acc32)
    IcpProgress *progress;         // optional (chained launches): the host-mapped word the host steers its look-ahead by
};
struct LoopArgs {                  // k_loop's one argument: its passes re-read it from the kernel-argument segment
    IcpParams P;
    LoopParams L;
};
#ifndef SAGE_LOOP_OCC
#define SAGE_LOOP_OCC 7        // waves per SIMD k_loop's register allocation allows (72 registers), i.e. 28 waves per CU
#endif
constexpr int kLoopMaxWavesHost = 8;
constexpr int kLoopStripeHost = 4;
constexpr int kLoopStateWords = 24;        // per query: f[8] | pp[8] | prev key, prev offset, kx, ky | kz, occ, -, -
size_t loop_lds_bytes(int lw, int nw, int gpw);
// workgroups of `nw` waves and `lds` bytes resident per CU for the variant (lw, filter): 0 = the kernel
// cannot be used on this device; the grid must not exceed this number x CUs (less one slot for the solver)
int loop_blocks_per_cu(int lw, bool filter, int nw, size_t lds);
void launch_loop(const IcpParams &p, const LoopParams &l, int lw, hipStream_t s);
void launch_loop_solve(const LoopParams &l, const P2pParams &x, hipStream_t s);

constexpr int kMaxPartials = 1 << 16;
// (2^24 - 1 blocks of four: the lower digits of a block are below 2^39 in magnitude, their sum over the frame inside 63 bits)
constexpr uint64_t kMaxQueries = (1ull << 26) - 4;

void launch_tf(Point4 *pts, int n, const IcpState *st, hipStream_t s);
void launch_scatter_points(const uint32_t *idx, const Point4 *vals, uint32_t n, Point4 *pts,
                           hipStream_t s);
void launch_scatter_slots(const uint32_t *idx, const Slot *vals, uint32_t n, Slot *table,
                          hipStream_t s);
void launch_scatter_u32(const uint32_t *idx, const uint32_t *vals, uint32_t n, uint32_t *dst, hipStream_t s);
void launch_sum_counters(const unsigned long long *c, int n, IcpState *st, hipStream_t s);

// preprocess.hip: one level of per-label-group voxel down-sampling (optionally with the range crop)
struct VdsParams {
    const Point4 *in;
    int n;
    int do_crop;                       // 1: apply Preprocess() first (range crop + label zeroing)
    double max_range, min_range, label_max_range;
    int n_groups;                      // -1: no grouping / no voxel test (crop only)
    const int *group_counts;           // device: [n_groups]
    const int *group_labels;           // device: concatenated label lists
    double group_vs[8];                // voxel size per group
    double scale;                      // vox_scale
    unsigned long long *keys;          // device hash set: (group, voxel) keys, capacity mask + 1
    uint32_t *winner;                  // lowest original index per key
    uint32_t mask;
    Point4 *tmp;                       // [n] cropped points
    uint32_t *slot_of;                 // [n]
    uint32_t *sort_key;                // [2n]
    uint32_t *sort_val;                // [2n]
    int *overflow;                     // set when a voxel index does not fit the key
    unsigned long long *out_keys;      // optional [n]: (group << 60) | the reference's 20-bit VoxelHash of
                                       // every survivor's voxel, in output order
};
size_t vds_sort_temp_bytes(int n);
void launch_vds_permute(const Point4 *in, const uint32_t *perm, uint32_t n, Point4 *out, hipStream_t s);
hipError_t voxel_downsample_device(const VdsParams &P, void *sort_temp, size_t sort_temp_bytes,
                                   uint32_t *d_n_kept, Point4 *out, hipStream_t s);

// sort.hip: re-ordering of a frame along the Morton curve of its map-frame voxels
size_t sort_temp_bytes(int n);
// ... and the dispatch order of k_icp's stripes: `order` = the stripes by `work`, heaviest first (stable); `work` is zeroed
// for the next measurement.  iota: [stripes] 0, 1, 2, ... (stripe_order_init fills it and zeroes `work`).
size_t stripe_sort_temp_bytes(unsigned stripes);
void stripe_order_init(uint32_t *work, uint32_t *iota, unsigned stripes, hipStream_t s);
hipError_t stripe_order_sort(uint32_t *work, uint32_t *work_sorted, const uint32_t *iota, uint32_t *order, unsigned stripes,
                             void *temp, size_t temp_bytes, hipStream_t s);
int icp_stripes_for(int n, int lw);        // stripes of the k_icp launch of n queries
// A frame point with a coordinate or label that is not finite raises st->bad_input (the reference
// casts such values to int: undefined behaviour); with `stop_on_bad` it also ends the loop before its
// first iteration (st->done, the progress word) — not under a communicator, where every rank has to
// enqueue the same collectives.
hipError_t sort_frame(const Point4 *d_in, Point4 *d_out, int n, IcpState *st, bool apply_pose, bool stop_on_bad,
                      double voxel_size, uint32_t *keys, uint32_t *vals, void *temp,
                      size_t temp_bytes, hipStream_t s);
// The order is worth its launches from ~20 k points on (profiles/r06/nosort_ab.txt: 10 k points -4 %, 15 k -1 %, 30 k +3 %,
// 60 k +11 % without it: a small frame's rows and the map under it are cache-resident in any order, and the twelve launches
// of the sort are 27 us of a registration that takes 800).  Below, the frame is searched as it came — check_copy_frame: the
// same refusal of non-finite input, one launch.  (The bits of a pose depend on which four queries form a block of the
// exact sums, i.e. on the order: the rule looks at the frame's size only, so every form of the loop takes the same.)
constexpr int kSortFrameFrom = 16384;
hipError_t check_copy_frame(const Point4 *d_in, Point4 *d_out, int n, IcpState *st, bool stop_on_bad, hipStream_t s);

}  // namespace sageicp
