// Launch interface between capi.hip (host side) and kernels.hip / sort.hip (device side).
#pragma once

#include <hip/hip_runtime.h>

#include "sageicp_types.h"

namespace sageicp {

struct NnParams {
    const Point4 *frame;      // pristine (sorted) frame, or ready-made queries (apply_pose == 0)
    Point4 *src;              // out (optional): the queries as searched (pose applied), for k_gn
    int n;
    const IcpState *st;       // pose to apply and the done flag
    int check_done;           // 1 inside the ICP loop: later launches of a finished loop are no-ops
    int apply_pose;
    double voxel_size;
    unsigned chunk;           // group cap: queries per wave (a power of two <= 16); a group is a
                              // run of queries of one chunk that share a home voxel
    unsigned chunk_log2;
    unsigned cap_heads;       // forced heads inside a chunk: bit i set for every i that is a multiple
                              // of the group cap (1 when the cap equals the chunk)
    unsigned nchunks;         // ceil(n / chunk)
    int4 *tabkey;             // [n] home voxel each cached probe-table row was built for (y, z, w)
    uint2 *blks;              // [n][32] probe-table rows {candidate offset, first point}
    const Slot *table;        // the open-addressed voxel hash
    uint32_t mask;
    const Point4 *pts;
    uint32_t pts_bytes;       // size of the point array (< 4 GiB: k_nn addresses it by byte offset)
    int cap;
    double sem_th;
    double dist_init;         // DBL_MAX
    double prune_scale;       // min(sem_th, 1) * (1 - 1e-9): scaled distance >= this x squared
                              // distance to the voxel's cell (0 when pruning is off)
    unsigned keep_all;        // 0x7FFFFFF: visit every occupied voxel (pruning off: sem_th < 0 or
                              // not a number); 0: prune by the cell lower bound
    int32_t *nn_idx;          // out: block*cap+slot of the semantic nearest neighbour, -1 if the
                              //      27-voxel neighbourhood is empty (acceptance is applied later)
    unsigned long long *cand_counter;  // optional: [2 x nchunks] per-chunk running sums of
                                       // {C_q, pairs evaluated}
};

// k_nn's per-wave LDS layout, in 32-bit words from the wave's base (host and device agree on it)
struct NnLds {
    unsigned spt;             // chunk x {x, y, z, label} fp64
    unsigned gap;             // chunk x 6 fp64: scaled squared gaps to the faces of the home cell
    unsigned skey;            // 3 x chunk home voxel indices
    unsigned wave_words;
};
__host__ __device__ inline NnLds nn_lds_layout(unsigned chunk) {
    NnLds l;
    l.spt = 0u;
    l.gap = 8u * chunk;
    l.skey = l.gap + 12u * chunk;
    l.wave_words = (l.skey + 3u * chunk + 3u) & ~3u;
    return l;
}

struct GnParams {
    const Point4 *src;        // transformed queries (or the explicit sources of align_clouds)
    const Point4 *tgt_pairs;  // explicit targets (align_clouds entry) or nullptr
    int n;
    const IcpState *st;
    int check_done;
    const Point4 *pts;
    const int32_t *nn_idx;
    double kernel;
    double max_dist;          // acceptance threshold on the unscaled distance
    double *partials;         // [gridDim.x][kNumSums]
    int fuse_mode;            // -1: partials only; 0 / 1: the last workgroup runs finish_iteration;
                              //  3: it reduces, exchanges the sums with the peer GPUs (p2p), solves
    IcpState *st_rw;          // state written by finish_iteration
    unsigned *ticket;         // last-arriver ticket (zero before the first launch)
    P2pParams p2p;            // fuse_mode 3 only
};

constexpr int kMaxGnBlocks = 512;
constexpr uint64_t kMaxQueries = (1ull << 26) - 1;
constexpr uint64_t kMaxMapPoints = (1ull << 27) - 2;   // blocks x capacity: 32-B points under 4 GiB

void launch_nn(const NnParams &p, hipStream_t s);
int launch_gn(const GnParams &p, hipStream_t s);   // returns the number of partials written
void launch_fin(IcpState *st, const double *partials, int nparts, int mode, int standalone,
                hipStream_t s);
void launch_tf(Point4 *pts, int n, const IcpState *st, hipStream_t s);
void launch_scatter_points(const uint32_t *idx, const Point4 *vals, uint32_t n, Point4 *pts,
                           hipStream_t s);
void launch_scatter_slots(const uint32_t *idx, const Slot *vals, uint32_t n, Slot *table,
                          hipStream_t s);
void launch_sum_candidates(const unsigned long long *c, int n, IcpState *st, hipStream_t s);
int gn_grid_for(int n);

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
};
size_t vds_sort_temp_bytes(int n);
hipError_t voxel_downsample_device(const VdsParams &P, void *sort_temp, size_t sort_temp_bytes,
                                   uint32_t *d_n_kept, Point4 *out, hipStream_t s);

// sort.hip: re-ordering of a frame along the Morton curve of its map-frame voxels
size_t sort_temp_bytes(int n);
hipError_t sort_frame(const Point4 *d_in, Point4 *d_out, int n, const IcpState *st, bool apply_pose,
                      double voxel_size, uint32_t *keys, uint32_t *vals, void *temp,
                      size_t temp_bytes, hipStream_t s);

}  // namespace sageicp
