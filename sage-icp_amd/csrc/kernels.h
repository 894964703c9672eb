// Launch interface between capi.hip (host side) and kernels.hip (device side).
#pragma once

#include <hip/hip_runtime.h>

#include "sageicp_types.h"

namespace sageicp {

struct NnParams {
    const Point4 *frame;      // pristine frame (or already-transformed queries when !apply_pose)
    int n;
    const IcpState *st;
    const Slot *table;
    uint32_t mask;
    const Point4 *pts;
    int cap;
    double voxel_size;
    double sem_th;
    double max_dist;
    int32_t *nn_idx;          // out: block*cap+slot of the accepted neighbour, -1 if none
    unsigned long long *cand_counter;  // optional: += sum_q C_q (candidates visible to q)
};

struct GnParams {
    const Point4 *frame;
    const Point4 *tgt_pairs;  // explicit targets (align_clouds entry) or nullptr
    int n;
    const IcpState *st;
    const Point4 *pts;
    const int32_t *nn_idx;
    double kernel;
    double *partials;         // [gridDim.x][kNumSums]
    int apply_pose;
};

constexpr int kMaxGnBlocks = 512;

void launch_nn(const NnParams &p, bool apply_pose, hipStream_t s);
int launch_gn(const GnParams &p, hipStream_t s);   // returns the number of partials written
void launch_fin(IcpState *st, const double *partials, int nparts, int mode, int standalone,
                hipStream_t s);
void launch_tf(Point4 *pts, int n, const IcpState *st, hipStream_t s);
int gn_grid_for(int n);

}  // namespace sageicp
