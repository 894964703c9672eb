// Per-frame spatial ordering of the scan (gfx950).  The reference hands RegisterFrame a scan in
// hash-map iteration order (core/Preprocessing.cpp:75-82), i.e. spatially random.  Here the
// frame is re-ordered once per call along the Morton curve of the map-frame voxels its points
// fall into under the current pose, so that the queries of a k_icp wave are neighbours: they touch
// the same voxel blocks (L1/L2 hits) and cost about the same.  (Round 1 re-sorted when the pose had
// carried the points a fraction of a voxel away from that order; with a lane per query and cached
// neighbourhood rows that costs what it saves, so run_icp sorts once per call.)
// (Re-ordering the queries by the work the previous iteration measured was tried in four forms —
// frame-wide work classes; sorted, or dealt out evenly over the waves, inside chunks of 1024
// consecutive queries; sorted inside one workgroup's 64 queries — and is not kept: the lockstep
// efficiency of a wave rises from 0.42 to 0.96 and the kernel gets slower or stays where it was,
// because the queries of a wave then no longer share cache lines; profiles/README.md.)
// A stable sort keeps the result — and therefore the fp64 summation order of the Gauss-Newton
// sums — bit-reproducible from run to run.
#include <hip/hip_runtime.h>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.h"

namespace sageicp {

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

template <bool APPLY_POSE>
__global__ __launch_bounds__(256) void k_morton_keys(const Point4 *pts, int n, IcpState *st, int stop_on_bad,
                                                     double voxel_size, uint32_t *keys,
                                                     uint32_t *vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Point4 f = pts[i];
    // the one pass every frame goes through before it is searched: non-finite input is refused here
    // (fabs(v) <= DBL_MAX is false for NaN and for the infinities)
    if (!(fabs(f.x) <= 1.7976931348623157e308 && fabs(f.y) <= 1.7976931348623157e308 &&
          fabs(f.z) <= 1.7976931348623157e308 && fabs(f.l) <= 1.7976931348623157e308)) {
        st->bad_input = 1;
        if (stop_on_bad) {
            st->done = 1;
            if (IcpProgress *pg = st->progress)
                __hip_atomic_store(&pg->word, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    double x = f.x, y = f.y, z = f.z;
    if (APPLY_POSE) {
        const double *R = st->R;
        const double *T = st->T;
        x = R[0] * f.x + R[1] * f.y + R[2] * f.z + T[4];
        y = R[3] * f.x + R[4] * f.y + R[5] * f.z + T[5];
        z = R[6] * f.x + R[7] * f.y + R[8] * f.z + T[6];
    }
    // the same expression k_icp evaluates, so at the sorting pose equal keys <=> same home voxel
    // (10 bits per axis: voxels 1024 apart alias, which only costs locality)
    const int cx = static_cast<int>(x / voxel_size) + 512;
    const int cy = static_cast<int>(y / voxel_size) + 512;
    const int cz = static_cast<int>(z / voxel_size) + 512;
    keys[i] = spread10(static_cast<uint32_t>(cx)) | (spread10(static_cast<uint32_t>(cy)) << 1) |
              (spread10(static_cast<uint32_t>(cz)) << 2);
    vals[i] = static_cast<uint32_t>(i);
}

__global__ __launch_bounds__(256) void k_gather(const Point4 *in, const uint32_t *perm, int n,
                                                Point4 *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = in[perm[i]];
}

// A frame too small for the order to pay (frame_sort_pays): the same refusal of non-finite input, the frame as it came.
__global__ __launch_bounds__(256) void k_check_copy(const Point4 *pts, int n, IcpState *st, int stop_on_bad, Point4 *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Point4 f = pts[i];
    if (!(fabs(f.x) <= 1.7976931348623157e308 && fabs(f.y) <= 1.7976931348623157e308 &&
          fabs(f.z) <= 1.7976931348623157e308 && fabs(f.l) <= 1.7976931348623157e308)) {
        st->bad_input = 1;
        if (stop_on_bad) {
            st->done = 1;
            if (IcpProgress *pg = st->progress)
                __hip_atomic_store(&pg->word, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    out[i] = f;
}
hipError_t check_copy_frame(const Point4 *d_in, Point4 *d_out, int n, IcpState *st, bool stop_on_bad, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_check_copy, dim3((n + 255) / 256), dim3(256), 0, s, d_in, n, st, stop_on_bad ? 1 : 0, d_out);
    return hipGetLastError();
}

size_t sort_temp_bytes(int n) {
    size_t bytes = 0;
    uint32_t *k = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, static_cast<unsigned>(n), 0, 30);
    return bytes;
}

// keys/vals: 2*n uint32 each (in | out halves).  perm_out = vals + n afterwards.
hipError_t sort_frame(const Point4 *d_in, Point4 *d_out, int n, IcpState *st, bool apply_pose, bool stop_on_bad,
                      double voxel_size, uint32_t *keys, uint32_t *vals, void *temp,
                      size_t temp_bytes, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int grid = (n + 255) / 256;
    if (apply_pose)
        hipLaunchKernelGGL(k_morton_keys<true>, dim3(grid), dim3(256), 0, s, d_in, n, st, stop_on_bad ? 1 : 0,
                           voxel_size, keys, vals);
    else
        hipLaunchKernelGGL(k_morton_keys<false>, dim3(grid), dim3(256), 0, s, d_in, n, st, stop_on_bad ? 1 : 0,
                           voxel_size, keys, vals);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys + n, vals, vals + n,
                                             static_cast<unsigned>(n), 0, 30, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, s, d_in, vals + n, n, d_out);
    return hipGetLastError();
}

// ---- the dispatch order of k_icp's stripes (kernels.h) ----------------------------------------------
__global__ __launch_bounds__(256) void k_stripe_init(uint32_t *work, uint32_t *iota, unsigned n) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    work[i] = 0u;
    iota[i] = i;
}
size_t stripe_sort_temp_bytes(unsigned stripes) {
    size_t bytes = 0;
    uint32_t *k = nullptr;
    (void)rocprim::radix_sort_pairs_desc(nullptr, bytes, k, k, k, k, stripes, 0, 32);
    return bytes;
}
void stripe_order_init(uint32_t *work, uint32_t *iota, unsigned stripes, hipStream_t s) {
    if (stripes) hipLaunchKernelGGL(k_stripe_init, dim3((stripes + 255u) / 256u), dim3(256), 0, s, work, iota, stripes);
}
hipError_t stripe_order_sort(uint32_t *work, uint32_t *work_sorted, const uint32_t *iota, uint32_t *order, unsigned stripes,
                             void *temp, size_t temp_bytes, hipStream_t s) {
    if (!stripes) return hipSuccess;
    hipError_t e = rocprim::radix_sort_pairs_desc(temp, temp_bytes, work, work_sorted, iota, order, stripes, 0, 32, s);
    if (e != hipSuccess) return e;
    return hipMemsetAsync(work, 0, sizeof(uint32_t) * stripes, s);
}

}  // namespace sageicp
