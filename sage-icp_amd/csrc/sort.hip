// Per-frame spatial ordering of the scan (gfx950).  The reference hands RegisterFrame a scan in
// hash-map iteration order (core/Preprocessing.cpp:75-82), i.e. spatially random.  Here the
// frame is re-ordered once per call along the Morton curve of the map-frame voxels its points
// fall into under the current pose, so that queries sharing a home voxel are consecutive
// (k_nn cuts each chunk into such runs and serves a whole run with one candidate list) and
// neighbouring runs touch neighbouring voxel blocks (L1/L2 hits).  run_icp re-sorts when the
// pose has carried the points a fraction of a voxel away from the order they were sorted in.
// Work balance: a lane of k_icp owns a query, so a wave lasts as long as its heaviest query.  Once
// an iteration has told how many map points each query really had to look at, the order becomes
// (work class, descending) major / Morton minor: the queries of a wave then cost about the same,
// and the heaviest waves start first.
// A stable sort keeps the result — and therefore the fp64 summation order of the Gauss-Newton
// sums — bit-reproducible from run to run.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "kernels.h"

namespace sageicp {

// work class of a query that looked at `pairs` map points last iteration: half-octave classes
// 0..15 (0: unknown or fewer than 6)
__device__ __forceinline__ uint32_t work_class(uint32_t pairs) {
    const uint32_t x = pairs / 3u;
    if (x < 2u) return 0u;
    const uint32_t l = 31u - static_cast<uint32_t>(__builtin_clz(x));      // floor(log2 x) >= 1
    const uint32_t half = (x >> (l - 1u)) & 1u;                             // second-highest bit
    const uint32_t c = 2u * l + half - 1u;
    return c > 15u ? 15u : c;
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

template <bool APPLY_POSE>
__global__ __launch_bounds__(256) void k_morton_keys(const Point4 *pts, int n, const IcpState *st,
                                                     double voxel_size, const uint32_t *work,
                                                     uint32_t *keys, uint32_t *vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Point4 f = pts[i];
    double x = f.x, y = f.y, z = f.z;
    if (APPLY_POSE) {
        const double *R = st->R;
        const double *T = st->T;
        x = R[0] * f.x + R[1] * f.y + R[2] * f.z + T[4];
        y = R[3] * f.x + R[4] * f.y + R[5] * f.z + T[5];
        z = R[6] * f.x + R[7] * f.y + R[8] * f.z + T[6];
    }
    // the same expression k_icp evaluates, so at the sorting pose equal keys <=> same home voxel
    // (9 bits per axis: voxels 512 apart alias, which only costs locality); bits 27..30: the
    // inverted work class, so that heavy queries come first
    const int cx = static_cast<int>(x / voxel_size) + 256;
    const int cy = static_cast<int>(y / voxel_size) + 256;
    const int cz = static_cast<int>(z / voxel_size) + 256;
    const uint32_t morton = spread10(static_cast<uint32_t>(cx) & 511u) |
                            (spread10(static_cast<uint32_t>(cy) & 511u) << 1) |
                            (spread10(static_cast<uint32_t>(cz) & 511u) << 2);
    const uint32_t cls = work ? work_class(work[i]) : 0u;
    keys[i] = ((15u - cls) << 27) | morton;
    vals[i] = static_cast<uint32_t>(i);
}

__global__ __launch_bounds__(256) void k_gather(const Point4 *in, const uint32_t *perm, int n,
                                                Point4 *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = in[perm[i]];
}

// work[perm[pos]] = pairs the query at sorted position `pos` looked at (perm: sorted -> pristine)
__global__ __launch_bounds__(256) void k_scatter_work(const uint4 *prev, const uint32_t *perm, int n,
                                                      uint32_t *work) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    work[perm[i]] = prev[i].z;
}

size_t sort_temp_bytes(int n) {
    size_t bytes = 0;
    uint32_t *k = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, k, k, k, n, 0, 31);
    return bytes;
}

// keys/vals: 2*n uint32 each (in | out halves).  perm_out = vals + n afterwards.
// `work` (optional, pristine order): pairs every query looked at in the last iteration.
// `prev` (optional): the per-query record of the last iteration in the CURRENT sorted order; when
// given, `work` is refreshed from it first (through the permutation of the previous sort, which
// still sits in vals + n).
hipError_t sort_frame(const Point4 *d_in, Point4 *d_out, int n, const IcpState *st, bool apply_pose,
                      double voxel_size, uint32_t *keys, uint32_t *vals, void *temp,
                      size_t temp_bytes, uint32_t *work, const uint4 *prev, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int grid = (n + 255) / 256;
    if (work && prev)
        hipLaunchKernelGGL(k_scatter_work, dim3(grid), dim3(256), 0, s, prev, vals + n, n, work);
    if (apply_pose)
        hipLaunchKernelGGL(k_morton_keys<true>, dim3(grid), dim3(256), 0, s, d_in, n, st, voxel_size,
                           work, keys, vals);
    else
        hipLaunchKernelGGL(k_morton_keys<false>, dim3(grid), dim3(256), 0, s, d_in, n, st,
                           voxel_size, work, keys, vals);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys + n, vals,
                                                      vals + n, n, 0, 31, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, s, d_in, vals + n, n, d_out);
    return hipGetLastError();
}

}  // namespace sageicp
