// Per-frame preprocessing on the device: range crop + label zeroing
// and the per-label-group voxel down-sampling that produce the two clouds of a frame — the
// denser one that goes into the map and the sparser one that is registered.
//
// Reference (cpp/sage_icp/):
//   core/Preprocessing.cpp:173-187  Preprocess, dynamic_vehicle_filter == false branch
//   core/Preprocessing.cpp:44-84    VoxelDownsample: one hash grid per label group, the FIRST point
//                                   that falls into a voxel is kept
//   pipeline/sageICP.cpp:97-101     Voxelize(): scale 0.5 -> frame_downsample, then 1.5 -> source
//
// "First point per voxel" is order dependent on the CPU (sequential inserts).  Here every point
// claims its (group, voxel) key in an open-addressed device table with a 64-bit atomicCAS and
// lowers the voxel's winner with atomicMin(original index): the survivor is exactly the point a
// sequential pass would have kept.  The atomics go to ~n distinct addresses (no hot word).
// Survivors are emitted group by group in original order (stable 4-bit radix sort of the group
// id), which is the insertion order the host pipeline used; the reference itself emits
// tsl::robin_map bucket order (deviation D3).
#include <hip/hip_runtime.h>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.h"

namespace sageicp {

constexpr unsigned long long kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t kDropped = 15u;         // sort key of a point that is not kept

__device__ __forceinline__ uint32_t mix64(unsigned long long k) {
    k ^= k >> 33; k *= 0xFF51AFD7ED558CCDull; k ^= k >> 33; k *= 0xC4CEB9FE1A85EC53ull; k ^= k >> 33;
    return static_cast<uint32_t>(k);
}

// pass 1: crop (optional), group look-up, voxel key, claim + atomicMin
__global__ __launch_bounds__(256) void k_vds_insert(VdsParams P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n) return;
    Point4 p = P.in[i];
    uint32_t slot = 0xFFFFFFFFu;
    bool valid = true;
    if (P.do_crop) {
        // Preprocessing.cpp:176-178: norm < max_range && norm > min_range; label zeroed beyond
        const double norm = sqrt(SAGE_SQNORM3_CROP(p.x * p.x, p.y * p.y, p.z * p.z));   // point.head<3>().norm()
        valid = norm < P.max_range && norm > P.min_range;
        if (norm > P.label_max_range) p.l = 0.0;
    }
    // (a point the crop keeps has finite coordinates; its label, and without the crop its coordinates,
    // still have to be: the reference casts them to int next — undefined for NaN / Inf — so such a frame
    // is refused as a whole, flag bit 1)
    if (valid && !(fabs(p.x) <= 1.7976931348623157e308 && fabs(p.y) <= 1.7976931348623157e308 &&
                   fabs(p.z) <= 1.7976931348623157e308 && fabs(p.l) <= 1.7976931348623157e308)) {
        atomicOr(P.overflow, 2);
        valid = false;
    }
    int group = -1;
    if (valid && P.n_groups >= 0) {
        const int label = static_cast<int>(p.l);
        int off = 0;
        for (int g = 0; g < P.n_groups && group < 0; ++g) {       // first group that lists the label
            const int cnt = P.group_counts[g];
            for (int k = 0; k < cnt; ++k)
                if (P.group_labels[off + k] == label) { group = g; break; }
            off += cnt;
        }
        valid = group >= 0;
    }
    if (valid && P.n_groups >= 0) {
        const double vs = P.group_vs[group] * P.scale;
        const long long vx = static_cast<int>(p.x / vs), vy = static_cast<int>(p.y / vs),
                        vz = static_cast<int>(p.z / vs);
        const long long B = 1ll << 19;
        if (vx < -B || vx >= B || vy < -B || vy >= B || vz < -B || vz >= B) {
            atomicOr(P.overflow, 1);       // voxel index does not fit the 20-bit key fields
            valid = false;
        } else {
            const unsigned long long key = (static_cast<unsigned long long>(group) << 60) |
                                           (static_cast<unsigned long long>(vx + B) << 40) |
                                           (static_cast<unsigned long long>(vy + B) << 20) |
                                           static_cast<unsigned long long>(vz + B);
            uint32_t s = mix64(key) & P.mask;
            for (;;) {
                const unsigned long long old = atomicCAS(P.keys + s, kEmptyKey, key);
                if (old == kEmptyKey || old == key) break;
                s = (s + 1) & P.mask;
            }
            atomicMin(P.winner + s, static_cast<uint32_t>(i));
            slot = s;
        }
    }
    P.tmp[i] = p;
    P.slot_of[i] = slot;
    P.sort_key[i] = valid ? static_cast<uint32_t>(group < 0 ? 0 : group) : kDropped;
    P.sort_val[i] = static_cast<uint32_t>(i);
}

// pass 2: a point survives iff it is the lowest-index point of its voxel
__global__ __launch_bounds__(256) void k_vds_flag(VdsParams P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n) return;
    const uint32_t s = P.slot_of[i];
    if (s != 0xFFFFFFFFu && P.winner[s] != static_cast<uint32_t>(i)) P.sort_key[i] = kDropped;
}

// pass 3 (after the stable sort by group id): number of survivors = first index with key kDropped
__global__ __launch_bounds__(256) void k_vds_count(const uint32_t *sorted_key, int n, uint32_t *n_kept) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool here = sorted_key[i] == kDropped;
    const bool prev = (i > 0) && sorted_key[i - 1] == kDropped;
    if (here && !prev) *n_kept = static_cast<uint32_t>(i);
    if (i == n - 1 && !here) *n_kept = static_cast<uint32_t>(n);
}

// survivors in group-major arrival order; optionally, for the host's replay of the reference's
// emission order (robin_order.hpp), what it needs of each survivor's key: the label group (bits
// 60..63) and the reference's 20-bit VoxelHash of the voxel (core/VoxelHashMap.hpp:72-77; bits
// 0..19) — hashed here, one lane per survivor, instead of by the replaying host thread
__global__ __launch_bounds__(256) void k_vds_gather(const Point4 *tmp, const uint32_t *sorted_val,
                                                    const uint32_t *n_kept, Point4 *out,
                                                    const uint32_t *slot_of,
                                                    const unsigned long long *keys,
                                                    unsigned long long *out_keys) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= *n_kept) return;
    const uint32_t src = sorted_val[i];
    out[i] = tmp[src];
    if (out_keys) {
        const unsigned long long k = keys[slot_of[src]];
        const long long B = 1ll << 19;
        const uint32_t vx = static_cast<uint32_t>(static_cast<int32_t>(static_cast<long long>((k >> 40) & 0xFFFFFu) - B));
        const uint32_t vy = static_cast<uint32_t>(static_cast<int32_t>(static_cast<long long>((k >> 20) & 0xFFFFFu) - B));
        const uint32_t vz = static_cast<uint32_t>(static_cast<int32_t>(static_cast<long long>(k & 0xFFFFFu) - B));
        const uint32_t h = ((1u << 20) - 1u) & (vx * 73856093u ^ vy * 19349663u ^ vz * 83492791u);
        out_keys[i] = (k & 0xF000000000000000ull) | h;
    }
}

__global__ __launch_bounds__(256) void k_vds_permute(const Point4 *in, const uint32_t *perm, uint32_t n,
                                                     Point4 *out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}
void launch_vds_permute(const Point4 *in, const uint32_t *perm, uint32_t n, Point4 *out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_vds_permute, dim3((n + 255) / 256), dim3(256), 0, s, in, perm, n, out);
}

size_t vds_sort_temp_bytes(int n) {
    size_t bytes = 0;
    uint32_t *k = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, static_cast<unsigned>(n), 0, 4);
    return bytes;
}

// One down-sampling level, everything on stream s.  P.sort_key/sort_val hold 2*n entries
// (input | output halves); the survivors land in `out`, their number in *d_n_kept.
hipError_t voxel_downsample_device(const VdsParams &P, void *sort_temp, size_t sort_temp_bytes,
                                   uint32_t *d_n_kept, Point4 *out, hipStream_t s) {
    if (P.n <= 0) return hipMemsetAsync(d_n_kept, 0, sizeof(uint32_t), s);
    hipError_t e = hipMemsetAsync(P.keys, 0xFF, (static_cast<size_t>(P.mask) + 1) * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(P.winner, 0xFF, (static_cast<size_t>(P.mask) + 1) * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_n_kept, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const int grid = (P.n + 255) / 256;
    hipLaunchKernelGGL(k_vds_insert, dim3(grid), dim3(256), 0, s, P);
    hipLaunchKernelGGL(k_vds_flag, dim3(grid), dim3(256), 0, s, P);
    e = rocprim::radix_sort_pairs(sort_temp, sort_temp_bytes, P.sort_key, P.sort_key + P.n, P.sort_val, P.sort_val + P.n,
                                  static_cast<unsigned>(P.n), 0, 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_vds_count, dim3(grid), dim3(256), 0, s, P.sort_key + P.n, P.n, d_n_kept);
    hipLaunchKernelGGL(k_vds_gather, dim3(grid), dim3(256), 0, s, P.tmp, P.sort_val + P.n, d_n_kept, out,
                       P.slot_of, P.keys, P.n_groups >= 0 ? P.out_keys : nullptr);
    return hipGetLastError();
}

}  // namespace sageicp
