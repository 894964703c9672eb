// Device-side VoxelHashMap::Update (gfx950) — see map_update.h.
//
//   reference                                   here
//   Update(points, pose)     VoxelHashMap.cpp:149-160   map_update_device
//   AddPoints                VoxelHashMap.cpp:162-174   k_up_keys .. k_up_insert
//   VoxelBlock::AddPoint     VoxelHashMap.hpp:45-70     apply_policy (same decisions as HostMap)
//   RemovePointsFarFrom...   VoxelHashMap.cpp:176-184   k_far_flags .. k_far_apply
//
// AddPoints is sequential and order dependent, but only INSIDE a voxel: points of different
// voxels never interact.  So the frame is stably sorted by voxel (equal voxels keep their arrival
// order), one lane walks each voxel's run and applies the retention policy in arrival order, and
// the only cross-voxel dependency — which block a new voxel gets — is reproduced exactly: new
// voxels are ranked by the arrival index of their first point (a flag per point + an exclusive
// scan), and rank j takes the j-th entry from the top of the free-list stack, then fresh blocks,
// which is the order the sequential host loop hands them out.  No atomics on hot words: slot
// claims CAS distinct table words, the point count goes through one wave-reduced add.
//
// HBM-bound integer/byte work; per frame ~30-50k points: 32 B in, 32 B out, a 12-B sort record,
// one 16-B probe and one 32-B block write per point — tens of microseconds, not milliseconds.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

#include "map_update.h"

namespace sageicp {

namespace {

constexpr uint32_t kNonHead = 0xFFFFFFFDu;
constexpr int kKeyBias = 1 << 20;

struct Pose34 {
    double R[9];
    double t[3];
};

__device__ __forceinline__ unsigned long long pack_key(int x, int y, int z) {
    return (static_cast<unsigned long long>(static_cast<uint32_t>(x + kKeyBias)) << 42) |
           (static_cast<unsigned long long>(static_cast<uint32_t>(y + kKeyBias)) << 21) |
           static_cast<unsigned long long>(static_cast<uint32_t>(z + kKeyBias));
}
__device__ __forceinline__ void unpack_key(unsigned long long k, int &x, int &y, int &z) {
    x = static_cast<int>((k >> 42) & 0x1FFFFFull) - kKeyBias;
    y = static_cast<int>((k >> 21) & 0x1FFFFFull) - kKeyBias;
    z = static_cast<int>(k & 0x1FFFFFull) - kKeyBias;
}

// w = pose * p (the arithmetic of se3_math.h mat_apply, so host and device agree bit for bit),
// voxel index by fp64 divide + truncation (VoxelHashMap.cpp:165)
__global__ __launch_bounds__(256) void k_up_keys(const Point4 *raw, int n, Pose34 T, double voxel_size,
                                                 Point4 *w, unsigned long long *keys, uint32_t *idx,
                                                 uint32_t *flag, MapCounters *ctr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) flag[n] = 0u;
    if (i >= n) return;
    const Point4 p = raw[i];
    Point4 o;
    o.x = T.R[0] * p.x + T.R[1] * p.y + T.R[2] * p.z + T.t[0];
    o.y = T.R[3] * p.x + T.R[4] * p.y + T.R[5] * p.z + T.t[1];
    o.z = T.R[6] * p.x + T.R[7] * p.y + T.R[8] * p.z + T.t[2];
    o.l = p.l;
    w[i] = o;
    const int vx = static_cast<int>(o.x / voxel_size);
    const int vy = static_cast<int>(o.y / voxel_size);
    const int vz = static_cast<int>(o.z / voxel_size);
    const int lim = kKeyBias - 1;
    if (vx < -lim || vx > lim || vy < -lim || vy > lim || vz < -lim || vz > lim) ctr->overflow = 1u;
    keys[i] = pack_key(vx, vy, vz);
    idx[i] = static_cast<uint32_t>(i);
    flag[i] = 0u;
}

// run heads: look the voxel up (before anything is inserted, so "absent" is definitive) and mark
// the arrival index of the first point of every absent voxel
__global__ __launch_bounds__(256) void k_up_heads(const unsigned long long *keys, const uint32_t *idx,
                                                  int n, DevMap M, uint32_t *head_slot,
                                                  uint32_t *flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (M.ctr->overflow) return;
    const unsigned long long k = keys[i];
    if (i > 0 && keys[i - 1] == k) {
        head_slot[i] = kNonHead;
        return;
    }
    int vx, vy, vz;
    unpack_key(k, vx, vy, vz);
    uint32_t s = voxel_hash(vx, vy, vz) & M.mask;
    uint32_t found = kNoSlot;
    for (;;) {
        const Slot e = M.table[s];
        if (e.blk == kEmptySlot) break;
        if (e.x == vx && e.y == vy && e.z == vz) { found = s; break; }   // tombstones never match
        s = (s + 1) & M.mask;
    }
    head_slot[i] = found;
    if (found == kNoSlot) flag[idx[i]] = 1u;
}

__device__ __forceinline__ bool is_basic_label(const UpdatePolicy &P, int label) {
    for (int i = 0; i < P.n_labels; ++i)
        if (P.labels[i] == label) return true;
    return false;
}

// Regions for the lanes of a wave that need one (`want`: the class, -1: none), called by whole
// waves.  The insertion kernel only POPS from the class stacks (regions released in the pass are
// collected in M.freed and pushed after it), so one counter subtraction per wave and class hands
// out a private interval of stack entries; what the stack cannot cover comes from the bump
// pointer.  Which voxel gets which region is a race between waves — and unobservable: readers go
// through regions[], and the iteration order of the map is the order of its blocks, not of their
// storage.  (A counter update per LANE — thousands on one address — cost the pass 0.2 ms.)
__device__ __forceinline__ uint32_t alloc_regions(const DevMap &M, int want) {
    const unsigned lane = threadIdx.x & 63u;
    uint32_t reg = kDevNoRegion;
    for (int k = 0; k < M.n_classes; ++k) {
        const bool mine = want == k;
        const unsigned long long mask = __ballot(mine);
        if (!mask) continue;
        const int leader = __ffsll(static_cast<long long>(mask)) - 1;
        const int cnt = __popcll(mask);
        const uint32_t units = M.class_points[k] / kDevUnitPoints;
        int old = 0;
        uint32_t bump = 0;
        if (static_cast<int>(lane) == leader) {
            old = atomicSub(&M.ctr->free_units_count[k], cnt);
            const int take = old < 0 ? 0 : (old < cnt ? old : cnt);
            if (take < cnt) {
                atomicAdd(&M.ctr->free_units_count[k], cnt - take);
                bump = atomicAdd(&M.ctr->units_hi, static_cast<uint32_t>(cnt - take) * units);
                if (bump + static_cast<uint32_t>(cnt - take) * units > M.ctr->units_cap) M.ctr->unit_overflow = 1u;
            }
        }
        old = __shfl(old, leader, 64);
        bump = __shfl(bump, leader, 64);
        if (mine) {
            const int take = old < 0 ? 0 : (old < cnt ? old : cnt);
            const int i = __popcll(mask & ((1ull << lane) - 1ull));
            const uint32_t u = i < take ? M.free_units[k][old - 1 - i] : bump + static_cast<uint32_t>(i - take) * units;
            reg = (static_cast<uint32_t>(k) << 28) | u;
            if (u + units > M.ctr->units_cap) reg = kDevNoRegion;      // (flagged above: nothing is written)
        }
    }
    return reg;
}

// one lane per voxel run: claim a slot + block for a new voxel, then the retention policy of
// VoxelBlock::AddPoint (VoxelHashMap.hpp:45-70) over the run in arrival order.  The policy is run
// twice: first on the labels alone, for the count the voxel ends the pass with — it decides the
// size class of its region, allocated once per wave and class — then for real.
__global__ __launch_bounds__(256) void k_up_insert(const unsigned long long *keys, const uint32_t *idx,
                                                   int n, const Point4 *w, const uint32_t *head_slot,
                                                   const uint32_t *rank, DevMap M, UpdatePolicy P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned appended = 0;
    const bool active = !M.ctr->overflow && i < n && head_slot[i] != kNonHead;
    unsigned long long k = 0;
    uint32_t s = kNoSlot, b = 0, reg = kDevNoRegion;
    int c = 0, z = 0, want = -1;
    bool fresh = false;
    if (active) {
        k = keys[i];
        s = head_slot[i];
        fresh = (s == kNoSlot);
        if (!fresh) {
            const uint32_t blk = M.table[s].blk;
            b = blk >> 8;
            c = static_cast<int>(blk & 255u);
            z = M.zeros[b];
            reg = M.regions[b];
        }
        // the count after the run (labels only: what is appended depends on the count and the label)
        int cf = c;
        for (int t = i; t < n && keys[t] == k; ++t) {
            const int label = static_cast<int>(w[idx[t]].l);
            if ((fresh && t == i) || cf < P.basic ||
                (label != 0 && !is_basic_label(P, label) && cf < P.basic + P.critical))
                ++cf;
        }
        int kc = 0;
        while (static_cast<uint32_t>(cf) > M.class_points[kc]) ++kc;
        if (fresh || static_cast<uint32_t>(kc) > (reg >> 28)) want = kc;
    }
    const uint32_t nreg = alloc_regions(M, want);
    uint32_t released = kDevNoRegion;
    if (active && !(want >= 0 && nreg == kDevNoRegion)) {
        int vx, vy, vz;
        unpack_key(k, vx, vy, vz);
        if (fresh) {
            const uint32_t j = rank[idx[i]];
            const uint32_t fc = M.ctr->free_count;
            b = (j < fc) ? M.free_list[fc - 1u - j] : M.ctr->blocks_hi + (j - fc);
            s = voxel_hash(vx, vy, vz) & M.mask;
            for (;;) {     // no key is compared in this phase: the first free slot of the chain is ours
                if (atomicCAS(&M.table[s].blk, kEmptySlot, b << 8) == kEmptySlot) break;
                s = (s + 1) & M.mask;
            }
            M.table[s].x = vx;
            M.table[s].y = vy;
            M.table[s].z = vz;
            reg = nreg;
        } else if (want >= 0) {
            // the voxel outgrows its region: its points move to the new one
            const Point4 *from = M.pts + static_cast<size_t>(reg & 0x0FFFFFFFu) * kDevUnitPoints;
            Point4 *to = M.pts + static_cast<size_t>(nreg & 0x0FFFFFFFu) * kDevUnitPoints;
            for (int j = 0; j < c; ++j) to[j] = from[j];
            released = reg;
            reg = nreg;
        }
        Point4 *blkp = M.pts + static_cast<size_t>(reg & 0x0FFFFFFFu) * kDevUnitPoints;
        for (int t = i; t < n && keys[t] == k; ++t) {
            const Point4 p = w[idx[t]];
            const int label = static_cast<int>(p.l);
            bool append = false, replace = false;
            if (fresh && t == i) {
                append = true;          // a new voxel takes its first point unconditionally (:171)
            } else if (c < P.basic) {
                append = true;
            } else if (label != 0) {
                if (is_basic_label(P, label)) replace = true;
                else if (c < P.basic + P.critical) append = true;
                else replace = true;
            }
            if (append) {
                blkp[c] = p;
                if (label == 0) ++z;
                ++c;
                ++appended;
            } else if (replace && z > 0) {
                for (int j = 0; j < c; ++j)
                    if (static_cast<int>(blkp[j].l) == 0) {
                        blkp[j] = p;
                        --z;
                        break;
                    }
            }
        }
        M.table[s].blk = (b << 8) | static_cast<uint32_t>(c);
        M.zeros[b] = static_cast<uint8_t>(z);
        M.slot_of[b] = s;
        M.regions[b] = reg;
    }
    // released regions: one list append per wave
    {
        const unsigned lane = threadIdx.x & 63u;
        const bool mine = released != kDevNoRegion;
        const unsigned long long mask = __ballot(mine);
        if (mask) {
            const int leader = __ffsll(static_cast<long long>(mask)) - 1;
            uint32_t base = 0;
            if (static_cast<int>(lane) == leader) base = atomicAdd(&M.ctr->n_freed, static_cast<uint32_t>(__popcll(mask)));
            base = __shfl(base, leader, 64);
            if (mine) M.freed[base + __popcll(mask & ((1ull << lane) - 1ull))] = released;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) appended += __shfl_down(appended, off, 64);
    if ((threadIdx.x & 63) == 0 && appended)
        atomicAdd(reinterpret_cast<unsigned long long *>(&M.ctr->total_points),
                  static_cast<unsigned long long>(appended));
}

// Region r of each lane (kDevNoRegion: none) back onto its class's stack; whole waves call this
// together, in kernels that only push: one counter update per wave and class.
__device__ __forceinline__ void push_regions(const DevMap &M, uint32_t r) {
    const unsigned lane = threadIdx.x & 63u;
    for (int k = 0; k < M.n_classes; ++k) {
        const bool mine = r != kDevNoRegion && (r >> 28) == static_cast<uint32_t>(k);
        const unsigned long long mask = __ballot(mine);
        if (!mask) continue;
        const int leader = __ffsll(static_cast<long long>(mask)) - 1;
        int base = 0;
        if (static_cast<int>(lane) == leader) base = atomicAdd(&M.ctr->free_units_count[k], __popcll(mask));
        base = __shfl(base, leader, 64);
        if (mine) M.free_units[k][base + __popcll(mask & ((1ull << lane) - 1ull))] = r & 0x0FFFFFFFu;
    }
}

// regions released by the insertion pass (voxels that moved to a bigger class) onto their class stacks
__global__ __launch_bounds__(256) void k_up_push_freed(DevMap M, uint32_t bound) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    push_regions(M, (i < bound && i < M.ctr->n_freed) ? M.freed[i] : kDevNoRegion);
}

__global__ void k_up_after_insert(MapCounters *ctr, const uint32_t *rank, int n) {
    if (threadIdx.x || blockIdx.x) return;
    if (ctr->overflow) { ctr->n_new = 0; return; }
    const uint32_t n_new = rank[n];
    const uint32_t fc = ctr->free_count;
    const uint32_t from_free = n_new < fc ? n_new : fc;
    ctr->free_count = fc - from_free;
    ctr->blocks_hi += n_new - from_free;
    ctr->num_voxels += n_new;
    ctr->used_slots += n_new;
    ctr->n_new = n_new;
}

// eviction test on the voxel's FIRST point (VoxelHashMap.cpp:179-181), one lane per block
__global__ __launch_bounds__(256) void k_far_flags(DevMap M, UpdatePolicy P, double ox, double oy,
                                                   double oz, uint32_t bound, uint32_t *far_flag) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= bound) return;
    uint32_t far = 0;
    if (!M.ctr->overflow && b < M.ctr->blocks_hi && M.slot_of[b] != kNoSlot) {
        const Point4 p = M.pts[static_cast<size_t>(M.regions[b] & 0x0FFFFFFFu) * kDevUnitPoints];
        const double dx = p.x - ox, dy = p.y - oy, dz = p.z - oz;
        far = (SAGE_SQNORM3(dx * dx, dy * dy, dz * dz) > P.max_dist2) ? 1u : 0u;
    }
    far_flag[b] = far;
}

// evicted voxels in ascending block order (the order the host sweep pushes them on the free list)
__global__ __launch_bounds__(256) void k_far_apply(DevMap M, const uint32_t *sel, const uint32_t *n_sel,
                                                   uint32_t bound) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    unsigned removed = 0;
    uint32_t r = kDevNoRegion;
    if (j < bound && j < *n_sel) {
        const uint32_t b = sel[j];
        const uint32_t s = M.slot_of[b];
        removed = M.table[s].blk & 255u;
        Slot t;
        t.x = kTombKey; t.y = kTombKey; t.z = kTombKey; t.blk = kTombstone;
        M.table[s] = t;
        M.slot_of[b] = kNoSlot;
        M.free_list[M.ctr->free_count + j] = b;
        r = M.regions[b];
        M.regions[b] = kDevNoRegion;
    }
    push_regions(M, r);        // (pushes only in this kernel)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) removed += __shfl_down(removed, off, 64);
    if ((threadIdx.x & 63) == 0 && removed)
        atomicAdd(reinterpret_cast<unsigned long long *>(&M.ctr->total_points),
                  ~static_cast<unsigned long long>(removed) + 1ull);
}

__global__ void k_far_after(MapCounters *ctr, const uint32_t *n_sel) {
    if (threadIdx.x || blockIdx.x) return;
    const uint32_t nf = *n_sel;
    ctr->free_count += nf;
    ctr->num_voxels -= nf;
    ctr->n_far = nf;
}

__global__ __launch_bounds__(256) void k_rebuild(DevMap M, Slot *nt, uint32_t nmask, uint32_t bound) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= bound || b >= M.ctr->blocks_hi) return;
    const uint32_t s = M.slot_of[b];
    if (s == kNoSlot) return;
    const Slot e = M.table[s];
    uint32_t d = voxel_hash(e.x, e.y, e.z) & nmask;
    for (;;) {
        if (atomicCAS(&nt[d].blk, kEmptySlot, e.blk) == kEmptySlot) break;
        d = (d + 1) & nmask;
    }
    nt[d].x = e.x;
    nt[d].y = e.y;
    nt[d].z = e.z;
    M.slot_of[b] = d;
}

// ---- Pointcloud() from the HBM copy (VoxelHashMap.cpp:132-142) --------------------------------
// points held by block b (its voxel's slot word carries the count; a free block holds none)
__global__ __launch_bounds__(256) void k_pc_counts(DevMap M, uint32_t blocks_hi, uint32_t *counts) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b > blocks_hi) return;
    uint32_t c = 0;
    if (b < blocks_hi) {
        const uint32_t s = M.slot_of[b];
        if (s != kNoSlot) c = M.table[s].blk & 255u;
    }
    counts[b] = c;                 // counts[blocks_hi] = 0: its scan entry is the total
}
// lane per (block, slot of the largest class): block b's live points, found through its region,
// land at offsets[b] in block-pool order — the order HostMap::pointcloud emits (host_map.hpp)
__global__ __launch_bounds__(256) void k_pc_gather_regions(DevMap M, uint32_t blocks_hi, uint32_t span,
                                                           const uint32_t *counts, const uint32_t *offsets,
                                                           Point4 *out) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    const uint32_t b = static_cast<uint32_t>(i / span), j = static_cast<uint32_t>(i % span);
    if (b >= blocks_hi || j >= counts[b]) return;
    out[static_cast<size_t>(offsets[b]) + j] =
        M.pts[static_cast<size_t>(M.regions[b] & 0x0FFFFFFFu) * kDevUnitPoints + j];
}
__global__ void k_rebuild_after(MapCounters *ctr) {
    if (threadIdx.x || blockIdx.x) return;
    ctr->used_slots = ctr->num_voxels;
}

}  // namespace

size_t map_update_temp_bytes(int n, int nb) {
    size_t a = 0, b = 0, c = 0;
    unsigned long long *k = nullptr;
    uint32_t *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, a, k, k, v, v, static_cast<size_t>(n), 0, 63);
    (void)rocprim::exclusive_scan(nullptr, b, v, v, 0u, static_cast<size_t>(n) + 1, rocprim::plus<uint32_t>());
    (void)rocprim::select(nullptr, c, rocprim::counting_iterator<uint32_t>(0), v, v, v,
                          static_cast<size_t>(nb));
    size_t d = 0;                  // map_pointcloud_device scans nb + 1 block counts
    (void)rocprim::exclusive_scan(nullptr, d, v, v, 0u, static_cast<size_t>(nb) + 1, rocprim::plus<uint32_t>());
    return std::max(std::max(a, d), std::max(b, c)) + 256;
}

hipError_t map_update_device(const DevMap &M, const UpdatePolicy &P, const UpdateScratch &S, int n,
                             const double pose[7], uint32_t blocks_hi_bound, hipStream_t s) {
    hipError_t e;
    if (n > 0) {
        Pose34 T;
        {   // quat_to_mat (se3_math.h) — evaluated on the host like sageicp_map_update_pose does
            const double x = pose[0], y = pose[1], z = pose[2], w = pose[3];
            const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
            const double wx = w * x, wy = w * y, wz = w * z;
            T.R[0] = 1.0 - 2.0 * (yy + zz); T.R[1] = 2.0 * (xy - wz);       T.R[2] = 2.0 * (xz + wy);
            T.R[3] = 2.0 * (xy + wz);       T.R[4] = 1.0 - 2.0 * (xx + zz); T.R[5] = 2.0 * (yz - wx);
            T.R[6] = 2.0 * (xz - wy);       T.R[7] = 2.0 * (yz + wx);       T.R[8] = 1.0 - 2.0 * (xx + yy);
            T.t[0] = pose[4]; T.t[1] = pose[5]; T.t[2] = pose[6];
        }
        const int grid = (n + 255) / 256;
        hipLaunchKernelGGL(k_up_keys, dim3(grid), dim3(256), 0, s, S.raw, n, T, P.voxel_size, S.w, S.keys,
                           S.idx, S.flag, M.ctr);
        size_t tb = S.temp_bytes;
        e = rocprim::radix_sort_pairs(S.temp, tb, S.keys, S.keys_alt, S.idx, S.idx_alt,
                                      static_cast<size_t>(n), 0, 63, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_up_heads, dim3(grid), dim3(256), 0, s, S.keys_alt, S.idx_alt, n, M,
                           S.head_slot, S.flag);
        tb = S.temp_bytes;
        e = rocprim::exclusive_scan(S.temp, tb, S.flag, S.rank, 0u, static_cast<size_t>(n) + 1,
                                    rocprim::plus<uint32_t>(), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_up_insert, dim3(grid), dim3(256), 0, s, S.keys_alt, S.idx_alt, n, S.w,
                           S.head_slot, S.rank, M, P);
        hipLaunchKernelGGL(k_up_after_insert, dim3(1), dim3(64), 0, s, M.ctr, S.rank, n);
        hipLaunchKernelGGL(k_up_push_freed, dim3(grid), dim3(256), 0, s, M, static_cast<uint32_t>(n));
    }
    if (blocks_hi_bound > 0) {
        const int gb = static_cast<int>((blocks_hi_bound + 255u) / 256u);
        hipLaunchKernelGGL(k_far_flags, dim3(gb), dim3(256), 0, s, M, P, pose[4], pose[5], pose[6],
                           blocks_hi_bound, S.far_flag);
        size_t tb = S.temp_bytes;
        e = rocprim::select(S.temp, tb, rocprim::counting_iterator<uint32_t>(0), S.far_flag, S.far_sel,
                            S.n_sel, static_cast<size_t>(blocks_hi_bound), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_far_apply, dim3(gb), dim3(256), 0, s, M, S.far_sel, S.n_sel, blocks_hi_bound);
        hipLaunchKernelGGL(k_far_after, dim3(1), dim3(64), 0, s, M.ctr, S.n_sel);
    }
    return hipGetLastError();
}

hipError_t map_pointcloud_device(const DevMap &M, uint32_t blocks_hi, uint32_t *counts, uint32_t *offsets,
                                 void *temp, size_t temp_bytes, Point4 *out, hipStream_t s) {
    if (blocks_hi == 0) return hipSuccess;
    const int gb = static_cast<int>((blocks_hi + 1u + 255u) / 256u);
    hipLaunchKernelGGL(k_pc_counts, dim3(gb), dim3(256), 0, s, M, blocks_hi, counts);
    size_t tb = temp_bytes;
    hipError_t e = rocprim::exclusive_scan(temp, tb, counts, offsets, 0u, static_cast<size_t>(blocks_hi) + 1,
                                           rocprim::plus<uint32_t>(), s);
    if (e != hipSuccess) return e;
    const uint32_t span = static_cast<uint32_t>(M.cap);
    const uint64_t nslots = static_cast<uint64_t>(blocks_hi) * span;
    hipLaunchKernelGGL(k_pc_gather_regions, dim3(static_cast<unsigned>((nslots + 255) / 256)), dim3(256), 0, s, M,
                       blocks_hi, span, counts, offsets, out);
    return hipGetLastError();
}

hipError_t map_rebuild_table(const DevMap &M, Slot *new_table, uint32_t new_mask,
                             uint32_t blocks_hi_bound, hipStream_t s) {
    hipError_t e = hipMemsetAsync(new_table, 0xFF, (static_cast<size_t>(new_mask) + 1) * sizeof(Slot), s);
    if (e != hipSuccess) return e;
    if (blocks_hi_bound > 0) {
        const int gb = static_cast<int>((blocks_hi_bound + 255u) / 256u);
        hipLaunchKernelGGL(k_rebuild, dim3(gb), dim3(256), 0, s, M, new_table, new_mask, blocks_hi_bound);
    }
    hipLaunchKernelGGL(k_rebuild_after, dim3(1), dim3(64), 0, s, M.ctr);
    return hipGetLastError();
}

}  // namespace sageicp
