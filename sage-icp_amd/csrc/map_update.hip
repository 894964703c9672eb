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
#include <cstring>

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

// core/VoxelHashMap.hpp:72-77 (robin_order.hpp: reference_voxel_hash) — what the host's bucket array is keyed by
__device__ __forceinline__ uint32_t ref_voxel_hash(int x, int y, int z) {
    return ((1u << 20) - 1u) & (static_cast<uint32_t>(x) * 73856093u ^ static_cast<uint32_t>(y) * 19349663u ^
                                static_cast<uint32_t>(z) * 83492791u);
}
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
                                                 UpdateEvents *flag, MapCounters *ctr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) flag[n] = UpdateEvents{};
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
    // bit 1: a coordinate or label that is not finite (refused: the reference casts it to int, undefined
    // behaviour); bit 0: a voxel index beyond the key's range.  Either rejects the whole update.
    if (!(fabs(o.x) <= 1.7976931348623157e308 && fabs(o.y) <= 1.7976931348623157e308 &&
          fabs(o.z) <= 1.7976931348623157e308 && fabs(o.l) <= 1.7976931348623157e308))
        atomicOr(&ctr->overflow, 2u);
    else if (vx < -lim || vx > lim || vy < -lim || vy > lim || vz < -lim || vz > lim) atomicOr(&ctr->overflow, 1u);
    keys[i] = pack_key(vx, vy, vz);
    idx[i] = static_cast<uint32_t>(i);
    flag[i] = UpdateEvents{};
}

__device__ __forceinline__ bool is_basic_label(const UpdatePolicy &P, int label) {
    for (int i = 0; i < P.n_labels; ++i)
        if (P.labels[i] == label) return true;
    return false;
}

// what the retention policy asks of a label: unlabelled / one of basic_parts_labels / any other.
// Evaluated by every lane for the point at its own position (the label list is read with
// wave-uniform scalar loads: once per wave there, once per POINT inside a head's serial loop).
constexpr int kUnlabelled = 0, kBasicPart = 1, kCritical = 2;
__device__ __forceinline__ int label_code(const UpdatePolicy &P, int label) {
    return label == 0 ? kUnlabelled : (is_basic_label(P, label) ? kBasicPart : kCritical);
}

struct EventsPlus {
    __host__ __device__ UpdateEvents operator()(const UpdateEvents &a, const UpdateEvents &b) const {
        UpdateEvents r;
        r.nw = a.nw + b.nw;
        for (int k = 0; k < 4; ++k) r.c[k] = a.c[k] + b.c[k];
        r.mg = a.mg + b.mg;
        r.ap = a.ap + b.ap;
        return r;
    }
};

// The sorted positions of one workgroup staged in LDS: every lane fetches the key and the point at
// its own position (one parallel gather); run heads then walk their runs through LDS instead of a
// chain of dependent gathers.  A run that leaves the workgroup's 256 positions reads the rest from
// memory.
constexpr unsigned long long kNoKey = ~0ull;          // (no packed voxel key has bit 63 set)
struct RunStage {
    unsigned long long (&sk)[256];
    const unsigned long long *keys;
    const uint32_t *idx;
    const Point4 *w;
    int n, i, tid;
    __device__ __forceinline__ unsigned long long key_at(int r) const {
        const int u = tid + r;
        if (u < 256) return sk[u];
        return i + r < n ? keys[i + r] : kNoKey;
    }
};

// Run heads: look the voxel up (before anything is inserted, so "absent" is definitive), run the
// retention policy of VoxelBlock::AddPoint (VoxelHashMap.hpp:45-70) on the LABELS of the run for
// the count the voxel ends the pass with — it decides the size class of its region — and lay the
// head's requests (UpdateEvents) at the arrival index of its point.
__global__ __launch_bounds__(256) void k_up_heads(const unsigned long long *keys, const uint32_t *idx,
                                                  int n, const Point4 *w, DevMap M, UpdatePolicy P,
                                                  uint32_t *head_slot, int8_t *want_out, UpdateEvents *flag) {
    __shared__ unsigned long long sk[256];
    __shared__ int sl[256];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * 256 + tid;
    {
        unsigned long long ki = kNoKey;
        int li = 0;
        if (i < n) {
            ki = keys[i];
            li = static_cast<int>(w[idx[i]].l);
        }
        sk[tid] = ki;
        sl[tid] = label_code(P, li);
    }
    __syncthreads();
    if (i >= n) return;
    if (M.ctr->overflow) return;
    const unsigned long long k = sk[tid];
    if (i > 0 && (tid > 0 ? sk[tid - 1] : keys[i - 1]) == k) {
        head_slot[i] = kNonHead;
        return;
    }
    int vx, vy, vz;
    unpack_key(k, vx, vy, vz);
    uint32_t s = voxel_hash(vx, vy, vz) & M.mask;
    uint32_t found = kNoSlot;
    int c = 0, cur = -1;
    for (;;) {
        const Slot e = M.table[s];
        if (e.blk == kEmptySlot) break;
        if (e.x == vx && e.y == vy && e.z == vz) {        // tombstones never match
            found = s;
            c = static_cast<int>(e.blk & 255u);
            // its region's class is the smallest that holds its count (host and device grow a
            // voxel that way, and counts never shrink): no need to fetch regions[] for it.  (Were
            // the region bigger, the move asked for below would still land in one that fits.)
            cur = 0;
            while (static_cast<uint32_t>(c) > M.class_points[cur]) ++cur;
            break;
        }
        s = (s + 1) & M.mask;
    }
    const bool fresh = found == kNoSlot;
    // what is appended depends on the count and the label only
    const RunStage R{sk, keys, idx, w, n, i, tid};
    int cf = c;
    for (int r = 0; R.key_at(r) == k; ++r) {
        const int u = tid + r;
        const int code = u < 256 ? sl[u] : label_code(P, static_cast<int>(w[idx[i + r]].l));
        if ((fresh && r == 0) || cf < P.basic || (code == kCritical && cf < P.basic + P.critical)) ++cf;
    }
    int kc = 0;
    while (static_cast<uint32_t>(cf) > M.class_points[kc]) ++kc;
    const int want = (fresh || kc > cur) ? kc : -1;
    head_slot[i] = found;
    want_out[i] = static_cast<int8_t>(want);
    UpdateEvents ev{};
    ev.nw = fresh ? 1u : 0u;
    if (want >= 0) ev.c[want] = 1u;
    ev.mg = (!fresh && want >= 0) ? 1u : 0u;
    ev.ap = static_cast<uint32_t>(cf - c);
    flag[idx[i]] = ev;
}

// Where the regions of this pass come from: class k's requests are ranked 0 .. total_k-1; the first
// fu_k = (entries on the class stack) take the stack from the top, the rest fresh units — class 0's
// fresh range first, then class 1's ...  No counter is touched while the pass runs
// (k_up_after_insert settles them): a request's region is a function of its rank.
struct RegionPlan {
    uint32_t fu[4], base[4], units[4];
    uint32_t units_end;
    __device__ __forceinline__ RegionPlan(const DevMap &M, const UpdateEvents &total) {
        uint32_t at = M.ctr->units_hi;
        for (int k = 0; k < 4; ++k) {
            units[k] = k < M.n_classes ? M.class_points[k] / kDevUnitPoints : 0u;
            fu[k] = k < M.n_classes ? static_cast<uint32_t>(M.ctr->free_units_count[k]) : 0u;
            base[k] = at;
            at += (total.c[k] > fu[k] ? total.c[k] - fu[k] : 0u) * units[k];
        }
        units_end = at;
    }
};

// One lane per voxel run: claim a slot + block for a new voxel, take the region the run's rank
// stands for (RegionPlan), move the voxel's points when it outgrows its region, then the retention
// policy of VoxelBlock::AddPoint (VoxelHashMap.hpp:45-70) over the run in arrival order, the run's
// points read from LDS (RunStage).
__global__ __launch_bounds__(256) void k_up_insert(const unsigned long long *keys, const uint32_t *idx,
                                                   int n, const Point4 *w, const uint32_t *head_slot,
                                                   const int8_t *want_in, const UpdateEvents *rank,
                                                   DevMap M, UpdatePolicy P, uint2 *new_list) {
    __shared__ unsigned long long sk[256];
    __shared__ Point4 sp[256];
    __shared__ uint8_t sc[256];
    const int tid = threadIdx.x;
#ifdef SAGE_UP_TIMING
    unsigned long long tph[7];
    // (everything issued so far has completed when the stamp is taken, and nothing moves across it)
#define UP_STAMP(j) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tph[j]) : : "memory")
    UP_STAMP(0);
#else
#define UP_STAMP(j)
#endif
    const int i = blockIdx.x * 256 + tid;
    {
        unsigned long long ki = kNoKey;
        Point4 pi{0.0, 0.0, 0.0, 0.0};
        if (i < n) {
            ki = keys[i];
            pi = w[idx[i]];
        }
        sk[tid] = ki;
        sp[tid] = pi;
        sc[tid] = static_cast<uint8_t>(label_code(P, static_cast<int>(pi.l)));
    }
    __syncthreads();
    UP_STAMP(1);
    const RunStage R{sk, keys, idx, w, n, i, tid};
    const UpdateEvents total = rank[n];
    const RegionPlan plan(M, total);
    // (more units than the array holds: flagged by k_up_after_insert, nothing is written)
    const bool active = !M.ctr->overflow && plan.units_end <= M.ctr->units_cap && i < n && head_slot[i] != kNonHead;
    UP_STAMP(2);
    UP_STAMP(3);
    UP_STAMP(4);
    if (active) {
        const unsigned long long k = sk[tid];
        uint32_t s = head_slot[i], b = 0, reg = kDevNoRegion;
        int c = 0, z = 0;
        const int want = want_in[i];
        const bool fresh = (s == kNoSlot);
        const UpdateEvents my = rank[idx[i]];
        uint32_t nreg = kDevNoRegion;
        if (want >= 0) {
            const uint32_t j = my.c[want], fu = plan.fu[want];
            const uint32_t u = j < fu ? M.free_units[want][fu - 1u - j] : plan.base[want] + (j - fu) * plan.units[want];
            nreg = (static_cast<uint32_t>(want) << 28) | u;
        }
        UP_STAMP(3);
        if (fresh) {
            int vx, vy, vz;
            unpack_key(k, vx, vy, vz);
            const uint32_t j = my.nw;
            const uint32_t fc = M.ctr->free_count;
            b = (j < fc) ? M.free_list[fc - 1u - j] : M.ctr->blocks_hi + (j - fc);
            s = voxel_hash(vx, vy, vz) & M.mask;
            for (;;) {     // no key is compared in this phase: the first free slot of the chain is ours
                if (atomicCAS(&M.table[s].blk, kEmptySlot, (nreg & 0x0FFFFFFFu) << 8) == kEmptySlot) break;
                s = (s + 1) & M.mask;
            }
            M.table[s].x = vx;
            M.table[s].y = vy;
            M.table[s].z = vz;
            reg = nreg;
            M.block_of[reg & 0x0FFFFFFFu] = b;
            if (new_list) new_list[j] = make_uint2(b, ref_voxel_hash(vx, vy, vz));      // (j: its rank among the new voxels = arrival order)
        } else {
            const uint32_t blk = M.table[s].blk;
            b = M.block_of[blk >> 8];
            c = static_cast<int>(blk & 255u);
            z = M.zeros[b];
            reg = M.regions[b];
            if (want >= 0) {
                // the voxel outgrows its region: its points move to the new one, a unit (4 points)
                // at a time — four loads in flight, then four stores
                const Point4 *from = M.pts + static_cast<size_t>(reg & 0x0FFFFFFFu) * kDevUnitPoints;
                Point4 *to = M.pts + static_cast<size_t>(nreg & 0x0FFFFFFFu) * kDevUnitPoints;
                for (int j = 0; j < c; j += 4) {
                    const Point4 a0 = from[j], a1 = from[j + 1], a2 = from[j + 2], a3 = from[j + 3];
                    to[j] = a0; to[j + 1] = a1; to[j + 2] = a2; to[j + 3] = a3;
                }
                M.freed[my.mg] = reg;
                reg = nreg;
                M.block_of[reg & 0x0FFFFFFFu] = b;
            }
        }
        Point4 *blkp = M.pts + static_cast<size_t>(reg & 0x0FFFFFFFu) * kDevUnitPoints;
        UP_STAMP(4);
        for (int r = 0; R.key_at(r) == k; ++r) {
            const int u = tid + r;
            const Point4 p = u < 256 ? sp[u] : w[idx[i + r]];
            const int code = u < 256 ? sc[u] : label_code(P, static_cast<int>(p.l));
            bool append = false, replace = false;
            if (fresh && r == 0) {
                append = true;          // a new voxel takes its first point unconditionally (:171)
            } else if (c < P.basic) {
                append = true;
            } else if (code != kUnlabelled) {
                if (code == kBasicPart) replace = true;
                else if (c < P.basic + P.critical) append = true;
                else replace = true;
            }
            if (append) {
                blkp[c] = p;
                if (code == kUnlabelled) ++z;
                ++c;
            } else if (replace && z > 0) {
                // the first unlabelled point of the voxel gives way; four labels in flight per step
                // (the region holds whole units of 4 points: the reads stay inside it)
                for (int j = 0; j < c; j += 4) {
                    const double l0 = blkp[j].l, l1 = blkp[j + 1].l, l2 = blkp[j + 2].l, l3 = blkp[j + 3].l;
                    const int f = static_cast<int>(l0) == 0 ? 0 : static_cast<int>(l1) == 0 ? 1 : static_cast<int>(l2) == 0 ? 2
                                : static_cast<int>(l3) == 0 ? 3 : 4;
                    if (f < 4 && j + f < c) {
                        blkp[j + f] = p;
                        --z;
                        break;
                    }
                }
            }
        }
        M.table[s].blk = ((reg & 0x0FFFFFFFu) << 8) | static_cast<uint32_t>(c);
        M.zeros[b] = static_cast<uint8_t>(z);
        M.slot_of[b] = s;
        M.regions[b] = reg;
    }
    UP_STAMP(5);
#ifdef SAGE_UP_TIMING
    UP_STAMP(6);
    for (int j = 0; j < 6; ++j) {
        const unsigned long long d = tph[j + 1] - tph[j];
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&M.ctr->dbg_sum[j], d);
            atomicMax(&M.ctr->dbg_max[j], d);
        }
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&M.ctr->dbg_sum[6], tph[6] - tph[0]);
        atomicMax(&M.ctr->dbg_max[6], tph[6] - tph[0]);
        atomicAdd(&M.ctr->dbg_sum[7], 1ull);
    }
#endif
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

// Regions released by the insertion pass (voxels that moved to a bigger class) onto their class
// stacks: one workgroup, the stack tops kept in LDS while it runs (after k_up_after_insert has
// settled the counters of the pass).
__global__ __launch_bounds__(1024) void k_up_push_freed(DevMap M) {
    __shared__ int top[4];
    if (threadIdx.x < 4) top[threadIdx.x] = M.ctr->free_units_count[threadIdx.x];
    __syncthreads();
    const uint32_t nf = M.ctr->n_freed;
    for (uint32_t i = threadIdx.x; i < nf; i += 1024) {
        const uint32_t r = M.freed[i], k = r >> 28;
        M.free_units[k][atomicAdd(&top[k], 1)] = r & 0x0FFFFFFFu;
    }
    __syncthreads();
    if (threadIdx.x < 4) M.ctr->free_units_count[threadIdx.x] = top[threadIdx.x];
}

__global__ void k_up_after_insert(DevMap M, const UpdateEvents *rank, int n) {
    if (threadIdx.x || blockIdx.x) return;
    MapCounters *ctr = M.ctr;
    ctr->n_freed = 0;
    if (ctr->overflow) { ctr->n_new = 0; return; }
    const UpdateEvents total = rank[n];
    const RegionPlan plan(M, total);
    if (plan.units_end > ctr->units_cap) {      // (the host reserves the worst case)
        ctr->unit_overflow = 1u;
        ctr->n_new = 0;
        return;
    }
    const uint32_t n_new = total.nw;
    const uint32_t fc = ctr->free_count;
    const uint32_t from_free = n_new < fc ? n_new : fc;
    ctr->free_count = fc - from_free;
    ctr->blocks_hi += n_new - from_free;
    ctr->num_voxels += n_new;
    ctr->used_slots += n_new;
    ctr->n_new = n_new;
    for (int k = 0; k < M.n_classes; ++k)
        ctr->free_units_count[k] = static_cast<int32_t>(plan.fu[k] - (total.c[k] < plan.fu[k] ? total.c[k] : plan.fu[k]));
    ctr->units_hi = plan.units_end;
    ctr->n_freed = total.mg;
    ctr->total_points += total.ap;
}

// eviction test on the voxel's FIRST point (VoxelHashMap.cpp:179-181), one lane per block
__global__ __launch_bounds__(256) void k_far_flags(DevMap M, UpdatePolicy P, double ox, double oy,
                                                   double oz, uint32_t bound, uint32_t *far_flag) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= bound) return;
    uint32_t far = 0;
    if (!M.ctr->overflow && !M.ctr->unit_overflow && b < M.ctr->blocks_hi && M.slot_of[b] != kNoSlot) {
        const Point4 p = M.pts[static_cast<size_t>(M.regions[b] & 0x0FFFFFFFu) * kDevUnitPoints];
        const double dx = p.x - ox, dy = p.y - oy, dz = p.z - oz;
        far = (SAGE_SQNORM3_FAR(dx * dx, dy * dy, dz * dz) > P.max_dist2) ? 1u : 0u;
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

__global__ __launch_bounds__(256) void k_far_keys(DevMap M, const uint32_t *sel, const uint32_t *n_sel, uint32_t bound, uint2 *far_list) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= bound || j >= *n_sel) return;
    const uint32_t b = sel[j];
    const Slot e = M.table[M.slot_of[b]];
    far_list[j] = make_uint2(b, ref_voxel_hash(e.x, e.y, e.z));
}
__global__ __launch_bounds__(256) void k_pc_counts_listed(DevMap M, const uint32_t *list, uint32_t n_list, uint32_t *counts) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j > n_list) return;
    uint32_t c = 0;
    if (j < n_list) {
        const uint32_t s = M.slot_of[list[j]];
        if (s != kNoSlot) c = M.table[s].blk & 255u;
    }
    counts[j] = c;
}
__global__ __launch_bounds__(256) void k_pc_gather_listed(DevMap M, const uint32_t *list, uint32_t n_list, uint32_t span,
                                                          const uint32_t *counts, const uint32_t *offsets, Point4 *out) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    const uint32_t q = static_cast<uint32_t>(i / span), j = static_cast<uint32_t>(i % span);
    if (q >= n_list || j >= counts[q]) return;
    out[static_cast<size_t>(offsets[q]) + j] = M.pts[static_cast<size_t>(M.regions[list[q]] & 0x0FFFFFFFu) * kDevUnitPoints + j];
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
__global__ __launch_bounds__(256) void k_derive_block_of(DevMap M, uint32_t blocks_hi) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= blocks_hi) return;
    const uint32_t r = M.regions[b];
    if (r != kDevNoRegion) M.block_of[r & 0x0FFFFFFFu] = b;
}
__global__ void k_rebuild_after(MapCounters *ctr) {
    if (threadIdx.x || blockIdx.x) return;
    ctr->used_slots = ctr->num_voxels;
}

}  // namespace

void map_derive_block_of(const DevMap &M, uint32_t blocks_hi, hipStream_t s) {
    if (blocks_hi)
        hipLaunchKernelGGL(k_derive_block_of, dim3((blocks_hi + 255u) / 256u), dim3(256), 0, s, M, blocks_hi);
}

size_t map_update_temp_bytes(int n, int nb) {
    size_t a = 0, b = 0, c = 0;
    unsigned long long *k = nullptr;
    uint32_t *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, a, k, k, v, v, static_cast<size_t>(n), 0, 63);
    UpdateEvents *ev = nullptr;
    (void)rocprim::exclusive_scan(nullptr, b, ev, ev, UpdateEvents{}, static_cast<size_t>(n) + 1, EventsPlus());
    (void)rocprim::select(nullptr, c, rocprim::counting_iterator<uint32_t>(0), v, v, v,
                          static_cast<size_t>(nb));
    size_t d = 0;                  // map_pointcloud_device scans nb + 1 block counts
    (void)rocprim::exclusive_scan(nullptr, d, v, v, 0u, static_cast<size_t>(nb) + 1, rocprim::plus<uint32_t>());
    return std::max(std::max(a, d), std::max(b, c)) + 256;
}

static hipError_t update_insert(const DevMap &M, const UpdatePolicy &P, const UpdateScratch &S, int n, const double pose[7],
                                hipStream_t s) {
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
        hipLaunchKernelGGL(k_up_heads, dim3(grid), dim3(256), 0, s, S.keys_alt, S.idx_alt, n, S.w, M, P,
                           S.head_slot, S.want, S.flag);
        tb = S.temp_bytes;
        e = rocprim::exclusive_scan(S.temp, tb, S.flag, S.rank, UpdateEvents{}, static_cast<size_t>(n) + 1,
                                    EventsPlus(), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_up_insert, dim3(grid), dim3(256), 0, s, S.keys_alt, S.idx_alt, n, S.w,
                           S.head_slot, S.want, S.rank, M, P, S.new_list);
        hipLaunchKernelGGL(k_up_after_insert, dim3(1), dim3(64), 0, s, M, S.rank, n);
        hipLaunchKernelGGL(k_up_push_freed, dim3(1), dim3(1024), 0, s, M);
    }
    return hipSuccess;
}
static hipError_t find_far(const DevMap &M, const UpdatePolicy &P, const UpdateScratch &S, const double pose[7],
                           uint32_t blocks_hi_bound, hipStream_t s) {
    const int gb = static_cast<int>((blocks_hi_bound + 255u) / 256u);
    hipLaunchKernelGGL(k_far_flags, dim3(gb), dim3(256), 0, s, M, P, pose[4], pose[5], pose[6],
                       blocks_hi_bound, S.far_flag);
    size_t tb = S.temp_bytes;
    return rocprim::select(S.temp, tb, rocprim::counting_iterator<uint32_t>(0), S.far_flag, S.far_sel,
                           S.n_sel, static_cast<size_t>(blocks_hi_bound), s);
}
hipError_t map_evict_listed(const DevMap &M, const uint32_t *d_list, const uint32_t *d_n, uint32_t bound, hipStream_t s) {
    if (bound == 0) return hipSuccess;
    const int gb = static_cast<int>((bound + 255u) / 256u);
    hipLaunchKernelGGL(k_far_apply, dim3(gb), dim3(256), 0, s, M, d_list, d_n, bound);
    hipLaunchKernelGGL(k_far_after, dim3(1), dim3(64), 0, s, M.ctr, d_n);
    return hipGetLastError();
}

hipError_t map_update_device(const DevMap &M, const UpdatePolicy &P, const UpdateScratch &S, int n,
                             const double pose[7], uint32_t blocks_hi_bound, hipStream_t s) {
    hipError_t e = update_insert(M, P, S, n, pose, s);
    if (e != hipSuccess) return e;
    if (blocks_hi_bound > 0) {
        if ((e = find_far(M, P, S, pose, blocks_hi_bound, s)) != hipSuccess) return e;
        if ((e = map_evict_listed(M, S.far_sel, S.n_sel, blocks_hi_bound, s)) != hipSuccess) return e;
    }
    return hipGetLastError();
}

hipError_t map_update_insert_find_far(const DevMap &M, const UpdatePolicy &P, const UpdateScratch &S, int n,
                                      const double pose[7], uint32_t blocks_hi_bound, hipStream_t s) {
    hipError_t e = update_insert(M, P, S, n, pose, s);
    if (e != hipSuccess) return e;
    if (blocks_hi_bound > 0) {
        if ((e = find_far(M, P, S, pose, blocks_hi_bound, s)) != hipSuccess) return e;
        if (S.far_list) {
            const int gb = static_cast<int>((blocks_hi_bound + 255u) / 256u);
            hipLaunchKernelGGL(k_far_keys, dim3(gb), dim3(256), 0, s, M, S.far_sel, S.n_sel, blocks_hi_bound, S.far_list);
        }
    } else {
        e = hipMemsetAsync(S.n_sel, 0, sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

hipError_t map_pointcloud_listed(const DevMap &M, const uint32_t *d_list, uint32_t n_list, uint32_t *counts, uint32_t *offsets,
                                 void *temp, size_t temp_bytes, Point4 *out, hipStream_t s) {
    if (n_list == 0) return hipSuccess;
    const int gb = static_cast<int>((n_list + 1u + 255u) / 256u);
    hipLaunchKernelGGL(k_pc_counts_listed, dim3(gb), dim3(256), 0, s, M, d_list, n_list, counts);
    size_t tb = temp_bytes;
    hipError_t e = rocprim::exclusive_scan(temp, tb, counts, offsets, 0u, static_cast<size_t>(n_list) + 1,
                                           rocprim::plus<uint32_t>(), s);
    if (e != hipSuccess) return e;
    const uint32_t span = static_cast<uint32_t>(M.cap);
    const uint64_t nslots = static_cast<uint64_t>(n_list) * span;
    hipLaunchKernelGGL(k_pc_gather_listed, dim3(static_cast<unsigned>((nslots + 255) / 256)), dim3(256), 0, s, M,
                       d_list, n_list, span, counts, offsets, out);
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
