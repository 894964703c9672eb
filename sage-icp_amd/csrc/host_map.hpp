// Host-authoritative semantic voxel map: the per-frame map maintenance of
// sage_icp::VoxelHashMap (reference cpp/sage_icp/core/VoxelHashMap.{hpp,cpp}) re-designed as
// flat arrays whose memory IS the device layout (open-addressed slot table + fixed-capacity
// fp64 point blocks), so the device mirror is refreshed by plain dirty-range copies.
//
//   AddPoints                    VoxelHashMap.cpp:162-174  (sequential, order dependent)
//   VoxelBlock::AddPoint policy  VoxelHashMap.hpp:45-70
//   RemovePointsFarFromLocation  VoxelHashMap.cpp:176-184
//   Pointcloud                   VoxelHashMap.cpp:132-142
//   Clear / Empty                VoxelHashMap.hpp:93-94
//
// Size-classed storage (round 3).  A voxel is a BLOCK b (its identity: count, unlabelled count, key,
// place in the iteration order, exactly as before) whose points live in a REGION of the point
// array: `regions[b]` = (class << 28) | first unit, a unit being kUnitPoints consecutive points
// (128 B).  A region holds 4, 8, 16 or (basic + critical) points — the smallest class that holds
// the voxel's count; when a voxel outgrows its region its points move to one of the next class and
// the old region goes to that class's free list.  A two-point voxel of a 0.1 m map then costs
// 128 B instead of 1,280 B.  The SLOT WORD of a voxel is (first unit << 8) | count: a search goes
// from the hash slot straight to the points, without the block (readers never need the class: a
// region starts at unit x kUnitPoints and the count says how far it is filled); the map's own
// bookkeeping gets from a slot to its block through `block_of[unit]`.
//
// Iteration order (Pointcloud(), far-voxel sweep) is block-pool order, not tsl::robin_map
// bucket order; the far-voxel sweep removes EVERY voxel whose first point is out of range
// (the reference erases while iterating its robin_map, which may skip some until a later
// frame).  The search itself (kernels.hip) never depends on either.
// A map in REFERENCE-ORDER mode (`track_order`, set while the map is empty) keeps, next to all of
// the above, the bucket array the reference's tsl::robin_map would have (robin_order.hpp
// RobinTable, fed with every new voxel in arrival order): its sweep erases while iterating that
// array, as VoxelHashMap.cpp:176-184 does — the voxel shifted into a bucket just erased survives
// until a later frame —, and Pointcloud() lists the voxels in bucket order (:132-142).  Such a map
// is maintained on the host only (capi.hip routes its updates here).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <new>
#include <utility>
#include <vector>

#include <sys/mman.h>

#include "robin_order.hpp"
#include "sageicp_types.h"

namespace sageicp {

// The host copy of the voxel blocks: gigabytes for a big map (1,280 B per voxel at 20 + 20 points).
// A std::vector would copy and zero-fill the whole array at every doubling; this store grows with
// mremap (pages move, nothing is copied, untouched blocks are never faulted in) and asks for huge
// pages.  New elements read as zero, like vector::resize(n, Point4{}).
class PointStore {
public:
    PointStore() = default;
    PointStore(const PointStore &o) { *this = o; }
    PointStore(PointStore &&o) noexcept { swap(o); }
    PointStore &operator=(const PointStore &o) {
        if (this == &o) return *this;
        release();
        if (o.n_) {
            resize(o.n_, o.used_hint_);
            std::memcpy(p_, o.p_, std::min(o.n_, o.used_hint_) * sizeof(Point4));
        }
        return *this;
    }
    PointStore &operator=(PointStore &&o) noexcept {
        swap(o);
        return *this;
    }
    ~PointStore() { release(); }

    Point4 *data() { return p_; }
    const Point4 *data() const { return p_; }
    size_t size() const { return n_; }
    Point4 &operator[](size_t i) { return p_[i]; }
    const Point4 &operator[](size_t i) const { return p_[i]; }
    void clear() { release(); }
    // `used`: elements [0, used) may hold data (what a copy has to carry); grow-only otherwise
    void resize(size_t n, size_t used = ~static_cast<size_t>(0)) {
        used_hint_ = std::min(used, n);
        if (n == n_) return;
        const size_t bytes = round_up(n * sizeof(Point4));
        if (n == 0) { release(); return; }
        void *q;
        if (!p_) {
            q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        } else {
            q = mremap(p_, cap_bytes_, bytes, MREMAP_MAYMOVE);
        }
        if (q == MAP_FAILED) throw std::bad_alloc();
        (void)madvise(q, bytes, MADV_HUGEPAGE);
        if (n < n_)          // shrinking keeps zero semantics for a later growth
            std::memset(static_cast<char *>(q) + n * sizeof(Point4), 0,
                        std::min(bytes, n_ * sizeof(Point4)) - n * sizeof(Point4));
        p_ = static_cast<Point4 *>(q);
        cap_bytes_ = bytes;
        n_ = n;
    }
    void set_used(size_t used) { used_hint_ = std::min(used, n_); }

private:
    Point4 *p_ = nullptr;
    size_t n_ = 0, cap_bytes_ = 0, used_hint_ = 0;
    static size_t round_up(size_t b) { return (b + (2u << 20) - 1) & ~static_cast<size_t>((2u << 20) - 1); }
    void release() {
        if (p_) munmap(p_, cap_bytes_);
        p_ = nullptr;
        n_ = cap_bytes_ = used_hint_ = 0;
    }
    void swap(PointStore &o) {
        std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(cap_bytes_, o.cap_bytes_);
        std::swap(used_hint_, o.used_hint_);
    }
};

constexpr uint32_t region_unit(uint32_t r) { return r & 0x0FFFFFFFu; }
constexpr uint32_t region_class(uint32_t r) { return r >> 28; }

class HostMap {
public:
    double voxel_size = 1.0;
    double max_distance = 100.0;
    int basic = 20;
    int critical = 20;
    std::vector<int> basic_labels;
    int cap = 40;  // basic + critical

    std::vector<Slot> table;          // power-of-two capacity
    uint32_t mask = 0;
    uint32_t num_voxels = 0;
    PointStore pts;                   // block b owns pts[unit(b) * kUnitPoints ..) for class_size(class(b)) points
    std::vector<uint32_t> regions;    // per block: (class << 28) | first unit of its region (kNoRegion: free block)
    std::vector<uint32_t> block_of;   // per unit: the block whose region starts there (stale for free regions)
    int n_classes = 1;                // region sizes in points, ascending; the last one is >= cap
    uint32_t class_points[kMaxClasses] = {40, 0, 0, 0};
    std::vector<uint32_t> free_units[kMaxClasses];   // per class: first units of free regions (stacks)
    uint32_t units_hi = 0;            // high-water mark of the unit allocator
    bool regions_all_dirty = true;    // the device copy of `regions` needs a refresh
    std::vector<uint32_t> dirty_regions;             // blocks whose region word changed since the last sync
    std::vector<uint8_t> cnt;         // points in block b (0 = block is free)
    std::vector<uint8_t> zeros;       // how many of them are unlabelled ((int)label == 0)
    std::vector<int32_t> keys;        // 3 ints per block: its voxel key
    std::vector<uint32_t> free_blocks;
    uint32_t blocks_hi = 0;           // high-water mark of allocated block indices
    uint64_t total_points = 0;

    // dirty tracking for the device mirror: the indices of the points and slots written since
    // the last sync (duplicates allowed; values are read at sync time), or "everything"
    bool table_all_dirty = true;
    bool points_all_dirty = true;
    std::vector<uint32_t> dirty_pts;
    std::vector<uint32_t> dirty_slots;
    uint64_t generation = 0;          // bumps on every mutation
    bool track_order = false;         // reference-order mode (see the header of this file)
    RobinTable order;                 // block indices in the reference's bucket order (track_order only)

    HostMap() { reset_table(1024); }

    void configure(double vs, double md, int b, int c, const int *labels, int nl) {
        voxel_size = vs;
        max_distance = md;
        basic = b;
        critical = c;
        cap = b + c;
        basic_labels.assign(labels, labels + nl);
        // size classes 4 / 8 / 16 below the capacity, then the capacity (rounded up to whole units);
        // SAGEICP_SIZE_CLASSES=0 (read when the map is created) keeps the single full-size class
        const char *e = std::getenv("SAGEICP_SIZE_CLASSES");
        const bool classed = !(e && e[0] == '0');
        n_classes = 0;
        if (classed)
            for (uint32_t sz : {4u, 8u, 16u})
                if (sz < static_cast<uint32_t>(cap)) class_points[n_classes++] = sz;
        class_points[n_classes++] = (static_cast<uint32_t>(cap) + kUnitPoints - 1) / kUnitPoints * kUnitPoints;
    }
    uint32_t class_units(uint32_t k) const { return class_points[k] / kUnitPoints; }
    size_t first_point(uint32_t b) const { return static_cast<size_t>(region_unit(regions[b])) * kUnitPoints; }

    bool empty() const { return num_voxels == 0; }

    void clear() {
        reset_table(1024);
        pts.clear();
        regions.clear();
        for (auto &f : free_units) f.clear();
        block_of.clear();
        units_hi = 0;
        regions_all_dirty = true;
        dirty_regions.clear();
        cnt.clear();
        zeros.clear();
        keys.clear();
        free_blocks.clear();
        dirty_pts.clear();
        dirty_slots.clear();
        points_all_dirty = true;
        blocks_hi = 0;
        total_points = 0;
        table_all_dirty = true;
        order.clear();                // (tsl::robin_map::clear() keeps its bucket array: so does this)
        ++generation;
    }

    // Sequential by definition (the retention policy is order dependent), and bound by cache
    // misses on the slot table and the point blocks: a two-stage software prefetch runs ahead of
    // the insertion cursor (slot line for point i+16, then count + block lines for point i+8).
    // Returns 0, or the reason the insertion stopped before point `*stopped_at` (the map stays
    // consistent: the points before it are in): 1 = block / point-slot limit, 2 = voxel index
    // beyond +-2^20 (the device-side update and the tombstone key rely on that range).
    int add_points(const double *xyzl, uint64_t n, uint64_t *stopped_at = nullptr) {
        constexpr uint64_t kFar = 16, kNear = 8;
        int status = 0;
        for (uint64_t i = 0; i < n; ++i) {
            if (i + kFar < n) {
                const double *p = xyzl + 4 * (i + kFar);
                __builtin_prefetch(&table[voxel_hash(static_cast<int32_t>(p[0] / voxel_size),
                                                     static_cast<int32_t>(p[1] / voxel_size),
                                                     static_cast<int32_t>(p[2] / voxel_size)) & mask]);
            }
            if (i + kNear < n) {
                const double *p = xyzl + 4 * (i + kNear);
                const Slot &e = table[voxel_hash(static_cast<int32_t>(p[0] / voxel_size),
                                                 static_cast<int32_t>(p[1] / voxel_size),
                                                 static_cast<int32_t>(p[2] / voxel_size)) & mask];
                if (e.blk != kEmptySlot) {       // a hint only: the home slot may hold another voxel
                    const size_t u = e.blk >> 8, c = e.blk & 255u;
                    if (u < block_of.size()) {
                        __builtin_prefetch(&block_of[u]);
                        const size_t f0 = u * kUnitPoints;
                        __builtin_prefetch(&pts[f0]);
                        __builtin_prefetch(&pts[f0 + (c < static_cast<size_t>(cap) ? c : 0)]);
                    }
                }
            }
            status = add_point(xyzl + 4 * i);
            if (status) {
                if (stopped_at) *stopped_at = i;
                break;
            }
        }
        if (n) ++generation;
        return status;
    }

    void remove_far(const double origin[3]) {
        const double max2 = max_distance * max_distance;
        bool any = false;
        if (track_order) {
            // VoxelHashMap.cpp:177-183 as written: erase while iterating the robin_map
            order.sweep_erase(
                [&](uint32_t b) {
                    const Point4 &p = pts[first_point(b)];
                    const double dx = p.x - origin[0], dy = p.y - origin[1], dz = p.z - origin[2];
                    return SAGE_SQNORM3_FAR(dx * dx, dy * dy, dz * dz) > max2;
                },
                [&](uint32_t b) {
                    erase_block(b);
                    any = true;
                });
            if (any) ++generation;
            return;
        }
        for (uint32_t b = 0; b < blocks_hi; ++b) {
            if (cnt[b] == 0) continue;
            const Point4 &p = pts[first_point(b)];
            const double dx = p.x - origin[0], dy = p.y - origin[1], dz = p.z - origin[2];
            if (SAGE_SQNORM3_FAR(dx * dx, dy * dy, dz * dz) > max2) {
                erase_block(b);
                any = true;
            }
        }
        if (any) ++generation;
    }

    uint64_t pointcloud(double *out, uint64_t capacity) const {
        uint64_t k = 0;
        auto emit = [&](uint32_t b) {
            const Point4 *p = &pts[first_point(b)];
            for (int j = 0; j < cnt[b]; ++j, ++k)
                if (k < capacity) std::memcpy(out + 4 * k, &p[j], 32);
        };
        if (track_order) {
            order.for_each(emit);     // VoxelHashMap.cpp:136-140: bucket order
            return k;
        }
        for (uint32_t b = 0; b < blocks_hi; ++b)
            if (cnt[b]) emit(b);
        return k;
    }

    // Take over the state a device-side update left in HBM (capi.hip ensure_host): the device
    // table (any slot layout, tombstones allowed), the first `bhi` point blocks, the per-block
    // unlabelled counts, the free-list stack and the counters.  The host table is rebuilt
    // tombstone-free at the same capacity, so afterwards it must be uploaded as a whole.
    // `dregions`: the region words of blocks [0, bhi); `dfree_units[k]` / `nfree_units[k]`: the free
    // regions of class k; units [0, uhi) of the point array are filled by the caller afterwards.
    void adopt(const std::vector<Slot> &dtab, size_t blocks_cap, uint32_t bhi, const uint8_t *dzeros,
               const uint32_t *dfree, uint32_t nfree, uint32_t nvox, uint64_t total,
               const uint32_t *dregions, size_t units_cap, uint32_t uhi,
               const uint32_t *const dfree_units[kMaxClasses], const uint32_t nfree_units[kMaxClasses]) {
        reset_table(static_cast<uint32_t>(dtab.size()));
        cnt.assign(blocks_cap, 0);
        zeros.assign(blocks_cap, 0);
        keys.assign(3 * blocks_cap, 0);
        regions.assign(blocks_cap, kNoRegion);
        if (bhi) std::memcpy(regions.data(), dregions, static_cast<size_t>(bhi) * sizeof(uint32_t));
        for (int k = 0; k < kMaxClasses; ++k)
            free_units[k].assign(dfree_units[k], dfree_units[k] + nfree_units[k]);
        units_hi = uhi;
        regions_all_dirty = false;
        dirty_regions.clear();
        pts.resize(units_cap * kUnitPoints, static_cast<size_t>(uhi) * kUnitPoints);   // units [0, uhi) filled by the caller
        block_of.assign(units_cap, 0);
        for (uint32_t b = 0; b < bhi; ++b)
            if (regions[b] != kNoRegion) block_of[region_unit(regions[b])] = b;
        for (const Slot &e : dtab) {
            if (e.blk == kEmptySlot || e.blk == kTombstone) continue;
            table[probe(e.x, e.y, e.z)] = e;
            const uint32_t b = block_of[e.blk >> 8];
            cnt[b] = static_cast<uint8_t>(e.blk & 255u);
            keys[3 * b] = e.x; keys[3 * b + 1] = e.y; keys[3 * b + 2] = e.z;
        }
        if (bhi) std::memcpy(zeros.data(), dzeros, bhi);
        free_blocks.assign(dfree, dfree + nfree);
        blocks_hi = bhi;
        num_voxels = nvox;
        total_points = total;
        dirty_pts.clear();
        dirty_slots.clear();
        table_all_dirty = true;
        points_all_dirty = false;
        ++generation;
    }

    // block -> slot map of the current table (kNoSlot for free blocks), for the device-side update
    std::vector<uint32_t> slot_of_blocks() const {
        std::vector<uint32_t> so(cnt.size(), kNoSlot);
        for (uint32_t s = 0; s <= mask; ++s)
            if (table[s].blk != kEmptySlot) so[block_of[table[s].blk >> 8]] = s;
        return so;
    }

    void clear_dirty() {
        dirty_pts.clear();
        dirty_slots.clear();
        dirty_regions.clear();
        table_all_dirty = false;
        points_all_dirty = false;
        regions_all_dirty = false;
    }

private:
    void reset_table(uint32_t capacity) {
        table.assign(capacity, Slot{0, 0, 0, kEmptySlot});
        mask = capacity - 1;
        num_voxels = 0;
    }

    // returns slot index holding the key, or the empty slot where it would be inserted
    uint32_t probe(int32_t x, int32_t y, int32_t z) const {
        uint32_t s = voxel_hash(x, y, z) & mask;
        for (;;) {
            const Slot &e = table[s];
            if (e.blk == kEmptySlot) return s;
            if (e.x == x && e.y == y && e.z == z) return s;
            s = (s + 1) & mask;
        }
    }

    void grow_table() {
        std::vector<Slot> old;
        old.swap(table);
        const uint32_t capacity = static_cast<uint32_t>(old.size()) * 2;
        table.assign(capacity, Slot{0, 0, 0, kEmptySlot});
        mask = capacity - 1;
        for (const Slot &e : old) {
            if (e.blk == kEmptySlot) continue;
            table[probe(e.x, e.y, e.z)] = e;
        }
        table_all_dirty = true;
        dirty_slots.clear();
    }

    void mark_point(size_t idx) {
        if (points_all_dirty) return;
        if (dirty_pts.size() > pts.size() / 4) {      // cheaper to refresh everything
            points_all_dirty = true;
            dirty_pts.clear();
            return;
        }
        dirty_pts.push_back(static_cast<uint32_t>(idx));
    }
    void mark_slot(uint32_t s) {
        if (table_all_dirty) return;
        if (dirty_slots.size() > table.size() / 4) {
            table_all_dirty = true;
            dirty_slots.clear();
            return;
        }
        dirty_slots.push_back(s);
    }

    void mark_region(uint32_t b) {
        if (regions_all_dirty) return;
        if (dirty_regions.size() > regions.size() / 4) {
            regions_all_dirty = true;
            dirty_regions.clear();
            return;
        }
        dirty_regions.push_back(b);
    }

    uint32_t alloc_block() {
        uint32_t b;
        if (!free_blocks.empty()) {
            b = free_blocks.back();
            free_blocks.pop_back();
        } else {
            b = blocks_hi++;
            if (blocks_hi > cnt.size()) {
                const size_t nb = std::max<size_t>(1024, cnt.size() * 2);
                cnt.resize(nb, 0);
                zeros.resize(nb, 0);
                keys.resize(nb * 3, 0);
                regions.resize(nb, kNoRegion);
                regions_all_dirty = true;          // (the device array is re-allocated with it)
                dirty_regions.clear();
            }
        }
        return b;
    }
    // can a region of class k be had without crossing the unit limit?
    bool region_available(uint32_t k) const {
        return !free_units[k].empty() || static_cast<uint64_t>(units_hi) + class_units(k) <= kMaxUnits;
    }
    uint32_t alloc_region(uint32_t k) {
        uint32_t u;
        if (!free_units[k].empty()) {
            u = free_units[k].back();
            free_units[k].pop_back();
        } else {
            u = units_hi;
            units_hi += class_units(k);
            const size_t need = static_cast<size_t>(units_hi) * kUnitPoints;
            if (need > pts.size()) pts.resize(std::max<size_t>(4096 * kUnitPoints, std::max(need, pts.size() * 2)), need);
            pts.set_used(need);                    // what a copy of the map carries
            if (units_hi > block_of.size()) block_of.resize(pts.size() / kUnitPoints, 0);
        }
        return (k << 28) | u;
    }
    void free_region(uint32_t r) { free_units[region_class(r)].push_back(region_unit(r)); }
    // move block b's `c` points into a region of the next class (its region is full); false: no room
    bool grow_region(uint32_t b, int c) {
        const uint32_t old = regions[b], k = region_class(old) + 1;
        if (!region_available(k)) return false;
        const uint32_t nr = alloc_region(k);
        const size_t from = static_cast<size_t>(region_unit(old)) * kUnitPoints,
                     to = static_cast<size_t>(region_unit(nr)) * kUnitPoints;
        std::memcpy(&pts[to], &pts[from], static_cast<size_t>(c) * sizeof(Point4));
        for (int j = 0; j < c; ++j) mark_point(to + j);
        free_region(old);
        regions[b] = nr;
        block_of[region_unit(nr)] = b;
        mark_region(b);
        return true;
    }

    int add_point(const double *p) {
        // (v3point / voxel_size_).cast<int>(): fp64 divide, truncation toward zero
        const int32_t vx = static_cast<int32_t>(p[0] / voxel_size);
        const int32_t vy = static_cast<int32_t>(p[1] / voxel_size);
        const int32_t vz = static_cast<int32_t>(p[2] / voxel_size);
        constexpr int32_t kLim = 1 << 20;
        if (vx <= -kLim || vx >= kLim || vy <= -kLim || vy >= kLim || vz <= -kLim || vz >= kLim) return 2;
        uint32_t s = probe(vx, vy, vz);
        const Point4 np{p[0], p[1], p[2], p[3]};
        if (table[s].blk == kEmptySlot) {
            // new voxel: its first point is taken unconditionally (VoxelHashMap.cpp:171)
            if ((free_blocks.empty() && blocks_hi + 3u >= (1u << kMaxBlockBits)) || !region_available(0))
                return 1;                    // checked BEFORE anything is touched
            if ((static_cast<uint64_t>(num_voxels) + 1) * 4 > table.size()) {
                grow_table();
                s = probe(vx, vy, vz);
            }
            const uint32_t b = alloc_block();
            regions[b] = alloc_region(0);
            block_of[region_unit(regions[b])] = b;
            mark_region(b);
            pts[first_point(b)] = np;
            cnt[b] = 1;
            zeros[b] = static_cast<int>(p[3]) == 0 ? 1 : 0;
            keys[3 * b] = vx; keys[3 * b + 1] = vy; keys[3 * b + 2] = vz;
            table[s] = Slot{vx, vy, vz, (region_unit(regions[b]) << 8) | 1u};
            ++num_voxels;
            ++total_points;
            mark_slot(s);
            mark_point(first_point(b));
            if (track_order) order.insert(reference_voxel_hash(vx, vy, vz), b);
            return 0;
        }
        const uint32_t b = block_of[table[s].blk >> 8];
        Point4 *blk = &pts[first_point(b)];
        int c = cnt[b];
        bool full = false;                   // the map's point array (not the voxel) is out of room
        auto append = [&]() {
            if (static_cast<uint32_t>(c) == class_points[region_class(regions[b])]) {
                if (!grow_region(b, c)) { full = true; return; }     // nothing was touched
                blk = &pts[first_point(b)];
            }
            blk[c] = np;
            if (static_cast<int>(p[3]) == 0) ++zeros[b];
            cnt[b] = static_cast<uint8_t>(c + 1);
            table[s].blk = (region_unit(regions[b]) << 8) | static_cast<uint32_t>(c + 1);
            ++total_points;
            mark_slot(s);
            mark_point(first_point(b) + c);
        };
        // the incoming point has a non-zero label here; blocks without an unlabelled point (the
        // common case once a voxel is saturated) are skipped without touching their 1.3 KB
        auto replace_first_unlabelled = [&]() {
            if (zeros[b] == 0) return;
            for (int j = 0; j < c; ++j)
                if (static_cast<int>(blk[j].l) == 0) {
                    blk[j] = np;
                    --zeros[b];
                    mark_point(first_point(b) + j);
                    break;
                }
        };
        if (c < basic) {
            append();
            return full ? 1 : 0;
        }
        const int label = static_cast<int>(p[3]);
        if (label == 0) return 0;
        if (std::find(basic_labels.begin(), basic_labels.end(), label) != basic_labels.end()) {
            replace_first_unlabelled();
        } else if (c < basic + critical) {
            append();
        } else {
            replace_first_unlabelled();
        }
        return full ? 1 : 0;
    }

    void erase_block(uint32_t b) {
        uint32_t i = probe(keys[3 * b], keys[3 * b + 1], keys[3 * b + 2]);
        // backward-shift deletion keeps linear-probe chains free of tombstones
        uint32_t j = i;
        for (;;) {
            j = (j + 1) & mask;
            const Slot &e = table[j];
            if (e.blk == kEmptySlot) break;
            const uint32_t k = voxel_hash(e.x, e.y, e.z) & mask;
            const bool stays = (i <= j) ? (i < k && k <= j) : (i < k || k <= j);
            if (stays) continue;
            table[i] = e;
            mark_slot(i);
            i = j;
        }
        table[i] = Slot{0, 0, 0, kEmptySlot};
        mark_slot(i);
        total_points -= cnt[b];
        cnt[b] = 0;
        free_region(regions[b]);
        regions[b] = kNoRegion;
        mark_region(b);
        free_blocks.push_back(b);
        --num_voxels;
    }
};

}  // namespace sageicp
