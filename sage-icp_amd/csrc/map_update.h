// Device-side map maintenance: VoxelHashMap::Update(points, pose)
// = AddPoints + RemovePointsFarFromLocation (reference core/VoxelHashMap.cpp:149-184, retention
// policy core/VoxelHashMap.hpp:45-70) applied directly to the HBM-resident map.
//
// The result is bit-identical to HostMap (host_map.hpp) in everything that is observable: the
// point blocks (which voxel owns which block, which points it keeps, in which order), the free
// list and the counters.  Only the position of a voxel's slot inside the open-addressed table may
// differ (concurrent claims; evicted voxels leave tombstones instead of a backward shift), which
// no search result depends on.
#pragma once

#include <hip/hip_runtime.h>

#include "sageicp_types.h"

namespace sageicp {

constexpr int kMaxBasicLabels = 32;

struct MapCounters {
    uint32_t blocks_hi;      // high-water mark of block indices
    uint32_t free_count;     // entries of the free list (a stack: the last one is handed out first)
    uint32_t num_voxels;
    uint32_t used_slots;     // live + tombstoned slots (what the load factor is computed from)
    uint64_t total_points;
    uint32_t n_new;          // voxels created by the last update
    uint32_t n_far;          // voxels evicted by the last update
    uint32_t overflow;       // bit 0: a voxel index beyond +-2^20 was seen, bit 1: a non-finite coordinate or label (update rejected)
    uint32_t unit_overflow;  // the pass asks for more units than the point array holds (update rejected before a write)
    // size-classed regions (host_map.hpp): the unit allocator's state
    uint32_t units_hi;       // high-water mark
    uint32_t units_cap;      // units the point array holds
    int32_t free_units_count[4];     // entries of each class's stack of free regions
    uint32_t n_freed;        // regions released by the last insertion pass (pushed onto the stacks after it)
    uint32_t pad;
#ifdef SAGE_UP_TIMING
    unsigned long long dbg_sum[8], dbg_max[8];     // probe: k_up_insert's phases per wave, 10-ns ticks
#endif
};

struct DevMap {
    Slot *table;
    uint32_t mask;
    Point4 *pts;
    int cap;
    uint8_t *zeros;          // unlabelled points per block
    uint32_t *slot_of;       // block -> its slot (kNoSlot for a free block)
    uint32_t *free_list;
    MapCounters *ctr;
    // size-classed regions: block b's points start at unit (regions[b] & 0x0FFFFFFF), class in the top 4 bits
    uint32_t *regions;
    uint32_t *block_of;      // unit -> the block whose region starts there (a slot word carries the unit)
    uint32_t *free_units[4]; // per class: stack of the first units of free regions
    uint32_t *freed;         // scratch: regions released during an insertion pass
    uint32_t class_points[4];
    int n_classes;
};
constexpr uint32_t kDevUnitPoints = kUnitPoints;
constexpr uint32_t kDevNoRegion = kNoRegion;

struct UpdatePolicy {
    double voxel_size;
    double max_dist2;
    int basic, critical;
    int n_labels;
    int labels[kMaxBasicLabels];
};

// What the run heads of an update ask for, laid at the arrival index of the head's point; its
// exclusive prefix sum ranks every request: nw — a new voxel (a block: handed out in ARRIVAL order,
// as the host's sequential loop does), c[k] — a region of class k (new voxel, or one that outgrows
// its region), mg — a region released by such a move, ap — the points the run appends.
struct UpdateEvents {
    uint32_t nw, c[4], mg, ap;
};

struct UpdateScratch {       // device buffers sized for n points / nb blocks (capi.hip reserves them)
    Point4 *raw;             // [n] the caller's points
    Point4 *w;               // [n] transformed into the map frame
    unsigned long long *keys, *keys_alt;   // [n]
    uint32_t *idx, *idx_alt;               // [n]
    uint32_t *head_slot;     // [n]
    UpdateEvents *flag;      // [n + 1]
    UpdateEvents *rank;      // [n + 1] exclusive prefix sum of flag; [n]: the totals
    int8_t *want;            // [n] per run head: the class of the region it asks for, -1: none
    uint32_t *far_flag;      // [nb]
    uint32_t *far_sel;       // [nb]
    uint32_t *n_sel;         // [1]
    // reference-order maps (host_map.hpp: the bucket array of the reference's robin_map lives on the host): what the host
    // replays — the voxels this update created, in arrival order, and the voxels now out of range
    uint2 *new_list;         // optional [n]: {block, the reference's 20-bit hash of the voxel} of the j-th new voxel
    uint2 *far_list;         // optional [nb]: the same for far_sel[j]
    void *temp;
    size_t temp_bytes;
};

size_t map_update_temp_bytes(int n, int nb);

// block_of[] from regions[] (blocks [0, blocks_hi)): when the device-side bookkeeping is set up from
// the host's view of the map
void map_derive_block_of(const DevMap &M, uint32_t blocks_hi, hipStream_t s);

// Update(points, pose): w = pose * raw, insert in order, evict voxels far from pose.translation.
// `blocks_hi_bound` >= the map's block high-water mark after the insert (grid size only).
hipError_t map_update_device(const DevMap &M, const UpdatePolicy &P, const UpdateScratch &S, int n,
                             const double pose[7], uint32_t blocks_hi_bound, hipStream_t s);

// The two halves of map_update_device for a map in reference-order mode: (1) transform + insert, then the far voxels
// are FOUND (far_sel / n_sel / far_list) but nothing is evicted; the host decides which of them the reference's
// erase-while-iterating sweep reaches in this frame (RobinTable::sweep_erase_listed) and (2) evicts exactly those, in
// the order given (the order they go onto the free list).  `d_list` / `d_n`: device memory.
hipError_t map_update_insert_find_far(const DevMap &M, const UpdatePolicy &P, const UpdateScratch &S, int n,
                                      const double pose[7], uint32_t blocks_hi_bound, hipStream_t s);
hipError_t map_evict_listed(const DevMap &M, const uint32_t *d_list, const uint32_t *d_n, uint32_t bound, hipStream_t s);
// Pointcloud() in a given order of the voxels: the points of blocks list[0], list[1], ... (n_list of them) packed into
// `out`; counts / offsets: [n_list + 1] scratch.
hipError_t map_pointcloud_listed(const DevMap &M, const uint32_t *d_list, uint32_t n_list, uint32_t *counts, uint32_t *offsets,
                                 void *temp, size_t temp_bytes, Point4 *out, hipStream_t s);

// Pointcloud() of the HBM copy: the live points of blocks [0, blocks_hi) packed into `out` in
// block-pool order (what HostMap::pointcloud emits).  counts / offsets: [blocks_hi + 1] scratch;
// offsets[blocks_hi] is the number of points written.
hipError_t map_pointcloud_device(const DevMap &M, uint32_t blocks_hi, uint32_t *counts, uint32_t *offsets,
                                 void *temp, size_t temp_bytes, Point4 *out, hipStream_t s);

// Re-insert every live voxel into a fresh (larger or tombstone-free) table.
hipError_t map_rebuild_table(const DevMap &M, Slot *new_table, uint32_t new_mask,
                             uint32_t blocks_hi_bound, hipStream_t s);

}  // namespace sageicp
