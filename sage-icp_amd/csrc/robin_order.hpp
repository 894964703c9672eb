// Iteration order of a tsl::robin_map (v1.0.1, default policies) keyed by voxels with the
// reference's 20-bit VoxelHash — host code.
//
// Why the product needs it: sage_icp::VoxelDownsample returns its survivors in the bucket order of
// such a map (core/Preprocessing.cpp:76-82; the grids it fills are default-constructed, :50, and
// grow from zero buckets), the second down-sampling level keeps the FIRST point per voxel in that
// order, and VoxelHashMap::AddPoints' retention policy depends on arrival order.  Emitting the
// survivors in any other order changes which points are registered and kept — the poses of a
// free-running stream then differ from the reference's by centimetres (same accuracy, different
// noise; profiles/README.md).  So the device down-sampling (preprocess.hip), which finds the
// survivors in arrival order, hands their voxel keys to this routine and permutes them.
//
// The layout of a robin-hood table is not a function of the key set alone (growth re-inserts in
// bucket order, clusters wrap around the end of the array), so the insertions are replayed:
//   * 0 buckets at first; before an insertion the array doubles (2, 4, 8, ...) when
//     size >= size_t(float(buckets) * 0.5f)                          (max load factor 0.5)
//   * ideal bucket = hash & (buckets - 1); an entry walks forward until it is farther from its
//     ideal bucket than the resident, takes that bucket and pushes the resident on (a resident is
//     displaced only by an entry STRICTLY farther from home)
//   * growth re-inserts the old buckets in index order; iteration is bucket 0 .. buckets-1
// Keys are distinct by construction (one survivor per voxel), so no look-up is needed.
// O(n) expected; ~25 ns per key on the host.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace sageicp {

// core/VoxelHashMap.hpp:72-77 / core/Preprocessing.cpp:35-40
inline uint32_t reference_voxel_hash(int32_t x, int32_t y, int32_t z) {
    return ((1u << 20) - 1u) & (static_cast<uint32_t>(x) * 73856093u ^ static_cast<uint32_t>(y) * 19349663u ^
                                static_cast<uint32_t>(z) * 83492791u);
}

class RobinOrderReplay {
public:
    // hashes[i]: reference_voxel_hash of the i-th inserted (distinct) voxel, n < 2^28.  Appends to
    // `order` the insertion indices (+ base) in the table's iteration order.
    static void iteration_order(const uint32_t *hashes, size_t n, uint32_t base, std::vector<uint32_t> &order) {
        RobinOrderReplay t;
        for (size_t i = 0; i < n; ++i) t.insert(hashes[i], static_cast<uint32_t>(i));
        for (const uint64_t e : t.slot_)
            if (e) order.push_back(base + static_cast<uint32_t>(e & kValMask));
    }

private:
    // one 8-B word per bucket (one cache line per probe): distance from the ideal bucket + 1 in
    // bits 48..63 (0: empty), the 20-bit hash in bits 28..47, the insertion index in bits 0..27
    static constexpr uint64_t kValMask = (1ull << 28) - 1;
    static uint64_t pack(uint32_t h, uint32_t v, int64_t d) {
        return (static_cast<uint64_t>(d + 1) << 48) | (static_cast<uint64_t>(h) << 28) | v;
    }
    std::vector<uint64_t> slot_;
    size_t size_ = 0;

    void put(uint64_t e) {      // e carries its current distance
        const size_t mask = slot_.size() - 1;
        size_t b = (((e >> 28) & 0xFFFFFu) + ((e >> 48) - 1)) & mask;
        for (;;) {
            const uint64_t r = slot_[b];
            if ((e >> 48) > (r >> 48)) {          // strictly farther from home than the resident
                slot_[b] = e;
                if (!r) return;
                e = r;
            }
            e += 1ull << 48;
            b = (b + 1) & mask;
        }
    }
    void grow() {
        std::vector<uint64_t> old;
        old.swap(slot_);
        slot_.assign(old.empty() ? 2 : 2 * old.size(), 0ull);
        for (const uint64_t e : old)
            if (e) put((e & ((1ull << 48) - 1)) | (1ull << 48));       // distance 0 again
    }
    void insert(uint32_t h, uint32_t v) {
        if (size_ >= static_cast<size_t>(static_cast<float>(slot_.size()) * 0.5f)) grow();
        put(pack(h, v, 0));
        ++size_;
    }
};

}  // namespace sageicp
