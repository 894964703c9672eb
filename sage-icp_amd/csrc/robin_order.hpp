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

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace sageicp {

// core/VoxelHashMap.hpp:72-77 / core/Preprocessing.cpp:35-40
inline uint32_t reference_voxel_hash(int32_t x, int32_t y, int32_t z) {
    return ((1u << 20) - 1u) & (static_cast<uint32_t>(x) * 73856093u ^ static_cast<uint32_t>(y) * 19349663u ^
                                static_cast<uint32_t>(z) * 83492791u);
}

// Two bucket arrays that outlive a replay (the caller keeps one RobinScratch per label group): a
// growth step switches to the other array and clears only what it is about to use, instead of
// allocating, zero-filling and page-faulting a fresh block at every doubling of every frame.
struct RobinScratch {
    std::vector<uint64_t> a, b;
};

class RobinOrderReplay {
public:
    // hashes[i]: reference_voxel_hash of the i-th inserted (distinct) voxel, n < 2^28.  Appends to
    // `order` the insertion indices (+ base) in the table's iteration order.
    static void iteration_order(const uint32_t *hashes, size_t n, uint32_t base, std::vector<uint32_t> &order,
                                RobinScratch *scratch = nullptr) {
        RobinScratch local;
        RobinOrderReplay t(scratch ? *scratch : local, n);
        // a put is a dependent cache miss into a table of megabytes: the home bucket of the
        // insertion kAhead ahead is requested now
        for (size_t i = 0; i < n; ++i) {
            if (i + kAhead < n && t.buckets_) __builtin_prefetch(t.cur_ + (hashes[i + kAhead] & (t.buckets_ - 1)), 1);
            t.insert(hashes[i], static_cast<uint32_t>(i));
        }
        for (size_t b = 0; b < t.buckets_; ++b)
            if (const uint64_t e = t.cur_[b]) order.push_back(base + static_cast<uint32_t>(e & kValMask));
    }

private:
    // one 8-B word per bucket (one cache line per probe): distance from the ideal bucket + 1 in
    // bits 48..63 (0: empty), the 20-bit hash in bits 28..47, the insertion index in bits 0..27
    static constexpr uint64_t kValMask = (1ull << 28) - 1;
    static constexpr size_t kAhead = 8;
    static uint64_t pack(uint32_t h, uint32_t v, int64_t d) {
        return (static_cast<uint64_t>(d + 1) << 48) | (static_cast<uint64_t>(h) << 28) | v;
    }
    uint64_t *cur_ = nullptr, *other_ = nullptr;
    size_t buckets_ = 0, size_ = 0;

    RobinOrderReplay(RobinScratch &s, size_t n) {
        size_t cap = 2;                    // the bucket count the n-th insertion will have seen
        while (static_cast<size_t>(static_cast<float>(cap) * 0.5f) < n) cap *= 2;
        if (s.a.size() < cap) s.a.resize(cap);
        if (s.b.size() < cap) s.b.resize(cap);
        cur_ = s.a.data();
        other_ = s.b.data();
    }
    void put(uint64_t e) {      // e carries its current distance
        const size_t mask = buckets_ - 1;
        size_t b = (((e >> 28) & 0xFFFFFu) + ((e >> 48) - 1)) & mask;
        for (;;) {
            const uint64_t r = cur_[b];
            if ((e >> 48) > (r >> 48)) {          // strictly farther from home than the resident
                cur_[b] = e;
                if (!r) return;
                e = r;
            }
            e += 1ull << 48;
            b = (b + 1) & mask;
        }
    }
    // Growth re-inserts the old buckets in index order into an empty array of twice the size.
    // Without wrap-around the old order is sorted by home bucket, an entry's new home is its old one
    // or that plus old_n (one more hash bit), and an entry inserted after everything with a smaller
    // or equal home in its half displaces nobody: it lands on max(home, last position of its half
    // + 1).  That is one sequential pass with no probing.  It is only taken when it is provably the
    // replay's result — no entry of the old array wrapped past its end, and neither half of the new
    // one spills over its own end — otherwise the insertions are replayed one by one (the small
    // arrays at the start of a replay, mostly).
    bool grow_linear(const uint64_t *old, size_t old_n) {
        for (size_t b = 0; b < old_n && old[b]; ++b)          // the cluster at bucket 0, if any
            if ((old[b] >> 48) - 1 > b) return false;          // farther from home than its index: wrapped
        size_t last[2] = {static_cast<size_t>(-1), old_n - 1};
        const size_t mask = buckets_ - 1;
        unsigned sh = 0;
        while ((static_cast<size_t>(1) << sh) < old_n) ++sh;
        for (size_t j = 0; j < old_n; ++j) {
            const uint64_t e = old[j];
            if (!e) continue;
            const size_t home = (e >> 28) & 0xFFFFFu & mask;
            const size_t half = (home >> sh) & 1u;
            const size_t pos = std::max(home, last[half] + 1);
            last[half] = pos;
            if (pos >= (half + 1) * old_n) return false;       // spilled into the other half / past the end
            cur_[pos] = (e & ((1ull << 48) - 1)) | (static_cast<uint64_t>(pos - home + 1) << 48);
        }
        return true;
    }
    void grow() {
        const uint64_t *old = cur_;
        const size_t old_n = buckets_;
        std::swap(cur_, other_);
        buckets_ = old_n ? 2 * old_n : 2;
        std::memset(cur_, 0, buckets_ * sizeof(uint64_t));
        if (old_n >= 64) {
            if (grow_linear(old, old_n)) return;
            std::memset(cur_, 0, buckets_ * sizeof(uint64_t));
        }
        for (size_t j = 0; j < old_n; ++j)
            if (const uint64_t e = old[j]) put((e & ((1ull << 48) - 1)) | (1ull << 48));       // distance 0 again
    }
    void insert(uint32_t h, uint32_t v) {
        if (size_ >= static_cast<size_t>(static_cast<float>(buckets_) * 0.5f)) grow();
        put(pack(h, v, 0));
        ++size_;
    }
};

}  // namespace sageicp
