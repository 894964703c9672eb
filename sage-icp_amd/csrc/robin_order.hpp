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
// The layout of a robin-hood table is not a function of the key set alone (an entry pushed out of
// its bucket walks past the entries that share its home, so the order inside such a group records
// which insertions went through it; growth re-inserts in bucket order; clusters wrap around the end
// of the array — a closed form built on sorting by home bucket was tried twice and reproduces the
// order on no realistic input), so the insertions are replayed:
//   * 0 buckets at first; before an insertion the array doubles (2, 4, 8, ...) when
//     size >= size_t(float(buckets) * 0.5f)                          (max load factor 0.5)
//   * ideal bucket = hash & (buckets - 1); an entry walks forward until it is farther from its
//     ideal bucket than the resident, takes that bucket and pushes the resident on (a resident is
//     displaced only by an entry STRICTLY farther from home)
//   * growth re-inserts the old buckets in index order; iteration is bucket 0 .. buckets-1
// Keys are distinct by construction (one survivor per voxel), so no look-up is needed.
// O(n) expected; ~25 ns per key on the host.
//
// Not modelled — and therefore REFUSED rather than answered wrongly: tsl::robin_map also grows
// when an insertion's probe distance passes a limit (DIST_FROM_IDEAL_BUCKET_LIMIT; the library is
// not in this image and its releases have used limits between 128 and 8192, so the strictest is
// taken).  The replay tracks the largest distance it meets; once that reaches kProbeLimit the order
// it would return is no longer claimed to be the reference's and iteration_order() returns false
// (street scenes stay below 40).  With the reference's 20-bit hash this happens at the latest when
// a label group passes ~2^19 voxels (more buckets than hash values).
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <utility>
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
    std::vector<uint64_t> a, b, dense;
};

class RobinOrderReplay {
public:
    // hashes[i]: reference_voxel_hash of the i-th inserted (distinct) voxel, n < 2^28.  Appends to
    // `order` the insertion indices (+ base) in the table's iteration order.
    static constexpr uint64_t kProbeLimit = 128;
    // returns false (and leaves `order` as it was) when the replay met a probe distance it does
    // not model; max_probe (optional): the largest distance met
    static bool iteration_order(const uint32_t *hashes, size_t n, uint32_t base, std::vector<uint32_t> &order,
                                RobinScratch *scratch = nullptr, uint32_t *max_probe = nullptr) {
        if (n >= (static_cast<size_t>(1) << 27)) return false;     // 28-bit insertion indices, 2^28 buckets
        RobinScratch local;
        RobinOrderReplay t(scratch ? *scratch : local, n);
        // a put is a dependent cache miss into a table of megabytes: the home bucket of the
        // insertion kAhead ahead is requested now
        for (size_t i = 0; i < n; ++i) {
            if (i + kAhead < n && t.buckets_) __builtin_prefetch(t.cur_ + (hashes[i + kAhead] & (t.buckets_ - 1)), 1);
            t.insert(hashes[i], static_cast<uint32_t>(i));
            if (t.max_field_ > kProbeLimit) break;     // (the field holds distance + 1)
        }
        if (max_probe) *max_probe = static_cast<uint32_t>(t.max_field_ ? t.max_field_ - 1 : 0);
        if (t.max_field_ > kProbeLimit) return false;
        for (size_t b = 0; b < t.buckets_; ++b)
            if (const uint64_t e = t.cur_[b]) order.push_back(base + static_cast<uint32_t>(e & kValMask));
        return true;
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
    uint64_t max_field_ = 0;               // largest (distance + 1) any entry was stored with
    size_t thresh_ = 0;                    // size at which the next insertion grows the array first
    uint64_t *dense_ = nullptr;            // the occupied buckets of the array being re-inserted, compacted

    RobinOrderReplay(RobinScratch &s, size_t n) {
        size_t cap = 2;                    // the bucket count the n-th insertion will have seen
        while (static_cast<size_t>(static_cast<float>(cap) * 0.5f) < n) cap *= 2;
        if (s.a.size() < cap) s.a.resize(cap);
        if (s.b.size() < cap) s.b.resize(cap);
        if (s.dense.size() < cap / 2 + 1) s.dense.resize(cap / 2 + 1);
        dense_ = s.dense.data();
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
                if ((e >> 48) > max_field_) max_field_ = e >> 48;
                if (!r) return;
                e = r;
            }
            if ((e >> 48) > kProbeLimit) {        // not modelled beyond here (and the 16-bit field must not wrap)
                max_field_ = e >> 48;
                return;
            }
            e += 1ull << 48;
            b = (b + 1) & mask;
        }
    }
    // Growth re-inserts the old buckets in index order into an empty array of twice the size.
    // The old order is sorted by home bucket (robin-hood invariant; the entries that had wrapped
    // past the old end come first), an entry's new home is its old one or that plus old_n (one more
    // hash bit), so each half of the new array is filled front to back and almost every entry lands
    // without probing.  Two cases are decided without a walk, each provably what put() would do:
    //   * the home bucket is empty: the entry stays there;
    //   * the home bucket is taken, the previous entry placed in this half had a home <= this one's
    //     (not so right after the wrapped entries) and sits at `last` >= home, and bucket last + 1 is
    //     empty: [home, last] is then fully occupied (the previous entry walked through it or sits
    //     on it), every resident there is at least as far from its home as this entry would be (a
    //     cluster is sorted by home; entries that wrapped are farther still), so none yields, and
    //     the entry lands on last + 1.
    // Anything else — the wrapped entries' neighbourhoods, the ends of the halves — goes through
    // put(), after which `last` is not trusted until an entry has found its home bucket empty.
    void grow_fast(const uint64_t *old, size_t old_n) {
        const size_t mask = buckets_ - 1;
        unsigned sh = 0;
        while ((static_cast<size_t>(1) << sh) < old_n) ++sh;
        // (occupied buckets compacted first: testing tens of thousands of buckets that are full or
        // empty at random costs more in mispredicted branches than the placements do)
        uint64_t *dn = dense_;          // (an array of old_n buckets holds at most old_n / 2 entries when it grows)
        size_t k = 0;
        for (size_t j = 0; j < old_n; ++j) {
            dn[k] = old[j];
            k += old[j] != 0;
        }
        size_t last[2] = {0, 0}, last_home[2] = {0, 0};
        bool trust[2] = {false, false};
        for (size_t q = 0; q < k; ++q) {
            const uint64_t e = dn[q] & ((1ull << 48) - 1);
            const size_t home = (e >> 28) & 0xFFFFFu & mask;
            const size_t half = (home >> sh) & 1u;
            size_t pos;
            if (!cur_[home]) {
                pos = home;
            } else if (trust[half] && home >= last_home[half] && last[half] >= home && last[half] + 1 < buckets_ && !cur_[last[half] + 1] &&
                       last[half] + 1 - home < kProbeLimit) {
                pos = last[half] + 1;
            } else {
                put(e | (1ull << 48));
                trust[0] = trust[1] = false;
                continue;
            }
            last[half] = pos;
            last_home[half] = home;
            trust[half] = true;
            const uint64_t d1 = pos - home + 1;
            cur_[pos] = e | (d1 << 48);
            if (d1 > max_field_) max_field_ = d1;
        }
    }
    void grow() {
        const uint64_t *old = cur_;
        const size_t old_n = buckets_;
        std::swap(cur_, other_);
        buckets_ = old_n ? 2 * old_n : 2;
        std::memset(cur_, 0, buckets_ * sizeof(uint64_t));
        if (old_n >= 64) {
            grow_fast(old, old_n);
            return;
        }
        for (size_t j = 0; j < old_n; ++j)
            if (const uint64_t e = old[j]) put((e & ((1ull << 48) - 1)) | (1ull << 48));       // distance 0 again
    }
    void insert(uint32_t h, uint32_t v) {
        if (size_ >= thresh_) {
            grow();
            thresh_ = static_cast<size_t>(static_cast<float>(buckets_) * 0.5f);
        }
        put(pack(h, v, 0));
        ++size_;
    }
};

// The bucket array of a LIVING tsl::robin_map<Voxel, VoxelBlock, VoxelHash> (the reference's
// VoxelHashMap::map_, core/VoxelHashMap.hpp:106): insertions of new voxels in arrival order
// (VoxelHashMap.cpp:166-172), the far-voxel sweep that erases WHILE it iterates (:176-184), clear()
// (VoxelHashMap.hpp:93), copies, and iteration in bucket order (Pointcloud(), :132-142).  The payload
// is the caller's block index.  Used only by maps created in reference-order mode
// (sageicp_map_set_reference_order / SAGEICP_MAP_REFERENCE_ORDER=1, host_map.hpp): what is observable
// of the reference's container beyond the search — which far voxels survive a sweep for another frame,
// and the order Pointcloud() lists the voxels in — is then the reference's.
//   * erase(pos): the bucket is cleared and the entries behind it move one bucket back while their
//     distance from home is > 0 (backward-shift deletion, wrapping around the end of the array)
//   * `for (auto &[k, v] : map_) if (far) map_.erase(k);` — the iterator of the range-for then steps
//     past the bucket that was just refilled by the shift: that entry is not looked at in this sweep
//   * clear() empties the buckets and KEEPS the array (min load factor 0: no shrink), so a map that
//     is cleared and filled again iterates in another order than a fresh one
//   * a copy has the same array
// Same rules, same limit as the replay above: valid() turns false when a probe distance the replay
// does not model is met (kProbeLimit); the table keeps working as a container, its order is then no
// longer claimed to be the reference's.
class RobinTable {
public:
    static constexpr uint64_t kProbeLimit = RobinOrderReplay::kProbeLimit;
    bool valid() const { return max_field_ <= kProbeLimit; }
    size_t size() const { return size_; }
    size_t buckets() const { return b_.size(); }
    uint32_t max_probe() const { return static_cast<uint32_t>(max_field_ ? max_field_ - 1 : 0); }

    void clear() {
        std::fill(b_.begin(), b_.end(), 0ull);
        size_ = 0;
    }
    // hash: reference_voxel_hash of a voxel that is not in the table; val < 2^28
    void insert(uint32_t hash, uint32_t val) {
        if (size_ >= static_cast<size_t>(static_cast<float>(b_.size()) * 0.5f)) grow();
        put(pack(hash, val));
        ++size_;
    }
    // f(val) for every entry, bucket 0 .. buckets-1
    template <class F>
    void for_each(F f) const {
        for (const uint64_t e : b_)
            if (e) f(static_cast<uint32_t>(e & kValMask));
    }
    // the reference's sweep: pred(val) decides, on_erase(val) is told before the entry goes
    template <class P, class E>
    void sweep_erase(P pred, E on_erase) {
        for (size_t i = 0; i < b_.size(); ++i) {
            const uint64_t e = b_[i];
            if (!e) continue;
            const uint32_t v = static_cast<uint32_t>(e & kValMask);
            if (!pred(v)) continue;
            on_erase(v);
            erase_at(i);          // bucket i now holds what stood behind it; the loop moves on to i + 1
        }
    }

    // The same sweep when only the FAR entries are known (a map whose points live in HBM: the device finds the far
    // voxels, the host owns this array): `far` = (hash, val) of every entry the predicate holds for, any order.
    // The full sweep looks at every bucket in index order; an erase at bucket p moves the entries behind it one bucket
    // back, and the loop goes on at p + 1 — so exactly one kind of far entry is NOT erased in this sweep: the one that
    // stood at p + 1 and moved into p, behind the loop.  Every other entry that moved is still ahead of it.  Hence: the
    // far entries in bucket order; one that is found behind the position the loop has reached was skipped, every
    // other one is erased where it stands now.  O(far entries), not O(buckets).
    // That argument needs the array not to be a ring: a run of entries that wraps around its end (bucket 0 holding an
    // entry away from its home) lets a shift carry a skipped entry from bucket 0 to the last bucket — AHEAD of the
    // loop again, which then erases it after all.  Erasures only move entries towards their homes, so a table without
    // such a run before the sweep has none during it; one with it (a few in a thousand sweeps of a small table) takes
    // the sweep as written, with the list as its predicate.
    template <class E>
    void sweep_erase_listed(std::vector<std::pair<uint32_t, uint32_t>> far, E on_erase) {
        if (!b_.empty() && (b_[0] >> 48) > 1) {
            std::vector<uint32_t> vals(far.size());
            for (size_t k = 0; k < far.size(); ++k) vals[k] = far[k].second;
            std::sort(vals.begin(), vals.end());
            sweep_erase([&](uint32_t v) { return std::binary_search(vals.begin(), vals.end(), v); }, on_erase);
            return;
        }
        std::vector<std::pair<size_t, size_t>> at(far.size());          // (bucket before the sweep, index into far)
        for (size_t k = 0; k < far.size(); ++k) at[k] = {find(far[k].first, far[k].second), k};
        std::sort(at.begin(), at.end());
        size_t next = 0;                                                  // the bucket the loop looks at next
        for (const auto &pk : at) {
            const auto &f = far[pk.second];
            const size_t pos = find(f.first, f.second);
            if (pos >= b_.size()) {                                       // not in the table: the caller's list and this array have
                max_field_ = 0xFFFFu;                                     // parted ways — the order is no longer claimed (valid())
                continue;
            }
            if (pos < next) continue;                                     // moved into the bucket just erased: not looked at
            on_erase(f.second);
            erase_at(pos);
            next = pos + 1;
        }
    }
    // the bucket of the entry (hash, val), which must be in the table
    size_t find(uint32_t hash, uint32_t val) const {
        const size_t mask = b_.size() - 1;
        size_t i = hash & mask;
        for (size_t n = 0; n < b_.size(); ++n, i = (i + 1) & mask)
            if (b_[i] && static_cast<uint32_t>(b_[i] & kValMask) == val) return i;
        return b_.size();
    }

private:
    static constexpr uint64_t kValMask = (1ull << 28) - 1;
    std::vector<uint64_t> b_;     // distance + 1 in bits 48..63 (0: empty), hash in 28..47, value in 0..27
    size_t size_ = 0;
    uint64_t max_field_ = 0;

    static uint64_t pack(uint32_t h, uint32_t v) { return (1ull << 48) | (static_cast<uint64_t>(h) << 28) | v; }
    void put(uint64_t e) {
        const size_t mask = b_.size() - 1;
        size_t i = (((e >> 28) & 0xFFFFFu) + ((e >> 48) - 1)) & mask;
        for (;;) {
            const uint64_t r = b_[i];
            if ((e >> 48) > (r >> 48)) {          // strictly farther from home than the resident
                b_[i] = e;
                if ((e >> 48) > max_field_) max_field_ = e >> 48;
                if (!r) return;
                e = r;
            }
            if ((e >> 48) >= 0xFFFFu) {           // (the 16-bit field must not wrap; far beyond kProbeLimit)
                max_field_ = 0xFFFFu;
                // park it in the next empty bucket: the container stays whole, the order is void
                while (b_[i]) i = (i + 1) & mask;
                b_[i] = e;
                return;
            }
            e += 1ull << 48;
            i = (i + 1) & mask;
        }
    }
    void grow() {
        std::vector<uint64_t> old;
        old.swap(b_);
        b_.assign(old.empty() ? 2 : old.size() * 2, 0ull);
        for (const uint64_t e : old)
            if (e) put((e & ((1ull << 48) - 1)) | (1ull << 48));       // distance 0 again, old bucket order
    }
    void erase_at(size_t i) {
        const size_t mask = b_.size() - 1;
        b_[i] = 0;
        --size_;
        size_t prev = i, j = (i + 1) & mask;
        while ((b_[j] >> 48) > 1) {
            b_[prev] = b_[j] - (1ull << 48);
            b_[j] = 0;
            prev = j;
            j = (j + 1) & mask;
        }
    }
};

}  // namespace sageicp
