// The HBM mirror of a map (slot table, point array, compact copy: refreshed lazily by the next search) and Update()
// on the device (row f-2: core/VoxelHashMap.cpp:144-184 through map_update.hip).  Part of libsageicp_hip.so's host side:
// capi_internal.h.
#include "capi_internal.h"

namespace sageicp_impl {

// ---- device mirror ------------------------------------------------------------------------
int reserve_stage(const sageicp_map *m, size_t bytes) {
    if (bytes <= m->stage_bytes) return SAGEICP_OK;
    if (m->h_stage) HIPCHK(hipHostFree(m->h_stage));
    if (m->d_stage) HIPCHK(hipFree(m->d_stage));
    m->h_stage = nullptr; m->d_stage = nullptr; m->stage_bytes = 0;
    const size_t cap = bytes + bytes / 2 + (1u << 20);
    HIPCHK(hipHostMalloc(&m->h_stage, cap, hipHostMallocDefault));
    HIPCHK(hipMalloc(&m->d_stage, cap));
    m->stage_bytes = cap;
    return SAGEICP_OK;
}

// The point array on the device: at least `units` units (+ one NaN point after them: a harmless
// target for an offset of one past the end), the first `keep` units preserved.
int reserve_device_points(const sageicp_map *m, size_t units, size_t keep) {
    if (units <= m->d_units_cap) return SAGEICP_OK;
    hipStream_t s = m->sc.stream;
    Point4 *np_ = nullptr;
    const size_t bytes = units * kUnitPoints * sizeof(Point4);
    HIPCHK(hipMalloc(&np_, bytes + sizeof(Point4)));
    if (keep && m->d_pts)
        HIPCHK(hipMemcpyAsync(np_, m->d_pts, keep * kUnitPoints * sizeof(Point4), hipMemcpyDeviceToDevice, s));
    const double qnan = std::numeric_limits<double>::quiet_NaN();
    const Point4 pad{qnan, qnan, qnan, qnan};
    HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(np_) + bytes, &pad, sizeof(Point4), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    if (m->d_pts) HIPCHK(hipFree(m->d_pts));
    m->d_pts = np_;
    m->d_units_cap = units;
    m->cand_stale = true;
    return SAGEICP_OK;
}
// d_regions for at least `blocks` blocks, the first `keep` preserved, the rest marked free
int reserve_device_regions(const sageicp_map *m, size_t blocks, size_t keep) {
    if (blocks <= m->d_regions_cap) return SAGEICP_OK;
    hipStream_t s = m->sc.stream;
    uint32_t *nr = nullptr;
    HIPCHK(hipMalloc(&nr, blocks * sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(nr, 0xFF, blocks * sizeof(uint32_t), s));      // kNoRegion
    if (keep && m->d_regions)
        HIPCHK(hipMemcpyAsync(nr, m->d_regions, keep * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    if (m->d_regions) HIPCHK(hipFree(m->d_regions));
    m->d_regions = nr;
    m->d_regions_cap = blocks;
    return SAGEICP_OK;
}

// Refresh the HBM mirror from the host-authoritative map.  Everything after a (re)allocation,
// otherwise only the slots and points written since the last sync: they are packed into one
// pinned staging buffer, copied once and scattered by a kernel.
int sync_mirror(const sageicp_map *m) {
    int rc = m->sc.init(m->device);
    if (rc) return rc;
    HIPCHK(hipSetDevice(m->device));
    if (m->on_device) return SAGEICP_OK;      // the HBM copy is the map
    const HostMap &h = m->host;
    hipStream_t s = m->sc.stream;
    bool any = false;
    bool table_full = h.table_all_dirty || m->mirror_stale_all;
    if (h.table.size() != m->d_table_cap) {
        if (m->d_table) HIPCHK(hipFree(m->d_table));
        m->d_table = nullptr; m->d_table_cap = 0;
        HIPCHK(hipMalloc(&m->d_table, h.table.size() * sizeof(Slot)));
        m->d_table_cap = h.table.size();
        table_full = true;
    }
    // (the host arrays grow by doubling; the map itself never holds more than 2^24 units of 4
    // points — HostMap::add_point refuses the point that would cross the limit)
    // (a small map gets the host vector's doubled capacity — a growing map re-allocates rarely —, a
    // big one what it holds and an eighth)
    bool points_full = h.points_all_dirty || m->mirror_stale_all;
    if (h.units_hi > m->d_units_cap) {
        const size_t want = h.units_hi < (1u << 22)
                                ? std::max<size_t>(h.pts.size() / kUnitPoints, h.units_hi)
                                : std::min<size_t>(kMaxUnits, static_cast<size_t>(h.units_hi) + h.units_hi / 8 + 1024);
        if ((rc = reserve_device_points(m, want, 0))) return rc;
        points_full = true;
    }
    bool regions_full = h.regions_all_dirty || m->mirror_stale_all;
    if (h.regions.size() > m->d_regions_cap) {
        if ((rc = reserve_device_regions(m, h.regions.size(), 0))) return rc;
        regions_full = true;
    }
    if (regions_full && h.blocks_hi) {
        HIPCHK(hipMemcpyAsync(m->d_regions, h.regions.data(), h.blocks_hi * sizeof(uint32_t),
                              hipMemcpyHostToDevice, s));
        any = true;
    }
    if (table_full) {
        HIPCHK(hipMemcpyAsync(m->d_table, h.table.data(), h.table.size() * sizeof(Slot),
                              hipMemcpyHostToDevice, s));
        any = true;
    }
    if (points_full && h.units_hi) {
        HIPCHK(hipMemcpyAsync(m->d_pts, h.pts.data(), static_cast<size_t>(h.units_hi) * kUnitPoints * sizeof(Point4),
                              hipMemcpyHostToDevice, s));
        any = true;
    }
    const size_t ns = table_full ? 0 : h.dirty_slots.size();
    const size_t np = points_full ? 0 : h.dirty_pts.size();
    const size_t nr = regions_full ? 0 : h.dirty_regions.size();
    if (ns || np || nr) {
        // staging layout: [slot idx][point idx][region idx][region values][slot values][point values],
        // 32-B aligned parts
        auto up = [](size_t x) { return (x + 31) & ~static_cast<size_t>(31); };
        const size_t o_si = 0, o_pi = up(o_si + ns * 4), o_ri = up(o_pi + np * 4), o_rv = up(o_ri + nr * 4),
                     o_sv = up(o_rv + nr * 4),
                     o_pv = up(o_sv + ns * sizeof(Slot)), total = o_pv + np * sizeof(Point4);
        if ((rc = reserve_stage(m, total))) return rc;
        char *hs = static_cast<char *>(m->h_stage);
        uint32_t *si = reinterpret_cast<uint32_t *>(hs + o_si);
        uint32_t *pi = reinterpret_cast<uint32_t *>(hs + o_pi);
        Slot *sv = reinterpret_cast<Slot *>(hs + o_sv);
        Point4 *pv = reinterpret_cast<Point4 *>(hs + o_pv);
        for (size_t i = 0; i < ns; ++i) { si[i] = h.dirty_slots[i]; sv[i] = h.table[h.dirty_slots[i]]; }
        for (size_t i = 0; i < np; ++i) { pi[i] = h.dirty_pts[i]; pv[i] = h.pts[h.dirty_pts[i]]; }
        uint32_t *ri = reinterpret_cast<uint32_t *>(hs + o_ri), *rv = reinterpret_cast<uint32_t *>(hs + o_rv);
        for (size_t i = 0; i < nr; ++i) { ri[i] = h.dirty_regions[i]; rv[i] = h.regions[h.dirty_regions[i]]; }
        HIPCHK(hipMemcpyAsync(m->d_stage, m->h_stage, total, hipMemcpyHostToDevice, s));
        char *ds = static_cast<char *>(m->d_stage);
        launch_scatter_u32(reinterpret_cast<uint32_t *>(ds + o_ri), reinterpret_cast<uint32_t *>(ds + o_rv),
                           static_cast<uint32_t>(nr), m->d_regions, s);
        launch_scatter_slots(reinterpret_cast<uint32_t *>(ds + o_si), reinterpret_cast<Slot *>(ds + o_sv),
                             static_cast<uint32_t>(ns), m->d_table, s);
        launch_scatter_points(reinterpret_cast<uint32_t *>(ds + o_pi),
                              reinterpret_cast<Point4 *>(ds + o_pv), static_cast<uint32_t>(np),
                              m->d_pts, s);
        HIPCHK(hipGetLastError());
        any = true;
    }
    if (any) {
        HIPCHK(hipStreamSynchronize(s));
        m->cand_stale = true;
    }
    const_cast<HostMap &>(h).clear_dirty();
    m->mirror_stale_all = false;
    return SAGEICP_OK;
}

// The compact copy the scan reads (kernels.hip, k_derive_cand): rebuilt from the HBM copy of the
// map when that has changed (mirror refresh, device-side update, clone).  One pass over the hash
// table and the live points; the ICP loop that follows reads the map ~150 times.
// (`derive` false: the coming search scans the full records — small frames, sparse voxels — so only
// the allocation is kept in step and the copy stays marked stale for the search that wants it)
int ensure_cand(const sageicp_map *m, bool derive) {
    hipStream_t s = m->sc.stream;
    const size_t slots = m->d_units_cap * kUnitPoints;
    if (!m->d_cand_flags) {
        HIPCHK(hipMalloc(&m->d_cand_flags, 16));
        HIPCHK(hipMemsetAsync(m->d_cand_flags, 0, 16, s));
    }
    if (!derive) return SAGEICP_OK;             // (this search reads the full records: no copy is made for it)
    if (slots > m->d_cand_slots) {
        if (m->d_cand) HIPCHK(hipFree(m->d_cand));
        m->d_cand = nullptr; m->d_cand_slots = 0;
        HIPCHK(hipMalloc(&m->d_cand, (slots + 1) * sizeof(uint4)));
        m->d_cand_slots = slots;
        m->cand_stale = true;
    }
    if (!m->cand_stale) return SAGEICP_OK;
    HIPCHK(hipMemsetAsync(m->d_cand_flags, 0, 16, s));
    if (m->d_table && m->d_pts && slots)
        launch_derive_cand(m->d_table, static_cast<uint32_t>(m->d_table_cap), m->d_pts, m->d_cand, slots,
                           m->d_cand_flags, s);
    HIPCHK(hipGetLastError());
    m->cand_stale = false;
    return SAGEICP_OK;
}

// ---- device-side Update() (row f-2) -----------------------------------------------------------
bool map_is_empty(const sageicp_map *m) {
    return m->on_device ? m->ctr.num_voxels == 0 : m->host.empty();
}

// Bring `host` up to date after device-side updates: download table, blocks, counts and free list
// and let HostMap rebuild itself from them.  The host table is rebuilt without tombstones, so the
// device table (and the block -> slot map) is stale afterwards and is re-uploaded on next use.
int ensure_host(const sageicp_map *m) {
    if (!m->on_device) return SAGEICP_OK;
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = m->sc.stream;
    HostMap &h = const_cast<HostMap &>(m->host);
    const MapCounters c = m->ctr;
    std::vector<Slot> tab(m->d_table_cap);
    std::vector<uint8_t> zeros(std::max<uint32_t>(c.blocks_hi, 1));
    std::vector<uint32_t> fl(std::max<uint32_t>(c.free_count, 1));
    std::vector<uint32_t> regs(std::max<uint32_t>(c.blocks_hi, 1));
    std::vector<uint32_t> fu[kMaxClasses];
    const uint32_t *fu_ptr[kMaxClasses];
    uint32_t fu_n[kMaxClasses];
    for (int k = 0; k < kMaxClasses; ++k) {
        fu_n[k] = static_cast<uint32_t>(std::max(0, c.free_units_count[k]));
        fu[k].resize(std::max<uint32_t>(fu_n[k], 1));
        fu_ptr[k] = fu[k].data();
        if (fu_n[k])
            HIPCHK(hipMemcpyAsync(fu[k].data(), m->d_free_units[k], fu_n[k] * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    if (c.blocks_hi)
        HIPCHK(hipMemcpyAsync(regs.data(), m->d_regions, c.blocks_hi * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(tab.data(), m->d_table, tab.size() * sizeof(Slot), hipMemcpyDeviceToHost, s));
    if (c.blocks_hi)
        HIPCHK(hipMemcpyAsync(zeros.data(), m->d_zeros, c.blocks_hi, hipMemcpyDeviceToHost, s));
    if (c.free_count)
        HIPCHK(hipMemcpyAsync(fl.data(), m->d_free, c.free_count * sizeof(uint32_t),
                              hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    h.adopt(tab, std::max<size_t>(m->d_blocks_cap, c.blocks_hi), c.blocks_hi, zeros.data(), fl.data(), c.free_count,
            c.num_voxels, c.total_points, regs.data(), m->d_units_cap, c.units_hi, fu_ptr, fu_n);
    if (c.units_hi) {
        HIPCHK(hipMemcpyAsync(h.pts.data(), m->d_pts,
                              static_cast<size_t>(c.units_hi) * kUnitPoints * sizeof(Point4),
                              hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    m->on_device = false;
    m->aux_valid = false;
    return SAGEICP_OK;
}

int reserve_update_scratch(const sageicp_map *m, size_t n, size_t nb) {
    UpdateScratch &u = m->up;
    if (n > m->up_n) {
        const size_t c = n + n / 2 + 1024;
        void *olds[] = {u.raw, u.w, u.keys, u.keys_alt, u.idx, u.idx_alt, u.head_slot, u.flag, u.rank, u.want, u.new_list};
        for (void *q : olds)
            if (q) HIPCHK(hipFree(q));
        u.raw = u.w = nullptr; u.keys = u.keys_alt = nullptr; u.idx = u.idx_alt = u.head_slot = nullptr;
        u.flag = u.rank = nullptr; u.want = nullptr; u.new_list = nullptr;
        m->up_n = 0;
        HIPCHK(hipMalloc(&u.raw, c * sizeof(Point4)));
        HIPCHK(hipMalloc(&u.w, c * sizeof(Point4)));
        HIPCHK(hipMalloc(&u.keys, c * sizeof(unsigned long long)));
        HIPCHK(hipMalloc(&u.keys_alt, c * sizeof(unsigned long long)));
        HIPCHK(hipMalloc(&u.idx, c * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&u.idx_alt, c * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&u.head_slot, c * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&u.flag, (c + 1) * sizeof(UpdateEvents)));
        HIPCHK(hipMalloc(&u.rank, (c + 1) * sizeof(UpdateEvents)));
        HIPCHK(hipMalloc(&u.want, c));
        HIPCHK(hipMalloc(&u.new_list, c * sizeof(uint2)));
        m->up_n = c;
    }
    if (nb > m->up_nb) {
        const size_t c = nb + nb / 2 + 1024;
        if (u.far_flag) HIPCHK(hipFree(u.far_flag));
        if (u.far_sel) HIPCHK(hipFree(u.far_sel));
        if (u.far_list) HIPCHK(hipFree(u.far_list));
        u.far_flag = u.far_sel = nullptr;
        u.far_list = nullptr;
        m->up_nb = 0;
        HIPCHK(hipMalloc(&u.far_flag, c * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&u.far_sel, c * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&u.far_list, c * sizeof(uint2)));
        m->up_nb = c;
    }
    if (!u.n_sel) HIPCHK(hipMalloc(&u.n_sel, sizeof(uint32_t)));
    const size_t tb = map_update_temp_bytes(static_cast<int>(m->up_n), static_cast<int>(m->up_nb));
    if (tb > u.temp_bytes) {
        if (u.temp) HIPCHK(hipFree(u.temp));
        u.temp = nullptr; u.temp_bytes = 0;
        HIPCHK(hipMalloc(&u.temp, tb));
        u.temp_bytes = tb;
    }
    return SAGEICP_OK;
}

// (re)allocate the per-block device arrays for `blocks` blocks, keeping the first `keep` blocks
int grow_device_blocks(const sageicp_map *m, size_t blocks, size_t keep) {
    hipStream_t s = m->sc.stream;
    if (int rc = reserve_device_regions(m, blocks, keep)) return rc;
    if (blocks > m->d_blocks_cap) m->d_blocks_cap = blocks;
    if (m->d_blocks_cap > m->d_aux_cap) {
        const size_t nb = m->d_blocks_cap;
        uint8_t *z = nullptr;
        uint32_t *so = nullptr, *fl = nullptr;
        HIPCHK(hipMalloc(&z, nb));
        HIPCHK(hipMalloc(&so, nb * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&fl, nb * sizeof(uint32_t)));
        HIPCHK(hipMemsetAsync(z, 0, nb, s));
        HIPCHK(hipMemsetAsync(so, 0xFF, nb * sizeof(uint32_t), s));        // kNoSlot
        if (keep && m->d_zeros) {
            HIPCHK(hipMemcpyAsync(z, m->d_zeros, keep, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(so, m->d_slot_of, keep * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(fl, m->d_free, std::min(keep, m->d_aux_cap) * sizeof(uint32_t),
                                  hipMemcpyDeviceToDevice, s));
        }
        HIPCHK(hipStreamSynchronize(s));
        if (m->d_zeros) HIPCHK(hipFree(m->d_zeros));
        if (m->d_slot_of) HIPCHK(hipFree(m->d_slot_of));
        if (m->d_free) HIPCHK(hipFree(m->d_free));
        m->d_zeros = z; m->d_slot_of = so; m->d_free = fl;
        m->d_aux_cap = nb;
    }
    if (!m->d_ctr) {
        HIPCHK(hipMalloc(&m->d_ctr, sizeof(MapCounters)));
        HIPCHK(hipHostMalloc(&m->h_ctr, sizeof(MapCounters), hipHostMallocDefault));
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&m->h_ctr_aux), 16 * sizeof(uint32_t), hipHostMallocDefault));
    }
    return SAGEICP_OK;
}

// the unit allocator's device arrays: per-class stacks able to hold every region the point array
// can be cut into, and the scratch list of one pass's released regions (at most one per point)
int reserve_unit_stacks(const sageicp_map *m, size_t n) {
    hipStream_t s = m->sc.stream;
    const HostMap &h = m->host;
    for (int k = 0; k < h.n_classes; ++k) {
        const size_t need = m->d_units_cap / h.class_units(k) + 1;
        if (need <= m->d_free_units_cap[k]) continue;
        uint32_t *nf = nullptr;
        HIPCHK(hipMalloc(&nf, need * sizeof(uint32_t)));
        const size_t keep = m->on_device ? static_cast<size_t>(std::max(0, m->ctr.free_units_count[k])) : 0;
        if (keep)
            HIPCHK(hipMemcpyAsync(nf, m->d_free_units[k], keep * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        if (m->d_free_units[k]) HIPCHK(hipFree(m->d_free_units[k]));
        m->d_free_units[k] = nf;
        m->d_free_units_cap[k] = need;
    }
    if (m->d_units_cap > m->d_block_of_cap) {
        uint32_t *nb = nullptr;
        HIPCHK(hipMalloc(&nb, m->d_units_cap * sizeof(uint32_t)));
        if (m->on_device && m->d_block_of && m->ctr.units_hi)
            HIPCHK(hipMemcpyAsync(nb, m->d_block_of, static_cast<size_t>(m->ctr.units_hi) * sizeof(uint32_t),
                                  hipMemcpyDeviceToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        if (m->d_block_of) HIPCHK(hipFree(m->d_block_of));
        m->d_block_of = nb;
        m->d_block_of_cap = m->d_units_cap;
        if (!m->on_device) m->aux_valid = false;        // (derived from the host's view below)
    }
    if (n > m->d_freed_cap) {
        if (m->d_freed) HIPCHK(hipFree(m->d_freed));
        m->d_freed = nullptr; m->d_freed_cap = 0;
        const size_t c = n + n / 2 + 1024;
        HIPCHK(hipMalloc(&m->d_freed, c * sizeof(uint32_t)));
        m->d_freed_cap = c;
    }
    return SAGEICP_OK;
}

DevMap dev_map(const sageicp_map *m) {
    DevMap dm{};
    dm.table = m->d_table;
    dm.mask = static_cast<uint32_t>(m->d_table_cap - 1);
    dm.pts = m->d_pts;
    dm.cap = m->host.cap;
    dm.zeros = m->d_zeros;
    dm.slot_of = m->d_slot_of;
    dm.free_list = m->d_free;
    dm.ctr = m->d_ctr;
    dm.regions = m->d_regions;
    dm.block_of = m->d_block_of;
    for (int k = 0; k < kMaxClasses; ++k) {
        dm.free_units[k] = m->d_free_units[k];
        dm.class_points[k] = k < m->host.n_classes ? static_cast<uint32_t>(m->host.class_points[k]) : 0u;
    }
    dm.freed = m->d_freed;
    dm.n_classes = m->host.n_classes;
    return dm;
}

// VoxelHashMap::Update(points, pose) on the device.
// `d_points`: the points are already in HBM (the pipeline's down-sampled frame); else `xyzl` (host).
int device_update(sageicp_map *m, const double *xyzl, uint64_t n, const double pose[7],
                  const Point4 *d_points) {
    if (m->host.basic_labels.size() > static_cast<size_t>(kMaxBasicLabels))
        return fail(SAGEICP_ERR_INVALID, "device map update supports at most 32 basic_parts_labels");
    if (n > 0x3FFFFFFFull) return fail(SAGEICP_ERR_INVALID, "too many points");
    int rc = m->sc.init(m->device);
    if (rc) return rc;
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = m->sc.stream;
    const HostMap &h = m->host;
    if (!m->on_device) {
        if ((rc = sync_mirror(m))) return rc;       // table + points as the host has them
        m->ctr = MapCounters{};
        m->ctr.blocks_hi = h.blocks_hi;
        m->ctr.free_count = static_cast<uint32_t>(h.free_blocks.size());
        m->ctr.num_voxels = h.num_voxels;
        m->ctr.used_slots = h.num_voxels;
        m->ctr.total_points = h.total_points;
        m->ctr.units_hi = h.units_hi;
        for (int k = 0; k < h.n_classes; ++k) m->ctr.free_units_count[k] = static_cast<int32_t>(h.free_units[k].size());
    }
    // capacity for the worst case (every point opens a voxel); the host rule is load <= 1/4
    const uint64_t need_blocks = static_cast<uint64_t>(m->ctr.blocks_hi) + n;
    if (need_blocks + 3 >= (1ull << kMaxBlockBits)) return fail(SAGEICP_ERR_CAPACITY, "more than 2^24 voxels");
    size_t blocks = m->d_blocks_cap;
    // (growth: doubling while the arrays are small — a growing map re-allocates rarely —, by a quarter
    // beyond 4 M blocks / units, where a doubled array would be most of the map's footprint)
    auto grown = [](size_t cap) { return cap < (size_t{1} << 22) ? 2 * cap : cap + cap / 4; };
    if (need_blocks > blocks) blocks = std::max<size_t>(need_blocks, std::max<size_t>(1024, grown(blocks)));
    if ((rc = grow_device_blocks(m, blocks, m->ctr.blocks_hi))) return rc;
    // ... and of units.  One region per voxel run at most: a run into a new voxel takes at most
    // `per_point` units per point of it (one with the reference's capacities: 1 unit for 1-4
    // points, 2 for 5-8, 4 for 9-16, 10 beyond), a run into
    // an existing voxel at worst moves it into a region of the last class — and there are no more
    // such runs than voxels.  What the pass really needs is known on the device only
    // (k_up_heads); should it exceed an array already at its limit of 2^24 units, the pass flags
    // that before anything is written and the call fails below.
    uint64_t per_point = 1;      // (a region of class k is first taken by a run of class_points[k-1] + 1 points)
    for (int k = 0; k < h.n_classes; ++k) {
        const uint64_t least = k ? h.class_points[k - 1] + 1u : 1u;
        per_point = std::max<uint64_t>(per_point, (h.class_units(k) + least - 1) / least);
    }
    const uint64_t moving = std::min<uint64_t>(n, m->ctr.num_voxels);
    const uint64_t need_units = std::min<uint64_t>(
        kMaxUnits, static_cast<uint64_t>(m->ctr.units_hi) + n * per_point + moving * h.class_units(h.n_classes - 1));
    if (need_units > m->d_units_cap) {
        const size_t units = std::min<size_t>(kMaxUnits, std::max<size_t>(need_units, std::max<size_t>(4096, grown(m->d_units_cap))));
        if ((rc = reserve_device_points(m, units, m->ctr.units_hi))) return rc;
    }
    if ((rc = reserve_unit_stacks(m, n))) return rc;
    m->ctr.units_cap = static_cast<uint32_t>(m->d_units_cap);
    if (!m->on_device && !(m->aux_valid && m->aux_generation == h.generation)) {
        // auxiliary arrays from the host's view of the map
        const std::vector<uint32_t> so = h.slot_of_blocks();
        if (h.blocks_hi) {
            HIPCHK(hipMemcpyAsync(m->d_zeros, h.zeros.data(), h.blocks_hi, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(m->d_slot_of, so.data(), h.blocks_hi * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, s));
        }
        if (!h.free_blocks.empty())
            HIPCHK(hipMemcpyAsync(m->d_free, h.free_blocks.data(), h.free_blocks.size() * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, s));
        for (int k = 0; k < h.n_classes; ++k)
            if (!h.free_units[k].empty())
                HIPCHK(hipMemcpyAsync(m->d_free_units[k], h.free_units[k].data(),
                                      h.free_units[k].size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        map_derive_block_of(dev_map(m), h.blocks_hi, s);
        HIPCHK(hipStreamSynchronize(s));
        m->aux_valid = true;
        m->aux_generation = h.generation;
    }
    *m->h_ctr = m->ctr;
    m->h_ctr->n_new = m->h_ctr->n_far = m->h_ctr->overflow = 0;
    m->h_ctr->unit_overflow = m->h_ctr->n_freed = 0;
    HIPCHK(hipMemcpyAsync(m->d_ctr, m->h_ctr, sizeof(MapCounters), hipMemcpyHostToDevice, s));

    DevMap dm = dev_map(m);
    // table: (live + tombstoned + incoming) slots must stay within a quarter of the capacity
    if ((static_cast<uint64_t>(m->ctr.used_slots) + n) * 4 > m->d_table_cap) {
        size_t cap = 1024;
        while ((static_cast<uint64_t>(m->ctr.num_voxels) + n) * 4 > cap) cap *= 2;
        cap = std::max(cap, m->d_table_cap);
        Slot *nt = nullptr;
        HIPCHK(hipMalloc(&nt, cap * sizeof(Slot)));
        HIPCHK(map_rebuild_table(dm, nt, static_cast<uint32_t>(cap - 1), m->ctr.blocks_hi, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipFree(m->d_table));
        m->d_table = nt;
        m->d_table_cap = cap;
        m->ctr.used_slots = m->ctr.num_voxels;
        dm.table = nt;
        dm.mask = static_cast<uint32_t>(cap - 1);
    }
    const uint32_t bound = static_cast<uint32_t>(need_blocks);
    if ((rc = reserve_update_scratch(m, n, bound))) return rc;
    UpdateScratch us = m->up;
    if (d_points) us.raw = const_cast<Point4 *>(d_points);
    else if (n) HIPCHK(hipMemcpyAsync(m->up.raw, xyzl, n * sizeof(Point4), hipMemcpyHostToDevice, s));
    UpdatePolicy pol{};
    pol.voxel_size = h.voxel_size;
    pol.max_dist2 = h.max_distance * h.max_distance;
    pol.basic = h.basic;
    pol.critical = h.critical;
    pol.n_labels = static_cast<int>(h.basic_labels.size());
    for (int i = 0; i < pol.n_labels; ++i) pol.labels[i] = h.basic_labels[i];
    const bool ref_order = h.track_order;
    if (!ref_order) {
        us.new_list = nullptr;
        us.far_list = nullptr;
        HIPCHK(map_update_device(dm, pol, us, static_cast<int>(n), pose, bound, s));
    } else {
        // A map in reference-order mode: the bucket array of the reference's robin_map lives on the host (host_map.hpp,
        // RobinTable); the device inserts and FINDS the far voxels, the host replays the new voxels (arrival order) and the
        // erase-while-iterating sweep on that array (VoxelHashMap.cpp:166-172,176-184) — only the voxels concerned, a few
        // thousand words across PCIe — and the device evicts what the sweep reached.
        HIPCHK(map_update_insert_find_far(dm, pol, us, static_cast<int>(n), pose, bound, s));
    }
    HIPCHK(hipMemcpyAsync(m->h_ctr, m->d_ctr, sizeof(MapCounters), hipMemcpyDeviceToHost, s));
    if (ref_order) HIPCHK(hipMemcpyAsync(&m->h_ctr_aux[0], us.n_sel, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (m->h_ctr->overflow & 2u) {
        // nothing was inserted or evicted (every kernel checks the flag first)
        return fail(SAGEICP_ERR_INVALID, "Update: a coordinate or label is not finite (NaN / Inf); the map is unchanged");
    }
    if (m->h_ctr->overflow) {
        // nothing was inserted or evicted either
        return fail(SAGEICP_ERR_CAPACITY, "voxel index beyond +-2^20 in the device map update");
    }
    if (m->h_ctr->unit_overflow) {
        // nothing was inserted or evicted here either
        return fail(SAGEICP_ERR_CAPACITY, "voxel storage beyond 2^24 units of 4 points");
    }
#ifdef SAGE_UP_TIMING
    {
        const MapCounters &c = *m->h_ctr;
        const double w = static_cast<double>(c.dbg_sum[7] ? c.dbg_sum[7] : 1);
        std::fprintf(stderr, "k_up_insert phases, us (mean over %llu waves / slowest wave): stage %.2f/%.2f  dry run %.2f/%.2f  "
                             "alloc %.2f/%.2f  claim+move %.2f/%.2f  policy %.2f/%.2f  tail %.2f/%.2f  whole %.2f/%.2f\n",
                     c.dbg_sum[7], c.dbg_sum[0] / w / 100, c.dbg_max[0] / 100.0, c.dbg_sum[1] / w / 100, c.dbg_max[1] / 100.0,
                     c.dbg_sum[2] / w / 100, c.dbg_max[2] / 100.0, c.dbg_sum[3] / w / 100, c.dbg_max[3] / 100.0,
                     c.dbg_sum[4] / w / 100, c.dbg_max[4] / 100.0, c.dbg_sum[5] / w / 100, c.dbg_max[5] / 100.0,
                     c.dbg_sum[6] / w / 100, c.dbg_max[6] / 100.0);
        for (int j = 0; j < 8; ++j) m->h_ctr->dbg_sum[j] = m->h_ctr->dbg_max[j] = 0;
    }
#endif
    if (ref_order) {
        const uint32_t n_new = m->h_ctr->n_new, n_far = m->h_ctr_aux[0];
        if ((rc = m->reserve_lists(static_cast<size_t>(n_new) + n_far))) return rc;
        uint2 *hl = m->h_lists;
        if (n_new) HIPCHK(hipMemcpyAsync(hl, us.new_list, n_new * sizeof(uint2), hipMemcpyDeviceToHost, s));
        if (n_far) HIPCHK(hipMemcpyAsync(hl + n_new, us.far_list, n_far * sizeof(uint2), hipMemcpyDeviceToHost, s));
        if (n_new || n_far) HIPCHK(hipStreamSynchronize(s));
        RobinTable &order = const_cast<HostMap &>(h).order;
        for (uint32_t j = 0; j < n_new; ++j) order.insert(hl[j].y, hl[j].x);                 // arrival order
        std::vector<std::pair<uint32_t, uint32_t>> far(n_far);
        for (uint32_t j = 0; j < n_far; ++j) far[j] = {hl[n_new + j].y, hl[n_new + j].x};
        uint32_t *erased = reinterpret_cast<uint32_t *>(hl);        // (the lists are consumed: the erased blocks go back in their place)
        uint32_t n_er = 0;
        order.sweep_erase_listed(std::move(far), [&](uint32_t b) { erased[1 + n_er++] = b; });
        erased[0] = n_er;
        if (n_er) {
            // far_sel <- the erased blocks in the order of their erasure (the order they go onto the free list), n_sel <- their number
            HIPCHK(hipMemcpyAsync(us.n_sel, erased, sizeof(uint32_t), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(us.far_sel, erased + 1, n_er * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            HIPCHK(map_evict_listed(dm, us.far_sel, us.n_sel, n_er, s));
            HIPCHK(hipMemcpyAsync(m->h_ctr, m->d_ctr, sizeof(MapCounters), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        } else {
            m->h_ctr->n_far = 0;
        }
    }
    m->ctr = *m->h_ctr;
    m->on_device = true;
    m->cand_stale = true;
    const_cast<HostMap &>(h).clear_dirty();
    m->mirror_stale_all = false;
    return SAGEICP_OK;
}

// Non-finite input (NaN / Inf coordinates or labels).  The reference turns such values into voxel
// indices and label classes with static_cast<int> — undefined behaviour (INT_MIN on x86, 0 or a
// saturated value on gfx950) — so there is nothing to be faithful to: every entry that would cast one
// refuses the whole call with SAGEICP_ERR_INVALID before anything is changed (host buffers are checked
// here, device-resident frames by the first kernel that reads them: sort.hip, map_update.hip,
// preprocess.hip).  Where the reference's behaviour IS defined it is kept: Preprocess() drops a point
// whose norm is not finite (both range comparisons fail, Preprocessing.cpp:176-177), TransformPoints
// and AlignClouds propagate.
bool all_finite(const double *xyzl, uint64_t n) {
    // (x - x is 0 for every finite x and NaN otherwise: four of them summed stay 0 exactly)
    double acc = 0.0;
    for (uint64_t i = 0; i < 4 * n; ++i) acc += xyzl[i] - xyzl[i];
    return acc == 0.0;
}

}  // namespace sageicp_impl
