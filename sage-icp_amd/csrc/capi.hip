// C ABI of libsageicp_hip.so (include/sageicp.h): host side of the SAGE-ICP registration hot
// path on MI355X.  Owns the host-authoritative map, its HBM mirror, the per-map stream and
// scratch, the ICP launch loop (no host round trip per iteration: kernels early-exit on a
// device-resident `done` flag and the host polls it once per chunk of iterations) and the
// optional RCCL exchange of the Gauss-Newton sums for query-sharded multi-GPU runs.
//
// Reference call sites this file stands in for (cpp/sage_icp/):
//   core/Registration.cpp:113-141   RegisterFrame        -> run_icp()
//   core/VoxelHashMap.cpp:48-130    GetCorrespondences   -> sageicp_get_correspondences()
//   core/VoxelHashMap.cpp:144-184   Update/AddPoints/... -> HostMap (host_map.hpp)
// There is no CPU fallback: without a HIP device the compute entries fail with
// SAGEICP_ERR_NO_DEVICE.
#include "capi_internal.h"

namespace sageicp {
thread_local std::string g_err;
std::atomic<int> g_profiling{0};
std::atomic<int> g_counting{1};
std::atomic<unsigned> g_env_epoch{0};
const char *env_cached(const char *name) {
    // (text knobs — SAGEICP_DEVICES, SAGEICP_CU_SHARE — are asked for when a handle or its streams are created, not per call)
    static std::mutex mu;
    static unsigned epoch = 0xFFFFFFFFu;
    static std::map<std::string, std::pair<bool, std::string>> vals;
    std::lock_guard<std::mutex> lk(mu);
    const unsigned e = g_env_epoch.load();
    if (epoch != e) {
        vals.clear();
        epoch = e;
    }
    auto it = vals.find(name);
    if (it == vals.end()) {
        const char *v = std::getenv(name);
        it = vals.emplace(name, std::make_pair(v != nullptr, std::string(v ? v : ""))).first;
    }
    return it->second.first ? it->second.second.c_str() : nullptr;
}
std::atomic<int> g_reference_order{1};
}  // namespace sageicp


// =============================================================================================
extern "C" {

int sageicp_abi_version(void) { return SAGEICP_ABI_VERSION; }
const char *sageicp_last_error(void) { return g_err.c_str(); }
int sageicp_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}
void sageicp_set_profiling(int level) { g_profiling = level; }
void sageicp_set_counting(int on) { g_counting = on ? 1 : 0; }
void sageicp_reload_env(void) { sageicp::g_env_epoch.fetch_add(1u); }
void sageicp_set_downsample_order(int reference_order) { g_reference_order = reference_order ? 1 : 0; }
int sageicp_robin_iteration_order(const int32_t *vox_xyz, uint64_t n, uint32_t *order_out) {
    if (n && (!vox_xyz || !order_out)) return fail(SAGEICP_ERR_INVALID, "null argument");
    if (n >= (1ull << 27)) return fail(SAGEICP_ERR_INVALID, "too many voxels (2^27 max)");
    std::vector<uint32_t> h(n), order;
    for (uint64_t i = 0; i < n; ++i) h[i] = reference_voxel_hash(vox_xyz[3 * i], vox_xyz[3 * i + 1], vox_xyz[3 * i + 2]);
    order.reserve(n);
    static thread_local RobinScratch scratch;      // (exercises the reuse of the bucket arrays across calls)
    uint32_t max_probe = 0;
    if (!RobinOrderReplay::iteration_order(h.data(), n, 0u, order, &scratch, &max_probe))
        return fail(SAGEICP_ERR_CAPACITY, "a probe distance of " + std::to_string(max_probe) + " or more: beyond it "
                    "tsl::robin_map forces a growth this replay does not model (the reference's 20-bit hash: at "
                    "the latest from ~2^19 voxels)");
    std::memcpy(order_out, order.data(), n * sizeof(uint32_t));
    return SAGEICP_OK;
}

// ---- map ----------------------------------------------------------------------------------
int sageicp_robin_sweep(const int32_t *vox_xyz, uint64_t n, const uint8_t *far, int listed, uint32_t *erased_out,
                        uint64_t *n_erased, uint32_t *order_after, uint64_t *n_after) {
    if (n && (!vox_xyz || !far || !erased_out || !order_after)) return fail(SAGEICP_ERR_INVALID, "null argument");
    if (!n_erased || !n_after) return fail(SAGEICP_ERR_INVALID, "null argument");
    if (n >= (1ull << 27)) return fail(SAGEICP_ERR_INVALID, "too many voxels (2^27 max)");
    sageicp::RobinTable t;
    std::vector<uint32_t> h(n);
    for (uint64_t i = 0; i < n; ++i) {
        h[i] = reference_voxel_hash(vox_xyz[3 * i], vox_xyz[3 * i + 1], vox_xyz[3 * i + 2]);
        t.insert(h[i], static_cast<uint32_t>(i));
    }
    uint64_t k = 0;
    if (listed) {
        std::vector<std::pair<uint32_t, uint32_t>> lst;
        for (uint64_t i = n; i-- > 0;)          // (any order: here the reverse of arrival)
            if (far[i]) lst.emplace_back(h[i], static_cast<uint32_t>(i));
        t.sweep_erase_listed(std::move(lst), [&](uint32_t v) { erased_out[k++] = v; });
    } else {
        t.sweep_erase([&](uint32_t v) { return far[v] != 0; }, [&](uint32_t v) { erased_out[k++] = v; });
    }
    *n_erased = k;
    uint64_t a = 0;
    t.for_each([&](uint32_t v) { order_after[a++] = v; });
    *n_after = a;
    return t.valid() ? SAGEICP_OK : fail(SAGEICP_ERR_CAPACITY, "a probe distance the replay does not model");
}

sageicp_map *sageicp_map_create(double voxel_size, double max_distance, int basic, int critical,
                                const int *labels, int n_labels, int device) {
    if (!(voxel_size > 0.0) || basic < 0 || critical < 0 || basic + critical < 1 ||
        basic + critical > kMaxCap || n_labels < 0 || (n_labels > 0 && !labels)) {
        fail(SAGEICP_ERR_INVALID, "sageicp_map_create: invalid parameters (need voxel_size > 0, "
                                  "1 <= basic+critical <= 255)");
        return nullptr;
    }
    sageicp_map *m = new sageicp_map;
    m->host.configure(voxel_size, max_distance, basic, critical, labels, n_labels);
    m->device = device;
    // SAGEICP_DEVICES=0,1,2,3: every map of this process spans these devices (the knob for callers
    // that cannot be changed, e.g. the ROS node behind the header shim); the first is rank 0
    if (const char *list = env_cached("SAGEICP_DEVICES")) {
        std::vector<int> devs;
        for (const char *q = list; *q;) {
            char *end = nullptr;
            const long v = std::strtol(q, &end, 10);
            if (end == q) break;
            devs.push_back(static_cast<int>(v));
            q = (*end == ',') ? end + 1 : end;
        }
        if (devs.size() > 1 && sageicp_map_set_devices(m, devs.data(), static_cast<int>(devs.size()))) {
            sageicp_map_destroy(m);
            return nullptr;
        }
    }
    // SAGEICP_MAP_REFERENCE_ORDER=1: every map of this process is created in reference-order mode (the
    // knob for callers behind the header shim; sageicp_map_set_reference_order for everyone else)
    if (env_int("SAGEICP_MAP_REFERENCE_ORDER", 0)) (void)sageicp_map_set_reference_order(m, 1);
    return m;
}

int sageicp_map_set_reference_order(sageicp_map *m, int on) {
    if (!m) return fail(SAGEICP_ERR_INVALID, "null map");
    if (!map_is_empty(m) || m->on_device)
        return fail(SAGEICP_ERR_INVALID, "sageicp_map_set_reference_order: the map must be empty (the bucket order "
                                         "records every insertion since construction)");
    m->host.track_order = on != 0;
    m->host.order = sageicp::RobinTable();          // zero buckets, like a default-constructed robin_map
    for (sageicp_map *r : m->replicas)
        if (int rc = sageicp_map_set_reference_order(r, on)) return rc;
    return SAGEICP_OK;
}
int sageicp_map_reference_order(const sageicp_map *m) {
    if (!m || !m->host.track_order) return 0;
    return m->host.order.valid() ? 1 : -1;
}

int sageicp_map_set_devices(sageicp_map *m, const int *devices, int n) {
    if (!m || !devices || n < 1 || n > kMaxRanks)
        return fail(SAGEICP_ERR_INVALID, "sageicp_map_set_devices: 1..8 devices");
    if (m->sc.stream && devices[0] != m->device)
        return fail(SAGEICP_ERR_INVALID, "sageicp_map_set_devices: the map already lives on another first device");
    const int count = sageicp_device_count();
    for (int k = 0; k < n; ++k)
        if (count > 0 && (devices[k] < 0 || devices[k] >= count))
            return fail(SAGEICP_ERR_INVALID, "sageicp_map_set_devices: device ordinal out of range");
    if (int rc = ensure_host(m)) return rc;
    for (sageicp_map *r : m->replicas) sageicp_map_destroy(r);
    m->replicas.clear();
    for (sageicp_comm *c : m->ranks) sageicp_comm_destroy(c);
    m->ranks.clear();
    m->device = devices[0];
    for (int k = 1; k < n; ++k) {
        sageicp_map *r = new sageicp_map;
        r->device = devices[k];
        r->host = m->host;                 // the same map, mirrored on its own device at first use
        r->mirror_stale_all = true;
        m->replicas.push_back(r);
    }
    return SAGEICP_OK;
}
int sageicp_map_num_devices(const sageicp_map *m) { return m ? 1 + static_cast<int>(m->replicas.size()) : 0; }

void sageicp_map_destroy(sageicp_map *m) {
    if (!m) return;
    for (sageicp_comm *c : m->ranks) sageicp_comm_destroy(c);
    m->ranks.clear();
    for (sageicp_map *r : m->replicas) sageicp_map_destroy(r);
    m->replicas.clear();
    if (m->sc.stream) {
        (void)hipSetDevice(m->device);
        (void)hipStreamSynchronize(m->sc.stream);
        if (m->d_table) (void)hipFree(m->d_table);
        if (m->d_pts) (void)hipFree(m->d_pts);
        if (m->d_cand) (void)hipFree(m->d_cand);
        if (m->d_cand_flags) (void)hipFree(m->d_cand_flags);
        if (m->d_stage) (void)hipFree(m->d_stage);
        if (m->h_stage) (void)hipHostFree(m->h_stage);
        for (int k = 0; k < kMaxClasses; ++k)
            if (m->d_free_units[k]) (void)hipFree(m->d_free_units[k]);
        if (m->d_regions) (void)hipFree(m->d_regions);
        if (m->d_freed) (void)hipFree(m->d_freed);
        if (m->d_block_of) (void)hipFree(m->d_block_of);
        void *aux[] = {m->d_zeros, m->d_slot_of, m->d_free, m->d_ctr, m->up.raw, m->up.w, m->up.keys,
                       m->up.keys_alt, m->up.idx, m->up.idx_alt, m->up.head_slot, m->up.flag, m->up.rank, m->up.want,
                       m->up.far_flag, m->up.far_sel, m->up.n_sel, m->up.temp};
        for (void *q : aux)
            if (q) (void)hipFree(q);
        if (m->h_ctr) (void)hipHostFree(m->h_ctr);
        if (m->h_ctr_aux) (void)hipHostFree(m->h_ctr_aux);
        if (m->h_lists) (void)hipHostFree(m->h_lists);
        if (m->d_pc) (void)hipFree(m->d_pc);
    }
    m->sc.destroy();
    delete m;
}

// copy of a map whose authority is the HBM copy: device-to-device, the (stale) host side is not
// touched on either map
static int clone_on_device(const sageicp_map *src, sageicp_map *m) {
    const HostMap &h = src->host;
    m->host.configure(h.voxel_size, h.max_distance, h.basic, h.critical, h.basic_labels.data(),
                      static_cast<int>(h.basic_labels.size()));
    m->host.n_classes = h.n_classes;                    // (the source's size classes, whatever the environment says now)
    for (int k = 0; k < kMaxClasses; ++k) m->host.class_points[k] = h.class_points[k];
    // (a map in reference-order mode: the bucket array lives on the host whoever holds the points — "a copy has the same array")
    m->host.track_order = h.track_order;
    m->host.order = h.order;
    int rc = m->sc.init(m->device);
    if (rc) return rc;
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = m->sc.stream;
    HIPCHK(hipStreamSynchronize(src->sc.stream));
    HIPCHK(hipMalloc(&m->d_table, src->d_table_cap * sizeof(Slot)));
    m->d_table_cap = src->d_table_cap;
    HIPCHK(hipMemcpyAsync(m->d_table, src->d_table, src->d_table_cap * sizeof(Slot),
                          hipMemcpyDeviceToDevice, s));
    m->ctr = src->ctr;
    if ((rc = grow_device_blocks(m, src->d_blocks_cap, 0))) return rc;
    if ((rc = reserve_device_points(m, src->d_units_cap, 0))) return rc;
    m->on_device = false;       // (reserve_unit_stacks: nothing of this map's to keep yet)
    if ((rc = reserve_unit_stacks(m, 0))) return rc;
    HIPCHK(hipMemcpyAsync(m->d_pts, src->d_pts, static_cast<size_t>(m->ctr.units_hi) * kUnitPoints * sizeof(Point4),
                          hipMemcpyDeviceToDevice, s));
    for (int k = 0; k < h.n_classes; ++k)
        if (m->ctr.free_units_count[k] > 0)
            HIPCHK(hipMemcpyAsync(m->d_free_units[k], src->d_free_units[k],
                                  static_cast<size_t>(m->ctr.free_units_count[k]) * sizeof(uint32_t),
                                  hipMemcpyDeviceToDevice, s));
    if (m->ctr.units_hi)
        HIPCHK(hipMemcpyAsync(m->d_block_of, src->d_block_of, static_cast<size_t>(m->ctr.units_hi) * sizeof(uint32_t),
                              hipMemcpyDeviceToDevice, s));
    if (m->ctr.blocks_hi) {
        HIPCHK(hipMemcpyAsync(m->d_regions, src->d_regions, m->ctr.blocks_hi * sizeof(uint32_t),
                              hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(m->d_zeros, src->d_zeros, m->ctr.blocks_hi, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(m->d_slot_of, src->d_slot_of, m->ctr.blocks_hi * sizeof(uint32_t),
                              hipMemcpyDeviceToDevice, s));
    }
    if (m->ctr.free_count)
        HIPCHK(hipMemcpyAsync(m->d_free, src->d_free, m->ctr.free_count * sizeof(uint32_t),
                              hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    m->on_device = true;
    m->mirror_stale_all = false;
    return SAGEICP_OK;
}

static sageicp_map *clone_one(const sageicp_map *src) {
    sageicp_map *m = new sageicp_map;
    m->device = src->device;
    if (src->on_device) {
        if (clone_on_device(src, m)) {
            sageicp_map_destroy(m);
            return nullptr;
        }
        return m;
    }
    m->host = src->host;
    m->mirror_stale_all = true;   // the clone builds its own mirror on first use
    return m;
}

sageicp_map *sageicp_map_clone(const sageicp_map *src) {
    if (!src) return nullptr;
    sageicp_map *m = clone_one(src);
    if (!m) return nullptr;
    for (const sageicp_map *r : src->replicas) {
        sageicp_map *c = clone_one(r);
        if (!c) {
            sageicp_map_destroy(m);
            return nullptr;
        }
        m->replicas.push_back(c);
    }
    return m;
}

int sageicp_map_clear(sageicp_map *m) {
    if (!m) return fail(SAGEICP_ERR_INVALID, "null map");
    m->on_device = false;     // whatever the device holds is dropped with the rest
    m->aux_valid = false;
    m->host.clear();
    m->mirror_stale_all = true;
    for (sageicp_map *r : m->replicas) sageicp_map_clear(r);
    m->replicas_diverged = false;         // every copy is empty again
    return SAGEICP_OK;
}
int sageicp_map_empty(const sageicp_map *m) { return (!m || map_is_empty(m)) ? 1 : 0; }
uint64_t sageicp_map_size(const sageicp_map *m) {
    if (!m) return 0;
    return m->on_device ? m->ctr.total_points : m->host.total_points;
}
uint64_t sageicp_map_num_voxels(const sageicp_map *m) {
    if (!m) return 0;
    return m->on_device ? m->ctr.num_voxels : m->host.num_voxels;
}

int sageicp_map_add_points(sageicp_map *m, const double *xyzl, uint64_t n) {
    if (!m || (n && !xyzl)) return fail(SAGEICP_ERR_INVALID, "null argument");
    if (m->replicas_diverged)
        return fail(SAGEICP_ERR_INVALID, "the copies of this multi-device map diverged in an earlier failed update: Clear() it");
    if (!all_finite(xyzl, n))
        return fail(SAGEICP_ERR_INVALID, "AddPoints: a coordinate or label is not finite (NaN / Inf); nothing was inserted");
    if (int rc = ensure_host(m)) return rc;
    // A voxel index out of range is found by a dry pass: nothing is inserted then (like non-finite input
    // above).  Only the storage limits (2^24 voxels / 2^24 units of 4 points: 67 M point slots, DESIGN.md
    // section 1) can still stop a call part-way — before the point that does not fit, the points before it
    // in, the map consistent: which point that is depends on the retention policy's decisions on every
    // point before it.
    {
        const double vs = m->host.voxel_size;
        // (the range is tested on the quotient in double: casting a value beyond int32 is the undefined
        // behaviour D5 refuses to rely on; truncation toward zero keeps |index| < 2^20 exactly when |q| < 2^20)
        constexpr double kLim = 1048576.0;
        for (uint64_t i = 0; i < n; ++i) {
            const double qx = xyzl[4 * i] / vs, qy = xyzl[4 * i + 1] / vs, qz = xyzl[4 * i + 2] / vs;
            if (!(std::fabs(qx) < kLim && std::fabs(qy) < kLim && std::fabs(qz) < kLim))
                return fail(SAGEICP_ERR_CAPACITY, "AddPoints: voxel index beyond +-2^20 at point " + std::to_string(i) +
                                                      "; nothing was inserted");
        }
    }
    uint64_t at = 0;
    const int why = m->host.add_points(xyzl, n, &at);     // limits are checked before a point is taken
    // every copy of a multi-device map takes exactly the points rank 0 took: all of them, or the
    // prefix before the point a limit stopped at
    const uint64_t took = why ? at : n;
    for (sageicp_map *r : m->replicas)
        if (int rc = sageicp_map_add_points(r, xyzl, took)) {
            m->replicas_diverged = true;
            const std::string w2 = g_err;
            return fail(rc, "AddPoints reached only some devices of the map (" + w2 + "); the map must be cleared");
        }
    if (why == 1)
        return fail(SAGEICP_ERR_CAPACITY, "map full (2^24 voxels / 2^24 storage units of 4 points): stopped before point " +
                                              std::to_string(at) + ", the points before it are in");
    if (why == 2)
        return fail(SAGEICP_ERR_CAPACITY, "voxel index beyond +-2^20: stopped before point " +
                                              std::to_string(at) + ", the points before it are in");
    return SAGEICP_OK;
}

int sageicp_map_remove_far(sageicp_map *m, const double origin[3]) {
    if (!m || !origin) return fail(SAGEICP_ERR_INVALID, "null argument");
    if (int rc = ensure_host(m)) return rc;
    m->host.remove_far(origin);
    for (sageicp_map *r : m->replicas)
        if (int rc = sageicp_map_remove_far(r, origin)) return rc;
    return SAGEICP_OK;
}

int sageicp_map_update(sageicp_map *m, const double *xyzl, uint64_t n, const double origin[3]) {
    int rc = sageicp_map_add_points(m, xyzl, n);
    if (rc) return rc;
    return sageicp_map_remove_far(m, origin);
}

int sageicp_map_update_pose(sageicp_map *m, const double *xyzl, uint64_t n, const double pose[7]) {
    if (!m || (n && !xyzl) || !pose) return fail(SAGEICP_ERR_INVALID, "null argument");
    // Update(points, pose): transform into the map frame, origin = pose.translation()
    double R[9];
    quat_to_mat(pose, R);
    std::vector<double> w(4 * n);
    for (uint64_t i = 0; i < n; ++i) {
        mat_apply(R, pose + 4, xyzl + 4 * i, &w[4 * i]);
        w[4 * i + 3] = xyzl[4 * i + 3];
    }
    return sageicp_map_update(m, w.data(), n, pose + 4);
}

int sageicp_map_update_pose_device(sageicp_map *m, const double *xyzl, uint64_t n, const double pose[7]) {
    if (!m || (n && !xyzl) || !pose) return fail(SAGEICP_ERR_INVALID, "null argument");
    return device_update_all(m, xyzl, n, pose, nullptr);
}

// Pointcloud() while the HBM copy is the authority: packed on the device (block counts -> prefix
// sum -> gather, block-pool order like the host's), only size() x 32 B cross PCIe, and the map
// stays where it is — the node's per-frame LocalMap() (ros/ros2/OdometryServer.cpp:211-220 under
// publish_frame, the launch files' default) costs the copy of the live points and nothing else:
// no table rebuild, no re-upload before the next RegisterFrame.
// Make the pages of [p, p + bytes) exist — the range is about to be overwritten as a whole — from
// SAGEICP_TOUCH_THREADS (default 4; 0: off) parked threads, each a contiguous share populated with
// one madvise(MADV_POPULATE_WRITE) call (Linux 5.14) or, where that is refused, by a byte written
// into every page.  (Measured and not kept: asking for huge pages first — no better; the copy cut
// in pieces running behind the populating threads — slower than populate-then-copy, 3.1 vs 2.2 ms.)
static void pretouch(void *p, size_t bytes) {
    static const int threads = env_int("SAGEICP_TOUCH_THREADS", 4);
    constexpr size_t kPage = 4096;
    if (threads <= 0 || bytes < (size_t{4} << 20)) return;
    static ReplayPool pool;
    static std::mutex one_at_a_time;
    std::lock_guard<std::mutex> lk(one_at_a_time);
    char *base = static_cast<char *>(p);
    // whole pages inside the range go through madvise; the ragged ends are touched
    char *lo_al = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(base) + kPage - 1) & ~(kPage - 1));
    char *hi_al = reinterpret_cast<char *>(reinterpret_cast<uintptr_t>(base + bytes) & ~(kPage - 1));
    *static_cast<volatile char *>(base) = 0;
    *static_cast<volatile char *>(base + bytes - 1) = 0;
    if (hi_al <= lo_al) return;
#ifdef MADV_HUGEPAGE
    // where the kernel hands out transparent huge pages on request (.../transparent_hugepage/enabled = madvise, the
    // usual setting), the 2-MB-aligned inside of the range is faulted in as ~20 huge pages instead of ~11,000 small
    // ones: 2.95 against 3.11 ms per LocalMap() of 46 MB (SAGEICP_HUGEPAGES=0: off)
    static const int huge = env_int("SAGEICP_HUGEPAGES", 1);
    if (huge) {
        constexpr uintptr_t kHuge = uintptr_t{2} << 20;
        const uintptr_t h0 = (reinterpret_cast<uintptr_t>(lo_al) + kHuge - 1) & ~(kHuge - 1);
        const uintptr_t h1 = reinterpret_cast<uintptr_t>(hi_al) & ~(kHuge - 1);
        if (h1 > h0) (void)madvise(reinterpret_cast<void *>(h0), h1 - h0, MADV_HUGEPAGE);
    }
#endif
    const size_t pages = static_cast<size_t>(hi_al - lo_al) / kPage, share = (pages + threads - 1) / threads;
    const std::function<void(size_t)> job = [&](size_t t) {
        const size_t lo = t * share, hi = std::min(pages, lo + share);
        if (lo >= hi) return;
#ifdef MADV_POPULATE_WRITE
        if (madvise(lo_al + lo * kPage, (hi - lo) * kPage, MADV_POPULATE_WRITE) == 0) return;
#endif
        for (size_t i = lo; i < hi; ++i) *static_cast<volatile char *>(lo_al + i * kPage) = 0;
    };
    pool.run(static_cast<size_t>(threads), job, static_cast<size_t>(threads));
}

static int pointcloud_from_device(const sageicp_map *m, double *out, uint64_t cap, uint64_t *n_out) {
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = m->sc.stream;
    const uint64_t total = m->ctr.total_points;
    *n_out = total;
    const uint64_t want = out ? std::min(cap, total) : 0;
    if (!want) return SAGEICP_OK;
    int rc = reserve_update_scratch(m, 0, static_cast<size_t>(m->ctr.blocks_hi) + 1);
    if (rc) return rc;
    if (total > m->d_pc_cap) {
        if (m->d_pc) HIPCHK(hipFree(m->d_pc));
        m->d_pc = nullptr; m->d_pc_cap = 0;
        const size_t c = total + total / 4 + 1024;
        HIPCHK(hipMalloc(&m->d_pc, c * sizeof(Point4)));
        m->d_pc_cap = c;
    }
    const DevMap dm = dev_map(m);
    if (m->host.track_order) {
        // reference-order mode: the voxels in the bucket order of the host's array (VoxelHashMap.cpp:132-142), their points
        // packed on the device in that order
        std::vector<uint32_t> list;
        list.reserve(m->host.order.size());
        m->host.order.for_each([&](uint32_t b) { list.push_back(b); });
        if ((rc = reserve_update_scratch(m, 0, list.size() + 1))) return rc;
        uint32_t *d_list = reinterpret_cast<uint32_t *>(m->up.far_list);        // ([nb] uint2: room for the list)
        if (!list.empty()) HIPCHK(hipMemcpyAsync(d_list, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIPCHK(map_pointcloud_listed(dm, d_list, static_cast<uint32_t>(list.size()), m->up.far_flag, m->up.far_sel, m->up.temp,
                                     m->up.temp_bytes, m->d_pc, s));
        HIPCHK(hipStreamSynchronize(s));                                         // (`list` is pageable and leaves scope)
    } else {
        HIPCHK(map_pointcloud_device(dm, m->ctr.blocks_hi, m->up.far_flag, m->up.far_sel, m->up.temp,
                                     m->up.temp_bytes, m->d_pc, s));
    }
    // The destination is the caller's pageable buffer, and under the reference's interface a FRESH
    // one every call (`std::vector<Eigen::Vector4d> Pointcloud()` returns by value: tens of MB
    // straight from mmap).  The runtime's staged copy moves 63 MB in 1.2 ms into pages that exist —
    // and in 3.4 ms into pages that do not: two thirds of the call were first-touch faults taken one
    // by one inside the copy (profiles/pointcloud_probe.py).  So the pages are made to exist first, by a
    // few parked host threads side by side, while the device packs the points.
    pretouch(out, want * sizeof(Point4));
    HIPCHK(hipMemcpyAsync(out, m->d_pc, want * sizeof(Point4), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return SAGEICP_OK;
}

uint64_t sageicp_map_pointcloud(const sageicp_map *m, double *out, uint64_t cap) {
    if (!m) return 0;
    if (m->on_device) {
        uint64_t n = 0;
        if (pointcloud_from_device(m, out, cap, &n)) return 0;
        return n;
    }
    if (out) pretouch(out, static_cast<size_t>(std::min<uint64_t>(cap, m->host.total_points)) * sizeof(Point4));
    return m->host.pointcloud(out, out ? cap : 0);
}

int sageicp_map_resident(const sageicp_map *m) { return (m && m->on_device) ? 1 : 0; }

uint64_t sageicp_map_point_slots(const sageicp_map *m) {
    if (!m) return 0;
    return static_cast<uint64_t>(m->on_device ? m->ctr.units_hi : m->host.units_hi) * kUnitPoints;
}

int sageicp_map_sync(const sageicp_map *m) {
    if (!m) return fail(SAGEICP_ERR_INVALID, "null map");
    if (int rc = sync_mirror(m)) return rc;
    for (const sageicp_map *r : m->replicas)
        if (int rc = sync_mirror(r)) return rc;
    return SAGEICP_OK;
}

// ---- search ---------------------------------------------------------------------------------
int sageicp_get_correspondences(const sageicp_map *m, const double *q, uint64_t n, double max_dist,
                                double sem_th, double *src_out, double *tgt_out, uint64_t *n_out,
                                int64_t *query_idx_out) {
    if (!m || !n_out || (n && (!q || !src_out || !tgt_out)))
        return fail(SAGEICP_ERR_INVALID, "null argument");
    if (n > kMaxQueries) return fail(SAGEICP_ERR_INVALID, "too many queries (2^26 - 4 max)");
    *n_out = 0;
    if (!all_finite(q, n))
        return fail(SAGEICP_ERR_INVALID, "GetCorrespondences: a coordinate or label of a query is not finite (NaN / Inf)");
    int rc = ensure_host(m);      // the returned target points are read from the host copy
    if (rc) return rc;
    if ((rc = sync_mirror(m))) return rc;
    if (n == 0 || map_is_empty(m)) return SAGEICP_OK;
    Scratch &sc = m->sc;
    if ((rc = sc.reserve_frame(n))) return rc;
    if ((rc = sc.reserve_nn(n))) return rc;
    if ((rc = sc.reserve_sort(n))) return rc;
    hipStream_t s = sc.stream;
    HIPCHK(hipMemcpyAsync(sc.d_frame, q, n * sizeof(Point4), hipMemcpyHostToDevice, s));
    double I[7];
    identity_pose(I);
    fill_state(sc.h_state, I);
    HIPCHK(hipMemcpyAsync(sc.d_state, sc.h_state, sizeof(IcpState), hipMemcpyHostToDevice, s));
    // same pipeline as the ICP loop, pose = identity: sort, rows, search; results are mapped
    // back to the caller's query order through the sort permutation
    HIPCHK(sort_frame(sc.d_frame, sc.d_sorted, static_cast<int>(n), sc.d_state, false, false,
                      m->host.voxel_size, sc.d_keys, sc.d_vals, sc.d_sort_temp, sc.sort_temp_bytes_,
                      s));
    const int lw = icp_lw(n, sparse_voxels(m));
    if ((rc = ensure_cand(m, wants_filter(m, n, sem_th)))) return rc;
    const IcpParams ip = icp_params(m, sc.d_sorted, n, sem_th, lw);     // identity pose, no loop state
    launch_rows(ip, s);
    launch_icp(ip, lw, false, s);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> idx(n);
    std::vector<uint32_t> perm(n);
    HIPCHK(hipMemcpyAsync(idx.data(), sc.d_nn, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(perm.data(), sc.d_vals + n, n * sizeof(uint32_t), hipMemcpyDeviceToHost,
                          s));
    HIPCHK(hipStreamSynchronize(s));
    std::vector<int32_t> by_query(n);
    for (uint64_t i = 0; i < n; ++i) by_query[perm[i]] = idx[i];
    uint64_t k = 0;
    for (uint64_t i = 0; i < n; ++i) {     // pairs in query order (VoxelHashMap.cpp:119-127)
        if (by_query[i] < 0) continue;
        // acceptance on the unscaled distance: (nn - point).norm() < max (VoxelHashMap.cpp:111)
        const Point4 &t = m->host.pts[by_query[i]];
        const double dx = t.x - q[4 * i], dy = t.y - q[4 * i + 1], dz = t.z - q[4 * i + 2];
        if (!(std::sqrt(SAGE_SQNORM3_ACCEPT(dx * dx, dy * dy, dz * dz)) < max_dist)) continue;
        std::memcpy(src_out + 4 * k, q + 4 * i, 32);
        std::memcpy(tgt_out + 4 * k, &t, 32);
        if (query_idx_out) query_idx_out[k] = static_cast<int64_t>(i);
        ++k;
    }
    *n_out = k;
    return SAGEICP_OK;
}

// ---- AlignClouds ------------------------------------------------------------------------------
int sageicp_align_clouds(const double *src, const double *tgt, uint64_t n, double kernel,
                         double pose_out[7], double *JTJ_out, double *JTr_out, int device) {
    if (!pose_out || (n && (!src || !tgt))) return fail(SAGEICP_ERR_INVALID, "null argument");
    if (n > 0x7FFFFFFFull) return fail(SAGEICP_ERR_INVALID, "too many pairs");
    Scratch sc;
    int rc = sc.init(device);
    if (rc) return rc;
    auto body = [&]() -> int {
        HIPCHK(hipSetDevice(device));
        int r;
        if ((r = sc.reserve_frame(n))) return r;
        if ((r = sc.reserve_tgt(n))) return r;
        hipStream_t s = sc.stream;
        if (n) {
            HIPCHK(hipMemcpyAsync(sc.d_frame, src, n * sizeof(Point4), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(sc.d_tgt, tgt, n * sizeof(Point4), hipMemcpyHostToDevice, s));
        }
        double I[7];
        identity_pose(I);
        fill_state(sc.h_state, I);
        HIPCHK(hipMemcpyAsync(sc.d_state, sc.h_state, sizeof(IcpState), hipMemcpyHostToDevice, s));
        if ((r = sc.reserve_partials(128))) return r;
        GnParams gp{sc.d_frame, sc.d_tgt, static_cast<int>(n), kernel, sc.d_partials};
        FinParams fp{};
        fp.st = sc.d_state;
        fp.partials = sc.d_partials;
        fp.nparts = launch_gn(gp, s);
        fp.mode = 0;
        fp.standalone = 1;
        launch_fin(fp, s);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(sc.h_state, sc.d_state, sizeof(IcpState), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        // one step from identity: T_icp == est
        for (int i = 0; i < 7; ++i) pose_out[i] = sc.h_state->T_icp[i];
        if (JTJ_out || JTr_out) {
            double JTJ[36], JTr[6];
            assemble_normal_equations(sc.h_state->sums, JTJ, JTr);
            if (JTJ_out) std::memcpy(JTJ_out, JTJ, sizeof(JTJ));
            if (JTr_out) std::memcpy(JTr_out, JTr, sizeof(JTr));
        }
        return SAGEICP_OK;
    };
    rc = body();
    sc.destroy();
    return rc;
}

// ---- TransformPoints --------------------------------------------------------------------------
int sageicp_transform_points(const double pose[7], double *xyzl, uint64_t n, int device) {
    if (!pose || (n && !xyzl)) return fail(SAGEICP_ERR_INVALID, "null argument");
    if (n > 0x7FFFFFFFull) return fail(SAGEICP_ERR_INVALID, "too many points");
    Scratch sc;
    int rc = sc.init(device);
    if (rc) return rc;
    auto body = [&]() -> int {
        HIPCHK(hipSetDevice(device));
        int r;
        if ((r = sc.reserve_frame(n))) return r;
        hipStream_t s = sc.stream;
        fill_state(sc.h_state, pose);
        HIPCHK(hipMemcpyAsync(sc.d_state, sc.h_state, sizeof(IcpState), hipMemcpyHostToDevice, s));
        if (n) {
            HIPCHK(hipMemcpyAsync(sc.d_frame, xyzl, n * sizeof(Point4), hipMemcpyHostToDevice, s));
            launch_tf(sc.d_frame, static_cast<int>(n), sc.d_state, s);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(xyzl, sc.d_frame, n * sizeof(Point4), hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipStreamSynchronize(s));
        return SAGEICP_OK;
    };
    rc = body();
    sc.destroy();
    return rc;
}

// ---- RegisterFrame ----------------------------------------------------------------------------
int sageicp_register_frame(const sageicp_map *m, const double *frame, uint64_t n,
                           const double init[7], double max_dist, double kernel, double sem_th,
                           double pose_out[7], sageicp_stats *stats) {
    if (!m || !init || !pose_out || (n && !frame)) return fail(SAGEICP_ERR_INVALID, "null argument");
    const double t0 = now_us();
    if (map_is_empty(m)) {   // Registration.cpp:119
        std::memcpy(pose_out, init, 56);
        if (stats) { std::memset(stats, 0, sizeof(*stats)); stats->n_queries = n; }
        return SAGEICP_OK;
    }
    if (!m->replicas.empty())
        return register_sharded(m, frame, nullptr, n, init, max_dist, kernel, sem_th, pose_out, stats, t0);
    int rc = sync_mirror(m);
    if (rc) return rc;
    Scratch &sc = m->sc;
    if ((rc = sc.reserve_frame(n))) return rc;
    if (n) HIPCHK(hipMemcpyAsync(sc.d_frame, frame, n * sizeof(Point4), hipMemcpyHostToDevice,
                                 sc.stream));
    const double us_upload = now_us() - t0;
    return run_icp(m, sc.d_frame, n, init, max_dist, kernel, sem_th, nullptr, pose_out, stats,
                   us_upload, t0);
}

int sageicp_map_loop_status(const sageicp_map *m, sageicp_loop_status *out) {
    if (!m || !out) return fail(SAGEICP_ERR_INVALID, "null argument");
    const Scratch &sc = m->sc;
    out->calls_single_launch = sc.calls_single_launch;
    out->calls_per_iteration = sc.calls_per_iteration;
    out->calls_chained = sc.calls_chained;
    out->timeouts = sc.loop_timeouts;
    out->cooldown_calls = static_cast<uint32_t>(std::max(0, sc.loop_cooldown));
    out->derate_workgroups = 32u * static_cast<uint32_t>(std::max(0, sc.loop_derate));
    out->last_fallback = sc.last_fallback;
    return SAGEICP_OK;
}

sageicp_frame *sageicp_frame_upload(const sageicp_map *m, const double *frame, uint64_t n) {
    if (!m || (n && !frame)) { fail(SAGEICP_ERR_INVALID, "null argument"); return nullptr; }
    if (m->sc.init(m->device)) return nullptr;
    if (hipSetDevice(m->device) != hipSuccess) { fail(SAGEICP_ERR_HIP, "hipSetDevice"); return nullptr; }
    sageicp_frame *f = new sageicp_frame;
    f->device = m->device;
    f->n = n;
    if (hipMalloc(&f->d, std::max<uint64_t>(n, 1) * sizeof(Point4)) != hipSuccess) {
        fail(SAGEICP_ERR_HIP, "hipMalloc(frame)");
        delete f;
        return nullptr;
    }
    if (n && hipMemcpy(f->d, frame, n * sizeof(Point4), hipMemcpyHostToDevice) != hipSuccess) {
        fail(SAGEICP_ERR_HIP, "hipMemcpy(frame)");
        (void)hipFree(f->d);
        delete f;
        return nullptr;
    }
    return f;
}

void sageicp_frame_destroy(sageicp_frame *f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->d) (void)hipFree(f->d);
    delete f;
}

int sageicp_register_frame_resident(const sageicp_map *m, const sageicp_frame *f,
                                    const double init[7], double max_dist, double kernel,
                                    double sem_th, sageicp_comm *comm, double pose_out[7],
                                    sageicp_stats *stats) {
    if (!m || !f || !init || !pose_out) return fail(SAGEICP_ERR_INVALID, "null argument");
    if (f->device != m->device) return fail(SAGEICP_ERR_INVALID, "frame and map live on different devices");
    if (comm && comm->device != m->device) return fail(SAGEICP_ERR_INVALID, "comm and map live on different devices");
    const double t0 = now_us();
    if (map_is_empty(m)) {
        std::memcpy(pose_out, init, 56);
        if (stats) { std::memset(stats, 0, sizeof(*stats)); stats->n_queries = f->n; }
        return SAGEICP_OK;
    }
    if (!m->replicas.empty()) {
        if (comm) return fail(SAGEICP_ERR_INVALID, "a map that spans several devices shards the frame itself");
        return register_sharded(m, nullptr, f->d, f->n, init, max_dist, kernel, sem_th, pose_out, stats, t0);
    }
    int rc = sync_mirror(m);
    if (rc) return rc;
    const double us_upload = now_us() - t0;
    return run_icp(m, f->d, f->n, init, max_dist, kernel, sem_th, comm, pose_out, stats, us_upload,
                   t0);
}

// ---- RCCL communicator ------------------------------------------------------------------------
int sageicp_comm_unique_id(uint8_t id_out[SAGEICP_UNIQUE_ID_BYTES]) {
    if (!id_out) return fail(SAGEICP_ERR_INVALID, "null argument");
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == SAGEICP_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(SAGEICP_ERR_RCCL, "ncclGetUniqueId failed");
    std::memcpy(id_out, &id, sizeof(id));
    return SAGEICP_OK;
}

sageicp_comm *sageicp_comm_create(const uint8_t id_in[SAGEICP_UNIQUE_ID_BYTES], int rank,
                                  int nranks, int device) {
    if (!id_in || nranks < 1 || rank < 0 || rank >= nranks) {
        fail(SAGEICP_ERR_INVALID, "sageicp_comm_create: bad rank/nranks");
        return nullptr;
    }
    if (load_rccl()) return nullptr;
    if (hipSetDevice(device) != hipSuccess) { fail(SAGEICP_ERR_HIP, "hipSetDevice"); return nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id_in, sizeof(id));
    sageicp_comm *c = new sageicp_comm;
    c->rank = rank; c->nranks = nranks; c->device = device;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        fail(SAGEICP_ERR_RCCL, std::string("ncclCommInitRank: ") +
                                   (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
        delete c;
        return nullptr;
    }
    return c;
}

sageicp_comm *sageicp_comm_create_local(int rank, int nranks, int device) {
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) {
        fail(SAGEICP_ERR_INVALID, "sageicp_comm_create_local: bad rank/nranks (at most 8 ranks)");
        return nullptr;
    }
    sageicp_comm *c = new sageicp_comm;
    c->rank = rank; c->nranks = nranks; c->device = device;
    return c;
}

int sageicp_comm_p2p_export(sageicp_comm *c, uint8_t handle_out[SAGEICP_P2P_HANDLE_BYTES]) {
    if (!c || !handle_out) return fail(SAGEICP_ERR_INVALID, "null argument");
    if (c->nranks > kMaxRanks) return fail(SAGEICP_ERR_INVALID, "direct exchange: at most 8 ranks");
    static_assert(sizeof(hipIpcMemHandle_t) == SAGEICP_P2P_HANDLE_BYTES, "hipIpcMemHandle_t size");
    HIPCHK(hipSetDevice(c->device));
    if (!c->my_block) {
        // fine-grained: peers' stores and this rank's polling loads bypass the non-coherent caches
        HIPCHK(hipExtMallocWithFlags(reinterpret_cast<void **>(&c->my_block), sizeof(P2pBlock),
                                     hipDeviceMallocFinegrained));
        HIPCHK(hipMemset(c->my_block, 0, sizeof(P2pBlock)));
        HIPCHK(hipMalloc(&c->d_exchanges, sizeof(unsigned long long)));
        HIPCHK(hipMemset(c->d_exchanges, 0, sizeof(unsigned long long)));
        HIPCHK(hipDeviceSynchronize());
    }
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, c->my_block));
    std::memcpy(handle_out, &h, sizeof(h));
    return SAGEICP_OK;
}

int sageicp_comm_p2p_connect(sageicp_comm *c, const uint8_t *handles) {
    if (!c || !handles) return fail(SAGEICP_ERR_INVALID, "null argument");
    if (!c->my_block) return fail(SAGEICP_ERR_INVALID, "sageicp_comm_p2p_export first");
    HIPCHK(hipSetDevice(c->device));
    for (int r = 0; r < c->nranks; ++r) {
        if (r == c->rank) { c->blocks[r] = c->my_block; continue; }
        if (c->blocks[r]) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + static_cast<size_t>(r) * SAGEICP_P2P_HANDLE_BYTES, sizeof(h));
        void *p = nullptr;
        HIPCHK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->blocks[r] = static_cast<P2pBlock *>(p);
    }
    c->p2p = true;
    return SAGEICP_OK;
}

int sageicp_comm_p2p_enable(sageicp_comm *c, int on) {
    if (!c) return fail(SAGEICP_ERR_INVALID, "null argument");
    if (on) {
        if (c->poisoned)
            return fail(SAGEICP_ERR_INVALID, "direct exchange failed earlier on this communicator: "
                                             "create a new communicator and connect it");
        for (int r = 0; r < c->nranks; ++r)
            if (!c->blocks[r]) return fail(SAGEICP_ERR_INVALID, "direct exchange is not connected");
    }
    c->p2p = on != 0;
    return SAGEICP_OK;
}

int sageicp_comm_p2p_enabled(const sageicp_comm *c) { return c && c->p2p ? 1 : 0; }

int sageicp_comm_describe(const sageicp_comm *c, sageicp_comm_info *out) {
    if (!c || !out) return fail(SAGEICP_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    out->rank = c->rank;
    out->nranks = c->nranks;
    out->device = c->device;
    out->rccl_ranks = out->rccl_rank = -1;
    if (c->comm) {
        out->has_rccl = 1;
        int v = -1;
        if (g_rccl.CommCount && g_rccl.CommCount(c->comm, &v) == ncclSuccess) out->rccl_ranks = v;
        v = -1;
        if (g_rccl.CommUserRank && g_rccl.CommUserRank(c->comm, &v) == ncclSuccess) out->rccl_rank = v;
    }
    out->p2p_connected = 1;
    for (int r = 0; r < c->nranks && r < kMaxRanks; ++r)
        if (!c->blocks[r]) out->p2p_connected = 0;
    if (c->nranks > kMaxRanks) out->p2p_connected = 0;
    out->p2p_enabled = c->p2p ? 1 : 0;
    out->p2p_poisoned = c->poisoned ? 1 : 0;
    return SAGEICP_OK;
}

void sageicp_comm_destroy(sageicp_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (int r = 0; r < c->nranks && r < kMaxRanks; ++r)
        if (r != c->rank && c->blocks[r] && !c->peer_mapped) (void)hipIpcCloseMemHandle(c->blocks[r]);
    if (c->my_block) (void)hipFree(c->my_block);
    if (c->d_exchanges) (void)hipFree(c->d_exchanges);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    delete c;
}

// ---- Preprocess / VoxelDownsample on the device --------------------------------------------------
int sageicp_preprocess(const double *frame, uint64_t n, double max_range, double min_range,
                       double label_max_range, double *out, uint64_t *n_out, int device) {
    if (!n_out || (n && (!frame || !out))) return fail(SAGEICP_ERR_INVALID, "null argument");
    Prep pr;
    int rc = pr.init(device);
    if (rc) return rc;
    std::vector<std::vector<double>> res;
    const int crop = 1;
    const double scale = 0.0;       // crop only
    rc = pr.run(frame, n, max_range, min_range, label_max_range, 0, nullptr, nullptr, nullptr, &crop,
                &scale, 1, res);
    pr.destroy();
    if (rc) return rc;
    *n_out = res[0].size() / 4;
    if (!res[0].empty()) std::memcpy(out, res[0].data(), res[0].size() * sizeof(double));
    return SAGEICP_OK;
}

int sageicp_voxel_downsample(const double *frame, uint64_t n, int n_groups,
                             const int *group_label_counts, const int *group_labels,
                             const double *group_voxel_size, double vox_scale, double *out,
                             uint64_t *n_out, int device) {
    if (!n_out || (n && (!frame || !out)) || n_groups < 0 ||
        (n_groups && (!group_label_counts || !group_labels || !group_voxel_size)) || !(vox_scale > 0.0))
        return fail(SAGEICP_ERR_INVALID, "bad argument");
    Prep pr;
    int rc = pr.init(device);
    if (rc) return rc;
    std::vector<std::vector<double>> res;
    const int crop = 0;
    rc = pr.run(frame, n, 0, 0, 0, n_groups, group_label_counts, group_labels, group_voxel_size, &crop,
                &vox_scale, 1, res);
    pr.destroy();
    if (rc) return rc;
    *n_out = res[0].size() / 4;
    if (!res[0].empty()) std::memcpy(out, res[0].data(), res[0].size() * sizeof(double));
    return SAGEICP_OK;
}

// ---- pipeline counterpart -----------------------------------------------------------------------
struct sageicp_pipeline {
    sageicp::Pipeline impl;
    // Preprocess() + Voxelize() depend on the raw frame only (not on the pose, not on the map), so
    // the next frame's can run while this one registers (sageicp_pipeline_prefetch): two sets of
    // buffers and streams, `cur` the one the frame being registered lives in.
    sageicp::Prep prep[2];
    int cur = 0;
    int device;
    std::thread worker;                  // runs the announced frame's voxelize on prep[cur ^ 1]
    bool announced = false;              // _prefetch named the frame that follows the next one registered
    const double *an_frame = nullptr;
    uint64_t an_n = 0;
    bool ready = false;                  // prep[cur ^ 1] holds (or the worker is filling it with) pf_frame
    const double *pf_frame = nullptr;
    uint64_t pf_n = 0;
    uint64_t an_print = 0, pf_print = 0; // content fingerprints (a buffer re-used for other data is another frame)
    // FNV-1a over n and 64 rows spread over the frame: cheap, and enough to tell a buffer that was
    // refilled since it was announced from the frame that was announced
    static uint64_t fingerprint(const double *f, uint64_t m) {
        uint64_t h = 1469598103934665603ull ^ m;
        if (!f || !m) return h;
        const uint64_t step = std::max<uint64_t>(1, m / 64);
        for (uint64_t i = 0; i < m; i += step) {
            uint64_t w[4];
            std::memcpy(w, f + 4 * i, 32);
            for (uint64_t x : w) { h ^= x; h *= 1099511628211ull; }
        }
        uint64_t w[4];
        std::memcpy(w, f + 4 * (m - 1), 32);
        for (uint64_t x : w) { h ^= x; h *= 1099511628211ull; }
        return h;
    }
    int pf_rc = 0;
    std::string pf_err;
    explicit sageicp_pipeline(const sageicp_pipeline_config &c) : impl(c), device(c.device) {}
    ~sageicp_pipeline() {
        if (worker.joinable()) worker.join();
        prep[0].destroy();
        prep[1].destroy();
    }
    int voxelize_into(sageicp::Prep &pr, const double *f, uint64_t m) {
        int rc = pr.init(device);
        if (rc) return rc;
        std::vector<int> counts, labels;
        std::vector<double> vs;
        impl.group_tables(counts, labels, vs);
        const int crop[2] = {1, 0};
        const double scales[2] = {0.5, 1.5};
        std::vector<std::vector<double>> res;
        // level 0 (frame_downsample: it goes into the map, AddPoints depends on arrival order)
        // keeps the reference's emission order; level 1 (the registered source) does not need it
        pr.arrival_order_levels = env_int("SAGEICP_SOURCE_REFERENCE_ORDER", 0) ? 0u : 2u;
        return pr.run(f, m, impl.max_range_(), impl.min_range_(), impl.label_max_range_(),
                      static_cast<int>(counts.size()), counts.data(), labels.data(), vs.data(),
                      crop, scales, 2, res, false);
    }
};

sageicp_pipeline *sageicp_pipeline_create(const sageicp_pipeline_config *c) {
    if (!c || c->n_groups < 0 || (c->n_groups && (!c->group_label_counts || !c->group_voxel_size))) {
        fail(SAGEICP_ERR_INVALID, "sageicp_pipeline_create: bad config");
        return nullptr;
    }
    sageicp_pipeline *p = new sageicp_pipeline(*c);
    if (!p->impl.ok()) {
        delete p;
        return nullptr;
    }
    return p;
}
void sageicp_pipeline_destroy(sageicp_pipeline *p) { delete p; }
int sageicp_pipeline_register_frame(sageicp_pipeline *p, const double *frame, uint64_t n,
                                    double pose_out[7], double *icp_s, double *total_s,
                                    uint64_t *n_source, sageicp_stats *stats) {
    if (!p || !pose_out || (n && !frame)) return fail(SAGEICP_ERR_INVALID, "null argument");
    // Preprocess + Voxelize on the device (preprocess.hip): crop + scale 0.5, then scale 1.5.
    // Neither cloud comes back to the host: the source is registered and the down-sampled frame
    // inserted into the map from where the kernels left them (only a host-side map update
    // downloads its points).
    struct Backend {
        sageicp_pipeline *p;
        int voxelize(const double *f, uint64_t m, uint64_t &n_src) {
            if (p->worker.joinable()) p->worker.join();
            int r;
            if (p->ready && p->pf_frame == f && p->pf_n == m &&
                p->pf_print == sageicp_pipeline::fingerprint(f, m)) {   // prepared while the last frame registered
                r = p->pf_rc ? fail(p->pf_rc, p->pf_err) : SAGEICP_OK;
                p->cur ^= 1;
            } else {                                                 // none, or another frame: dropped
                r = p->voxelize_into(p->prep[p->cur], f, m);
            }
            p->ready = false;
            n_src = p->prep[p->cur].kept_levels[1];
            if (r == SAGEICP_OK && p->announced) {
                // the frame after this one: its Preprocess() + Voxelize() run on the other set of
                // buffers (own stream, own host thread) under this frame's ICP loop and map update
                p->ready = true;
                p->pf_frame = p->an_frame;
                p->pf_n = p->an_n;
                p->pf_print = p->an_print;
                p->pf_rc = 0;
                p->pf_err.clear();
                sageicp_pipeline *q = p;
                sageicp::Prep *dst = &p->prep[p->cur ^ 1];
                const double *nf = p->an_frame;
                const uint64_t nn = p->an_n;
                p->worker = std::thread([q, dst, nf, nn] {
                    (void)hipSetDevice(q->device);
                    q->pf_rc = q->voxelize_into(*dst, nf, nn);
                    if (q->pf_rc) q->pf_err = g_err;                 // the error text is per thread
                });
            }
            p->announced = false;
            return r;
        }
        int register_source(const double guess[7], double max_dist, double kernel, double sem_th,
                            double pose[7], sageicp_stats *stats) {
            sageicp_frame view;                 // non-owning: the source cloud in the Prep buffers
            view.device = p->device;
            view.d = p->prep[p->cur].d_src;
            view.n = p->prep[p->cur].kept_levels[1];
            return sageicp_register_frame_resident(p->impl.map, &view, guess, max_dist, kernel, sem_th,
                                                   nullptr, pose, stats);
        }
        int update_map(const double pose[7]) {
            const sageicp::Prep &pr = p->prep[p->cur];
            const uint64_t n_fd = pr.kept_levels[0];
            if (p->impl.map_update_on_device_())
                return device_update_all(p->impl.map, nullptr, n_fd, pose, pr.d_fd);
            std::vector<double> fd(4 * n_fd);
            if (n_fd) {
                HIPCHK(hipSetDevice(p->device));
                HIPCHK(hipMemcpy(fd.data(), pr.d_fd, n_fd * sizeof(Point4), hipMemcpyDeviceToHost));
            }
            return sageicp_map_update_pose(p->impl.map, fd.data(), n_fd, pose);
        }
    };
    const int rc = p->impl.register_frame(frame, n, pose_out, icp_s, total_s, n_source, stats, Backend{p});
    p->announced = false;       // an announcement is consumed by this call, also when it failed or the frame was empty
    return rc;
}
int sageicp_pipeline_prefetch(sageicp_pipeline *p, const double *frame, uint64_t n) {
    if (!p || (n && !frame)) return fail(SAGEICP_ERR_INVALID, "null argument");
    p->announced = true;
    p->an_frame = frame;
    p->an_n = n;
    p->an_print = sageicp_pipeline::fingerprint(frame, n);
    return SAGEICP_OK;
}
int sageicp_pipeline_prefetch_wait(sageicp_pipeline *p) {
    if (!p) return fail(SAGEICP_ERR_INVALID, "null pipeline");
    if (p->worker.joinable()) p->worker.join();     // what it prepared stays (`ready`)
    return SAGEICP_OK;
}
int sageicp_pipeline_prefetch_cancel(sageicp_pipeline *p) {
    if (!p) return fail(SAGEICP_ERR_INVALID, "null pipeline");
    if (p->worker.joinable()) p->worker.join();     // nothing reads an announced buffer after this
    p->announced = false;
    p->ready = false;
    return SAGEICP_OK;
}
int sageicp_pipeline_reinitialize(sageicp_pipeline *p) {
    if (!p) return fail(SAGEICP_ERR_INVALID, "null pipeline");
    p->impl.reinitialize();
    return SAGEICP_OK;
}
uint64_t sageicp_pipeline_num_poses(const sageicp_pipeline *p) { return p ? p->impl.poses.size() : 0; }
int sageicp_pipeline_pose(const sageicp_pipeline *p, uint64_t i, double out[7]) {
    if (!p || !out || i >= p->impl.poses.size()) return fail(SAGEICP_ERR_INVALID, "bad pose index");
    for (int k = 0; k < 7; ++k) out[k] = p->impl.poses[i].v[k];
    return SAGEICP_OK;
}
const sageicp_map *sageicp_pipeline_local_map(const sageicp_pipeline *p) {
    return p ? p->impl.map : nullptr;
}

// ---- KITTI trajectory metrics (metrics/Metrics.cpp) ------------------------------------------------
static std::vector<sageicp::metrics::M4> to_m4(const double *p, uint64_t n) {
    std::vector<sageicp::metrics::M4> v(n);
    for (uint64_t i = 0; i < n; ++i) std::memcpy(v[i].m, p + 16 * i, 16 * sizeof(double));
    return v;
}
int sageicp_metrics_seq_error(const double *poses_gt, const double *poses_result, uint64_t n,
                              float *avg_trans_error, float *avg_rot_error) {
    if (!avg_trans_error || !avg_rot_error || (n && (!poses_gt || !poses_result)))
        return fail(SAGEICP_ERR_INVALID, "null argument");
    sageicp::metrics::seq_error(to_m4(poses_gt, n), to_m4(poses_result, n), avg_trans_error, avg_rot_error);
    return SAGEICP_OK;
}
int sageicp_metrics_absolute_trajectory_error(const double *poses_gt, const double *poses_result,
                                              uint64_t n, float *ate_rot, float *ate_trans) {
    if (!ate_rot || !ate_trans || !n || !poses_gt || !poses_result)
        return fail(SAGEICP_ERR_INVALID, "null argument or empty trajectory");
    sageicp::metrics::absolute_trajectory_error(to_m4(poses_gt, n), to_m4(poses_result, n), ate_rot, ate_trans);
    return SAGEICP_OK;
}

}  // extern "C"

// probes: the loop state the last RegisterFrame of this map left on the host (pose T[7], T_icp[7], the last
// reduced sums [20], iterations, last step)
extern "C" int sageicp_debug_last_state(const sageicp_map *m, double *out36) {
    if (!m || !m->sc.h_state) return SAGEICP_ERR_INVALID;
    const sageicp::IcpState &st = *m->sc.h_state;
    for (int i = 0; i < 7; ++i) out36[i] = st.T[i];
    for (int i = 0; i < 7; ++i) out36[7 + i] = st.T_icp[i];
    for (int i = 0; i < sageicp::kNumSums; ++i) out36[14 + i] = st.sums[i];
    out36[34] = st.iter;
    out36[35] = st.last_step_norm;
    return SAGEICP_OK;
}

#ifdef SAGE_NN_TIMING
// probe: map points handed to every query in the last iteration (sorted order)
extern "C" int sageicp_debug_work(const sageicp_map *m, uint32_t *out, size_t n) {
    if (!m || !m->sc.d_work) return SAGEICP_ERR_INVALID;
    return hipMemcpy(out, m->sc.d_work, n * 4, hipMemcpyDeviceToHost) == hipSuccess ? SAGEICP_OK : SAGEICP_ERR_HIP;
}
#endif
