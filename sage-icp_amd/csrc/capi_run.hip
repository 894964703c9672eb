// The ICP loop of core/Registration.cpp:113-141 on the device: which form a frame takes (plan_loop: the whole loop in
// one launch wherever the frame fits), run_icp, the RCCL binding, and the single-process multi-GPU mode.  Part of
// libsageicp_hip.so's host side: capi_internal.h.
#include "capi_internal.h"

namespace sageicp_impl {

// ---- RCCL, bound at run time (only multi-GPU runs need it) ------------------------------------
Rccl g_rccl;
static std::mutex g_rccl_mu;


int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return SAGEICP_OK;
    // Prefer an RCCL the process already holds (torch ships one), then the ROCm install.
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (h) break;
    }
    for (int i = 0; i < 3 && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(SAGEICP_ERR_RCCL, std::string("cannot load librccl: ") + dlerror());
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
    g_rccl.CommUserRank = reinterpret_cast<decltype(g_rccl.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce)
        return fail(SAGEICP_ERR_RCCL, "librccl lacks a required symbol");
    g_rccl.h = h;
    return SAGEICP_OK;
}

void identity_pose(double T[7]) {
    T[0] = T[1] = T[2] = 0.0; T[3] = 1.0; T[4] = T[5] = T[6] = 0.0;
}

void fill_state(IcpState *st, const double init[7]) {
    std::memset(st, 0, sizeof(IcpState));
    for (int i = 0; i < 7; ++i) st->T[i] = init[i];
    quat_to_mat(init, st->R);
    identity_pose(st->T_icp);
}

// largest r2 with sqrt(r2) < max_dist: the acceptance test (nn - p).norm() < max_dist
// (VoxelHashMap.cpp:111) without a device square root, exact for the IEEE sqrt the CPU evaluates
double accept_threshold(double max_dist) {
    if (!(max_dist > 0.0)) return -1.0;                       // nothing passes (also NaN)
    double x = max_dist * max_dist;
    if (std::isinf(x)) x = std::numeric_limits<double>::max();
    while (x > 0.0 && !(std::sqrt(x) < max_dist)) x = std::nextafter(x, 0.0);
    for (;;) {
        const double up = std::nextafter(x, std::numeric_limits<double>::infinity());
        if (std::isinf(up) || !(std::sqrt(up) < max_dist)) break;
        x = up;
    }
    return std::sqrt(x) < max_dist ? x : -1.0;
}

// fewer than six points per voxel on average
bool sparse_voxels(const sageicp_map *m) {
    const uint64_t mp = m->on_device ? m->ctr.total_points : m->host.total_points;
    const uint64_t mv = m->on_device ? m->ctr.num_voxels : m->host.num_voxels;
    return mp < 6 * mv;
}

// Does the scan of `n` queries read the compact copy behind its fp32 filter?  Worth it where scans
// are long and bytes are what the kernel is made of: frames of 40k+ points against voxels holding
// 6+ points on average (c2: +6 %, c4: +10 %; c1, c5 and the streamed 24k-point frames lose 4-5 %
// with it; SAGEICP_FILTER=0/1 overrides).  Off for a negative or NaN sem_th, where a larger
// distance can scale to a smaller one and the filter's thresholds do not exist.
bool wants_filter(const sageicp_map *m, uint64_t n, double sem_th) {
    const int want = env_int("SAGEICP_FILTER", (n >= 40000 && !sparse_voxels(m)) ? 1 : 0);
    return sem_th >= 0.0 && want != 0;
}

// With 2 or 4 lanes per query: do the lanes stride through a query's voxels as one sequence (kernels.hip,
// "flat order")?  Where the voxels hold few points relative to the lanes — fewer than 2 W on average —,
// restarting in every voxel leaves lanes idle and makes the heaviest query's chain the longer one (c5:
// +4.6 %, c1 through the launch-per-iteration loop: +13 %); against c2's and c4's ~12 points per voxel the
// restart is faster by 1.5 and 5 %.  (8 and 16 lanes always stride flat; SAGEICP_FLAT=0/1 overrides.)
static bool wants_flat(const sageicp_map *m, int lw) {
    const uint64_t mp = m->on_device ? m->ctr.total_points : m->host.total_points;
    const uint64_t mv = m->on_device ? m->ctr.num_voxels : m->host.num_voxels;
    return env_int("SAGEICP_FLAT", mp < (2ull << lw) * mv ? 1 : 0) != 0;
}

// (raised while a frame whose sums left the range of the fixed-point accumulators is registered again at a
// coarser scale: the sums are accumulated at 2^(-24 g_acc_shift) of their value — see the end of run_icp)
static thread_local int g_acc_shift = 0;
// (a frame started again because a peer rank left its one-launch loop: see the end of run_icp)
static thread_local int g_restarts = 0;
static thread_local bool g_no_loop = false;
static thread_local bool g_no_chain = false;     // (a frame registered again after its chained launches timed out)

// k_icp's arguments for a search of `n` queries against the HBM copy of `m`
IcpParams icp_params(const sageicp_map *m, const Point4 *d_queries, uint64_t n, double sem_th, int lw) {
    const Scratch &sc = m->sc;
    IcpParams ip{};
    ip.frame = d_queries;
    ip.n = static_cast<int>(n);
    ip.st = sc.d_state;
    ip.check_done = 0;
    ip.apply_pose = 0;
    ip.voxel_size = m->host.voxel_size;
    ip.inv_voxel_size = env_int("SAGEICP_EXACT_DIVIDE", 0) ? 0.0 : 1.0 / m->host.voxel_size;
    ip.rows = sc.d_rows;
    ip.table = m->d_table;
    ip.mask = static_cast<uint32_t>(m->d_table_cap - 1);
    ip.pts = m->d_pts;
    const uint64_t pts_bytes = (static_cast<uint64_t>(m->d_units_cap) * kUnitPoints + 1) * sizeof(Point4);
    ip.pts_bytes = static_cast<uint32_t>(pts_bytes);        // (< 4 GiB: kMaxUnits units of 128 B)
    ip.cand = m->d_cand;
    ip.cand_bytes = m->d_cand_slots >= m->d_units_cap * kUnitPoints ? static_cast<uint32_t>(pts_bytes / 2) : 0u;
    ip.cand_flags = m->d_cand_flags;
    // fp32 thresholds of the scan's filter (kernels.hip): off (infinite) for a negative or NaN
    // sem_th, where a larger distance can scale to a smaller one
    {
        const double k1 = (1.0 + 1.0 / 1024.0) * (1.0 + 1e-6);
        const double inf = std::numeric_limits<double>::infinity();
        const bool filt = wants_filter(m, n, sem_th) && env_int("SAGEICP_NO_FILTER", 0) == 0;
        ip.filter = wants_filter(m, n, sem_th) ? 1 : 0;
        ip.flat = wants_flat(m, lw) ? 1 : 0;
        ip.filt_inv_diff = filt ? k1 : inf;
        ip.filt_inv_same = filt ? (sem_th > 0.0 ? k1 / sem_th : inf) : inf;
        ip.filt_slack = std::ldexp(1.0, -44) * 1025.0 * (1.0 + 1e-6);
    }
    ip.sem_th = sem_th;
    ip.dist_init = DBL_MAX;
    // scaled distance = d2 * sem_th for matching labels, d2 otherwise: >= min(sem_th, 1) * d2.
    // A negative or NaN sem_th gives no usable bound: every occupied voxel is visited.
    const bool prune = sem_th >= 0.0 && env_int("SAGEICP_NO_PRUNE", 0) == 0;
    ip.prune_scale = prune ? std::min(sem_th, 1.0) * (1.0 - 1e-9) : 0.0;
    ip.keep_all = prune ? 0u : 0x7FFFFFFu;
    ip.nn_idx = sc.d_nn;
    ip.kernel = 0.0;
    ip.accept_r2 = -1.0;
    ip.nn_prev = sc.d_prev;
    ip.work = sc.d_work;
    ip.acc_scale = std::ldexp(1.0, -24 * std::max(g_acc_shift, std::min(2, std::max(0, env_int("SAGEICP_ACC_SHIFT", 0)))));
    {
        // k_fin adds the accumulator copies in 64-bit integers: blocks x limit < 2^62 (kernels.hip, kDigitLimitCounted)
        const uint64_t blocks = std::max<uint64_t>(1, (n + 3) / 4);
        int bits = 0;
        while ((1ull << bits) < blocks) ++bits;              // ceil(log2(blocks)) <= 24
        ip.digit_limit = std::ldexp(1.0, std::min(46, 62 - bits));
    }
    ip.counters = nullptr;
    ip.stripe_work = nullptr;
    ip.stripe_order = nullptr;
    const uint64_t qw = 64u >> lw;
    ip.nwaves = static_cast<unsigned>((n + qw - 1) / qw);
#ifdef SAGE_ICP_DELAY_PROBE
    ip.dbg_delay = static_cast<unsigned>(env_int("SAGEICP_DBG_DELAY", 0));
    ip.dbg_repeat = static_cast<unsigned>(env_int("SAGEICP_DBG_REPEAT", 0));
#endif
    return ip;
}

// Shape of the one-launch loop (k_loop) for a frame of n points, or false when the frame does not fit
// the machine in that form.  The frame is cut into groups of 64 >> lw queries; a workgroup of nw waves
// owns gpw of them for the whole call, rows and per-query state in LDS (kernels.h), and every workgroup
// has to be resident at once: what bounds a frame is the LDS of the machine (232 B per query: ~170k
// queries on 256 CUs), not its wave slots.
struct LoopPlan {
    int lw, nw, gpw, wgs;      // lanes per query (log2), waves per workgroup, units of 64 >> lw queries per workgroup
                               // (at most), query workgroups
    bool filter;
};
static int loop_wgs_per_cu(const Scratch &sc, int lw, bool filter, int nw, size_t lds) {
    // (cached per shape: the occupancy query costs microseconds)
    static std::mutex mu;
    static std::map<std::array<long, 5>, int> cache;
    const std::array<long, 5> key{sc.device, lw, filter ? 1 : 0, nw, static_cast<long>(lds)};
    int v;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it == cache.end()) it = cache.emplace(key, loop_blocks_per_cu(lw, filter, nw, lds)).first;
        v = it->second;
    }
    // The kernel is built for SAGE_LOOP_OCC waves per SIMD, and the occupancy query assumes that the waves of the
    // resident workgroups spread evenly over the four SIMDs of a CU.  They do not: four workgroups of seven waves
    // (28 waves: seven per SIMD by the query) are NOT resident together on gfx950 (profiles/r05: the launch timed
    // out) — a workgroup's waves go to the SIMDs in turn, so what fits is what fits when every workgroup puts
    // its ceil(nw / 4) waves on the same SIMD.  Four waves per workgroup, seven workgroups per CU fill the CU.
    return std::min(v, SAGE_LOOP_OCC / ((nw + 3) / 4));
}
static bool plan_loop(const sageicp_map *m, uint64_t n, double sem_th, LoopPlan *out) {
    const Scratch &sc = m->sc;
    const int mode = env_int("SAGEICP_LOOP", 1);       // 0: never, 1 / 2: wherever the frame fits
    if (mode == 0 || n == 0 || sc.num_cus < 8) return false;
    const bool sparse = sparse_voxels(m);
    const bool filter = wants_filter(m, n, sem_th);
    const uint64_t cus = static_cast<uint64_t>(sc.num_cus);
    const int env_nw = std::min(kLoopMaxWavesHost, std::max(0, env_int("SAGEICP_LOOP_WAVES", 0)));
    const int env_gpw = std::max(0, env_int("SAGEICP_LOOP_GPW", 0));
    auto groups_at = [n](int l) { return (n + (64u >> l) - 1) / (64u >> l); };
    // the workgroup slots a launch may count on: k per CU less a sixteenth (below) — and behind a CU mask (SAGEICP_CU_SHARE:
    // ranks sharing one GPU) one per CU fewer for every halving of the device: on half of the CUs six of seven workgroups per
    // CU are resident (832 of 896 time out, 768 hold), on a quarter five (384 time out, 320 hold): profiles/r06/run48.sh, run50.sh
    auto slots = [&](int k) {
        const uint64_t a = static_cast<uint64_t>(k) * cus * 15 / 16;
        int less = 0;
        for (int sh = 1; sh < sc.cu_share_k; sh *= 2) ++less;
        return (less > 0 && k > less) ? std::min(a, static_cast<uint64_t>(k - less) * cus) : a;
    };
    auto round32 = [](uint64_t w) { return std::max<uint64_t>(32, (w + 31) / 32 * 32); };   // (XCD stripes: 8 x kLoopStripe)
    // the accumulator words count their workgroups in 8 bits, and (digit << 8) summed over the blocks of four
    // queries of a copy's workgroups must stay inside 63 bits: |digit| < 2^40 per block (kernels.hip, to_digits)
    auto countable = [](uint64_t wgs, uint64_t gpw, int lw) {
        return wgs / kLoopReplicas <= 255 && (wgs / kLoopReplicas) * gpw * ((64u >> lw) / 4u) <= 8192;
    };
    // one wave per group: nw groups per workgroup of nw waves
    auto one_pass = [&](int lw, int nw, LoopPlan *pl) {
        const uint64_t wgs = round32((groups_at(lw) + nw - 1) / nw);
        const size_t lds = loop_lds_bytes(lw, nw, nw);
        const int k = loop_wgs_per_cu(sc, lw, filter, nw, lds);
        if (k < 1 || wgs + 32ull * static_cast<uint64_t>(sc.loop_derate) > slots(k) || !countable(wgs, nw, lw)) return false;
        *pl = LoopPlan{lw, nw, nw, static_cast<int>(wgs), filter};
        return true;
    };
    // the waves of a workgroup take several groups each, one after another: as many resident waves as the
    // registers allow, the groups spread over all the workgroups that fit
    auto multi_pass = [&](int lw, LoopPlan *pl) {
        const uint64_t groups = groups_at(lw);
        int best = -1;
        const int order[] = {4, 8, 7, 6, 5, 3, 2, 1};
        for (int nw : order) {
            if (env_nw && nw != env_nw) continue;
            if (!env_nw && nw < 4) continue;
            for (int k = SAGE_LOOP_OCC / ((nw + 3) / 4); k >= 1; --k) {
                if (static_cast<uint64_t>(k) * cus < 34) break;
                // (measured, profiles/r05/resident_probe: of the 7 x 256 = 1,792 slots for workgroups of four waves
                // 1,696 are resident together beside the solving wave, 1,728 are not; a sixteenth stays free, and
                // a launch that still times out takes another 32 workgroups off every later plan of this handle)
                uint64_t cap = slots(k) / 32 * 32;
                cap = cap > 32ull * static_cast<uint64_t>(sc.loop_derate) ? cap - 32ull * static_cast<uint64_t>(sc.loop_derate) : 0;
                if (cap < 32) continue;
                if (const int e = env_int("SAGEICP_LOOP_MAX_WGS", 0)) cap = std::min<uint64_t>(cap, static_cast<uint64_t>(e) / 32 * 32);   // (probes)
                // (every resident workgroup slot is used: the groups are dealt out evenly over the workgroups,
                // so more workgroups mean fewer waves that have to make a second pass)
                uint64_t gpw = env_gpw ? static_cast<uint64_t>(env_gpw) : (groups + cap - 1) / cap;
                const uint64_t wgs = env_gpw ? round32((groups + gpw - 1) / gpw) : std::min(cap, round32(groups));
                if (wgs > cap) continue;
                if (!env_gpw) gpw = (groups + wgs - 1) / wgs;
                const size_t lds = loop_lds_bytes(lw, nw, static_cast<int>(gpw));
                if (lds > 160 * 1024 || !countable(wgs, gpw, lw)) continue;
                if (loop_wgs_per_cu(sc, lw, filter, nw, lds) < k) continue;
                if (k * nw > best) {
                    best = k * nw;
                    *pl = LoopPlan{lw, nw, static_cast<int>(gpw), static_cast<int>(wgs), filter};
                }
                break;                          // (fewer workgroups per CU only mean fewer resident waves)
            }
        }
        return best > 0;
    };
    const int forced = env_int("SAGEICP_LW", -1);
    int lw = forced >= 0 ? std::min(forced, 4) : icp_lw(n, sparse);
    if (lw < 1) return false;                          // (k_loop is built for 2..16 lanes per query)
    if (forced < 0 && !env_gpw) {
        // An iteration of k_loop ends with its slowest wave, and with eight or more lanes per query the
        // lanes stride through a query's voxels in flat order (kernels.hip): while every group still gets a
        // wave of its own, more lanes than the launch-per-iteration loop would take pay — 8 where it
        // would take 4 (c1: 17.8 -> 13.4 us per iteration), 16 against dense voxels (15k queries: 16.3 ->
        // 15.8; c1's sparse ones: 13.4 -> 14.4); profiles/r04/flat_where.txt
        const uint64_t few = 15 * cus;                 // (3,840 waves on 256 CUs: where round 4 measured it)
        if (lw < 3 && groups_at(3) <= few) lw = 3;
        if (lw == 3 && !sparse && groups_at(4) <= few) lw = 4;
        // ... and fewer once the units outnumber the waves the machine holds (7 per SIMD less the residency margin):
        // a second pass of some waves costs more than a longer chain in everybody's first — 60k queries against
        // dense voxels: 24.5 us per iteration with four lanes, 26.0 with eight; 30k: 21.7 / 19.6 (profiles/r05)
        // (round 6, second scene family — profiles/r06/ring_probe.txt: 52k queries of ring geometry, 6,592 units at eight
        // lanes on 6,656 waves: four lanes 3 % faster; the first family's 60k: 24.5 against 26.0 — the switch sits at four
        // fifths of the resident waves, ~42k queries)
        const uint64_t resident_waves = 4ull * SAGE_LOOP_OCC * cus * 15 / 16;
        while (lw > 2 && groups_at(lw) > resident_waves * 4 / 5) --lw;
    }
    const bool dbg = env_int("SAGEICP_LOOP_DEBUG", 0) != 0;
    bool ok = !env_gpw && one_pass(lw, env_nw ? env_nw : 4, out);
    if (!ok) ok = multi_pass(lw, out);
    if (dbg) {
        if (ok)
            std::fprintf(stderr, "sageicp: one-launch loop for %llu queries: %d lanes/query, %d workgroups of %d waves, <= %d units of %d queries each, "
                                 "%zu B of LDS (%d workgroups per CU by the occupancy query, %d CUs)\n",
                         static_cast<unsigned long long>(n), 1 << out->lw, out->wgs, out->nw,
                         out->gpw, 64 >> out->lw, loop_lds_bytes(out->lw, out->nw, out->gpw),
                         loop_wgs_per_cu(sc, out->lw, filter, out->nw, loop_lds_bytes(out->lw, out->nw, out->gpw)), sc.num_cus);
        else
            std::fprintf(stderr, "sageicp: %llu queries at %d lanes/query do not fit the one-launch loop (7 waves x 4 workgroups "
                                 "of 36 KB per CU by the occupancy query: %d)\n",
                         static_cast<unsigned long long>(n), 1 << lw,
                         loop_wgs_per_cu(sc, lw, filter, 7, 36 * 1024));
    }
    return ok;
}

// The ICP loop of Registration.cpp:127-138 as a stream of launches: k_icp (search + accumulation)
// and k_fin (reduce, solve, compose, test) per iteration — or, for a frame that fits the machine
// and is not sharded over GPUs, as ONE launch (k_loop).
// (raised while a frame whose sums left the range of the fixed-point accumulators is registered again
// at a coarser scale — see the end of run_icp)


int run_icp(const sageicp_map *m, const Point4 *d_frame, uint64_t n, const double init[7],
            double max_dist, double kernel, double sem_th, sageicp_comm *comm, double out[7],
            sageicp_stats *stats, double us_upload, double t_begin) {
    Scratch &sc = m->sc;
    hipStream_t s = sc.stream;
    if (n > kMaxQueries) return fail(SAGEICP_ERR_INVALID, "frame too large (2^26 - 4 points max)");
    int rc;
    const bool prof = g_profiling != 0;
    const bool prof2 = g_profiling >= 2;
    // Single GPU: iterations are enqueued a few ahead of the GPU, which reports its progress
    // through a host-mapped word (no stream synchronisation inside the loop).  With an RCCL
    // communicator every rank must enqueue the same number of all-reduces, so the loop advances
    // in fixed chunks (4, 8, 16, 16, ...) with one synchronisation per chunk instead.
    const bool p2p = comm && comm->p2p;
    if (comm && !p2p && !comm->comm)
        return fail(SAGEICP_ERR_INVALID, "communicator without RCCL needs a connected p2p exchange");
    // (the direct exchange enqueues no collective, so its loop can be polled like the 1-GPU one)
    const bool polled = (!comm || p2p) && env_int("SAGEICP_CHUNKED", 0) == 0;
    if (prof && (rc = sc.reserve_events(polled ? kMaxIterations : kChunkMax))) return rc;
    // (probes only: SAGEICP_MAX_ITER stops either loop early — the launch-per-iteration loop then simply runs out of launches)
    const int max_it = std::min(kMaxIterations, std::max(1, env_int("SAGEICP_MAX_ITER", kMaxIterations)));

    fill_state(sc.h_state, init);
    if (polled) {
        std::memset(sc.h_prog, 0, sizeof(IcpProgress));
        sc.h_state->progress = sc.d_prog;
    }
    HIPCHK(hipMemcpyAsync(sc.d_state, sc.h_state, sizeof(IcpState), hipMemcpyHostToDevice, s));

    // Lanes per query are decided in ONE place, whichever loop then runs: a frame that fits the one-launch
    // loop takes that loop's choice also when the launch-per-iteration loop registers it (a launch that timed
    // out, the calls of the cool-down after it) — the fixed-point sums are rounded once per group of queries,
    // so their bits depend on the lanes per query and on nothing else, and a call repeated gives the same bits.
    LoopPlan plan{};
    // (SAGEICP_CHUNKED=1 asks for the chunked launch-per-iteration loop by name)
    const bool loop_shape = (!comm || (p2p && !comm->device_shared && env_int("SAGEICP_CHUNKED", 0) == 0)) &&
                            plan_loop(m, n, sem_th, &plan);
    const bool restarted = g_no_loop;          // (lanes per query as the plan says; the loop form not again for this frame)
    const int lw = loop_shape ? plan.lw : icp_lw(n, sparse_voxels(m));
    // a launch that timed out (its grid was not resident as a whole: the GPU is shared with other work)
    // cost 50 ms before the frame went through the other loop: the next calls do not try again
    bool use_loop = loop_shape && !restarted;
    sc.last_fallback = !loop_shape ? SAGEICP_LOOP_FALLBACK_DOES_NOT_FIT : (restarted ? SAGEICP_LOOP_FALLBACK_PEER : SAGEICP_LOOP_FALLBACK_NONE);
    if (use_loop && sc.loop_cooldown > 0) {
        --sc.loop_cooldown;
        use_loop = false;
        sc.last_fallback = SAGEICP_LOOP_FALLBACK_COOLDOWN;
    }
    const unsigned loop_waves = use_loop ? static_cast<unsigned>((n + (64u >> plan.lw) - 1) / (64u >> plan.lw)) : 0u;
    if ((rc = ensure_cand(m, wants_filter(m, n, sem_th)))) return rc;
    if ((rc = sc.reserve_sort(n))) return rc;
    IcpParams ip = icp_params(m, sc.d_sorted, n, sem_th, lw);
    ip.check_done = 1;
    ip.apply_pose = 1;
    ip.kernel = kernel;
    ip.accept_r2 = accept_threshold(max_dist);
    // (the counters behind sum_candidates / pairs_evaluated cost ~45 vector instructions per pass, a memset and a
    // launch per frame: a caller that wants the other statistics only — bench.py's timed region — switches them off)
    const bool counting = stats && g_counting != 0;
    ip.counters = counting ? sc.d_cand : nullptr;
    if (counting) HIPCHK(hipMemsetAsync(sc.d_cand, 0, sizeof(unsigned long long) * 2 * (std::max(ip.nwaves, loop_waves) + 1), s));

    // direct exchange of the sums with the peer GPUs (k_fin mode 3, or the solving wave of the one-launch loop)
    P2pParams xp{};
    xp.nranks = 1;
    if (p2p) {
        xp.nranks = comm->nranks;
        xp.rank = comm->rank;
        for (int r = 0; r < comm->nranks; ++r) xp.block[r] = comm->blocks[r];
        xp.exchanges = comm->d_exchanges;
        // a peer's sums normally arrive within microseconds, but its FIRST launches of a process (code
        // object loading) or a GPU shared with other work can take a second: five seconds of in-kernel
        // waiting is a failure (SAGEICP_P2P_TIMEOUT_S overrides, e.g. under a debugger)
        xp.timeout_ticks = 100000000ull * static_cast<unsigned long long>(
                               std::max(1, env_int("SAGEICP_P2P_TIMEOUT_S", 5)));
        if (const int ticks = env_int("SAGEICP_P2P_TIMEOUT_TICKS", 0))      // tests: provoke a timeout
            xp.timeout_ticks = static_cast<unsigned long long>(ticks);
    }
    LoopParams L{};
    // The solving wave is launched well before its grid; should this call leave in between (an allocation or a launch
    // failing), it must not sit there waiting for a grid that never comes (and write its abort into the state of a
    // later call): the guard sends it home with the word the grid would have sent for a frame it refuses.
    struct SolverGuard {
        Scratch *sc = nullptr;
        unsigned long long epoch = 0;
        ~SolverGuard() {
            if (!sc) return;
            const unsigned long long word = epoch | 0x8000000000000000ull;
            (void)hipMemcpyAsync(&sc->d_loop->go[0], &word, sizeof(word), hipMemcpyHostToDevice, sc->stream);
            (void)hipStreamSynchronize(sc->stream);
            (void)hipStreamSynchronize(sc->stream2);
        }
    } solver_guard;
    // Frames beyond the LDS on one GPU: the launches of the iterations CHAINED (kernels.h, IcpParams::chain) — no k_fin between
    // them, the solving wave of the one-launch loop resident beside them.  (Not after a one-launch loop that timed out in this
    // call or its cool-down: a GPU that did not hold that grid is not asked to hold a resident solving wave either.)
    const int chain_grid = n > 0 ? icp_blocks_for(static_cast<int>(n), lw) : 0;
    // (profiling level 2 asks for the time of every kernel of every iteration, k_fin's too: the form with k_fin)
    // (Under a communicator with the direct exchange the form exists too — the exchange is the solving wave's, as in the
    // one-launch loop; parity-green and in step when a rank loses it, tests/test_gpu_parity.py — but it is OFF unless
    // SAGEICP_CHAIN_COMM=1: the only place it could be measured, two processes sharing one GPU, runs c4 twice as slowly
    // with it (42.7 against 21.0 ms per frame: the waiting launches of two processes on one device's queues,
    // profiles/r06/run63.sh); on a GPU per rank it has never run.  Never for several ranks of one process on one device.)
    const bool chain_comm = comm && p2p && !comm->device_shared && env_int("SAGEICP_CHAIN_COMM", 0) != 0;
    const bool chain = !loop_shape && (!comm || chain_comm) && polled && n > 0 &&
                       chain_grid / kChainReplicas <= 255 && !g_no_chain && !prof2 && env_int("SAGEICP_CHAIN", 1) != 0;
    if ((use_loop || chain) && (rc = sc.loop_streams())) return rc;
    if (chain) {
        L.sh = sc.d_loop;
        L.st = sc.d_state;
        L.wgs = chain_grid;                       // every workgroup of a launch sends its sums, also the ones past the frame's end
        L.copies = kChainReplicas;
        L.progress = sc.d_prog;
        L.timeout_ticks = 100000ull * static_cast<unsigned long long>(std::max(1, env_int("SAGEICP_LOOP_TIMEOUT_MS", 50)));
        if (const int ticks = env_int("SAGEICP_LOOP_TIMEOUT_TICKS", 0)) L.timeout_ticks = static_cast<unsigned long long>(ticks);
        // (the solving wave waits for a whole LAUNCH here, not for resident workgroups: the first launch of a process
        // loads code objects, a big frame's launch takes its hundred microseconds)
        L.count_timeout_ticks = std::max<unsigned long long>(L.timeout_ticks, 20ull * 100000ull) * 10ull;
        if (comm) {
            // (as for the one-launch loop: a launch waits for a pose that waits for the peers' sums — its patience has to
            // outlast the exchange's; the solving wave's wait for its OWN launch stays local)
            L.shared_loop = 1;
            if (env_int("SAGEICP_LOOP_TIMEOUT_TICKS", 0) == 0)
                L.timeout_ticks = std::max(L.timeout_ticks, xp.timeout_ticks + 100000000ull);
            // (tests: ONE rank of a communicator loses its launches — the others must follow it out of the exchange)
            if (env_int("SAGEICP_LOOP_COUNT_TIMEOUT_RANK", -1) == comm->rank)
                L.count_timeout_ticks = static_cast<unsigned long long>(std::max(1, env_int("SAGEICP_LOOP_COUNT_TIMEOUT_TICKS", 1)));
        }
        L.max_iterations = max_it;
        L.epoch = ++sc.loop_epoch;
        for (int i = 0; i < 7; ++i) L.T0[i] = init[i];
        L.acc_unscale = 1.0 / ip.acc_scale;
        launch_loop_solve(L, xp, sc.stream2);
        HIPCHK(hipGetLastError());
        solver_guard.sc = &sc;
        solver_guard.epoch = L.epoch;
        HIPCHK(hipEventRecord(sc.ev_solve, sc.stream2));
    }
    if (use_loop) {
        // ---- the whole loop in one launch (kernels.hip, k_loop): first its solving wave, on its own stream —
        // it has to hold its registers before the grid fills the machine; it waits for the grid's go
        L.sh = sc.d_loop;
        L.st = sc.d_state;
        L.nw = plan.nw;
        L.gpw = plan.gpw;
        L.wgs = plan.wgs;
        L.copies = kLoopReplicas;
        L.progress = nullptr;
        L.contiguous = env_int("SAGEICP_LOOP_CONTIGUOUS", 0) ? 1 : 0;
        {
            const uint64_t qw = 64u >> plan.lw, groups = (n + qw - 1) / qw;
            for (int x = 0; x <= 8; ++x) L.xcd_first[x] = static_cast<uint32_t>(groups * x / 8);
            // (an XCD's workgroups must be able to hold its range)
            const uint64_t nwg = static_cast<uint64_t>(plan.wgs / 8);
            for (int x = 0; x < 8; ++x)
                if (L.contiguous == 1 && (L.xcd_first[x + 1] - L.xcd_first[x] + nwg - 1) / nwg > static_cast<uint64_t>(plan.gpw)) L.contiguous = 0;
        }
        // a wait inside the launch normally takes microseconds; 50 ms of it means the grid is not
        // resident as a whole (SAGEICP_LOOP_TIMEOUT_MS overrides, e.g. under a debugger)
        L.timeout_ticks = 100000ull * static_cast<unsigned long long>(std::max(1, env_int("SAGEICP_LOOP_TIMEOUT_MS", 50)));
        if (const int ticks = env_int("SAGEICP_LOOP_TIMEOUT_TICKS", 0))      // tests: provoke a timeout
            L.timeout_ticks = static_cast<unsigned long long>(ticks);
        // (under a communicator the workgroups wait for a pose that waits for the peers' sums: their patience has
        // to outlast the exchange's — a peer's first launches of a process can take a second)
        L.count_timeout_ticks = L.timeout_ticks;       // (the solving wave's wait for its own workgroups: local, short)
        if (comm && env_int("SAGEICP_LOOP_TIMEOUT_TICKS", 0) == 0)
            L.timeout_ticks = std::max(L.timeout_ticks, xp.timeout_ticks + 100000000ull);
        // (tests: ONE rank of a communicator loses its grid — the others must follow it out of the launch)
        if (comm && env_int("SAGEICP_LOOP_COUNT_TIMEOUT_RANK", -1) == comm->rank)
            L.count_timeout_ticks = static_cast<unsigned long long>(std::max(1, env_int("SAGEICP_LOOP_COUNT_TIMEOUT_TICKS", 1)));
        L.max_iterations = max_it;
        L.epoch = ++sc.loop_epoch;
        for (int i = 0; i < 7; ++i) L.T0[i] = init[i];
        L.acc_unscale = 1.0 / ip.acc_scale;
        L.shared_loop = comm ? 1 : 0;
        L.prio = std::min(3, std::max(0, env_int("SAGEICP_LOOP_PRIO", 3)));
        L.deal = env_int("SAGEICP_LOOP_DEAL", 1) ? 1 : 0;
#ifndef SAGE_LOOP_INGRID          // (the counter-collection twin keeps the solving wave inside the grid: kernels.hip)
        launch_loop_solve(L, xp, sc.stream2);
        HIPCHK(hipGetLastError());
        solver_guard.sc = &sc;
        solver_guard.epoch = L.epoch;
        HIPCHK(hipEventRecord(sc.ev_solve, sc.stream2));
#endif
    }

    // Spatial ordering of the frame: the loop runs on a copy sorted by map-frame voxel under the
    // initial guess, so that the queries of a wave share home voxels and neighbouring waves touch
    // neighbouring voxel blocks (L1 / L2 hits, similar work per lane), and every query's
    // neighbourhood row is built for that order; inside the loop a row is redone only when its
    // query crosses a voxel face.  (Round 1 re-sorted when the pose had drifted half a voxel; with a
    // lane per query that no longer pays for its ~90 us: 45.3 against 47.2 us per iteration on the
    // c2 cold start, profiles/README.md.)
    // ... from kSortFrameFrom points on (kernels.h); SAGEICP_SORT_FROM overrides the size (0: always sorted)
    if (n > 0 && n < static_cast<uint64_t>(std::max(0, env_int("SAGEICP_SORT_FROM", kSortFrameFrom))))
        HIPCHK(check_copy_frame(d_frame, sc.d_sorted, static_cast<int>(n), sc.d_state, comm == nullptr, s));
    else if (n > 0)
        HIPCHK(sort_frame(d_frame, sc.d_sorted, static_cast<int>(n), sc.d_state, true, comm == nullptr,
                          m->host.voxel_size, sc.d_keys, sc.d_vals, sc.d_sort_temp,
                          sc.sort_temp_bytes_, s));

    double us_nn = 0, us_fin = 0;
    uint32_t nn_launches = 0;
    bool looped = false;
    if (use_loop) {
        // ---- ... then the grid (the shared block zeroed first: the solving wave starts on the grid's go)
        if (prof && (rc = sc.reserve_events(1))) return rc;
        IcpParams lp = ip;
        lp.filter = plan.filter ? ip.filter : 0;
        lp.nwaves = loop_waves;
        HIPCHK(hipMemsetAsync(sc.d_loop, 0, offsetof(LoopShared, acc32), s));      // (the chained launches' copies are not this loop's)
        if (prof) HIPCHK(hipEventRecord(sc.events[1], s));
        launch_loop(lp, L, plan.lw, s);
        if (hipPeekAtLastError() == hipSuccess) solver_guard.sc = nullptr;      // the grid is on its way: it will say go
        if (prof) HIPCHK(hipEventRecord(sc.events[2], s));
#ifndef SAGE_LOOP_INGRID
        HIPCHK(hipStreamWaitEvent(s, sc.ev_solve, 0));             // the solving wave writes the final state
#endif
        if (counting) launch_sum_counters(sc.d_cand, static_cast<int>(lp.nwaves), sc.d_state, s);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(sc.h_state, sc.d_state, sizeof(IcpState), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (sc.h_state->bad_input && !comm) {
            looped = true;                     // (reported below)
        } else if (sc.h_state->exchange_failed) {
            looped = true;                     // (reported below)
        } else if (sc.h_state->loop_aborted || !sc.h_state->done) {
            // a wait inside the launch timed out (the grid was not resident as a whole: another stream
            // or process held CUs): the launch-per-iteration loop below registers the frame instead,
            // with the same lanes per query
            if (sc.h_state->peer_aborted) {
                // not this rank's grid: a peer lost its one-launch loop and every rank left the same exchange with it
                // (sageicp_types.h, P2pBlock::abort_tag) — this frame goes through the other form on every rank, in step; no cool-down here
                sc.last_fallback = SAGEICP_LOOP_FALLBACK_PEER;
            } else {
            sc.loop_cooldown = std::max(0, env_int("SAGEICP_LOOP_COOLDOWN", 256));
            if (env_int("SAGEICP_LOOP_TIMEOUT_TICKS", 0) == 0 && env_int("SAGEICP_LOOP_COUNT_TIMEOUT_RANK", -1) < 0 && sc.loop_derate < 16) ++sc.loop_derate;
            ++sc.loop_timeouts;
            sc.last_fallback = SAGEICP_LOOP_FALLBACK_TIMEOUT;
            {
                // (said out loud once per process: the fast path was lost, and what it cost)
                static std::atomic<int> told{0};
                if (told.exchange(1) == 0 && env_int("SAGEICP_QUIET", 0) == 0)
                    std::fprintf(stderr, "sageicp: a wait inside the one-launch ICP loop timed out (the GPU is shared with other work, or "
                                         "fewer workgroups are resident than planned): this frame goes through the launch-per-iteration "
                                         "loop, the next %d calls of this map too, later plans take %d workgroups fewer "
                                         "(sageicp_map_loop_status() reports the state)\n", sc.loop_cooldown, 32 * sc.loop_derate);
            }
            }
            // (under a communicator the solving wave told the peers through the exchange it was about to make: they left
            // it with this rank, the exchange counts as made on every rank, and all of them register the frame again below)
            fill_state(sc.h_state, init);
            if (polled) {
                std::memset(sc.h_prog, 0, sizeof(IcpProgress));
                sc.h_state->progress = sc.d_prog;
            }
            HIPCHK(hipMemcpyAsync(sc.d_state, sc.h_state, sizeof(IcpState), hipMemcpyHostToDevice, s));
            if (counting) HIPCHK(hipMemsetAsync(sc.d_cand, 0, sizeof(unsigned long long) * 2 * (ip.nwaves + 1), s));
            use_loop = false;
        } else {
            looped = true;
            if (prof) {
                float a = 0;
                (void)hipEventElapsedTime(&a, sc.events[1], sc.events[2]);
                us_nn = 1e3 * a;
                nn_launches = static_cast<uint32_t>(std::max(1, sc.h_state->iter));   // per iteration
            }
        }
    }
    if (n > 0 && !looped) {
        launch_rows(ip, s);
        HIPCHK(hipMemsetAsync(sc.d_prev, 0xFF, n * sizeof(uint2), s));     // no previous answers yet
    }

    // The workgroups of k_icp add their sums into fixed-point accumulators (kernels.h) that k_fin
    // reads in one round trip.
    HIPCHK(hipMemsetAsync(sc.d_acc, 0, sizeof(long long) * kAccReplicas * kAccWords, s));
    ip.acc = sc.d_acc;
    if (chain) {
        // ... or, chained, into the counted accumulators of the shared block the solving wave reads (zeroed first: the
        // solving wave starts on launch 0's go)
        HIPCHK(hipMemsetAsync(sc.d_loop, 0, sizeof(LoopShared), s));
        ip.chain = sc.d_loop;
        ip.chain_timeout = L.timeout_ticks;
        ip.chain_epoch = L.epoch;
        ip.digit_limit = std::min(ip.digit_limit, std::ldexp(1.0, 40));       // (counted words: kernels.hip, kDigitLimitCounted)
    }
    FinParams fp{};
    fp.st = sc.d_state;
    fp.partials = nullptr;
    fp.acc = ip.acc;
    fp.acc_unscale = 1.0 / ip.acc_scale;
    fp.nparts = 0;
    fp.mode = p2p ? 3 : (comm ? 1 : 0);
    fp.standalone = 0;
    if (p2p) fp.p2p = xp;

    // one iteration; `slot` indexes its 5 profiling events
    // Profiling level 1 brackets k_icp in one iteration out of 8 (two event records cost ~6 us of
    // stream time): the roofline figure is the mean over that sample; level 2: every kernel of
    // every iteration.
    auto sampled = [&](int iteration) { return prof2 || (prof && (iteration & 7) == 4); };
    // Heaviest first (kernels.h, IcpParams::stripe_order): the stripes of k_icp are dispatched in the order of the work
    // iterations 0, 3 and 15 measured (a cold registration's first passes are not its later ones; from then on the heavy
    // regions stay the heavy regions).  The sort's buffers are the frame sort's, free once the frame is in order.
    const unsigned stripes = (n > 0 && !looped) ? static_cast<unsigned>(icp_stripes_for(static_cast<int>(n), lw)) : 0u;
    // (three small sorts per frame: worth it from ~40k points on — c1 through this loop: +4.5 % with them)
    const bool lpt = stripes >= 16 && stripes <= sc.sort_cap && stripe_sort_temp_bytes(stripes) <= sc.sort_temp_bytes_ &&
                     env_int("SAGEICP_LPT", n >= 40000 ? 1 : 0) != 0;
    uint32_t *st_work = sc.d_keys, *st_sorted = sc.d_keys + stripes, *st_iota = sc.d_vals, *st_order = sc.d_vals + stripes;
    if (lpt) stripe_order_init(st_work, st_iota, stripes, s);
    auto measures = [&](int iteration) { return lpt && (iteration == 0 || iteration == 3 || iteration == 15); };
    auto enqueue_iteration = [&](int slot, int iteration) -> int {
        const bool ev = sampled(iteration);
        if (ev) HIPCHK(hipEventRecord(sc.events[5 * slot + 1], s));
        ip.stripe_work = measures(iteration) ? st_work : nullptr;
        ip.chain_iter = iteration;
        launch_icp(ip, lw, true, s);
        if (chain && iteration == 0 && hipPeekAtLastError() == hipSuccess) solver_guard.sc = nullptr;      // launch 0 will say go
        if (measures(iteration)) {
            HIPCHK(stripe_order_sort(st_work, st_sorted, st_iota, st_order, stripes, sc.d_sort_temp, sc.sort_temp_bytes_, s));
            ip.stripe_order = st_order;
        }
        if (ev) HIPCHK(hipEventRecord(sc.events[5 * slot + 2], s));
        if (chain) return SAGEICP_OK;                  // (the solving wave is already waiting for this launch's sums)
        launch_fin(fp, s);
        if (comm && !p2p) {     // k_fin left the local sums in state->sums
            ncclResult_t r = g_rccl.AllReduce(sc.d_state->sums, sc.d_state->sums, kNumSums,
                                              ncclDouble, ncclSum, comm->comm, s);
            if (r != ncclSuccess)
                return fail(SAGEICP_ERR_RCCL, std::string("ncclAllReduce: ") +
                                                  (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
            FinParams f2 = fp;
            f2.mode = 2;
            launch_fin(f2, s);
        }
        if (prof2) HIPCHK(hipEventRecord(sc.events[5 * slot + 3], s));
        return SAGEICP_OK;
    };
    auto harvest = [&](int slot) {
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, sc.events[5 * slot + 1], sc.events[5 * slot + 2]);
        if (prof2) (void)hipEventElapsedTime(&b, sc.events[5 * slot + 2], sc.events[5 * slot + 3]);
        us_nn += 1e3 * a; us_fin += 1e3 * b;
        ++nn_launches;
    };
    // (nothing the host does depends on WHEN it looks at the progress word: a call repeated gives
    // the same bits)
    if (looped) {
        // (the one-launch loop has run; the state is on the host)
    } else if (polled) {
        const int depth = std::min(8, std::max(1, env_int("SAGEICP_DEPTH", 4)));
        volatile unsigned long long *word = &sc.h_prog->word;
        int enq = 0;
        unsigned spins = 0;
        unsigned long long idle_word = ~0ull;       // (chained: the progress word at the last look that found the stream idle)
        for (;;) {
            const unsigned long long w = *word;
            const int comp = static_cast<int>(w & 0xFFFFFFFFull);
            if (w >> 32) break;                                  // converged or out of iterations
            if (enq < max_it && enq - comp < depth) {
                if ((rc = enqueue_iteration(enq, enq))) return rc;
                ++enq;
                spins = 0;
                continue;
            }
            __builtin_ia32_pause();
            if ((++spins & 0xFFFFu) == 0) {                      // every ~ms: is the stream alive?
                const hipError_t q = hipStreamQuery(s);
                if (q != hipSuccess && q != hipErrorNotReady)
                    return fail(SAGEICP_ERR_HIP, std::string("ICP loop: ") + hipGetErrorString(q));
                if (q == hipSuccess && (*word >> 32) == 0 && enq >= max_it)
                    break;     // everything ran and nothing flagged the end: read the state below
                if (chain && q == hipSuccess) {
                    // chained: every launch enqueued has run, and the solving wave has said nothing new for a whole
                    // interval (~1 ms; a solve takes microseconds) — it is gone (an exit that did not reach the progress
                    // word): the state below says why, and the frame is registered again with k_fin
                    if (idle_word == *word) break;
                    idle_word = *word;
                } else {
                    idle_word = ~0ull;
                }
            }
        }
        if (chain && solver_guard.sc) {
            // no launch was enqueued at all (the sort refused the frame — a non-finite point — before the host got to
            // iteration 0): nobody will tell the solving wave to start, so it is sent home here, not when this call returns
            sc.go_word = solver_guard.epoch | 0x8000000000000000ull;
            HIPCHK(hipMemcpyAsync(&sc.d_loop->go[0], &sc.go_word, sizeof(sc.go_word), hipMemcpyHostToDevice, s));
            solver_guard.sc = nullptr;
        }
        if (chain) HIPCHK(hipStreamWaitEvent(s, sc.ev_solve, 0));      // the solving wave writes the final state
        if (counting) launch_sum_counters(sc.d_cand, static_cast<int>(ip.nwaves), sc.d_state, s);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(sc.h_state, sc.d_state, sizeof(IcpState), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (chain && (sc.h_state->loop_aborted || !sc.h_state->done) && !sc.h_state->bad_input) {
            // a wait timed out (the solving wave was not resident beside the launches, or a launch took longer than its
            // patience): the frame again with k_fin between the launches — same lanes per query, same bits
            static std::atomic<int> told{0};
            if (told.exchange(1) == 0 && env_int("SAGEICP_QUIET", 0) == 0)
                std::fprintf(stderr, "sageicp: a wait inside the chained ICP launches timed out: this frame is registered again with "
                                     "k_fin between the launches\n");
            g_no_chain = true;
            const int rc2 = run_icp(m, d_frame, n, init, max_dist, kernel, sem_th, comm, out, stats, us_upload, t_begin);
            g_no_chain = false;
            return rc2;
        }
        if (prof)
            for (int k = 0; k < sc.h_state->iter && k < enq; ++k)      // the rest were no-ops
                if (sampled(k)) harvest(k);
    } else {
        int launched = 0;
        int chunk = 4;
        for (;;) {
            const int todo = std::min(chunk, kMaxIterations - launched);
            for (int k = 0; k < todo; ++k)
                if ((rc = enqueue_iteration(k, launched + k))) return rc;
            if (counting) launch_sum_counters(sc.d_cand, static_cast<int>(ip.nwaves), sc.d_state, s);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(sc.h_state, sc.d_state, sizeof(IcpState), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (prof) {
                const int executed = std::min(todo, sc.h_state->iter - launched);   // the rest were no-ops
                for (int k = 0; k < executed; ++k)
                    if (sampled(launched + k)) harvest(k);
            }
            launched += todo;
            if (sc.h_state->done || launched >= kMaxIterations) break;
            chunk = std::min(kChunkMax, chunk * 2);   // 4, 8, 16, 16, ... : few syncs, bounded no-op tail
        }
    }
    const IcpState &st = *sc.h_state;
    if (st.peer_aborted && !looped && comm && g_restarts < 4) {
        // a peer gave up its one-launch loop at an exchange this rank made from k_fin: every rank starts the frame again
        // (this one in the form it already had)
        ++g_restarts;
        g_no_loop = true;
        const int rc2 = run_icp(m, d_frame, n, init, max_dist, kernel, sem_th, comm, out, stats, us_upload, t_begin);
        g_no_loop = false;
        --g_restarts;
        return rc2;
    }
    if (st.bad_input)
        return fail(SAGEICP_ERR_INVALID, "the frame holds a coordinate or label that is not finite (NaN / Inf)");
    if (st.acc_overflow && !comm && g_acc_shift < 2) {
        // |sum over four queries| >= 2^46 or 2^62 / blocks of the frame (2^40 in the one-launch loop): georeferenced coordinates (UTM: ~3e6 m,
        // 4 s^2 = 4e13; 10^7 m beyond) do that.  The reference has no such limit: the frame is registered again
        // with the sums accumulated at 2^-24, then 2^-48 of their value — the same exact integer arithmetic on
        // digits of weight 2^24, 2^-16, 2^-56 (what is dropped lies 2^80 below the limit either way).
        ++g_acc_shift;
        const int rc2 = run_icp(m, d_frame, n, init, max_dist, kernel, sem_th, comm, out, stats, us_upload, t_begin);
        --g_acc_shift;
        return rc2;
    }
    if (st.acc_overflow)
        return fail(SAGEICP_ERR_CAPACITY, "a Gauss-Newton sum left the range of the fixed-point accumulators "
                                          "(coordinates beyond ~10^13 m, a pose guess that is not finite — or, under a communicator, "
                                          "|sum over four queries| >= 2^40: every rank would have to take the same decision)");
    if (st.exchange_failed) {
        // the ranks' exchange counters may now differ by one: a later exchange could pass its wait
        // on a stale tag and add rows of another iteration.  The blocks are dead until every rank
        // exports and connects fresh ones.
        if (comm) {
            comm->p2p = false;
            comm->poisoned = true;
        }
        return fail(SAGEICP_ERR_RCCL, "direct exchange: a peer's sums did not arrive in time");
    }
    for (int i = 0; i < 7; ++i) out[i] = st.T[i];
    if (looped) ++sc.calls_single_launch; else ++sc.calls_per_iteration;
    if (!looped && chain) ++sc.calls_chained;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->iterations = st.iter;
        stats->converged = st.converged;
        stats->n_queries = n;
        stats->n_corr_first = st.iter > 0 ? st.n_corr[0] : 0;
        stats->n_corr_last = st.iter > 0 ? st.n_corr[std::min(st.iter, kHistory) - 1] : 0;
        stats->last_step_norm = st.last_step_norm;
        stats->us_upload = us_upload;
        stats->us_nn = us_nn; stats->us_fin = us_fin;
        stats->nn_launches = nn_launches;
        stats->sum_candidates = st.sum_candidates;
        stats->pairs_evaluated = st.sum_pairs;
        stats->lanes_per_query = 1u << lw;
        stats->compact_scan = (looped ? plan.filter && ip.filter : ip.filter != 0) ? 1u : 0u;
        stats->single_launch = looped ? 1u : 0u;
        for (int i = 0; i < 64 && i < st.iter; ++i) stats->n_corr_hist[i] = st.n_corr[i];
        stats->us_wall = now_us() - t_begin;
    }
    return SAGEICP_OK;
}


// ---- single-process multi-GPU mode -------------------------------------------------------------
// Update(points, pose) on every copy of the map.  `d_points` (optional) lives on rank 0's device.
int device_update_all(sageicp_map *m, const double *xyzl, uint64_t n, const double pose[7],
                      const Point4 *d_points) {
    if (m->replicas_diverged)
        return fail(SAGEICP_ERR_INVALID, "the copies of this multi-device map diverged in an earlier failed update: Clear() it");
    int rc = device_update(m, xyzl, n, pose, d_points);
    if (rc || m->replicas.empty()) return rc;       // (a failed device update changes nothing on its device)
    std::vector<double> host;
    if (d_points) {                               // the other devices take the points from the host
        host.resize(4 * n);
        HIPCHK(hipSetDevice(m->device));
        if (n) HIPCHK(hipMemcpy(host.data(), d_points, n * sizeof(Point4), hipMemcpyDeviceToHost));
        xyzl = host.data();
    }
    for (sageicp_map *r : m->replicas)
        if ((rc = device_update(r, xyzl, n, pose))) {
            m->replicas_diverged = true;            // rank 0 (and maybe others) took the update, this copy did not
            const std::string why = g_err;
            return fail(rc, "update reached only some devices of the map (" + why + "); the map must be cleared");
        }
    return SAGEICP_OK;
}

// exchange blocks of the ranks of one process: fine-grained device memory, reached by the other
// devices through peer access (no IPC)
int create_ranks(const sageicp_map *m) {
    const int N = 1 + static_cast<int>(m->replicas.size());
    if (static_cast<int>(m->ranks.size()) == N) {
        for (sageicp_comm *c : m->ranks)
            if (c->poisoned)
                return fail(SAGEICP_ERR_RCCL, "an earlier exchange between the devices of this map timed out: "
                                              "call sageicp_map_set_devices again");
        return SAGEICP_OK;
    }
    std::vector<int> dev(N);
    dev[0] = m->device;
    for (int k = 1; k < N; ++k) dev[k] = m->replicas[k - 1]->device;
    for (int a = 0; a < N; ++a)
        for (int b = 0; b < N; ++b) {
            if (dev[a] == dev[b]) continue;
            int can = 0;
            HIPCHK(hipDeviceCanAccessPeer(&can, dev[a], dev[b]));
            if (!can) return fail(SAGEICP_ERR_HIP, "devices of one map need peer access to each other");
            HIPCHK(hipSetDevice(dev[a]));
            const hipError_t e = hipDeviceEnablePeerAccess(dev[b], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                return fail(SAGEICP_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            (void)hipGetLastError();
        }
    std::vector<sageicp_comm *> ranks(N, nullptr);
    auto undo = [&]() {
        for (sageicp_comm *c : ranks) sageicp_comm_destroy(c);
    };
    for (int k = 0; k < N; ++k) {
        sageicp_comm *c = new sageicp_comm;
        ranks[k] = c;
        c->rank = k; c->nranks = N; c->device = dev[k];
        c->peer_mapped = true;
        for (int r = 0; r < N; ++r)
            if (r != k && dev[r] == dev[k]) c->device_shared = true;
        if (hipSetDevice(dev[k]) != hipSuccess ||
            hipExtMallocWithFlags(reinterpret_cast<void **>(&c->my_block), sizeof(P2pBlock),
                                  hipDeviceMallocFinegrained) != hipSuccess ||
            hipMemset(c->my_block, 0, sizeof(P2pBlock)) != hipSuccess ||
            hipMalloc(&c->d_exchanges, sizeof(unsigned long long)) != hipSuccess ||
            hipMemset(c->d_exchanges, 0, sizeof(unsigned long long)) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) {
            undo();
            return fail(SAGEICP_ERR_HIP, "allocating the exchange blocks failed");
        }
    }
    for (int k = 0; k < N; ++k) {
        for (int r = 0; r < N; ++r) ranks[k]->blocks[r] = ranks[r]->my_block;
        ranks[k]->p2p = true;
    }
    m->ranks = ranks;
    return SAGEICP_OK;
}

// RegisterFrame over all devices of the map: rank k registers block k of the frame (contiguous
// blocks of ceil(n / N) points) against its copy of the map, on its own host thread and
// stream; the sums meet in k_fin (direct exchange).  Exactly one of h_frame / d_frame is given
// (d_frame on rank 0's device).
int register_sharded(const sageicp_map *m, const double *h_frame, const Point4 *d_frame, uint64_t n,
                     const double init[7], double max_dist, double kernel, double sem_th,
                     double pose_out[7], sageicp_stats *stats, double t0) {
    if (m->replicas_diverged)
        return fail(SAGEICP_ERR_INVALID, "the copies of this multi-device map diverged in an earlier failed update: Clear() it");
    int rc = create_ranks(m);
    if (rc) return rc;
    const int N = 1 + static_cast<int>(m->replicas.size());
    std::vector<const sageicp_map *> maps(N);
    maps[0] = m;
    for (int k = 1; k < N; ++k) maps[k] = m->replicas[k - 1];
    const uint64_t per = (n + N - 1) / N;
    std::vector<int> codes(N, SAGEICP_OK);
    std::vector<std::string> errors(N);
    std::vector<std::array<double, 7>> poses(N);
    std::vector<sageicp_stats> st(N);
    // Everything that can fail before the loop (mirror refresh, buffers, the copy of the shard) is
    // done by every rank first; the ranks meet, and enter the loop only if all of them are ready —
    // a rank that failed alone would leave the others waiting in k_fin for sums that never come.
    std::mutex gate_mu;
    std::condition_variable gate_cv;
    int gate_arrived = 0;
    bool gate_ok = true;
    auto work = [&](int k) {
        const sageicp_map *mk = maps[k];
        const uint64_t lo = std::min<uint64_t>(n, k * per), cnt = std::min<uint64_t>(n, lo + per) - lo;
        const Point4 *mine = nullptr;
        auto setup = [&]() -> int {
            HIPCHK(hipSetDevice(mk->device));
            int r = sync_mirror(mk);
            if (r) return r;
            Scratch &sc = mk->sc;
            if ((r = sc.reserve_frame(cnt))) return r;
            if ((r = ensure_cand(mk, wants_filter(mk, cnt, sem_th)))) return r;
            if ((r = sc.reserve_sort(cnt))) return r;
            mine = sc.d_frame;
            if (cnt) {
                if (h_frame)
                    HIPCHK(hipMemcpyAsync(sc.d_frame, h_frame + 4 * lo, cnt * sizeof(Point4),
                                          hipMemcpyHostToDevice, sc.stream));
                else if (k == 0)
                    mine = d_frame + lo;
                else
                    HIPCHK(hipMemcpyPeerAsync(sc.d_frame, mk->device, d_frame + lo, m->device,
                                              cnt * sizeof(Point4), sc.stream));
                HIPCHK(hipStreamSynchronize(sc.stream));     // the shard has arrived (or the copy failed: here, not in the loop)
            }
            return SAGEICP_OK;
        };
        codes[k] = setup();
        if (codes[k]) errors[k] = g_err;          // g_err is per thread
        {
            std::unique_lock<std::mutex> lk(gate_mu);
            if (codes[k]) gate_ok = false;
            if (++gate_arrived == N) gate_cv.notify_all();
            else gate_cv.wait(lk, [&] { return gate_arrived == N; });
            if (!gate_ok) {
                if (!codes[k]) {
                    codes[k] = SAGEICP_ERR_HIP;
                    errors[k] = "not started: another device rank failed its set-up";
                }
                return;
            }
        }
        codes[k] = run_icp(mk, mine, cnt, init, max_dist, kernel, sem_th, m->ranks[k], poses[k].data(),
                           &st[k], now_us() - t0, t0);
        if (codes[k]) errors[k] = g_err;
    };
    std::vector<std::thread> th;
    for (int k = 1; k < N; ++k) th.emplace_back(work, k);
    work(0);
    for (auto &t : th) t.join();
    (void)hipSetDevice(m->device);
    for (int k = 0; k < N; ++k)        // the rank that failed on its own first, then the ones it stopped
        if (codes[k] && errors[k].rfind("not started", 0) != 0)
            return fail(codes[k], "device rank " + std::to_string(k) + ": " + errors[k]);
    for (int k = 0; k < N; ++k)
        if (codes[k]) return fail(codes[k], "device rank " + std::to_string(k) + ": " + errors[k]);
    std::memcpy(pose_out, poses[0].data(), 56);
    if (stats) {
        *stats = st[0];
        stats->n_queries = n;
        for (int k = 1; k < N; ++k) {
            stats->sum_candidates += st[k].sum_candidates;
            stats->pairs_evaluated += st[k].pairs_evaluated;
        }
        stats->us_wall = now_us() - t0;
    }
    return SAGEICP_OK;
}

}  // namespace sageicp_impl
