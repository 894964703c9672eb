// Internals shared by the translation units of libsageicp_hip.so's host side (capi.hip, capi_mirror.hip,
// capi_run.hip): error channel, tuning knobs, the per-handle device scratch, the pipeline's buffers, the opaque
// handles of include/sageicp.h.  Not part of the C ABI.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <limits>
#include <memory>
#include <cstring>
#include <map>
#include <mutex>
#include <array>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/sageicp.h"
#include "host_map.hpp"
#include "kernels.h"
#include "map_update.h"
#include "metrics.hpp"
#include "pipeline.hpp"
#include "robin_order.hpp"
#include "se3_math.h"
#include "sageicp_types.h"

namespace sageicp {

// ---- errors --------------------------------------------------------------------------
extern thread_local std::string g_err;
// (switches set through the C ABI, read by every call: several host threads may be inside the library)
extern std::atomic<int> g_profiling;
extern std::atomic<int> g_counting;                // sageicp_set_counting: the per-wave candidate / pair counters behind sageicp_stats
// VoxelDownsample emits its survivors in the reference's order (the bucket order of its
// tsl::robin_map, replayed on the host: robin_order.hpp) unless switched to arrival order
extern std::atomic<int> g_reference_order;

// tuning knobs (defaults chosen by measurement on MI355X; the environment overrides are for
// experiments only)
// The process environment is read ONCE per knob and call site (the first time it is asked for) and again only after
// sageicp_reload_env(): a registration asks for ~35 knobs, and a ROS node's environment does not change under it.
// (Tests and probes that flip knobs between calls go through the Python binding, which calls sageicp_reload_env()
// whenever the SAGEICP_* part of os.environ changed.)
extern std::atomic<unsigned> g_env_epoch;          // bumped by sageicp_reload_env()
struct EnvKnob {
    const char *name;
    std::atomic<unsigned> epoch{0xFFFFFFFFu};
    std::atomic<int> has{0}, val{0};
};
inline int env_knob(EnvKnob &k, int dflt) {
    const unsigned e = g_env_epoch.load(std::memory_order_relaxed);
    if (k.epoch.load(std::memory_order_acquire) != e) {
        const char *v = std::getenv(k.name);
        k.val.store(v ? std::atoi(v) : 0, std::memory_order_relaxed);
        k.has.store(v ? 1 : 0, std::memory_order_relaxed);
        k.epoch.store(e, std::memory_order_release);
    }
    return k.has.load(std::memory_order_relaxed) ? k.val.load(std::memory_order_relaxed) : dflt;
}
#define env_int(name, dflt) ([&]() -> int { static ::sageicp::EnvKnob _knob{name}; return ::sageicp::env_knob(_knob, (dflt)); }())
// a knob that is text: the value at the last (re)load (capi.hip), or nullptr
const char *env_cached(const char *name);
// Lanes per query in k_icp (log2).  One lane per query needs the fewest instructions per query
// but gives a frame of n points only n / 64 waves with long dependent chains; small frames and
// shards spread each query over more lanes.  Thresholds measured on MI355X (profiles/README.md);
// SAGEICP_LW overrides for experiments.
inline int icp_lw(uint64_t n, bool sparse_voxels) {
    const int e = env_int("SAGEICP_LW", -1);
    if (e >= 0) return e > 4 ? 4 : e;
    // (all of this re-measured after the flat-order scan, profiles/r04/lanes_probe2.txt (dense voxels) and
    // lanes_probe3.txt (sparse ones); us per iteration)
    // the biggest frames are bound by instruction issue, not by the length of a wave's chain: two
    // lanes per query halve the per-query share of the fixed work (prologue, bounds, epilogue) — against
    // dense voxels only at c4's size (500k: 90.7 against 92.9 with four, 400k a tie), against sparse ones
    // from ~150k (c5, 200k: 42.8 / 43.2; 100k: 33.3 / 31.6)
    if (n >= (sparse_voxels ? 150000u : 400000u)) return 1;
    // eight lanes stride through a query's voxels in flat order (kernels.hip) and hold against dense voxels
    // up to ~110k queries (50k: 25.7 against 31.1 with four; 60k: 28.0 / 31.5; 80k: 32.3 / 34.5; 100k: 36.2 /
    // 37.1; 120k: 40.8 / 40.6), against sparse ones up to ~60k (25k: 19.8 / 22.9; 50k: 24.7 / 25.1; 100k:
    // 34.9 / 31.6); until late round 4 the switch to four sat at 50k and, for sparse voxels, at 4k
    if (n >= (sparse_voxels ? 60000u : 110000u)) return 2;
    // sixteen lanes only for small frames against dense voxels (in flat order they hold up to ~20k queries:
    // 10k 17.8 against 18.9 with eight, 15k 19.1 / 20.0, 30k 24.7 / 22.6 — lanes_probe4.txt; the switch used
    // to sit at 10k): a scan against sparse ones is a handful of points whatever the split (c1, 10k: 17.6
    // with eight, 20.7 with four)
    if (n >= (sparse_voxels ? 4096u : 20000u)) return 3;
    return 4;
}


inline int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return fail(SAGEICP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

inline double now_us() {
    using namespace std::chrono;
    return duration<double, std::micro>(steady_clock::now().time_since_epoch()).count();
}

// ---- per-handle device scratch -----------------------------------------------------------
constexpr int kChunkMax = 16;

struct Scratch {
    int device = -1;
    hipStream_t stream = nullptr;
    Point4 *d_frame = nullptr; size_t frame_cap = 0;
    Point4 *d_tgt = nullptr; size_t tgt_cap = 0;
    int32_t *d_nn = nullptr; size_t nn_cap = 0;
    // Morton re-ordering of the frame (sort.hip)
    Point4 *d_sorted = nullptr; uint32_t *d_keys = nullptr; uint32_t *d_vals = nullptr;
    void *d_sort_temp = nullptr; size_t sort_cap = 0; size_t sort_temp_bytes_ = 0;
    // per-call work buffers: the queries' cached neighbourhood rows, the workgroup partials
    uint32_t *d_rows = nullptr;
    uint2 *d_prev = nullptr;       // every query's record of the previous iteration (kernels.h)
    uint32_t *d_work = nullptr;    // instrumented builds: points handed to each query
    double *d_partials = nullptr; size_t partials_cap = 0;
    long long *d_acc = nullptr;    // fixed-point accumulators of the Gauss-Newton sums (kernels.h, kAcc*)
    LoopShared *d_loop = nullptr;  // what the workgroups of k_loop share inside its launch (kernels.h)
    unsigned long long go_word = 0; // (host source of a `go` word sent by a copy: run_icp)
    hipStream_t stream2 = nullptr; // the solving wave of the one-launch loop runs here, beside the grid on `stream`
                                   // (created with the first such launch: a process has few hardware queues, and
                                   // streams that never run anything still take their turn on them)
    std::vector<uint32_t> cu_mask; // of both streams (empty: the whole device)
    hipEvent_t ev_solve = nullptr; // ... and this says that it has finished
    unsigned long long loop_epoch = 0;
    int num_cus = 0;               // CUs the streams of this handle may use (the whole device, or its share: below)
    int cu_share_i = 0, cu_share_k = 1;   // SAGEICP_CU_SHARE=i/k: the i-th of k equal parts of the device's CUs (several
                                   // ranks on ONE GPU — tests, or a small node — each keep a persistent grid resident)
    int loop_cooldown = 0;         // calls that stay away from k_loop after one of its launches timed out
    int loop_derate = 0;           // x 32 workgroups fewer than the residency rule allows: one more after every time-out
    // what sageicp_map_loop_status reports
    uint64_t calls_single_launch = 0, calls_per_iteration = 0, calls_chained = 0;
    uint32_t loop_timeouts = 0;
    int last_fallback = 0;
    unsigned long long *d_cand = nullptr;      // per-wave counters of k_icp [2 x sort_cap]
    IcpState *d_state = nullptr;
    IcpState *h_state = nullptr;   // pinned
    IcpProgress *h_prog = nullptr; // pinned + host-mapped: written by the device every iteration
    IcpProgress *d_prog = nullptr; // its device address
    std::vector<hipEvent_t> events;  // 5 per profiled iteration

    int init(int dev) {
        if (stream) return SAGEICP_OK;
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            return fail(SAGEICP_ERR_NO_DEVICE, "no HIP device visible (gfx950 required; no CPU fallback)");
        if (dev < 0 || dev >= count) return fail(SAGEICP_ERR_INVALID, "device ordinal out of range");
        device = dev;
        HIPCHK(hipSetDevice(device));
        {
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) num_cus = cus;
        }
        if (cu_share_k <= 1) {
            if (const char *e = env_cached("SAGEICP_CU_SHARE")) {
                int i = 0, k = 1;
                if (std::sscanf(e, "%d/%d", &i, &k) == 2 && k >= 1 && k <= 16 && i >= 0 && i < k) {
                    cu_share_i = i;
                    cu_share_k = k;
                }
            }
        }
        if (cu_share_k > 1 && num_cus >= 8 * cu_share_k) {
            // this handle's kernels run on CUs [i, i + 1) * num_cus / k only: the persistent grids of k ranks
            // that share one GPU are then resident side by side instead of waiting for each other
            const int per = num_cus / cu_share_k, lo = cu_share_i * per;
            std::vector<uint32_t> mask((num_cus + 31) / 32, 0u);
            for (int c = lo; c < lo + per; ++c) mask[c / 32] |= 1u << (c % 32);
            HIPCHK(hipExtStreamCreateWithCUMask(&stream, static_cast<uint32_t>(mask.size()), mask.data()));
            cu_mask = mask;
            num_cus = per;
        } else {
            cu_share_i = 0;
            cu_share_k = 1;
            HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        }
        HIPCHK(hipMalloc(&d_state, sizeof(IcpState)));
        HIPCHK(hipMalloc(&d_acc, sizeof(long long) * kAccReplicas * kAccWords));
        HIPCHK(hipMalloc(&d_loop, sizeof(LoopShared)));
        HIPCHK(hipMemset(d_loop, 0, sizeof(LoopShared)));

        HIPCHK(hipHostMalloc(&h_state, sizeof(IcpState), hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&h_prog, sizeof(IcpProgress), hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_prog), h_prog, 0));
        return SAGEICP_OK;
    }
    int loop_streams() {
        if (stream2) return SAGEICP_OK;
        HIPCHK(hipSetDevice(device));
        if (!cu_mask.empty()) HIPCHK(hipExtStreamCreateWithCUMask(&stream2, static_cast<uint32_t>(cu_mask.size()), cu_mask.data()));
        else {
            // a stream of its own priority gets a hardware queue of its own: the solving wave runs for the whole
            // loop, and whatever shared its queue (a process has four) would wait behind it — the pipeline's
            // prefetch stream did (2.74 against 2.17 ms per streamed frame, profiles/r05/stream.txt)
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, greatest) != hipSuccess) {
                (void)hipGetLastError();
                HIPCHK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
            }
        }
        HIPCHK(hipEventCreateWithFlags(&ev_solve, hipEventDisableTiming));
        return SAGEICP_OK;
    }
    int reserve_frame(size_t n) {
        if (n <= frame_cap) return SAGEICP_OK;
        if (d_frame) HIPCHK(hipFree(d_frame));
        d_frame = nullptr; frame_cap = 0;
        const size_t cap = n + n / 4 + 1024;
        HIPCHK(hipMalloc(&d_frame, cap * sizeof(Point4)));
        frame_cap = cap;
        return SAGEICP_OK;
    }
    int reserve_tgt(size_t n) {
        if (n <= tgt_cap) return SAGEICP_OK;
        if (d_tgt) HIPCHK(hipFree(d_tgt));
        d_tgt = nullptr; tgt_cap = 0;
        const size_t cap = n + n / 4 + 1024;
        HIPCHK(hipMalloc(&d_tgt, cap * sizeof(Point4)));
        tgt_cap = cap;
        return SAGEICP_OK;
    }
    int reserve_nn(size_t n) {
        if (n <= nn_cap) return SAGEICP_OK;
        if (d_nn) HIPCHK(hipFree(d_nn));
        d_nn = nullptr; nn_cap = 0;
        const size_t cap = n + n / 4 + 1024;
        HIPCHK(hipMalloc(&d_nn, cap * sizeof(int32_t)));
        nn_cap = cap;
        return SAGEICP_OK;
    }
    int reserve_sort(size_t n) {
        if (n <= sort_cap) return SAGEICP_OK;
        if (d_sorted) HIPCHK(hipFree(d_sorted));
        if (d_keys) HIPCHK(hipFree(d_keys));
        if (d_vals) HIPCHK(hipFree(d_vals));
        if (d_sort_temp) HIPCHK(hipFree(d_sort_temp));
        if (d_rows) HIPCHK(hipFree(d_rows));
        if (d_prev) HIPCHK(hipFree(d_prev));
        d_rows = nullptr; d_prev = nullptr;
        d_sorted = nullptr; d_keys = d_vals = nullptr; d_sort_temp = nullptr; sort_cap = 0;
        const size_t cap = n + n / 4 + 1024;
        HIPCHK(hipMalloc(&d_sorted, cap * sizeof(Point4)));
        HIPCHK(hipMalloc(&d_rows, cap * kRowWords * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&d_prev, cap * sizeof(uint2)));
#ifdef SAGE_NN_TIMING
        if (d_work) HIPCHK(hipFree(d_work));
        d_work = nullptr;
        HIPCHK(hipMalloc(&d_work, cap * sizeof(uint32_t)));
#endif
        if (d_cand) HIPCHK(hipFree(d_cand));
        d_cand = nullptr;
        HIPCHK(hipMalloc(&d_cand, 2 * cap * sizeof(unsigned long long)));
        HIPCHK(hipMalloc(&d_keys, 2 * cap * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&d_vals, 2 * cap * sizeof(uint32_t)));
        sort_temp_bytes_ = sort_temp_bytes(static_cast<int>(cap));
        HIPCHK(hipMalloc(&d_sort_temp, sort_temp_bytes_));
        sort_cap = cap;
        return SAGEICP_OK;
    }
    int reserve_partials(size_t blocks) {
        if (blocks <= partials_cap) return SAGEICP_OK;
        if (d_partials) HIPCHK(hipFree(d_partials));
        d_partials = nullptr; partials_cap = 0;
        const size_t cap = blocks + blocks / 4 + 256;
        HIPCHK(hipMalloc(&d_partials, cap * kNumSums * sizeof(double)));
        partials_cap = cap;
        return SAGEICP_OK;
    }
    int reserve_events(size_t iterations) {
        while (events.size() < 5 * iterations) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            events.push_back(e);
        }
        return SAGEICP_OK;
    }
    void destroy() {
        if (!stream) return;
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(stream);
        if (stream2) (void)hipStreamSynchronize(stream2);
        if (ev_solve) (void)hipEventDestroy(ev_solve);
        if (stream2) (void)hipStreamDestroy(stream2);
        for (auto &e : events) (void)hipEventDestroy(e);
        events.clear();
        if (d_frame) (void)hipFree(d_frame);
        if (d_tgt) (void)hipFree(d_tgt);
        if (d_nn) (void)hipFree(d_nn);
        if (d_sorted) (void)hipFree(d_sorted);
        if (d_keys) (void)hipFree(d_keys);
        if (d_vals) (void)hipFree(d_vals);
        if (d_sort_temp) (void)hipFree(d_sort_temp);
        if (d_rows) (void)hipFree(d_rows);
        if (d_prev) (void)hipFree(d_prev);
        if (d_work) (void)hipFree(d_work);
        if (d_partials) (void)hipFree(d_partials);
        if (d_state) (void)hipFree(d_state);
        if (d_acc) (void)hipFree(d_acc);
        if (d_loop) (void)hipFree(d_loop);
        if (d_cand) (void)hipFree(d_cand);
        if (h_state) (void)hipHostFree(h_state);
        if (h_prog) (void)hipHostFree(h_prog);
        (void)hipStreamDestroy(stream);
        *this = Scratch();
    }
};


// A few parked host threads for the order replays of one Prep (one per label group at most): a
// replay of a few thousand keys costs no more than starting a thread does, and the replays of a
// level are the critical path of a streamed frame.  run(count, f) executes f(0..count-1), each index
// once, on the workers and the calling thread; indices are handed out in order (largest job first
// if the caller sorted them so).
class ReplayPool {
public:
    ~ReplayPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void run(size_t count, const std::function<void(size_t)> &f, size_t want_threads) {
        if (count <= 1 || want_threads <= 1) {
            for (size_t i = 0; i < count; ++i) f(i);
            return;
        }
        while (th_.size() + 1 < std::min(want_threads, count)) th_.emplace_back([this] { worker(); });
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &f;
            total_ = count;
            next_ = 0;
            pending_ = count;
            ++epoch_;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
        total_ = next_ = 0;
    }

private:
    void drain() {
        for (;;) {
            size_t i;
            const std::function<void(size_t)> *job;
            {   // (a handful of jobs per level: the lock is not contended, and a worker still between
                // two jobs when the next run() starts sees that run's state consistently)
                std::lock_guard<std::mutex> lk(mu_);
                if (next_ >= total_) return;
                i = next_++;
                job = job_;
            }
            (*job)(i);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || (epoch_ != seen && job_); });
                if (stop_) return;
                seen = epoch_;
            }
            drain();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)> *job_ = nullptr;
    size_t total_ = 0, pending_ = 0, next_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

// ---- device preprocessing (preprocess.hip): buffers of one pipeline ------------------------------
struct Prep {
    int device = -1;
    hipStream_t stream = nullptr;
    size_t cap = 0;                     // points
    Point4 *d_in = nullptr, *d_tmp = nullptr, *d_fd = nullptr, *d_src = nullptr;
    uint32_t *d_slot = nullptr, *d_skey = nullptr, *d_sval = nullptr, *d_winner = nullptr;
    unsigned long long *d_keys = nullptr;
    uint32_t table_cap = 0;
    void *d_sort_temp = nullptr;
    size_t sort_bytes = 0;
    unsigned long long *d_okeys = nullptr;   // survivors' voxel keys (reference-order emission)
    uint32_t *d_perm = nullptr;
    unsigned long long *h_keys = nullptr;    // pinned
    uint32_t *h_perm = nullptr;              // pinned
    std::vector<uint32_t> h_hash;
    RobinScratch rscratch[8];                // bucket arrays of the order replay, one pair per label group
    std::unique_ptr<ReplayPool> pool;        // parked helper threads of the order replays
    double us_order = 0;                // host time of the last run's order replays
    // levels whose survivors are emitted in arrival order even under g_reference_order (bit l): the
    // pipeline's second level — its cloud is only registered, and registration sorts its frame
    // spatially first, so its emission order reaches nothing but the order of fp64 summation
    unsigned arrival_order_levels = 0;
    uint32_t *d_nkept = nullptr;        // [2]
    int *d_overflow = nullptr;
    int *d_gcounts = nullptr, *d_glabels = nullptr;
    size_t glabels_cap = 0;
    void *h_pin = nullptr;              // pinned staging for the raw frame and the results
    size_t pin_bytes = 0;
    uint32_t kept_levels[2] = {0, 0};   // points the last run left in d_fd / d_src

    int init(int dev) {
        if (stream) return SAGEICP_OK;
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            return fail(SAGEICP_ERR_NO_DEVICE, "no HIP device visible (gfx950 required; no CPU fallback)");
        if (dev < 0 || dev >= count) return fail(SAGEICP_ERR_INVALID, "device ordinal out of range");
        device = dev;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIPCHK(hipMalloc(&d_nkept, 2 * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&d_overflow, sizeof(int)));
        HIPCHK(hipMalloc(&d_gcounts, 8 * sizeof(int)));
        return SAGEICP_OK;
    }
    int reserve(size_t n, size_t nlabels) {
        if (nlabels > glabels_cap) {
            if (d_glabels) HIPCHK(hipFree(d_glabels));
            d_glabels = nullptr;
            HIPCHK(hipMalloc(&d_glabels, (nlabels + 16) * sizeof(int)));
            glabels_cap = nlabels + 16;
        }
        if (n <= cap) return SAGEICP_OK;
        free_points();
        const size_t c = n + n / 4 + 1024;
        uint32_t t = 1024;
        while (t < 2 * c) t <<= 1;
        HIPCHK(hipMalloc(&d_in, c * sizeof(Point4)));
        HIPCHK(hipMalloc(&d_tmp, c * sizeof(Point4)));
        HIPCHK(hipMalloc(&d_fd, c * sizeof(Point4)));
        HIPCHK(hipMalloc(&d_src, c * sizeof(Point4)));
        HIPCHK(hipMalloc(&d_slot, c * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&d_skey, 2 * c * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&d_sval, 2 * c * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&d_keys, static_cast<size_t>(t) * sizeof(unsigned long long)));
        HIPCHK(hipMalloc(&d_winner, static_cast<size_t>(t) * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&d_okeys, c * sizeof(unsigned long long)));
        HIPCHK(hipMalloc(&d_perm, c * sizeof(uint32_t)));
        if (h_keys) (void)hipHostFree(h_keys);
        if (h_perm) (void)hipHostFree(h_perm);
        h_keys = nullptr; h_perm = nullptr;
        HIPCHK(hipHostMalloc(&h_keys, c * sizeof(unsigned long long), hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&h_perm, c * sizeof(uint32_t), hipHostMallocDefault));
        table_cap = t;
        sort_bytes = vds_sort_temp_bytes(static_cast<int>(c));
        HIPCHK(hipMalloc(&d_sort_temp, sort_bytes));
        HIPCHK(hipHostMalloc(&h_pin, 3 * c * sizeof(Point4), hipHostMallocDefault));
        pin_bytes = 3 * c * sizeof(Point4);
        cap = c;
        return SAGEICP_OK;
    }
    void free_points() {
        if (d_in) (void)hipFree(d_in);
        if (d_tmp) (void)hipFree(d_tmp);
        if (d_fd) (void)hipFree(d_fd);
        if (d_src) (void)hipFree(d_src);
        if (d_slot) (void)hipFree(d_slot);
        if (d_skey) (void)hipFree(d_skey);
        if (d_sval) (void)hipFree(d_sval);
        if (d_keys) (void)hipFree(d_keys);
        if (d_winner) (void)hipFree(d_winner);
        if (d_okeys) (void)hipFree(d_okeys);
        if (d_perm) (void)hipFree(d_perm);
        if (h_keys) (void)hipHostFree(h_keys);
        if (h_perm) (void)hipHostFree(h_perm);
        h_keys = nullptr; h_perm = nullptr;
        d_okeys = nullptr; d_perm = nullptr;
        if (d_sort_temp) (void)hipFree(d_sort_temp);
        if (h_pin) (void)hipHostFree(h_pin);
        d_in = d_tmp = d_fd = d_src = nullptr;
        d_slot = d_skey = d_sval = d_winner = nullptr;
        d_keys = nullptr; d_sort_temp = nullptr; h_pin = nullptr;
        cap = 0;
    }
    void destroy() {
        if (!stream) return;
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(stream);
        free_points();
        if (d_nkept) (void)hipFree(d_nkept);
        if (d_overflow) (void)hipFree(d_overflow);
        if (d_gcounts) (void)hipFree(d_gcounts);
        if (d_glabels) (void)hipFree(d_glabels);
        (void)hipStreamDestroy(stream);
        *this = Prep();
    }

    // levels: each {do_crop, scale}; a scale <= 0 means "crop only" (no voxel test).  Runs the
    // levels in sequence on the device, each feeding the next, and returns every level's cloud.
    int run(const double *frame, uint64_t n, double max_range, double min_range,
            double label_max_range, int n_groups, const int *gcounts, const int *glabels,
            const double *gvs, const int *crop, const double *scales, int n_levels,
            std::vector<std::vector<double>> &out, bool download = true) {
        kept_levels[0] = kept_levels[1] = 0;
        us_order = 0;
        if (n > kMaxQueries) return fail(SAGEICP_ERR_INVALID, "frame too large (2^26 - 4 points max)");
        if (n_groups > 8) return fail(SAGEICP_ERR_INVALID, "at most 8 label groups");
        size_t nlabels = 0;
        for (int g = 0; g < n_groups; ++g) nlabels += static_cast<size_t>(gcounts[g]);
        int rc = reserve(n, nlabels);
        if (rc) return rc;
        HIPCHK(hipSetDevice(device));
        out.assign(n_levels, std::vector<double>());
        if (n == 0) return SAGEICP_OK;
        if (n_groups > 0) {
            HIPCHK(hipMemcpyAsync(d_gcounts, gcounts, n_groups * sizeof(int), hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemcpyAsync(d_glabels, glabels, nlabels * sizeof(int), hipMemcpyHostToDevice, stream));
        }
        HIPCHK(hipMemsetAsync(d_overflow, 0, sizeof(int), stream));
        std::memcpy(h_pin, frame, n * sizeof(Point4));
        HIPCHK(hipMemcpyAsync(d_in, h_pin, n * sizeof(Point4), hipMemcpyHostToDevice, stream));
        const Point4 *in = d_in;
        Point4 *outs[2] = {d_fd, d_src};
        uint64_t cur = n;
        for (int l = 0; l < n_levels; ++l) {
            VdsParams P{};
            P.in = in; P.n = static_cast<int>(cur); P.do_crop = crop[l];
            P.max_range = max_range; P.min_range = min_range; P.label_max_range = label_max_range;
            P.n_groups = scales[l] > 0.0 ? n_groups : -1;
            P.group_counts = d_gcounts; P.group_labels = d_glabels;
            for (int g = 0; g < n_groups; ++g) P.group_vs[g] = gvs[g];
            P.scale = scales[l];
            P.keys = d_keys; P.winner = d_winner; P.mask = table_cap - 1;
            P.tmp = d_tmp; P.slot_of = d_slot; P.sort_key = d_skey; P.sort_val = d_sval;
            P.overflow = d_overflow;
            const bool reorder = g_reference_order && P.n_groups > 0 && !((arrival_order_levels >> l) & 1u);
            P.out_keys = reorder ? d_okeys : nullptr;
            Point4 *dst = outs[l & 1];
            HIPCHK(voxel_downsample_device(P, d_sort_temp, sort_bytes, d_nkept + (l & 1), dst, stream));
            uint32_t kept = 0;
            HIPCHK(hipMemcpyAsync(&kept, d_nkept + (l & 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            kept_levels[l & 1] = kept;
            if (reorder && kept) {
                // the reference's emission order (Preprocessing.cpp:76-82): replay, group by
                // group, the insertions into its robin_map and permute the survivors
                const double t0 = now_us();
                HIPCHK(hipMemcpyAsync(h_keys, d_okeys, kept * sizeof(unsigned long long),
                                      hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                const double t1 = now_us();
                h_hash.resize(kept);
                // survivors are grouped (stable sort by group); the groups' tables are independent:
                // one host thread per group hashes and replays its run and writes its part of the
                // permutation in place
                std::vector<std::pair<uint32_t, uint32_t>> runs;
                for (uint32_t a = 0; a < kept;) {
                    const unsigned long long g = h_keys[a] >> 60;
                    uint32_t lo = a, hi = kept;            // first index of another group (binary search: the runs are long)
                    while (hi - lo > 1) {
                        const uint32_t mid = lo + (hi - lo) / 2;
                        if ((h_keys[mid] >> 60) == g) lo = mid; else hi = mid;
                    }
                    runs.emplace_back(a, hi);
                    a = hi;
                }
                auto replay = [&](size_t r) {
                    const uint32_t a = runs[r].first, b = runs[r].second;
                    for (uint32_t i = a; i < b; ++i) h_hash[i] = static_cast<uint32_t>(h_keys[i] & 0xFFFFFu);   // hashed on the device
                    std::vector<uint32_t> part;
                    part.reserve(b - a);
                    if (!RobinOrderReplay::iteration_order(h_hash.data() + a, b - a, a, part, &rscratch[r & 7])) {
                        // a probe distance the replay does not model (robin_order.hpp): this group keeps
                        // its arrival order — said once, loudly, because the poses of a stream then
                        // differ from the reference's by centimetres
                        static std::atomic<bool> told{false};
                        if (!told.exchange(true))
                            std::fprintf(stderr, "sageicp: VoxelDownsample: a label group of %u voxels exceeds the probe "
                                                 "distance the tsl::robin_map replay models; it is emitted in arrival order\n",
                                         b - a);
                        part.resize(b - a);
                        for (uint32_t i = a; i < b; ++i) part[i - a] = i;
                    }
                    std::memcpy(h_perm + a, part.data(), (b - a) * sizeof(uint32_t));
                };
                // The groups' replays are independent and the largest (half of the survivors on street
                // scenes) is the critical path: every group gets its own thread — parked helpers of
                // this Prep, woken per level (starting threads costs what a small replay does) —
                // largest first, the calling thread takes part.
                std::vector<size_t> by_size(runs.size());
                for (size_t r = 0; r < runs.size(); ++r) by_size[r] = r;
                std::sort(by_size.begin(), by_size.end(), [&](size_t x, size_t y) {
                    return runs[x].second - runs[x].first > runs[y].second - runs[y].first;
                });
                if (kept > 8192 && runs.size() > 1) {
                    if (!pool) pool.reset(new ReplayPool);
                    const size_t hw = std::max(1u, std::thread::hardware_concurrency());
                    pool->run(runs.size(), [&](size_t k) { replay(by_size[k]); },
                              std::min<size_t>(hw, static_cast<size_t>(std::max(1, env_int("SAGEICP_REPLAY_THREADS", 8)))));
                } else {
                    for (size_t r = 0; r < runs.size(); ++r) replay(r);
                }
                const double t2 = now_us();
                HIPCHK(hipMemcpyAsync(d_perm, h_perm, kept * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                launch_vds_permute(dst, d_perm, kept, d_tmp, stream);
                HIPCHK(hipMemcpyAsync(dst, d_tmp, kept * sizeof(Point4), hipMemcpyDeviceToDevice, stream));
                us_order += now_us() - t0;
                if (env_int("SAGEICP_DEBUG_ORDER", 0)) {
                    std::string rs;
                    for (auto &r : runs) rs += " " + std::to_string(r.second - r.first);
                    std::fprintf(stderr, "order level %d: kept %u, fetch keys %.0f us, replay %.0f us (runs:%s), rest %.0f us\n",
                                 l, kept, t1 - t0, t2 - t1, rs.c_str(), now_us() - t2);
                }
            }
            if (download) {       // otherwise the level's cloud stays in d_fd / d_src for the caller
                char *hp = static_cast<char *>(h_pin) + static_cast<size_t>(1 + (l & 1)) * cap * sizeof(Point4);
                if (kept) HIPCHK(hipMemcpyAsync(hp, dst, kept * sizeof(Point4), hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                out[l].resize(4 * static_cast<size_t>(kept));
                if (kept) std::memcpy(out[l].data(), hp, kept * sizeof(Point4));
            }
            in = dst;
            cur = kept;
        }
        int ovf = 0;
        HIPCHK(hipMemcpy(&ovf, d_overflow, sizeof(int), hipMemcpyDeviceToHost));
        if (ovf & 2) return fail(SAGEICP_ERR_INVALID, "a label (or, without the range crop, a coordinate) is not finite (NaN / Inf)");
        if (ovf) return fail(SAGEICP_ERR_CAPACITY, "voxel index beyond +-2^19 in VoxelDownsample");
        return SAGEICP_OK;
    }
};

}  // namespace sageicp

using namespace sageicp;

// ---- opaque handles -----------------------------------------------------------------------
struct sageicp_map {
    HostMap host;
    int device = 0;
    // device mirror + scratch: logically a cache of `host`, refreshed lazily by const searches
    mutable Scratch sc;
    mutable Slot *d_table = nullptr;
    mutable size_t d_table_cap = 0;      // slots
    mutable Point4 *d_pts = nullptr;
    mutable size_t d_units_cap = 0;      // units (4 points) the point array holds
    mutable size_t d_blocks_cap = 0;     // blocks the per-block arrays (d_regions, and the update's aux arrays) hold
    mutable uint32_t *d_regions = nullptr;              // per block: (class << 28) | first unit of its region
    mutable size_t d_regions_cap = 0;
    mutable uint32_t *d_free_units[kMaxClasses] = {};   // device-side update: per-class stacks of free regions
    mutable size_t d_free_units_cap[kMaxClasses] = {};
    mutable uint32_t *d_freed = nullptr;                // regions released by one insertion pass
    mutable size_t d_freed_cap = 0;
    mutable uint32_t *d_block_of = nullptr;             // device-side update: unit -> block (slot words carry units)
    mutable size_t d_block_of_cap = 0;
    mutable bool mirror_stale_all = true;
    // compact copy of d_pts for k_icp's scan (fp32 x, y, z, label), derived on the device whenever
    // the HBM copy of the map has changed since the last search
    mutable uint4 *d_cand = nullptr;
    mutable size_t d_cand_slots = 0;     // point slots it holds
    mutable uint32_t *d_cand_flags = nullptr;
    mutable bool cand_stale = true;
    // pinned staging + device landing buffers for the scattered refresh of changed records
    mutable void *h_stage = nullptr;
    mutable void *d_stage = nullptr;
    mutable size_t stage_bytes = 0;
    // Device-side Update() (map_update.hip).  After one the HBM copy is the authority
    // (`on_device`) and `host` is stale until ensure_host() downloads it; `ctr` is the host's
    // shadow of the device counters.  The auxiliary arrays are valid for the host generation
    // they were uploaded at.
    mutable bool on_device = false;
    mutable uint8_t *d_zeros = nullptr;
    mutable uint32_t *d_slot_of = nullptr;
    mutable uint32_t *d_free = nullptr;
    mutable MapCounters *d_ctr = nullptr;
    mutable MapCounters *h_ctr = nullptr;       // pinned
    mutable uint32_t *h_ctr_aux = nullptr;      // pinned, 16 words: what else a device update hands back (far voxels found)
    // reference-order maps: the lists a device update exchanges with the host's bucket array (pinned)
    mutable uint2 *h_lists = nullptr;
    mutable size_t h_lists_cap = 0;
    int reserve_lists(size_t n) const {
        if (n + 2 <= h_lists_cap) return SAGEICP_OK;
        if (h_lists) HIPCHK(hipHostFree(h_lists));
        h_lists = nullptr; h_lists_cap = 0;
        const size_t c = n + n / 2 + 4096;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h_lists), c * sizeof(uint2), hipHostMallocDefault));
        h_lists_cap = c;
        return SAGEICP_OK;
    }
    mutable size_t d_aux_cap = 0;               // blocks the auxiliary arrays hold
    mutable bool aux_valid = false;
    mutable uint64_t aux_generation = 0;
    mutable MapCounters ctr{};
    mutable UpdateScratch up{};
    mutable size_t up_n = 0, up_nb = 0;
    // Pointcloud() served from the HBM copy: the packed points before they cross PCIe
    mutable Point4 *d_pc = nullptr;
    mutable size_t d_pc_cap = 0;
    // Single-process multi-GPU mode (SAGEICP_DEVICES / sageicp_map_set_devices): one more complete
    // copy of the map per extra device.  Every mutation is applied to all of them, RegisterFrame
    // shards the frame over them (one host thread and one stream per device) and the ranks'
    // Gauss-Newton sums meet in peer-mapped exchange blocks.  `this` is rank 0.
    std::vector<sageicp_map *> replicas;
    mutable std::vector<struct sageicp_comm *> ranks;   // created at the first sharded registration
    // a mutation reached some copies of the map but not all (a device ran out of memory, ...): the
    // ranks would sum Gauss-Newton terms computed against different maps, so every later entry
    // refuses the handle until Clear() has emptied all copies
    bool replicas_diverged = false;
};

struct sageicp_frame {
    int device = 0;
    Point4 *d = nullptr;
    uint64_t n = 0;
};

struct sageicp_comm {
    ncclComm_t comm = nullptr;           // RCCL (may be absent when only the direct exchange is used)
    int rank = 0, nranks = 1, device = 0;
    // direct exchange of the sums over xGMI (P2pBlock, sageicp_types.h)
    bool p2p = false;
    bool poisoned = false;               // an exchange timed out: the ranks' exchange counters may
                                         // differ, so the blocks must not be used again
    P2pBlock *my_block = nullptr;        // fine-grained device memory, exported through HIP IPC
    P2pBlock *blocks[kMaxRanks] = {};    // every rank's block as mapped here (blocks[rank] == my_block)
    unsigned long long *d_exchanges = nullptr;
    bool peer_mapped = false;            // blocks[] are plain peer pointers of this process (no IPC handles to close)
    bool device_shared = false;          // several ranks of ONE process run on this device (tests on a 1-GPU box): their
                                         // streams share the process's few hardware queues, where a solving wave that
                                         // waits for its peer can sit in front of that very peer's grid — such ranks
                                         // stay with the launch-per-iteration loop
};

// ---- the library's translation units call each other through these --------------------------------------
// capi.hip: the C ABI.  capi_mirror.hip: the HBM mirror of a map and Update() on the device.  capi_run.hip: the ICP
// loop (plan_loop, run_icp), the RCCL binding and the single-process multi-GPU mode.
namespace sageicp_impl {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;        // optional: what RCCL itself reports
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
};
extern Rccl g_rccl;
int load_rccl();
// capi_mirror.hip
int reserve_device_points(const sageicp_map *m, size_t units, size_t keep);
int sync_mirror(const sageicp_map *m);
int ensure_cand(const sageicp_map *m, bool derive = true);
bool map_is_empty(const sageicp_map *m);
int ensure_host(const sageicp_map *m);
int reserve_update_scratch(const sageicp_map *m, size_t n, size_t nb);
int grow_device_blocks(const sageicp_map *m, size_t blocks, size_t keep);
int reserve_unit_stacks(const sageicp_map *m, size_t n);
DevMap dev_map(const sageicp_map *m);
int device_update(sageicp_map *m, const double *xyzl, uint64_t n, const double pose[7], const Point4 *d_points = nullptr);
bool all_finite(const double *xyzl, uint64_t n);
// capi_run.hip
void identity_pose(double T[7]);
void fill_state(IcpState *st, const double init[7]);
bool sparse_voxels(const sageicp_map *m);
bool wants_filter(const sageicp_map *m, uint64_t n, double sem_th);
IcpParams icp_params(const sageicp_map *m, const Point4 *d_queries, uint64_t n, double sem_th, int lw);
int run_icp(const sageicp_map *m, const Point4 *d_frame, uint64_t n, const double init[7], double max_dist, double kernel,
            double sem_th, sageicp_comm *comm, double out[7], sageicp_stats *stats, double us_upload, double t_begin);
int device_update_all(sageicp_map *m, const double *xyzl, uint64_t n, const double pose[7], const Point4 *d_points);
int register_sharded(const sageicp_map *m, const double *h_frame, const Point4 *d_frame, uint64_t n, const double init[7],
                     double max_dist, double kernel, double sem_th, double pose_out[7], sageicp_stats *stats, double t0);
}  // namespace sageicp_impl
using namespace sageicp_impl;
