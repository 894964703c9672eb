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
static thread_local std::string g_err;
static int g_profiling = 0;
static int g_counting = 1;                // sageicp_set_counting: the per-wave candidate / pair counters behind sageicp_stats
// VoxelDownsample emits its survivors in the reference's order (the bucket order of its
// tsl::robin_map, replayed on the host: robin_order.hpp) unless switched to arrival order
static int g_reference_order = 1;

// tuning knobs (defaults chosen by measurement on MI355X; the environment overrides are for
// experiments only)
static int env_int(const char *name, int dflt) {
    const char *v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}
// Lanes per query in k_icp (log2).  One lane per query needs the fewest instructions per query
// but gives a frame of n points only n / 64 waves with long dependent chains; small frames and
// shards spread each query over more lanes.  Thresholds measured on MI355X (profiles/README.md);
// SAGEICP_LW overrides for experiments.
static int icp_lw(uint64_t n, bool sparse_voxels) {
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


static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return fail(SAGEICP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

static double now_us() {
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
            if (const char *e = std::getenv("SAGEICP_CU_SHARE")) {
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
        if (n > kMaxQueries) return fail(SAGEICP_ERR_INVALID, "frame too large (2^26 points max)");
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

// ---- RCCL, bound at run time (only multi-GPU runs need it) ------------------------------------
namespace {
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
Rccl g_rccl;
std::mutex g_rccl_mu;

int device_update_all(sageicp_map *m, const double *xyzl, uint64_t n, const double pose[7],
                      const Point4 *d_points);
int register_sharded(const sageicp_map *m, const double *h_frame, const Point4 *d_frame, uint64_t n,
                     const double init[7], double max_dist, double kernel, double sem_th,
                     double pose_out[7], sageicp_stats *stats, double t0);

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
int ensure_cand(const sageicp_map *m, bool derive = true) {
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

static int reserve_update_scratch(const sageicp_map *m, size_t n, size_t nb) {
    UpdateScratch &u = m->up;
    if (n > m->up_n) {
        const size_t c = n + n / 2 + 1024;
        void *olds[] = {u.raw, u.w, u.keys, u.keys_alt, u.idx, u.idx_alt, u.head_slot, u.flag, u.rank, u.want};
        for (void *q : olds)
            if (q) HIPCHK(hipFree(q));
        u = UpdateScratch{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                          u.far_flag, u.far_sel, u.n_sel, u.temp, u.temp_bytes};
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
        m->up_n = c;
    }
    if (nb > m->up_nb) {
        const size_t c = nb + nb / 2 + 1024;
        if (u.far_flag) HIPCHK(hipFree(u.far_flag));
        if (u.far_sel) HIPCHK(hipFree(u.far_sel));
        u.far_flag = u.far_sel = nullptr;
        m->up_nb = 0;
        HIPCHK(hipMalloc(&u.far_flag, c * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&u.far_sel, c * sizeof(uint32_t)));
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
static int grow_device_blocks(const sageicp_map *m, size_t blocks, size_t keep) {
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
    }
    return SAGEICP_OK;
}

// the unit allocator's device arrays: per-class stacks able to hold every region the point array
// can be cut into, and the scratch list of one pass's released regions (at most one per point)
static int reserve_unit_stacks(const sageicp_map *m, size_t n) {
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

static DevMap dev_map(const sageicp_map *m) {
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
                  const Point4 *d_points = nullptr) {
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
    HIPCHK(map_update_device(dm, pol, us, static_cast<int>(n), pose, bound, s));
    HIPCHK(hipMemcpyAsync(m->h_ctr, m->d_ctr, sizeof(MapCounters), hipMemcpyDeviceToHost, s));
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
static bool all_finite(const double *xyzl, uint64_t n) {
    // (x - x is 0 for every finite x and NaN otherwise: four of them summed stay 0 exactly)
    double acc = 0.0;
    for (uint64_t i = 0; i < 4 * n; ++i) acc += xyzl[i] - xyzl[i];
    return acc == 0.0;
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
static bool sparse_voxels(const sageicp_map *m) {
    const uint64_t mp = m->on_device ? m->ctr.total_points : m->host.total_points;
    const uint64_t mv = m->on_device ? m->ctr.num_voxels : m->host.num_voxels;
    return mp < 6 * mv;
}

// Does the scan of `n` queries read the compact copy behind its fp32 filter?  Worth it where scans
// are long and bytes are what the kernel is made of: frames of 40k+ points against voxels holding
// 6+ points on average (c2: +6 %, c4: +10 %; c1, c5 and the streamed 24k-point frames lose 4-5 %
// with it; SAGEICP_FILTER=0/1 overrides).  Off for a negative or NaN sem_th, where a larger
// distance can scale to a smaller one and the filter's thresholds do not exist.
static bool wants_filter(const sageicp_map *m, uint64_t n, double sem_th) {
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
    ip.counters = nullptr;
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
    auto round32 = [](uint64_t w) { return std::max<uint64_t>(32, (w + 31) / 32 * 32); };   // (XCD stripes: 8 x kLoopStripe)
    // the accumulator words count their workgroups in 8 bits, and (digit << 8) summed over the blocks of four
    // queries of a copy's workgroups must stay inside 63 bits: |digit| < 2^40 per block (kernels.hip, to_digits)
    auto countable = [](uint64_t wgs, uint64_t gpw, int lw) {
        return wgs / 8 <= 255 && (wgs / 8) * gpw * ((64u >> lw) / 4u) <= 8192;
    };
    // one wave per group: nw groups per workgroup of nw waves
    auto one_pass = [&](int lw, int nw, LoopPlan *pl) {
        const uint64_t wgs = round32((groups_at(lw) + nw - 1) / nw);
        const size_t lds = loop_lds_bytes(lw, nw, nw);
        const int k = loop_wgs_per_cu(sc, lw, filter, nw, lds);
        if (k < 1 || wgs + 32ull * static_cast<uint64_t>(sc.loop_derate) > static_cast<uint64_t>(k) * cus * 15 / 16 || !countable(wgs, nw, lw)) return false;
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
                uint64_t cap = static_cast<uint64_t>(k) * cus * 15 / 16 / 32 * 32;
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
        const uint64_t resident_waves = 4ull * SAGE_LOOP_OCC * cus * 15 / 16;
        while (lw > 2 && groups_at(lw) > resident_waves) --lw;
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
    if (n > kMaxQueries) return fail(SAGEICP_ERR_INVALID, "frame too large (2^26 points max)");
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
    const int lw = loop_shape ? plan.lw : icp_lw(n, sparse_voxels(m));
    // a launch that timed out (its grid was not resident as a whole: the GPU is shared with other work)
    // cost 50 ms before the frame went through the other loop: the next calls do not try again
    bool use_loop = loop_shape;
    if (use_loop && sc.loop_cooldown > 0) {
        --sc.loop_cooldown;
        use_loop = false;
    }
    const unsigned loop_waves = use_loop ? static_cast<unsigned>((n + (64u >> plan.lw) - 1) / (64u >> plan.lw)) : 0u;
    const int blocks = n ? icp_blocks_for(static_cast<int>(n), lw) : 1;
    if ((rc = ensure_cand(m, wants_filter(m, n, sem_th)))) return rc;
    if ((rc = sc.reserve_sort(n))) return rc;
    if ((rc = sc.reserve_partials(static_cast<size_t>(blocks)))) return rc;
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
    if (use_loop && (rc = sc.loop_streams())) return rc;
    if (use_loop) {
        // ---- the whole loop in one launch (kernels.hip, k_loop): first its solving wave, on its own stream —
        // it has to hold its registers before the grid fills the machine; it waits for the grid's go
        L.sh = sc.d_loop;
        L.st = sc.d_state;
        L.nw = plan.nw;
        L.gpw = plan.gpw;
        L.wgs = plan.wgs;
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
        if (comm && env_int("SAGEICP_LOOP_TIMEOUT_TICKS", 0) == 0)
            L.timeout_ticks = std::max(L.timeout_ticks, xp.timeout_ticks + 100000000ull);
        L.max_iterations = max_it;
        L.epoch = ++sc.loop_epoch;
        for (int i = 0; i < 7; ++i) L.T0[i] = init[i];
        L.acc_unscale = 1.0 / ip.acc_scale;
        L.shared_loop = comm ? 1 : 0;
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
    if (n > 0)
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
        HIPCHK(hipMemsetAsync(sc.d_loop, 0, sizeof(LoopShared), s));
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
            sc.loop_cooldown = std::max(0, env_int("SAGEICP_LOOP_COOLDOWN", 256));
            if (env_int("SAGEICP_LOOP_TIMEOUT_TICKS", 0) == 0 && sc.loop_derate < 16) ++sc.loop_derate;
            if (comm) {
                // (the peers are somewhere inside their loops: there is no starting again in step)
                comm->p2p = false;
                comm->poisoned = true;
                return fail(SAGEICP_ERR_RCCL, "one-launch loop under a communicator: a wait inside the launch timed out "
                                              "(the GPU is shared with other work?); SAGEICP_LOOP=0 selects the launch-per-iteration loop");
            }
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
    auto enqueue_iteration = [&](int slot, int iteration) -> int {
        const bool ev = sampled(iteration);
        if (ev) HIPCHK(hipEventRecord(sc.events[5 * slot + 1], s));
        launch_icp(ip, lw, true, s);
        if (ev) HIPCHK(hipEventRecord(sc.events[5 * slot + 2], s));
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
            }
        }
        if (counting) launch_sum_counters(sc.d_cand, static_cast<int>(ip.nwaves), sc.d_state, s);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(sc.h_state, sc.d_state, sizeof(IcpState), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
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
    if (st.bad_input)
        return fail(SAGEICP_ERR_INVALID, "the frame holds a coordinate or label that is not finite (NaN / Inf)");
    if (st.acc_overflow && !comm && g_acc_shift < 2) {
        // |sum over four queries| >= 2^46 (2^40 in the one-launch loop): georeferenced coordinates (UTM: ~3e6 m,
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
    if (m->host.track_order) {
        // a reference-order map is maintained on the host (host_map.hpp: the bucket array of the
        // reference's robin_map is host state); the device mirror follows by dirty ranges
        std::vector<double> pts_host;
        if (d_points) {
            pts_host.resize(4 * n);
            HIPCHK(hipSetDevice(m->device));
            if (n) HIPCHK(hipMemcpy(pts_host.data(), d_points, n * sizeof(Point4), hipMemcpyDeviceToHost));
            xyzl = pts_host.data();
        }
        return sageicp_map_update_pose(m, xyzl, n, pose);
    }
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
            const int lw = icp_lw(cnt, sparse_voxels(mk));
            if ((r = ensure_cand(mk, wants_filter(mk, cnt, sem_th)))) return r;
            if ((r = sc.reserve_sort(cnt))) return r;
            if ((r = sc.reserve_partials(static_cast<size_t>(cnt ? icp_blocks_for(static_cast<int>(cnt), lw) : 1)))) return r;
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

}  // namespace

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
    if (const char *list = std::getenv("SAGEICP_DEVICES")) {
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
    HIPCHK(map_pointcloud_device(dm, m->ctr.blocks_hi, m->up.far_flag, m->up.far_sel, m->up.temp,
                                 m->up.temp_bytes, m->d_pc, s));
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
    if (n > kMaxQueries) return fail(SAGEICP_ERR_INVALID, "too many queries (2^26 max)");
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
