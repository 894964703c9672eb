/* sageicp.h — C ABI of libsageicp_hip.so: the MI355X (gfx950) implementation of SAGE-ICP's
 * per-scan registration hot path.  Plain pointers and sizes only; no Eigen/Sophus/torch types.
 *
 * Every entry point below is what a binding of the reference's C++ interface for this path
 * would call; the reference interface each one replaces is cited as file:line relative to
 * NeSC-IV/sage-icp @ 2024_10_08, directory cpp/sage_icp/.  INTEGRATION.md shows the header
 * shim (sage-icp_amd/shim/sage_icp/core/{VoxelHashMap,Registration}.hpp) that keeps the
 * reference's C++ signatures on top of this ABI so ros/ros2/OdometryServer.cpp and
 * pipeline/sageICP.cpp compile unchanged.
 *
 * Conventions
 *   points   double[n][4] row-major = (x, y, z, label); identical to the memory of
 *            std::vector<Eigen::Vector4d>::data().
 *   pose     double[7] = (qx, qy, qz, qw, tx, ty, tz); identical to Sophus::SE3d::data().
 *   return   0 on success, negative on error (never throws); sageicp_last_error() gives text.
 *   threads  one call at a time per map handle (the reference's single ROS executor thread,
 *            ros/ros2/OdometryServer.cpp:356).  Different handles may be used concurrently.
 *   device   the HIP path is the only path: without a gfx950 device every compute entry
 *            returns SAGEICP_ERR_NO_DEVICE.  There is no CPU fallback in this library.
 *   NaN/Inf  the reference turns coordinates into voxel indices and labels into classes with
 *            static_cast<int> (VoxelHashMap.cpp:52-54,87-88,165; Preprocessing.cpp:58-64): undefined
 *            for values that are not finite, so there is no behaviour to reproduce.  Every entry that
 *            would cast one — AddPoints / Update (host and device), GetCorrespondences, RegisterFrame
 *            (host buffer or resident frame), VoxelDownsample, the pipeline — refuses the WHOLE call
 *            with SAGEICP_ERR_INVALID and changes nothing.  Where the reference IS defined it is
 *            followed: Preprocess() drops a point whose norm is not finite (both range tests fail,
 *            Preprocessing.cpp:176-177) — so the pipeline drops such points like the reference and only
 *            refuses non-finite LABELS —, TransformPoints and AlignClouds propagate NaN.
 */
#ifndef SAGEICP_H_
#define SAGEICP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAGEICP_OK 0
#define SAGEICP_ERR_INVALID (-1)    /* bad argument */
#define SAGEICP_ERR_NO_DEVICE (-2)  /* no usable gfx950 device / HIP runtime failure at init */
#define SAGEICP_ERR_HIP (-3)        /* a HIP call failed */
#define SAGEICP_ERR_RCCL (-4)       /* RCCL could not be loaded or a collective failed */
#define SAGEICP_ERR_CAPACITY (-5)   /* > 2^24 voxels, > 255 points per voxel, > 2^26 point slots, or a voxel
                                     * index beyond +-2^20 */

typedef struct sageicp_map sageicp_map;       /* opaque: host map + device mirror + scratch */
typedef struct sageicp_frame sageicp_frame;   /* opaque: a scan resident in HBM */
typedef struct sageicp_comm sageicp_comm;     /* opaque: RCCL communicator for query sharding */

/* Version of this header's ABI; sageicp_abi_version() returns the library's.  A binding must refuse
 * a library whose version differs (the header shim and the Python loader do): struct layouts
 * below are part of it.   2: sageicp_stats without the three fields that had become constant
 * zeros (us_group, us_gn, resorts), with pairs_evaluated / lanes_per_query / compact_scan;
 * capacity limits 2^24 voxels / 2^26 point slots (sageicp_map_point_slots) / |voxel index| < 2^20 (SAGEICP_ERR_CAPACITY);
 * sageicp_comm_describe, sageicp_map_pointcloud served from the HBM copy, sageicp_map_point_slots
 * (size-classed voxel storage).   3: sageicp_stats names its loop form (single_launch in the slot of
 * reserved0), sageicp_pipeline_prefetch_wait, non-finite input refused (SAGEICP_ERR_INVALID) at every entry
 * that would cast it.   4: sageicp_map_loop_status, sageicp_reload_env (additions only). */
#define SAGEICP_ABI_VERSION 4

/* Filled by sageicp_register_frame*.  Times are microseconds. */
typedef struct sageicp_stats {
    int32_t iterations;         /* ICP iterations executed (<= 500, Registration.cpp:96) */
    int32_t converged;          /* 1 if ||log(est)|| < 1e-4 ended the loop (Registration.cpp:137) */
    uint64_t n_queries;         /* points of the frame (this rank's shard) */
    uint64_t n_corr_first;      /* accepted correspondences, first iteration (all ranks) */
    uint64_t n_corr_last;       /* accepted correspondences, last iteration (all ranks) */
    double last_step_norm;      /* ||log(est)|| of the last iteration */
    double us_wall;             /* host wall time of the call */
    double us_upload;           /* frame H2D + lazy map-mirror refresh inside the call */
    /* device time per kernel summed over the executed iterations (HIP events on the launch
     * stream); filled only when profiling is enabled with sageicp_set_profiling(). */
    double us_nn;               /* k_icp: pose apply + correspondence search + Gauss-Newton sums */
    double us_fin;              /* k_fin: reduction of the partials, solve, pose update */
    uint32_t nn_launches;       /* k_icp launches that were timed (us_nn / nn_launches = mean duration) */
    uint32_t single_launch;     /* 1: the whole loop ran inside one launch (k_loop: a frame that fits the machine, one GPU) */
    uint64_t sum_candidates;    /* sum over iterations and queries of C_q: map points stored in the
                                 * <=27 existing neighbour voxels of each query (this rank) */
    uint32_t n_corr_hist[64];   /* accepted correspondences of the first 64 iterations */
    uint64_t pairs_evaluated;   /* (query, map point) pairs the search actually evaluated, all
                                 * iterations: sum_candidates minus what the cell lower bound pruned */
    uint32_t lanes_per_query;   /* lanes that shared one query in the search kernel (1..16) */
    uint32_t compact_scan;      /* 1: the search scanned the 16-B compact copy of the map behind its fp32
                                 * filter (big frames, dense voxels); 0: the full fp64 records */
} sageicp_stats;

/* ---- library ------------------------------------------------------------------------ */
int sageicp_abi_version(void);
const char *sageicp_last_error(void);
int sageicp_device_count(void);               /* number of visible HIP devices (0: none) */
void sageicp_set_profiling(int level);        /* 0 off; 1 HIP events around k_icp in one iteration
                                               * out of 8; 2 around every kernel of every iteration */
void sageicp_reload_env(void);                /* the SAGEICP_* tuning knobs are read from the environment once per process (at
                                               * first use); this makes the next calls read them again.  Not while calls of
                                               * other threads are in flight. */
void sageicp_set_counting(int on);            /* 1 (default): a call given a sageicp_stats counts C_q and the pairs it
                                               * evaluates (sum_candidates, pairs_evaluated: ~3 % of the search);
                                               * 0: those two fields stay zero, the others are filled as before.
                                               * A call without stats (the C++ shim's) never counts. */

/* ---- map: sage_icp::VoxelHashMap (core/VoxelHashMap.hpp:35-107) ------------------------ */
/* ctor, VoxelHashMap.hpp:79-88.  device: HIP device ordinal that will hold the mirror. */
sageicp_map *sageicp_map_create(double voxel_size, double max_distance,
                                int basic_points_per_voxel, int critical_points_per_voxel,
                                const int *basic_parts_labels, int n_labels, int device);
void sageicp_map_destroy(sageicp_map *map);
/* Single-process multi-GPU mode: the map spans `n` (1..8) devices — a full copy per device, every
 * mutation applied to all of them — and sageicp_register_frame[_resident] shards the frame over
 * them (contiguous blocks, one host thread and stream per device; the Gauss-Newton sums meet in
 * peer-mapped exchange blocks), so the unchanged caller of sage_icp::RegisterFrame() — the ROS2
 * node's single executor thread, ros/ros2/OdometryServer.cpp:104,356 — uses several GPUs.
 * devices[0] is rank 0 (where a resident frame and the pipeline's buffers live).  The
 * environment variable SAGEICP_DEVICES=0,1,2,3 applies the same to every map the process creates. */
int sageicp_map_set_devices(sageicp_map *map, const int *devices, int n);
int sageicp_map_num_devices(const sageicp_map *map);
/* Reference-order mode (off by default; the map must be empty when it is switched).  The search never
 * depends on the iteration order of the reference's tsl::robin_map (core/VoxelHashMap.hpp:106), but two
 * things a caller can observe do: RemovePointsFarFromLocation erases WHILE iterating the container
 * (VoxelHashMap.cpp:176-184), so the voxel that the backward-shift deletion moves into the bucket just
 * erased is skipped and survives until a later frame, and Pointcloud() lists the voxels in bucket order
 * (:132-142).  By default this library removes every far voxel and lists block-pool order.  With the
 * mode on, the map also keeps the bucket array the reference's container would have (robin_map v1.0.1:
 * growth from zero buckets at load 0.5, robin-hood displacement, backward-shift deletion, clear() keeps
 * the array, the reference's 20-bit VoxelHash) and both behaviours are the reference's.  Such a map is
 * maintained on the host (sageicp_map_update_pose_device and the pipeline fall back to the host update:
 * ~3 ms instead of ~0.3 ms per streamed frame).  SAGEICP_MAP_REFERENCE_ORDER=1 applies the mode to every
 * map the process creates (for callers behind the header shim).
 * sageicp_map_reference_order: 0 off; 1 on; -1 on, but an insertion met a probe distance the emulation
 * does not model (>= 128: not before a map holds several 10^5 voxels, where the reference's own table
 * degenerates with its 20-bit hash) — the map works, its order is no longer claimed to be the
 * reference's. */
int sageicp_map_set_reference_order(sageicp_map *map, int on);
int sageicp_map_reference_order(const sageicp_map *map);
/* copy construction / copy assignment (ros/ros2/OdometryServer.cpp:104 copy-assigns the pipeline) */
sageicp_map *sageicp_map_clone(const sageicp_map *map);
int sageicp_map_clear(sageicp_map *map);                    /* Clear(), VoxelHashMap.hpp:93 */
int sageicp_map_empty(const sageicp_map *map);              /* Empty(), VoxelHashMap.hpp:94 */
uint64_t sageicp_map_size(const sageicp_map *map);          /* points held (== Pointcloud().size()) */
uint64_t sageicp_map_num_voxels(const sageicp_map *map);
/* AddPoints, VoxelHashMap.cpp:162-174 (sequential semantic retention policy, .hpp:45-70) */
int sageicp_map_add_points(sageicp_map *map, const double *xyzl, uint64_t n);
/* RemovePointsFarFromLocation, VoxelHashMap.cpp:176-184 */
int sageicp_map_remove_far(sageicp_map *map, const double origin[3]);
/* Update(points, origin), VoxelHashMap.cpp:144-147 */
int sageicp_map_update(sageicp_map *map, const double *xyzl, uint64_t n, const double origin[3]);
/* Update(points, pose), VoxelHashMap.cpp:149-160 */
int sageicp_map_update_pose(sageicp_map *map, const double *xyzl, uint64_t n,
                            const double pose[7]);
/* The same Update(points, pose) executed on the GPU against the HBM-resident map (stable sort by
 * voxel, one lane per voxel run applying VoxelBlock::AddPoint's policy in arrival order, eviction
 * by first-point distance).  Same result as the host entry: identical voxel blocks, free list and
 * counts.  Afterwards the HBM copy is the authority; host-side entries (AddPoints, Pointcloud,
 * clone ...) download it first.  SAGEICP_ERR_CAPACITY if a voxel index exceeds +-2^20. */
int sageicp_map_update_pose_device(sageicp_map *map, const double *xyzl, uint64_t n,
                                   const double pose[7]);
/* Pointcloud(), VoxelHashMap.cpp:132-142.  Returns the number of points the map holds; writes
 * at most `cap` of them (block-pool order; a map in reference-order mode: the reference's bucket order).  While
 * the HBM copy is the authority (after a device-side update) the points are packed on the device
 * and only they cross PCIe: the map stays resident, the next RegisterFrame uploads nothing.
 * `out_xyzl` is typically a fresh allocation (the reference's Pointcloud() returns a new vector):
 * its pages are populated by SAGEICP_TOUCH_THREADS (environment, default 4, 0: off) parked host
 * threads before the copy, which otherwise spends two thirds of its time in first-touch faults. */
uint64_t sageicp_map_pointcloud(const sageicp_map *map, double *out_xyzl, uint64_t cap);
/* 1 while the HBM copy of the map is the authority (device-side updates; Pointcloud() and
 * RegisterFrame keep it so), 0 while the host copy is (AddPoints / Update on the host, Clear). */
int sageicp_map_resident(const sageicp_map *map);
/* Point slots (32 B each) the map's voxel storage occupies, free regions included — the footprint of
 * the point array in HBM and on the host.  Voxels live in size-classed regions (4 / 8 / 16 points,
 * then max_points_per_voxel): a voxel starts in the smallest and moves up when it fills, so a map
 * of sparsely filled voxels does not pay max_points_per_voxel slots for each.  SAGEICP_SIZE_CLASSES=0
 * in the environment when the map is created gives every voxel a full-size region instead. */
uint64_t sageicp_map_point_slots(const sageicp_map *map);
/* Push pending host-side changes to the HBM mirror now (otherwise done lazily by the next
 * search).  Lets a caller keep the refresh out of a timed region. */
int sageicp_map_sync(const sageicp_map *map);

/* ---- search: VoxelHashMap::GetCorrespondences (core/VoxelHashMap.cpp:48-130) ------------ */
/* src_out / tgt_out: capacity n*4 doubles each; pairs are emitted in query order.
 * query_idx_out (optional, capacity n): index of the query behind each pair. */
int sageicp_get_correspondences(const sageicp_map *map, const double *q_xyzl, uint64_t n,
                                double max_correspondence_distance, double sem_th,
                                double *src_out, double *tgt_out, uint64_t *n_out,
                                int64_t *query_idx_out);

/* ---- Gauss-Newton step: AlignClouds (core/Registration.cpp:59-94) ----------------------- */
/* JTJ_out (36, row-major) and JTr_out (6) are optional.  device: HIP device ordinal. */
int sageicp_align_clouds(const double *src_xyzl, const double *tgt_xyzl, uint64_t n, double kernel,
                         double pose_out[7], double *JTJ_out, double *JTr_out, int device);

/* ---- TransformPoints (core/Registration.hpp:32, Registration.cpp:103-111) ---------------- */
int sageicp_transform_points(const double pose[7], double *xyzl, uint64_t n, int device);

/* ---- RegisterFrame (core/Registration.hpp:34-39, Registration.cpp:113-141) --------------- */
int sageicp_register_frame(const sageicp_map *map, const double *frame_xyzl, uint64_t n,
                           const double initial_guess[7], double max_correspondence_distance,
                           double kernel, double sem_th, double pose_out[7],
                           sageicp_stats *stats /* optional */);

/* The same call on a scan already resident in HBM (what bench.py times), optionally sharded
 * over the ranks of `comm` (NULL: single GPU).  Each rank passes ITS contiguous block of the
 * frame; the map is replicated; the 17 Gauss-Newton sums are all-reduced over RCCL/xGMI every
 * iteration, so every rank returns the same pose. */
sageicp_frame *sageicp_frame_upload(const sageicp_map *map, const double *frame_xyzl, uint64_t n);
void sageicp_frame_destroy(sageicp_frame *frame);
int sageicp_register_frame_resident(const sageicp_map *map, const sageicp_frame *frame,
                                    const double initial_guess[7],
                                    double max_correspondence_distance, double kernel,
                                    double sem_th, sageicp_comm *comm /* optional */,
                                    double pose_out[7], sageicp_stats *stats /* optional */);

/* Which form of the loop the calls of a map handle take, and why.  A frame that fits the machine's LDS runs its whole loop
 * in ONE launch (Registration.cpp:127-138 without a kernel boundary); that launch needs every workgroup resident at once,
 * and when a wait inside it times out (another tenant on the GPU, a driver that places fewer workgroups) the frame is
 * registered through the launch-per-iteration form instead, the handle stays away from the one-launch form for
 * `cooldown_calls` calls and plans `derate_workgroups` fewer workgroups from then on.  The first such event of a process
 * also writes one line to stderr.  Poses are the same to the bit in either form. */
#define SAGEICP_LOOP_FALLBACK_NONE 0
#define SAGEICP_LOOP_FALLBACK_TIMEOUT 1          /* a wait inside the one launch timed out: the grid was not resident as a whole */
#define SAGEICP_LOOP_FALLBACK_COOLDOWN 2         /* the call fell into the cool-down after such a time-out */
#define SAGEICP_LOOP_FALLBACK_DOES_NOT_FIT 3     /* the frame needs more LDS than the machine has (~170k points), or the form is
                                                  * switched off (SAGEICP_LOOP=0), or the call runs under an RCCL communicator */
#define SAGEICP_LOOP_FALLBACK_PEER 4             /* multi-GPU: another rank's one-launch loop timed out; every rank left the same
                                                  * exchange with it and registered the frame again, in step */
typedef struct sageicp_loop_status {
    uint64_t calls_single_launch;       /* registrations of this handle that ran in one launch */
    uint64_t calls_per_iteration;       /* ... through the launch-per-iteration form */
    uint64_t calls_chained;             /* ... of those, with the launches chained: no k_fin between them, the solving wave of the
                                         * one-launch form resident beside them (frames beyond the LDS on one GPU) */
    uint32_t timeouts;                  /* launches that gave up (each cost its time-out, 50 ms by default, before the fall-back) */
    uint32_t cooldown_calls;            /* calls that will still stay away from the one-launch form */
    uint32_t derate_workgroups;         /* workgroups taken off every later plan of this handle (32 per time-out, at most 512) */
    int32_t last_fallback;              /* SAGEICP_LOOP_FALLBACK_*: why the LAST call did not run in one launch (NONE: it did) */
} sageicp_loop_status;
int sageicp_map_loop_status(const sageicp_map *map, sageicp_loop_status *out);

/* ---- query sharding across GPUs (one process per GPU, RCCL over xGMI) -------------------- */
#define SAGEICP_UNIQUE_ID_BYTES 128
int sageicp_comm_unique_id(uint8_t id_out[SAGEICP_UNIQUE_ID_BYTES]);   /* rank 0, then broadcast */
sageicp_comm *sageicp_comm_create(const uint8_t id[SAGEICP_UNIQUE_ID_BYTES], int rank, int nranks,
                                  int device);
void sageicp_comm_destroy(sageicp_comm *comm);

/* Direct exchange of the 20 Gauss-Newton sums between the GPUs of one node, without a collective
 * launch: every rank exports a small block of fine-grained device memory through HIP IPC, maps its
 * peers' blocks, and the workgroup that finishes an iteration stores its sums into every block
 * over xGMI, waits for the others' and solves (replaces ncclAllReduce + a second launch; <= 8
 * ranks).  Usage: export on every rank, all-gather the handles (rank order), connect.  A
 * communicator made by sageicp_comm_create_local() has no RCCL side and only works connected. */
#define SAGEICP_P2P_HANDLE_BYTES 64
sageicp_comm *sageicp_comm_create_local(int rank, int nranks, int device);
int sageicp_comm_p2p_export(sageicp_comm *comm, uint8_t handle_out[SAGEICP_P2P_HANDLE_BYTES]);
int sageicp_comm_p2p_connect(sageicp_comm *comm, const uint8_t *handles /* nranks x 64 B */);
int sageicp_comm_p2p_enable(sageicp_comm *comm, int on);   /* fall back to RCCL with on = 0 */
int sageicp_comm_p2p_enabled(const sageicp_comm *comm);
/* What a communicator is made of, for logs and the bench line: the ranks as given at creation,
 * the ranks RCCL itself reports (ncclCommCount / ncclCommUserRank; -1 without an RCCL side), and the
 * state of the direct exchange. */
typedef struct sageicp_comm_info {
    int32_t rank, nranks, device;
    int32_t has_rccl;        /* 1: created by sageicp_comm_create (an ncclComm_t exists) */
    int32_t rccl_ranks;      /* ncclCommCount, -1: no RCCL side */
    int32_t rccl_rank;       /* ncclCommUserRank, -1: no RCCL side */
    int32_t p2p_connected;   /* every peer's exchange block is mapped */
    int32_t p2p_enabled;     /* the next registration uses the direct exchange */
    int32_t p2p_poisoned;    /* an exchange timed out: the blocks must not be used again */
    int32_t reserved[7];
} sageicp_comm_info;
int sageicp_comm_describe(const sageicp_comm *comm, sageicp_comm_info *out);

/* ---- Preprocess / VoxelDownsample (core/Preprocessing.hpp:33-45) on the device ------------------
 * sageicp_preprocess: core/Preprocessing.cpp:173-187 (dynamic_vehicle_filter == false): keep points
 * with min_range < |p| < max_range, zero the label beyond label_max_range; order preserved.
 * sageicp_voxel_downsample: core/Preprocessing.cpp:44-84: per label group g, the first point that
 * falls into a voxel of size group_voxel_size[g] * vox_scale is kept; points whose label is in no
 * group are dropped.  Output order: see sageicp_set_downsample_order (default: the reference's).
 * out: capacity n*4 doubles.  At most 8 groups; voxel indices must fit +-2^19. */
/* Emission order of sageicp_voxel_downsample and of the pipeline's two down-sampling levels:
 * 1 (default) = the reference's — the bucket order of the tsl::robin_map v1.0.1 its VoxelDownsample
 * iterates (Preprocessing.cpp:76-82), replayed on the host from the survivors' voxel keys, so that
 * the registered cloud and the map get exactly the points the reference's would; 0 = group by group
 * in input order (no host step, ~1 ms less per 120k-pt frame; the poses of a free-running stream
 * then differ from the reference's by centimetres at unchanged accuracy). */
void sageicp_set_downsample_order(int reference_order);
/* The replay itself (host code, no device needed): the iteration order of a tsl::robin_map v1.0.1
 * (core/Preprocessing.cpp:50,76-82: default-constructed, the reference's 20-bit VoxelHash) after the
 * n distinct voxels vox_xyz[3*i..] were inserted in this order; order_out[j] = insertion index of the
 * j-th entry met by the map's iterator.
 * Limits: n < 2^27; SAGEICP_ERR_CAPACITY when an insertion's probe distance reaches 128 — beyond its
 * DIST_FROM_IDEAL_BUCKET_LIMIT tsl::robin_map forces a growth the replay does not model (with the
 * reference's 20-bit hash: at the latest from ~2^19 voxels of one label group; street scenes stay
 * below 40).  Inside sageicp_voxel_downsample / the pipeline such a group is emitted in arrival
 * order and a warning goes to stderr once. */
int sageicp_robin_iteration_order(const int32_t *vox_xyz, uint64_t n, uint32_t *order_out);
/* The far-voxel sweep of VoxelHashMap::RemovePointsFarFromLocation (core/VoxelHashMap.cpp:176-184: erase WHILE iterating the
 * robin_map) on a table filled with the n distinct voxels in arrival order; far[i] != 0: voxel i is out of range.  listed = 0:
 * the sweep as written (every bucket looked at); 1: the form a map whose points live in HBM uses (only the far voxels are
 * known to the host) — the two must agree.  erased_out (capacity n): the voxels erased, in the order of their erasure;
 * order_after (capacity n): the iteration order of what is left. */
int sageicp_robin_sweep(const int32_t *vox_xyz, uint64_t n, const uint8_t *far, int listed, uint32_t *erased_out,
                        uint64_t *n_erased, uint32_t *order_after, uint64_t *n_after);
int sageicp_preprocess(const double *frame_xyzl, uint64_t n, double max_range, double min_range,
                       double label_max_range, double *out_xyzl, uint64_t *n_out, int device);
int sageicp_voxel_downsample(const double *frame_xyzl, uint64_t n, int n_groups,
                             const int *group_label_counts, const int *group_labels,
                             const double *group_voxel_size, double vox_scale, double *out_xyzl,
                             uint64_t *n_out, int device);

/* ---- per-frame pipeline counterpart: sage_icp::pipeline::sageICP (pipeline/sageICP.{hpp,cpp}) --
 * Host-side orchestration around the hot path for a stream of scans:
 * range crop + label zeroing (core/Preprocessing.cpp:173-187), two-level semantic voxel
 * down-sampling (core/Preprocessing.cpp:44-84, pipeline/sageICP.cpp:97-101), adaptive threshold
 * (core/Threshold.cpp:29-50), constant-velocity guess (pipeline/sageICP.cpp:110-115), RegisterFrame,
 * map update.  The reference's own pipeline compiles unchanged against the header shims; this
 * entry exists so that streams can be driven through the C ABI (tests, bench).  The PCL dynamic
 * vehicle filter and deskewing are not reproduced (both off in the pre-labelled configurations). */
typedef struct sageicp_pipeline sageicp_pipeline;
typedef struct sageicp_pipeline_config {   /* sageConfig, pipeline/sageICP.hpp:39-65 */
    double voxel_size_map, max_range, min_range, label_max_range, local_map_range;
    int basic_points_per_voxel, critical_points_per_voxel;
    const int *basic_parts_labels;
    int n_basic_parts_labels;
    double min_motion_th, initial_threshold, sem_th;
    int n_groups;                     /* voxel_labels.size() == voxel_size.size() */
    const int *group_label_counts;    /* [n_groups] */
    const int *group_labels;          /* concatenated label lists of the groups */
    const double *group_voxel_size;   /* [n_groups] */
    int device;
    int map_update_on_device;         /* 1: the per-frame map update runs on the GPU
                                       * (sageicp_map_update_pose_device); 0: on the host */
} sageicp_pipeline_config;

sageicp_pipeline *sageicp_pipeline_create(const sageicp_pipeline_config *config);
void sageicp_pipeline_destroy(sageicp_pipeline *p);
/* sageICP::RegisterFrame(frame), pipeline/sageICP.cpp:54-95.  icp_seconds is the span the
 * reference times around the hot path (:79-88); n_source the size of the registered cloud. */
int sageicp_pipeline_register_frame(sageicp_pipeline *p, const double *frame_xyzl, uint64_t n,
                                    double pose_out[7], double *icp_seconds, double *total_seconds,
                                    uint64_t *n_source, sageicp_stats *stats /* optional */);
/* Streams: announce the frame that FOLLOWS the next one registered.  Preprocess() + Voxelize()
 * (pipeline/sageICP.cpp:57-67) read the raw frame only — not the pose, not the map — so
 * sageicp_pipeline_register_frame() runs the announced frame's on a second set of device buffers
 * (own stream, own host thread) under the ICP loop and the map update of the frame it was called
 * for, and the call that later registers the announced frame (same pointer, same n) takes the
 * prepared clouds instead of computing them.  Results are bit-identical with and without; a frame
 * that was announced but not registered next is dropped.  One frame ahead, no queue.
 * Lifetime: a helper thread reads the announced buffer from inside the NEXT
 * sageicp_pipeline_register_frame() call on; it has finished when the register call after that one,
 * sageicp_pipeline_prefetch_cancel() or sageicp_pipeline_destroy() returns — until then the buffer
 * must stay valid and unchanged, also when the announced frame ends up not being registered.
 * A frame is recognised by pointer, size AND a fingerprint of its content taken here (64 rows spread
 * over the frame and the last one: best effort — it tells a buffer refilled with another scan from
 * the scan that was announced, not a buffer edited in a few places; do not edit announced buffers). */
int sageicp_pipeline_prefetch(sageicp_pipeline *p, const double *next_frame_xyzl, uint64_t n);
/* Wait for the helper thread without dropping what it prepared: afterwards nothing reads the
 * announced buffer any more (it may be released or refilled), and the prepared clouds are still
 * used if the announced frame — same pointer, size and content — is registered next. */
int sageicp_pipeline_prefetch_wait(sageicp_pipeline *p);
/* Drop an announcement / a prepared frame and wait for the helper thread: afterwards nothing
 * reads any announced buffer. */
int sageicp_pipeline_prefetch_cancel(sageicp_pipeline *p);
int sageicp_pipeline_reinitialize(sageicp_pipeline *p);          /* pipeline/sageICP.hpp:94-99 */
uint64_t sageicp_pipeline_num_poses(const sageicp_pipeline *p);  /* poses().size() */
int sageicp_pipeline_pose(const sageicp_pipeline *p, uint64_t index, double pose_out[7]);
const sageicp_map *sageicp_pipeline_local_map(const sageicp_pipeline *p);   /* LocalMap() */

/* ---- KITTI trajectory metrics: sage_icp::metrics (metrics/Metrics.hpp:33-37) ----------------------
 * Host-only (the reference's are CPU code too).  Poses are 4x4 homogeneous matrices, ROW-major
 * (a KITTI poses.txt row padded with 0 0 0 1), n of them, 16 doubles each.
 * sageicp_metrics_seq_error: SeqError, Metrics.cpp:140-155 — KITTI devkit relative error over
 *   segments of 100..800 m starting every 10th frame: (translation %, rotation deg/100 m with
 *   the reference's 180/3.14).  NaN when no segment is long enough, as in the reference.
 * sageicp_metrics_absolute_trajectory_error: AbsoluteTrajectoryError, Metrics.cpp:157-191 — RMSE
 *   of the rotation angle [rad] and of the translation [m] after a rigid Umeyama alignment. */
int sageicp_metrics_seq_error(const double *poses_gt, const double *poses_result, uint64_t n,
                              float *avg_trans_error, float *avg_rot_error);
int sageicp_metrics_absolute_trajectory_error(const double *poses_gt, const double *poses_result,
                                              uint64_t n, float *ate_rot, float *ate_trans);

#ifdef __cplusplus
}
#endif
#endif /* SAGEICP_H_ */
