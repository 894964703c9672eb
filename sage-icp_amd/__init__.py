"""sage_icp_amd — Python binding (ctypes) of libsageicp_hip.so, the MI355X implementation of
SAGE-ICP's registration hot path, for tests and bench.py.

The product is the C-ABI shared library (include/sageicp.h) and the C++ header shim in
shim/; this module only mirrors the reference's interface names on top of it:

    VoxelHashMap            sage_icp::VoxelHashMap   (cpp/sage_icp/core/VoxelHashMap.hpp:35-107)
    register_frame()        sage_icp::RegisterFrame  (cpp/sage_icp/core/Registration.hpp:34-39)
    transform_points()      sage_icp::TransformPoints(cpp/sage_icp/core/Registration.hpp:32)
    align_clouds()          AlignClouds              (cpp/sage_icp/core/Registration.cpp:59-94)

There is no CPU fallback: if the library has not been built, importing a compute symbol raises;
if no HIP device is present every compute call raises SageIcpError(SAGEICP_ERR_NO_DEVICE).
The directory is named `sage-icp_amd`; import it as `sage_icp_amd` through the loader module
/sage_icp_amd.py at the repo root.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAGE_SQNORM3_ORDER=0 in the environment loads the build with the rounds-1-3 association of the 3-term
# squared norms (csrc/sageicp_types.h; default 2 = what Eigen 3.4's reductions evaluate, per call site):
# libsageicp_hip.v0.so, built by build.py next to the default
SQNORM3_ORDER = 0 if os.environ.get("SAGE_SQNORM3_ORDER", "2") == "0" else 2
if os.environ.get("SAGE_SQNORM3_ORDER", "2") not in ("0", "2"):
    import warnings
    warnings.warn("SAGE_SQNORM3_ORDER=%r: only the builds 0 and 2 exist — using 2" % os.environ["SAGE_SQNORM3_ORDER"])
LIB_PATH = os.path.join(_HERE, "libsageicp_hip.v0.so" if SQNORM3_ORDER == 0 else "libsageicp_hip.so")
if os.environ.get("SAGEICP_VARIANT_LIB"):      # measurement only: a variant build of the same library (profiles/)
    LIB_PATH = os.path.abspath(os.environ["SAGEICP_VARIANT_LIB"])

_dp = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)

ABI_VERSION = 4          # SAGEICP_ABI_VERSION of include/sageicp.h
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_RCCL, ERR_CAPACITY = -1, -2, -3, -4, -5
UNIQUE_ID_BYTES = 128
P2P_HANDLE_BYTES = 64

IDENTITY = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64)


class SageIcpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sageicp error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("converged", C.c_int32),
        ("n_queries", C.c_uint64),
        ("n_corr_first", C.c_uint64),
        ("n_corr_last", C.c_uint64),
        ("last_step_norm", C.c_double),
        ("us_wall", C.c_double),
        ("us_upload", C.c_double),
        ("us_nn", C.c_double),
        ("us_fin", C.c_double),
        ("nn_launches", C.c_uint32),
        ("single_launch", C.c_uint32),
        ("sum_candidates", C.c_uint64),
        ("n_corr_hist", C.c_uint32 * 64),
        ("pairs_evaluated", C.c_uint64),
        ("lanes_per_query", C.c_uint32),
        ("compact_scan", C.c_uint32),
    ]


class CommInfo(C.Structure):
    """sageicp_comm_info"""
    _fields_ = [("rank", C.c_int32), ("nranks", C.c_int32), ("device", C.c_int32),
                ("has_rccl", C.c_int32), ("rccl_ranks", C.c_int32), ("rccl_rank", C.c_int32),
                ("p2p_connected", C.c_int32), ("p2p_enabled", C.c_int32), ("p2p_poisoned", C.c_int32),
                ("reserved", C.c_int32 * 7)]


class PipelineConfig(C.Structure):
    """sageicp_pipeline_config == sageConfig (pipeline/sageICP.hpp:39-65)"""
    _fields_ = [
        ("voxel_size_map", C.c_double), ("max_range", C.c_double), ("min_range", C.c_double),
        ("label_max_range", C.c_double), ("local_map_range", C.c_double),
        ("basic_points_per_voxel", C.c_int), ("critical_points_per_voxel", C.c_int),
        ("basic_parts_labels", C.POINTER(C.c_int)), ("n_basic_parts_labels", C.c_int),
        ("min_motion_th", C.c_double), ("initial_threshold", C.c_double), ("sem_th", C.c_double),
        ("n_groups", C.c_int),
        ("group_label_counts", C.POINTER(C.c_int)), ("group_labels", C.POINTER(C.c_int)),
        ("group_voxel_size", C.POINTER(C.c_double)),
        ("device", C.c_int),
        ("map_update_on_device", C.c_int),
    ]


# the SemanticKITTI parameter sets of ros/launch/odometry*.launch.py
KITTI_VOXEL_LABELS = [[40, 44, 48, 49], [50, 51, 52], [70, 72], [60, 71, 80, 81, 99], [0],
                      [10, 11, 13, 15, 16, 18, 20]]
KITTI_VOXEL_SIZE = [0.6, 1.0, 0.9, 0.8, 1.0, 0.6]


def make_pipeline_config(voxel_size_map=0.8, max_range=100.0, min_range=5.0, label_max_range=50.0,
                         local_map_range=100.0, basic=20, critical=20,
                         basic_parts_labels=(40, 44, 48, 49, 50, 70, 72), min_motion_th=0.1,
                         initial_threshold=2.0, sem_th=0.05, voxel_labels=None, voxel_size=None,
                         device=0, map_update_on_device=True):
    """defaults: ros/launch/odometry_gt.launch.py (pre-labelled scans, dynamic filter off)"""
    voxel_labels = KITTI_VOXEL_LABELS if voxel_labels is None else voxel_labels
    voxel_size = KITTI_VOXEL_SIZE if voxel_size is None else voxel_size
    assert len(voxel_labels) == len(voxel_size)
    keep = {}
    keep["basic"] = (C.c_int * len(basic_parts_labels))(*basic_parts_labels)
    keep["counts"] = (C.c_int * len(voxel_labels))(*[len(g) for g in voxel_labels])
    flat = [l for g in voxel_labels for l in g]
    keep["labels"] = (C.c_int * len(flat))(*flat)
    keep["sizes"] = (C.c_double * len(voxel_size))(*voxel_size)
    cfg = PipelineConfig(voxel_size_map, max_range, min_range, label_max_range, local_map_range,
                         basic, critical, keep["basic"], len(basic_parts_labels), min_motion_th,
                         initial_threshold, sem_th, len(voxel_labels), keep["counts"],
                         keep["labels"], keep["sizes"], device, 1 if map_update_on_device else 0)
    cfg._keep = keep          # the arrays must outlive the struct
    return cfg


# every symbol include/sageicp.h declares: (name, restype, argtypes)
_SIGNATURES = [
    ("sageicp_abi_version", C.c_int, []),
    ("sageicp_last_error", C.c_char_p, []),
    ("sageicp_device_count", C.c_int, []),
    ("sageicp_set_profiling", None, [C.c_int]),
    ("sageicp_set_counting", None, [C.c_int]),
    ("sageicp_reload_env", None, []),
    ("sageicp_map_loop_status", C.c_int, [C.c_void_p, C.c_void_p]),
    ("sageicp_set_downsample_order", None, [C.c_int]),
    ("sageicp_robin_iteration_order", C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    ("sageicp_robin_sweep", C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, _u64p, C.c_void_p, _u64p]),
    ("sageicp_map_create", C.c_void_p,
     [C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]),
    ("sageicp_map_destroy", None, [C.c_void_p]),
    ("sageicp_map_set_devices", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    ("sageicp_map_num_devices", C.c_int, [C.c_void_p]),
    ("sageicp_map_set_reference_order", C.c_int, [C.c_void_p, C.c_int]),
    ("sageicp_map_reference_order", C.c_int, [C.c_void_p]),
    ("sageicp_map_clone", C.c_void_p, [C.c_void_p]),
    ("sageicp_map_clear", C.c_int, [C.c_void_p]),
    ("sageicp_map_empty", C.c_int, [C.c_void_p]),
    ("sageicp_map_size", C.c_uint64, [C.c_void_p]),
    ("sageicp_map_num_voxels", C.c_uint64, [C.c_void_p]),
    ("sageicp_map_add_points", C.c_int, [C.c_void_p, _dp, C.c_uint64]),
    ("sageicp_map_remove_far", C.c_int, [C.c_void_p, _dp]),
    ("sageicp_map_update", C.c_int, [C.c_void_p, _dp, C.c_uint64, _dp]),
    ("sageicp_map_update_pose", C.c_int, [C.c_void_p, _dp, C.c_uint64, _dp]),
    ("sageicp_map_update_pose_device", C.c_int, [C.c_void_p, _dp, C.c_uint64, _dp]),
    ("sageicp_map_pointcloud", C.c_uint64, [C.c_void_p, _dp, C.c_uint64]),
    ("sageicp_map_resident", C.c_int, [C.c_void_p]),
    ("sageicp_map_point_slots", C.c_uint64, [C.c_void_p]),
    ("sageicp_map_sync", C.c_int, [C.c_void_p]),
    ("sageicp_get_correspondences", C.c_int,
     [C.c_void_p, _dp, C.c_uint64, C.c_double, C.c_double, _dp, _dp, _u64p, _i64p]),
    ("sageicp_align_clouds", C.c_int, [_dp, _dp, C.c_uint64, C.c_double, _dp, _dp, _dp, C.c_int]),
    ("sageicp_transform_points", C.c_int, [_dp, _dp, C.c_uint64, C.c_int]),
    ("sageicp_register_frame", C.c_int,
     [C.c_void_p, _dp, C.c_uint64, _dp, C.c_double, C.c_double, C.c_double, _dp,
      C.POINTER(Stats)]),
    ("sageicp_frame_upload", C.c_void_p, [C.c_void_p, _dp, C.c_uint64]),
    ("sageicp_frame_destroy", None, [C.c_void_p]),
    ("sageicp_register_frame_resident", C.c_int,
     [C.c_void_p, C.c_void_p, _dp, C.c_double, C.c_double, C.c_double, C.c_void_p, _dp,
      C.POINTER(Stats)]),
    ("sageicp_comm_unique_id", C.c_int, [_u8p]),
    ("sageicp_comm_create", C.c_void_p, [_u8p, C.c_int, C.c_int, C.c_int]),
    ("sageicp_comm_create_local", C.c_void_p, [C.c_int, C.c_int, C.c_int]),
    ("sageicp_comm_p2p_export", C.c_int, [C.c_void_p, C.POINTER(C.c_uint8)]),
    ("sageicp_comm_p2p_connect", C.c_int, [C.c_void_p, C.POINTER(C.c_uint8)]),
    ("sageicp_comm_p2p_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("sageicp_comm_p2p_enabled", C.c_int, [C.c_void_p]),
    ("sageicp_comm_describe", C.c_int, [C.c_void_p, C.POINTER(CommInfo)]),
    ("sageicp_comm_destroy", None, [C.c_void_p]),
    ("sageicp_preprocess", C.c_int,
     [_dp, C.c_uint64, C.c_double, C.c_double, C.c_double, _dp, _u64p, C.c_int]),
    ("sageicp_voxel_downsample", C.c_int,
     [_dp, C.c_uint64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
      C.c_double, _dp, _u64p, C.c_int]),
    ("sageicp_pipeline_create", C.c_void_p, [C.POINTER(PipelineConfig)]),
    ("sageicp_pipeline_destroy", None, [C.c_void_p]),
    ("sageicp_pipeline_register_frame", C.c_int,
     [C.c_void_p, _dp, C.c_uint64, _dp, _dp, _dp, _u64p, C.POINTER(Stats)]),
    ("sageicp_pipeline_prefetch", C.c_int, [C.c_void_p, _dp, C.c_uint64]),
    ("sageicp_pipeline_prefetch_cancel", C.c_int, [C.c_void_p]),
    ("sageicp_pipeline_prefetch_wait", C.c_int, [C.c_void_p]),
    ("sageicp_pipeline_reinitialize", C.c_int, [C.c_void_p]),
    ("sageicp_pipeline_num_poses", C.c_uint64, [C.c_void_p]),
    ("sageicp_pipeline_pose", C.c_int, [C.c_void_p, C.c_uint64, _dp]),
    ("sageicp_pipeline_local_map", C.c_void_p, [C.c_void_p]),
    ("sageicp_metrics_seq_error", C.c_int, [_dp, _dp, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("sageicp_metrics_absolute_trajectory_error", C.c_int,
     [_dp, _dp, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
]

EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None

# The library reads its SAGEICP_* knobs from the environment once (sageicp_reload_env() makes it look again).  Tests and
# probes flip knobs between calls through os.environ: every change of a SAGEICP_* name marks the cache stale, and the next
# call through lib() reloads.
_env_stale = [False]
_Environ = type(os.environ)
if not getattr(_Environ, "_sageicp_hooked", False):
    _set0, _del0 = _Environ.__setitem__, _Environ.__delitem__

    def _set1(self, k, v):
        _set0(self, k, v)
        if str(k).startswith("SAGEICP_"):
            _env_stale[0] = True

    def _del1(self, k):
        _del0(self, k)
        if str(k).startswith("SAGEICP_"):
            _env_stale[0] = True

    _Environ.__setitem__, _Environ.__delitem__, _Environ._sageicp_hooked = _set1, _del1, True


class LoopStatus(C.Structure):
    _fields_ = [("calls_single_launch", C.c_uint64), ("calls_per_iteration", C.c_uint64), ("calls_chained", C.c_uint64), ("timeouts", C.c_uint32),
                ("cooldown_calls", C.c_uint32), ("derate_workgroups", C.c_uint32), ("last_fallback", C.c_int32)]


def lib():
    """Load libsageicp_hip.so.  Raises (loudly) if it has not been built — no fallback."""
    global _lib
    if _lib is not None and _env_stale[0]:
        _env_stale[0] = False
        _lib.sageicp_reload_env()
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        L.sageicp_abi_version.restype = C.c_int
        if L.sageicp_abi_version() != ABI_VERSION:      # struct layouts are part of the ABI
            raise ImportError("%s speaks ABI version %d, this binding %d: rebuild it"
                              % (LIB_PATH, L.sageicp_abi_version(), ABI_VERSION))
        for name, res, args in _SIGNATURES:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise SageIcpError(rc, (lib().sageicp_last_error() or b"").decode())


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def device_count():
    return int(lib().sageicp_device_count())


def robin_sweep(vox, far, listed):
    """sageicp_robin_sweep: (erased voxels in erasure order, iteration order of the rest)"""
    v = np.ascontiguousarray(vox, dtype=np.int32).reshape(-1, 3)
    f = np.ascontiguousarray(far, dtype=np.uint8)
    n = len(v)
    er, af = np.zeros(max(n, 1), dtype=np.uint32), np.zeros(max(n, 1), dtype=np.uint32)
    ne, na = C.c_uint64(0), C.c_uint64(0)
    _check(lib().sageicp_robin_sweep(v.ctypes.data_as(C.c_void_p), n, f.ctypes.data_as(C.c_void_p), 1 if listed else 0,
                                     er.ctypes.data_as(C.c_void_p), C.byref(ne), af.ctypes.data_as(C.c_void_p), C.byref(na)))
    return er[:ne.value].copy(), af[:na.value].copy()


def set_downsample_order(reference_order=True):
    """True (default): VoxelDownsample emits in the reference's robin_map bucket order; False: arrival order"""
    lib().sageicp_set_downsample_order(1 if reference_order else 0)


def set_profiling(level):
    """0 off; 1 (or True) HIP events around k_icp in one iteration out of 8 (what bench.py times
    with); 2 around every kernel of every iteration"""
    lib().sageicp_set_profiling(int(level))


def set_counting(on):
    """True (default): calls that return statistics count candidates and evaluated pairs (sum_candidates,
    pairs_evaluated); False: those two stay zero and the search runs as it does for a caller without statistics"""
    lib().sageicp_set_counting(1 if on else 0)


class Frame:
    """A scan resident in HBM (sageicp_frame)."""

    def __init__(self, vmap, pts):
        pts, pp = _d(pts)
        self.n = pts.reshape(-1, 4).shape[0]
        self._h = lib().sageicp_frame_upload(vmap._h, pp, self.n)
        if not self._h:
            raise SageIcpError(ERR_HIP, (lib().sageicp_last_error() or b"").decode())

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:       # (None: interpreter shutdown)
            lib().sageicp_frame_destroy(self._h)
            self._h = None


class Comm:
    """Communicator for query-sharded registration (one process per GPU): RCCL all-reduce, or the
    direct exchange over xGMI once p2p_connect() has been called."""

    def __init__(self, unique_id, rank, nranks, device):
        self.rank, self.nranks = rank, nranks
        if unique_id is None:                       # no RCCL side: direct exchange only
            self._h = lib().sageicp_comm_create_local(rank, nranks, device)
        else:
            buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
            self._h = lib().sageicp_comm_create(buf, rank, nranks, device)
        if not self._h:
            raise SageIcpError(ERR_RCCL, (lib().sageicp_last_error() or b"").decode())

    def p2p_export(self):
        buf = (C.c_uint8 * P2P_HANDLE_BYTES)()
        _check(lib().sageicp_comm_p2p_export(self._h, buf))
        return bytes(buf)

    def p2p_connect(self, handles):
        """handles: the p2p_export() of every rank, in rank order"""
        assert len(handles) == self.nranks
        flat = b"".join(bytes(h) for h in handles)
        buf = (C.c_uint8 * len(flat)).from_buffer_copy(flat)
        _check(lib().sageicp_comm_p2p_connect(self._h, buf))

    def p2p_enable(self, on=True):
        _check(lib().sageicp_comm_p2p_enable(self._h, 1 if on else 0))

    @property
    def p2p_enabled(self):
        return bool(lib().sageicp_comm_p2p_enabled(self._h))

    def describe(self):
        """dict: the ranks as created, the ranks RCCL itself reports (-1 without an RCCL side), the
        state of the direct exchange"""
        info = CommInfo()
        _check(lib().sageicp_comm_describe(self._h, C.byref(info)))
        return {k: int(getattr(info, k)) for k, _ in CommInfo._fields_ if k != "reserved"}

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
        _check(lib().sageicp_comm_unique_id(buf))
        return bytes(buf)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:       # (None: interpreter shutdown)
            lib().sageicp_comm_destroy(self._h)
            self._h = None


class VoxelHashMap:
    """sage_icp::VoxelHashMap (core/VoxelHashMap.hpp:35-107) over the C ABI."""

    def __init__(self, voxel_size, max_distance, basic_points_per_voxel=20,
                 critical_points_per_voxel=20, basic_parts_labels=(40, 44, 48, 49, 50, 70, 72),
                 device=0, _handle=None):
        self.voxel_size_ = voxel_size
        self.max_distance_ = max_distance
        self.basic_points_per_voxel_ = basic_points_per_voxel
        self.critical_points_per_voxel_ = critical_points_per_voxel
        self.basic_parts_labels_ = list(basic_parts_labels)
        self.device = device
        if _handle is not None:
            self._h = _handle
            return
        labels = (C.c_int * len(self.basic_parts_labels_))(*self.basic_parts_labels_)
        self._h = lib().sageicp_map_create(voxel_size, max_distance, basic_points_per_voxel,
                                           critical_points_per_voxel, labels,
                                           len(self.basic_parts_labels_), device)
        if not self._h:
            raise SageIcpError(ERR_INVALID, (lib().sageicp_last_error() or b"").decode())

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:       # (None: interpreter shutdown)
            lib().sageicp_map_destroy(self._h)
            self._h = None

    def clone(self):
        h = lib().sageicp_map_clone(self._h)
        return VoxelHashMap(self.voxel_size_, self.max_distance_, self.basic_points_per_voxel_,
                            self.critical_points_per_voxel_, self.basic_parts_labels_,
                            self.device, _handle=h)

    def set_reference_order(self, on=True):
        """Reference-order mode (include/sageicp.h): the far-voxel sweep erases while iterating the
        reference's robin_map bucket array and Pointcloud() lists the voxels in bucket order.  The map
        must be empty."""
        _check(lib().sageicp_map_set_reference_order(self._h, 1 if on else 0))
        return self

    def reference_order(self):
        """0 off, 1 on, -1 on but beyond what the emulation models"""
        return int(lib().sageicp_map_reference_order(self._h))

    def set_devices(self, devices):
        """single-process multi-GPU mode: the map spans these devices, RegisterFrame shards over them"""
        arr = (C.c_int * len(devices))(*devices)
        _check(lib().sageicp_map_set_devices(self._h, arr, len(devices)))

    def num_devices(self):
        return int(lib().sageicp_map_num_devices(self._h))

    # names follow the reference's members
    def Clear(self):
        _check(lib().sageicp_map_clear(self._h))

    def Empty(self):
        return bool(lib().sageicp_map_empty(self._h))

    def size(self):
        return int(lib().sageicp_map_size(self._h))

    def num_voxels(self):
        return int(lib().sageicp_map_num_voxels(self._h))

    def AddPoints(self, pts):
        pts, pp = _d(pts)
        _check(lib().sageicp_map_add_points(self._h, pp, pts.reshape(-1, 4).shape[0]))

    def RemovePointsFarFromLocation(self, origin):
        o, op = _d(origin)
        _check(lib().sageicp_map_remove_far(self._h, op))

    def UpdateOnDevice(self, pts, pose):
        """Update(points, pose) executed on the GPU against the HBM-resident map (row f-2)."""
        pts, pp = _d(pts)
        x, xp = _d(pose)
        assert x.size == 7
        _check(lib().sageicp_map_update_pose_device(self._h, pp, pts.reshape(-1, 4).shape[0], xp))

    def Update(self, pts, pose_or_origin):
        pts, pp = _d(pts)
        x, xp = _d(pose_or_origin)
        n = pts.reshape(-1, 4).shape[0]
        if x.size == 7:
            _check(lib().sageicp_map_update_pose(self._h, pp, n, xp))
        elif x.size == 3:
            _check(lib().sageicp_map_update(self._h, pp, n, xp))
        else:
            raise ValueError("Update takes a pose[7] or an origin[3]")

    def Pointcloud(self):
        n = self.size()
        out = np.empty((n, 4))
        lib().sageicp_map_pointcloud(self._h, out.ctypes.data_as(_dp), n)
        return out

    def resident(self):
        """True while the HBM copy of the map is the authority (after a device-side update)"""
        return bool(lib().sageicp_map_resident(self._h))

    def point_slots(self):
        """32-B point slots the voxel storage occupies (size-classed regions, free ones included)"""
        return int(lib().sageicp_map_point_slots(self._h))

    def sync(self):
        _check(lib().sageicp_map_sync(self._h))

    def loop_status(self):
        """sageicp_map_loop_status: which form of the ICP loop this handle's calls took, and why"""
        st = LoopStatus()
        _check(lib().sageicp_map_loop_status(self._h, C.byref(st)))
        return st

    def GetCorrespondences(self, pts, max_correspondance_distance, th, with_index=False):
        pts, pp = _d(pts)
        n = pts.reshape(-1, 4).shape[0]
        src = np.empty((n, 4))
        tgt = np.empty((n, 4))
        idx = np.empty(n, dtype=np.int64)
        nout = C.c_uint64(0)
        _check(lib().sageicp_get_correspondences(
            self._h, pp, n, max_correspondance_distance, th, src.ctypes.data_as(_dp),
            tgt.ctypes.data_as(_dp), C.byref(nout), idx.ctypes.data_as(_i64p)))
        k = nout.value
        if with_index:
            return src[:k].copy(), tgt[:k].copy(), idx[:k].copy()
        return src[:k].copy(), tgt[:k].copy()


def register_frame(frame, voxel_map, initial_guess, max_correspondence_distance, kernel, sem_th,
                   comm=None, return_stats=False):
    """sage_icp::RegisterFrame (core/Registration.cpp:113-141).  `frame` is an (n,4) array or a
    resident Frame; with `comm` the call is one rank of a query-sharded registration."""
    init, ip = _d(initial_guess)
    out = np.empty(7)
    st = Stats()
    if isinstance(frame, Frame):
        _check(lib().sageicp_register_frame_resident(
            voxel_map._h, frame._h, ip, max_correspondence_distance, kernel, sem_th,
            comm._h if comm is not None else None, out.ctypes.data_as(_dp), C.byref(st)))
    else:
        if comm is not None:
            raise ValueError("sharded registration needs a resident Frame")
        pts, pp = _d(frame)
        _check(lib().sageicp_register_frame(
            voxel_map._h, pp, pts.reshape(-1, 4).shape[0], ip, max_correspondence_distance,
            kernel, sem_th, out.ctypes.data_as(_dp), C.byref(st)))
    return (out, st) if return_stats else out


def transform_points(pose, pts, device=0):
    T, tp = _d(pose)
    out = np.array(pts, dtype=np.float64, order="C", copy=True).reshape(-1, 4)
    _check(lib().sageicp_transform_points(tp, out.ctypes.data_as(_dp), out.shape[0], device))
    return out


def align_clouds(src, tgt, kernel, device=0):
    src, sp = _d(src)
    tgt, gp = _d(tgt)
    T = np.empty(7)
    JTJ = np.empty(36)
    JTr = np.empty(6)
    _check(lib().sageicp_align_clouds(sp, gp, src.reshape(-1, 4).shape[0], kernel,
                                      T.ctypes.data_as(_dp), JTJ.ctypes.data_as(_dp),
                                      JTr.ctypes.data_as(_dp), device))
    return T, JTJ.reshape(6, 6), JTr


class SageICP:
    """sage_icp::pipeline::sageICP (pipeline/sageICP.hpp:67-109) over the C ABI."""

    def __init__(self, config=None, **kw):
        self.config = config if config is not None else make_pipeline_config(**kw)
        self._h = lib().sageicp_pipeline_create(C.byref(self.config))
        if not self._h:
            raise SageIcpError(ERR_INVALID, (lib().sageicp_last_error() or b"").decode())

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:       # (None: interpreter shutdown)
            lib().sageicp_pipeline_destroy(self._h)
            self._h = None

    def RegisterFrame(self, frame):
        """returns (pose[7], icp_seconds, total_seconds, n_source, stats)"""
        pts, pp = _d(frame)
        out = np.empty(7)
        icp, tot, ns = C.c_double(0), C.c_double(0), C.c_uint64(0)
        st = Stats()
        _check(lib().sageicp_pipeline_register_frame(self._h, pp, pts.reshape(-1, 4).shape[0],
                                                     out.ctypes.data_as(_dp), C.byref(icp),
                                                     C.byref(tot), C.byref(ns), C.byref(st)))
        return out, icp.value, tot.value, ns.value, st

    def prefetch(self, next_frame):
        """Announce the frame after the next one registered: its Preprocess() + Voxelize() then run
        under that frame's ICP loop (sageicp_pipeline_prefetch).  Returns the array to pass to
        RegisterFrame() later (the same buffer: it is matched by address) — keep it alive."""
        pts, pp = _d(next_frame)
        # keep the buffer alive until it has been registered (the one announced before it may
        # still be waiting for its RegisterFrame)
        self._announced = (getattr(self, "_announced", ()) + ((pts, pp),))[-2:]
        _check(lib().sageicp_pipeline_prefetch(self._h, pp, pts.reshape(-1, 4).shape[0]))
        return pts

    def prefetch_wait(self):
        """wait for the helper thread; what it prepared is kept"""
        _check(lib().sageicp_pipeline_prefetch_wait(self._h))

    def prefetch_cancel(self):
        """drop an announced / prepared frame and wait for the helper thread"""
        _check(lib().sageicp_pipeline_prefetch_cancel(self._h))
        self._announced = ()

    def reinitialize(self):
        _check(lib().sageicp_pipeline_reinitialize(self._h))

    def poses(self):
        n = int(lib().sageicp_pipeline_num_poses(self._h))
        out = np.empty((n, 7))
        for i in range(n):
            _check(lib().sageicp_pipeline_pose(self._h, i, out[i].ctypes.data_as(_dp)))
        return out

    def LocalMap(self):
        h = lib().sageicp_pipeline_local_map(self._h)
        n = int(lib().sageicp_map_size(h))
        out = np.empty((n, 4))
        lib().sageicp_map_pointcloud(h, out.ctypes.data_as(_dp), n)
        return out


def preprocess(frame, max_range, min_range, label_max_range, device=0):
    """sage_icp::Preprocess with dynamic_vehicle_filter off (core/Preprocessing.cpp:173-187)"""
    pts, pp = _d(frame)
    n = pts.reshape(-1, 4).shape[0]
    out = np.empty((n, 4))
    k = C.c_uint64(0)
    _check(lib().sageicp_preprocess(pp, n, max_range, min_range, label_max_range,
                                    out.ctypes.data_as(_dp), C.byref(k), device))
    return out[:k.value].copy()


def voxel_downsample(frame, voxel_labels, voxel_size, vox_scale, device=0):
    """sage_icp::VoxelDownsample (core/Preprocessing.cpp:44-84)"""
    pts, pp = _d(frame)
    n = pts.reshape(-1, 4).shape[0]
    counts = (C.c_int * len(voxel_labels))(*[len(g) for g in voxel_labels])
    flat = [l for g in voxel_labels for l in g]
    labels = (C.c_int * max(len(flat), 1))(*flat)
    sizes = (C.c_double * len(voxel_size))(*voxel_size)
    out = np.empty((n, 4))
    k = C.c_uint64(0)
    _check(lib().sageicp_voxel_downsample(pp, n, len(voxel_labels), counts, labels, sizes, vox_scale,
                                          out.ctypes.data_as(_dp), C.byref(k), device))
    return out[:k.value].copy()


def _poses44(p):
    a = np.ascontiguousarray(p, dtype=np.float64).reshape(-1, 4, 4)
    return a, a.ctypes.data_as(_dp)


def seq_error(poses_gt, poses_result):
    """sage_icp::metrics::SeqError (metrics/Metrics.cpp:140-155): (avg translation error %,
    avg rotation error deg/100 m) over the KITTI devkit segments; poses are (n, 4, 4)"""
    g, gp = _poses44(poses_gt)
    r, rp = _poses44(poses_result)
    assert g.shape == r.shape
    t, o = C.c_float(0), C.c_float(0)
    _check(lib().sageicp_metrics_seq_error(gp, rp, g.shape[0], C.byref(t), C.byref(o)))
    return t.value, o.value


def absolute_trajectory_error(poses_gt, poses_result):
    """sage_icp::metrics::AbsoluteTrajectoryError (metrics/Metrics.cpp:157-191): (ATE rotation
    [rad], ATE translation [m]) after a rigid Umeyama alignment; poses are (n, 4, 4)"""
    g, gp = _poses44(poses_gt)
    r, rp = _poses44(poses_result)
    assert g.shape == r.shape
    a, b = C.c_float(0), C.c_float(0)
    _check(lib().sageicp_metrics_absolute_trajectory_error(gp, rp, g.shape[0], C.byref(a), C.byref(b)))
    return a.value, b.value
