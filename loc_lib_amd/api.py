"""ctypes binding of ``liblocgpu.so`` (the C ABI in ``include/locgpu.h``).

This is test/bench plumbing around the product library: every call goes straight through the C ABI to the
HIP kernels. There is no CPU fallback — if the library or a GPU is missing the calls raise ``LocGpuError``.
Class and method names mirror the reference's matcher interface
(``LocUtils::IcpRegistration`` / ``NdtRegistration``: SetInputTarget, ScanMatch, CaculateMatrixHAndB —
LocUtils/include/LocUtils/model/matching/3d/matching_interface.h:13-54).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LOCGPU_LIB") or os.path.join(_HERE, "liblocgpu.so")  # LOCGPU_LIB: A/B builds of the same library
CSRC = os.path.join(_HERE, "csrc")

P2P, P2LINE, P2PLANE = 0, 1, 2
SEARCH_TREE_FAITHFUL, SEARCH_GRID_EXACT = 0, 1
CENTER, NEARBY6 = 0, 1
DIRECT_NDT, INCREMENTAL_NDT = 1, 2

# every symbol include/locgpu.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "locgpu_icp_opts_default", "locgpu_ndt_opts_default", "locgpu_create", "locgpu_destroy", "locgpu_last_error",
    "locgpu_device_count", "locgpu_icp_set_target", "locgpu_icp_set_target_async", "locgpu_icp_target_info", "locgpu_knn", "locgpu_icp_hb", "locgpu_icp_align",
    "locgpu_transform_cloud", "locgpu_batch_create", "locgpu_batch_destroy", "locgpu_icp_align_batch", "locgpu_ndt_align_batch",
    "locgpu_icp_hb_batch", "locgpu_gn_update", "locgpu_ndt_set_target", "locgpu_ndt_target_info", "locgpu_ndt_dump",
    "locgpu_ndt_align", "locgpu_profile_enable", "locgpu_profile_read", "locgpu_visit_count_enable", "locgpu_visit_count_read",
    "locgpu_search_stats_read", "locgpu_debug_batch_nn", "locgpu_graph_enable",
    "locgpu_cloud_create", "locgpu_cloud_destroy", "locgpu_cloud_upload", "locgpu_cloud_info", "locgpu_cloud_download", "locgpu_cloud_copy",
    "locgpu_cloud_remove_nan", "locgpu_cloud_voxel_filter", "locgpu_cloud_crop_box", "locgpu_cloud_transform", "locgpu_cloud_append",
    "locgpu_icp_set_target_cloud", "locgpu_icp_set_target_cloud_async", "locgpu_ndt_set_target_cloud", "locgpu_icp_align_cloud", "locgpu_ndt_align_cloud",
    "locgpu_voxel_filter", "locgpu_crop_box", "locgpu_remove_nan",
    "locgpu_submap_create", "locgpu_submap_destroy", "locgpu_submap_add_keyframe", "locgpu_submap_cloud", "locgpu_submap_last_keyframe",
    "locgpu_submap_info", "locgpu_cloud_loam_extract", "locgpu_loam_extract",
    "locgpu_batch_create_empty", "locgpu_batch_upload_async", "locgpu_batch_upload_wait",
    "locgpu_bfnn_set_target", "locgpu_bfnn_knn",
    "locgpu_icp_align_batch_begin", "locgpu_ndt_align_batch_begin", "locgpu_align_batch_end",
    "locgpu_comm_unique_id", "locgpu_comm_init", "locgpu_comm_info", "locgpu_batch_create_sharded", "locgpu_icp_set_target_bcast",
    "locgpu_pool_opts_default", "locgpu_pool_create", "locgpu_pool_destroy", "locgpu_pool_submit", "locgpu_pool_wait", "locgpu_pool_info",
    "locgpu_pool_profile_read", "locgpu_pool_step", "locgpu_pool_done",
    "locgpu_icp_scan_match", "locgpu_ndt_scan_match",
]
COMM_ID_BYTES = 128
NO_INTENSITY = ctypes.c_size_t(-1).value


class LocGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("locgpu error %d: %s" % (code, msg))
        self.code = code


class IcpOpts(ctypes.Structure):
    _fields_ = [("method", ctypes.c_int32), ("max_iteration", ctypes.c_int32), ("max_nn_distance", ctypes.c_double),
                ("max_plane_distance", ctypes.c_double), ("max_line_distance", ctypes.c_double), ("min_effective_pts", ctypes.c_int32),
                ("eps", ctypes.c_double), ("approximate", ctypes.c_int32), ("ann_alpha", ctypes.c_float), ("search_mode", ctypes.c_int32)]


class NdtOpts(ctypes.Structure):
    _fields_ = [("max_iteration", ctypes.c_int32), ("voxel_size", ctypes.c_double), ("min_effective_pts", ctypes.c_int32),
                ("min_pts_in_voxel", ctypes.c_int32), ("eps", ctypes.c_double), ("res_outlier_th", ctypes.c_double),
                ("nearby_type", ctypes.c_int32), ("method", ctypes.c_int32), ("capacity", ctypes.c_int64)]


class PoolOpts(ctypes.Structure):
    _fields_ = [("slots", ctypes.c_int32), ("prefetch", ctypes.c_int32), ("scans_per_job", ctypes.c_int32), ("chunk", ctypes.c_int32), ("matcher", ctypes.c_int32),
                ("max_points", ctypes.c_uint64), ("icp", IcpOpts)]


class AlignStats(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int32), ("converged", ctypes.c_int32), ("status", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("last_effective_num", ctypes.c_int64), ("last_dx_norm", ctypes.c_double)]


def build(force=False):
    """Compile liblocgpu.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    """Load liblocgpu.so. Raises if it has not been built — the product path never falls back to the CPU."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LocGpuError(-2, "liblocgpu.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); no CPU fallback exists")
        L = ctypes.CDLL(LIB_PATH)
        vp, sz, i32, dbl, f32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double, ctypes.c_float
        sig = {
            "locgpu_icp_opts_default": (None, [vp]), "locgpu_ndt_opts_default": (None, [vp]),
            "locgpu_create": (i32, [i32, vp]), "locgpu_destroy": (None, [vp]), "locgpu_last_error": (ctypes.c_char_p, [vp]),
            "locgpu_device_count": (i32, []),
            "locgpu_icp_set_target": (i32, [vp, vp, sz, sz]), "locgpu_icp_set_target_async": (i32, [vp, vp, sz, sz]), "locgpu_icp_target_info": (i32, [vp, vp]),
            "locgpu_knn": (i32, [vp, vp, sz, i32, i32, f32, i32, vp, vp]),
            "locgpu_icp_hb": (i32, [vp, vp, sz, sz, vp, vp, vp, vp, vp, vp]),
            "locgpu_icp_align": (i32, [vp, vp, sz, sz, vp, vp, vp, vp]),
            "locgpu_transform_cloud": (i32, [vp, vp, vp, sz, sz, vp, sz]),
            "locgpu_batch_create": (i32, [vp, vp, vp, sz, i32, vp]), "locgpu_batch_destroy": (None, [vp]),
            "locgpu_icp_align_batch": (i32, [vp, vp, vp, vp, vp, vp]), "locgpu_ndt_align_batch": (i32, [vp, vp, vp, vp, vp]),
            "locgpu_icp_align_batch_begin": (i32, [vp, vp, vp, vp]), "locgpu_ndt_align_batch_begin": (i32, [vp, vp, vp]),
            "locgpu_align_batch_end": (i32, [vp, vp, vp, vp]),
            "locgpu_icp_hb_batch": (i32, [vp, vp, vp, vp, vp]),
            "locgpu_gn_update": (i32, [vp, i32, i32, dbl, vp, vp, vp, vp]),
            "locgpu_ndt_set_target": (i32, [vp, vp, sz, sz, vp]), "locgpu_ndt_target_info": (i32, [vp, vp]),
            "locgpu_ndt_dump": (i32, [vp, vp, vp, vp, sz, vp]), "locgpu_ndt_align": (i32, [vp, vp, sz, sz, vp, vp, vp]),
            "locgpu_profile_enable": (i32, [vp, i32]), "locgpu_profile_read": (i32, [vp, vp, i32]),
            "locgpu_visit_count_enable": (i32, [vp, i32]), "locgpu_visit_count_read": (i32, [vp, vp, i32]),
            "locgpu_search_stats_read": (i32, [vp, vp, i32]), "locgpu_debug_batch_nn": (i32, [vp, vp, i32, vp]), "locgpu_graph_enable": (i32, [vp, i32]),
            "locgpu_cloud_create": (i32, [vp, vp]), "locgpu_cloud_destroy": (None, [vp]),
            "locgpu_cloud_upload": (i32, [vp, vp, sz, sz, sz, i32]), "locgpu_cloud_info": (i32, [vp, vp, vp]),
            "locgpu_cloud_download": (i32, [vp, vp, sz, sz, sz]), "locgpu_cloud_copy": (i32, [vp, vp]),
            "locgpu_cloud_remove_nan": (i32, [vp, vp]), "locgpu_cloud_voxel_filter": (i32, [vp, f32, vp, vp]),
            "locgpu_cloud_crop_box": (i32, [vp, vp, vp, vp]), "locgpu_cloud_transform": (i32, [vp, vp, vp]),
            "locgpu_cloud_append": (i32, [vp, vp]),
            "locgpu_icp_set_target_cloud": (i32, [vp, vp]), "locgpu_icp_set_target_cloud_async": (i32, [vp, vp]), "locgpu_ndt_set_target_cloud": (i32, [vp, vp, vp]),
            "locgpu_icp_align_cloud": (i32, [vp, vp, vp, vp, vp, vp]), "locgpu_ndt_align_cloud": (i32, [vp, vp, vp, vp, vp]),
            "locgpu_voxel_filter": (i32, [vp, vp, sz, sz, sz, i32, f32, vp, vp, vp]),
            "locgpu_crop_box": (i32, [vp, vp, sz, sz, sz, i32, vp, vp, vp, vp, vp]),
            "locgpu_remove_nan": (i32, [vp, vp, sz, sz, sz, i32, vp, vp, vp]),
            "locgpu_submap_create": (i32, [vp, i32, f32, vp]), "locgpu_submap_destroy": (None, [vp]),
            "locgpu_submap_add_keyframe": (i32, [vp, vp, vp]), "locgpu_submap_cloud": (i32, [vp, vp]),
            "locgpu_submap_last_keyframe": (i32, [vp, vp]), "locgpu_submap_info": (i32, [vp, vp, vp]),
            "locgpu_cloud_loam_extract": (i32, [vp, vp, i32, vp, vp]),
            "locgpu_loam_extract": (i32, [vp, vp, sz, sz, sz, i32, sz, i32, vp, vp, vp, vp, sz, sz]),
            "locgpu_batch_create_empty": (i32, [vp, i32, sz, vp]), "locgpu_batch_upload_async": (i32, [vp, vp, vp, sz]),
            "locgpu_batch_upload_wait": (i32, [vp]),
            "locgpu_comm_unique_id": (i32, [vp]), "locgpu_comm_init": (i32, [vp, i32, i32, vp]), "locgpu_comm_info": (i32, [vp, vp, vp]),
            "locgpu_batch_create_sharded": (i32, [vp, vp, vp, sz, i32, i32, i32, vp]),
            "locgpu_icp_set_target_bcast": (i32, [vp, vp, sz, sz, i32]),
            "locgpu_bfnn_set_target": (i32, [vp, vp, sz, sz]), "locgpu_bfnn_knn": (i32, [vp, vp, sz, i32, vp]),
            "locgpu_pool_opts_default": (None, [vp]), "locgpu_pool_create": (i32, [vp, vp, vp]), "locgpu_pool_destroy": (None, [vp]),
            "locgpu_pool_submit": (i32, [vp, vp, vp, sz, i32, i32, i32, vp, vp]), "locgpu_pool_wait": (i32, [vp, ctypes.c_int64, vp, vp]),
            "locgpu_pool_info": (i32, [vp, vp]), "locgpu_pool_profile_read": (i32, [vp, vp, i32]),
            "locgpu_pool_step": (i32, [vp, i32]), "locgpu_pool_done": (i32, [vp, ctypes.c_int64, vp]),
            "locgpu_icp_scan_match": (i32, [vp, vp, sz, sz, vp, vp, vp, vp, vp, sz, vp, vp]),
            "locgpu_ndt_scan_match": (i32, [vp, vp, sz, sz, vp, vp, vp, vp, sz, vp, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def device_count():
    return int(lib().locgpu_device_count())


def comm_unique_id():
    """RCCL unique id (COMM_ID_BYTES bytes): make it on one rank, hand it to every rank's Context.comm_init."""
    buf = (ctypes.c_char * COMM_ID_BYTES)()
    rc = lib().locgpu_comm_unique_id(buf)
    if rc != 0:
        raise LocGpuError(rc, "locgpu_comm_unique_id failed")
    return bytes(buf)


def _cloud(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("cloud must be [n, >=3] float32")
    return a


def _pose(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    if p.shape[-1] != 7:
        raise ValueError("pose must have 7 doubles (quaternion xyzw + translation)")
    return p


def icp_opts(method=P2P, **kw):
    o = IcpOpts()
    lib().locgpu_icp_opts_default(ctypes.byref(o))
    o.method = method
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown ICP option %r" % k)
        setattr(o, k, v)
    return o


def ndt_opts(**kw):
    o = NdtOpts()
    lib().locgpu_ndt_opts_default(ctypes.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown NDT option %r" % k)
        setattr(o, k, v)
    return o


def _stats_dict(s):
    return dict(iterations=s.iterations, converged=bool(s.converged), status=s.status, last_effective_num=s.last_effective_num,
                last_dx_norm=s.last_dx_norm)


class Context:
    """One matcher instance on one GPU (what an IcpRegistration / NdtRegistration object owns)."""

    def __init__(self, device_id=0):
        self._h = ctypes.c_void_p()
        rc = lib().locgpu_create(device_id, ctypes.byref(self._h))
        if rc != 0:
            raise LocGpuError(rc, lib().locgpu_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().locgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise LocGpuError(rc, lib().locgpu_last_error(self._h).decode())

    # ---- IcpRegistration::SetInputTarget
    def icp_set_target(self, cloud, wait=True):
        """wait=False: returns once the points are copied; the host tree build runs on a worker thread and the next call that reads
        the ICP target completes the ingest."""
        c = _cloud(cloud)
        fn = lib().locgpu_icp_set_target if wait else lib().locgpu_icp_set_target_async
        self._check(fn(self._h, c.ctypes.data, c.shape[0], c.strides[0]))

    def icp_target_info(self):
        out = np.zeros(4, dtype=np.int64)
        self._check(lib().locgpu_icp_target_info(self._h, out.ctypes.data))
        return dict(num_leaves=int(out[0]), num_nodes=int(out[1]), depth=int(out[2]), bytes=int(out[3]))

    # ---- SearchPointInterface::FindNearstPoints, many queries
    def knn(self, queries, k=5, approximate=True, alpha=0.1, search_mode=SEARCH_TREE_FAITHFUL, with_visits=False):
        q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32)[:, :3])
        out = np.empty((q.shape[0], k), dtype=np.int32)
        vis = np.zeros((q.shape[0], 2), dtype=np.uint32) if with_visits else None
        self._check(lib().locgpu_knn(self._h, q.ctypes.data, q.shape[0], k, int(approximate), alpha, search_mode, out.ctypes.data,
                                     vis.ctypes.data if with_visits else None))
        return (out, vis) if with_visits else out

    # ---- BfnnRegistration (brute-force SearchPointInterface)
    def bfnn_set_target(self, cloud):
        c = _cloud(cloud)
        self._check(lib().locgpu_bfnn_set_target(self._h, c.ctypes.data, c.shape[0], c.strides[0]))

    def bfnn_knn(self, queries, k=5):
        q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32)[:, :3])
        out = np.empty((q.shape[0], k), dtype=np.int32)
        self._check(lib().locgpu_bfnn_knn(self._h, q.ctypes.data, q.shape[0], k, out.ctypes.data))
        return out

    # ---- MatchingInterface::CaculateMatrixHAndB
    def icp_hb(self, src, pose, opts):
        s = _cloud(src)
        H, B = np.zeros(36), np.zeros(6)
        eff, ok = ctypes.c_int64(0), ctypes.c_int(0)
        self._check(lib().locgpu_icp_hb(self._h, s.ctypes.data, s.shape[0], s.strides[0], _pose(pose).ctypes.data, ctypes.byref(opts),
                                        H.ctypes.data, B.ctypes.data, ctypes.byref(eff), ctypes.byref(ok)))
        return bool(ok.value), H.reshape(6, 6), B, int(eff.value)

    # ---- IcpRegistration::ScanMatch (pose part)
    def icp_align(self, src, init_pose, opts):
        s = _cloud(src)
        out = np.zeros(7)
        st = AlignStats()
        self._check(lib().locgpu_icp_align(self._h, s.ctypes.data, s.shape[0], s.strides[0], _pose(init_pose).ctypes.data, ctypes.byref(opts),
                                           out.ctypes.data, ctypes.byref(st)))
        return out, _stats_dict(st)

    # ---- IcpRegistration::ScanMatch whole: pose + output cloud (icp_registration.cpp:216-244)
    def icp_scan_match(self, src, init_pose, opts, in_place=False):
        """Returns (pose, stats, output cloud). The output cloud has the source's layout: every field of the source point, x, y, z
        replaced (pcl::transformPointCloud); in_place=True hands the source array itself as the output cloud."""
        s = _cloud(src)
        cloud = s if in_place else np.full_like(s, np.nan)
        out = np.zeros(7)
        st = AlignStats()
        self._check(lib().locgpu_icp_scan_match(self._h, s.ctypes.data, s.shape[0], s.strides[0], _pose(init_pose).ctypes.data, ctypes.byref(opts),
                                                out.ctypes.data, ctypes.byref(st), cloud.ctypes.data, cloud.strides[0], None, None))
        return out, _stats_dict(st), cloud

    # ---- NdtRegistration::ScanMatch whole (ndt_registration.cpp:238-261); result_pose in-out (status 1 leaves it as handed in)
    def ndt_scan_match(self, src, init_pose, result_pose=None):
        s = _cloud(src)
        cloud = np.full_like(s, np.nan)
        out = np.array(_pose(result_pose if result_pose is not None else init_pose), copy=True)
        st = AlignStats()
        self._check(lib().locgpu_ndt_scan_match(self._h, s.ctypes.data, s.shape[0], s.strides[0], _pose(init_pose).ctypes.data, out.ctypes.data,
                                                ctypes.byref(st), cloud.ctypes.data, cloud.strides[0], None, None))
        return out, _stats_dict(st), cloud

    # ---- pcl::transformPointCloud(src, out, pose.matrix().cast<float>())
    def transform_cloud(self, pose, src):
        s = _cloud(src)
        out = s.copy()
        self._check(lib().locgpu_transform_cloud(self._h, _pose(pose).ctypes.data, s.ctypes.data, s.shape[0], s.strides[0], out.ctypes.data,
                                                 out.strides[0]))
        return out

    # ---- filters either side of the matcher, host-pointer one-shots ([n, 4] float32 x, y, z, intensity)
    def _one_shot(self, fn, cloud, is_dense, *extra):
        a = _xyzi(cloud)
        out = np.zeros_like(a)
        n, d = ctypes.c_size_t(0), ctypes.c_int(0)
        self._check(fn(self._h, a.ctypes.data, a.shape[0], 16, 12, int(is_dense), *extra, out.ctypes.data, ctypes.byref(n), ctypes.byref(d)))
        return out[:n.value].copy(), bool(d.value)

    def voxel_filter(self, cloud, leaf, is_dense=True):
        return self._one_shot(lib().locgpu_voxel_filter, cloud, is_dense, float(leaf))

    def crop_box(self, cloud, mn, mx, is_dense=True):
        mn, mx = np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(mx, np.float32)
        return self._one_shot(lib().locgpu_crop_box, cloud, is_dense, mn.ctypes.data, mx.ctypes.data)

    def remove_nan(self, cloud, is_dense):
        return self._one_shot(lib().locgpu_remove_nan, cloud, is_dense)

    def loam_extract_full(self, full_points, num_scan=16):
        """One-shot on the reference's FullPointType records (64-byte structured array: x,y,z @0, uint8 intensity @24, ring @25)."""
        a = np.ascontiguousarray(full_points)
        if a.dtype.itemsize != 64:
            raise ValueError("FullPointType records are 64 bytes")
        n = len(a)
        edge, surf = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
        ne, ns = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._check(lib().locgpu_loam_extract(self._h, a.ctypes.data, n, 64, 24, 1, 25, int(num_scan), edge.ctypes.data, ctypes.byref(ne),
                                              surf.ctypes.data, ctypes.byref(ns), 16, 12))
        return edge[:ne.value].copy(), surf[:ns.value].copy()

    # ---- matcher entry points on resident clouds
    def icp_set_target_cloud(self, cloud, wait=True):
        """wait=False: the host tree build runs on a worker thread; the next call that reads the ICP target completes the ingest."""
        fn = lib().locgpu_icp_set_target_cloud if wait else lib().locgpu_icp_set_target_cloud_async
        self._check(fn(self._h, cloud._h))

    def ndt_set_target_cloud(self, cloud, opts=None):
        self._check(lib().locgpu_ndt_set_target_cloud(self._h, cloud._h, ctypes.byref(opts) if opts is not None else None))

    def icp_align_cloud(self, cloud, init_pose, opts):
        out = np.zeros(7)
        st = AlignStats()
        self._check(lib().locgpu_icp_align_cloud(self._h, cloud._h, _pose(init_pose).ctypes.data, ctypes.byref(opts), out.ctypes.data, ctypes.byref(st)))
        return out, _stats_dict(st)

    def ndt_align_cloud(self, cloud, init_pose):
        out = np.array(_pose(init_pose), copy=True)
        st = AlignStats()
        self._check(lib().locgpu_ndt_align_cloud(self._h, cloud._h, _pose(init_pose).ctypes.data, out.ctypes.data, ctypes.byref(st)))
        return out, _stats_dict(st)

    # ---- several GPUs of one node: RCCL communicator + sharded batches
    def comm_init(self, rank, world, uid):
        """uid: COMM_ID_BYTES bytes from comm_unique_id() of one rank, handed to all (e.g. torch.distributed.broadcast)."""
        buf = (ctypes.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
        self._check(lib().locgpu_comm_init(self._h, int(rank), int(world), buf))

    def comm_info(self):
        r, w = ctypes.c_int(0), ctypes.c_int(1)
        self._check(lib().locgpu_comm_info(self._h, ctypes.byref(r), ctypes.byref(w)))
        return int(r.value), int(w.value)

    def icp_set_target_bcast(self, cloud, root=0):
        """Collective SetInputTarget: rank `root` builds the tree from its cloud and broadcasts it (other ranks may pass None)."""
        if cloud is None:
            self._check(lib().locgpu_icp_set_target_bcast(self._h, None, 0, 12, int(root)))
        else:
            c = _cloud(cloud)
            self._check(lib().locgpu_icp_set_target_bcast(self._h, c.ctypes.data, c.shape[0], c.strides[0], int(root)))

    # ---- batches
    def batch(self, scans, first=None, n_total=None):
        return Batch(self, scans, first=first, n_total=n_total)

    def batch_empty(self, n_scans, max_points):
        return Batch(self, None, n_scans=n_scans, max_points=max_points)

    def icp_align_batch(self, batch, init_poses, opts):
        batch.upload_wait()
        ip = _pose(init_poses).reshape(batch.n_scans, 7)
        out = np.zeros_like(ip)
        st = (AlignStats * batch.n_scans)()
        self._check(lib().locgpu_icp_align_batch(self._h, batch._h, ip.ctypes.data, ctypes.byref(opts), out.ctypes.data, st))
        return out, [_stats_dict(s) for s in st]

    def ndt_align_batch(self, batch, init_poses):
        batch.upload_wait()
        ip = _pose(init_poses).reshape(batch.n_scans, 7)
        out = np.zeros_like(ip)
        st = (AlignStats * batch.n_scans)()
        self._check(lib().locgpu_ndt_align_batch(self._h, batch._h, ip.ctypes.data, out.ctypes.data, st))
        return out, [_stats_dict(s) for s in st]

    # two batches in flight (locgpu.h: *_align_batch_begin / locgpu_align_batch_end)
    def icp_align_batch_begin(self, batch, init_poses, opts):
        batch.upload_wait()
        ip = _pose(init_poses).reshape(batch.n_scans, 7)
        self._check(lib().locgpu_icp_align_batch_begin(self._h, batch._h, ip.ctypes.data, ctypes.byref(opts)))

    def ndt_align_batch_begin(self, batch, init_poses):
        batch.upload_wait()
        ip = _pose(init_poses).reshape(batch.n_scans, 7)
        self._check(lib().locgpu_ndt_align_batch_begin(self._h, batch._h, ip.ctypes.data))

    def align_batch_end(self, batch):
        out = np.zeros((batch.n_scans, 7))
        st = (AlignStats * batch.n_scans)()
        self._check(lib().locgpu_align_batch_end(self._h, batch._h, out.ctypes.data, st))
        return out, [_stats_dict(s) for s in st]

    def icp_hb_batch(self, batch, poses, opts):
        batch.upload_wait()
        p = _pose(poses).reshape(batch.n_scans, 7)
        hb = np.zeros((batch.n_scans, 44))
        self._check(lib().locgpu_icp_hb_batch(self._h, batch._h, p.ctypes.data, ctypes.byref(opts), hb.ctypes.data))
        return hb

    # ---- NdtRegistration
    def ndt_set_target(self, cloud, opts=None):
        c = _cloud(cloud)
        self._check(lib().locgpu_ndt_set_target(self._h, c.ctypes.data, c.shape[0], c.strides[0], ctypes.byref(opts) if opts else None))

    def ndt_target_info(self):
        out = np.zeros(3, dtype=np.int64)
        self._check(lib().locgpu_ndt_target_info(self._h, out.ctypes.data))
        return dict(num_voxels=int(out[0]), capacity=int(out[1]), bytes=int(out[2]))

    def ndt_dump(self):
        n = self.ndt_target_info()["num_voxels"]
        keys = np.zeros((max(n, 1), 3), dtype=np.int32)
        mu = np.zeros((max(n, 1), 3))
        info = np.zeros((max(n, 1), 9))
        n_out = ctypes.c_size_t(0)
        self._check(lib().locgpu_ndt_dump(self._h, keys.ctypes.data, mu.ctypes.data, info.ctypes.data, n, ctypes.byref(n_out)))
        return keys[:n], mu[:n], info[:n].reshape(n, 3, 3)

    def ndt_align(self, src, init_pose):
        s = _cloud(src)
        out = np.zeros(7)
        st = AlignStats()
        self._check(lib().locgpu_ndt_align(self._h, s.ctypes.data, s.shape[0], s.strides[0], _pose(init_pose).ctypes.data, out.ctypes.data,
                                           ctypes.byref(st)))
        return out, _stats_dict(st)

    def graph_enable(self, on=True):
        """Replay a captured hipGraph of all Gauss–Newton iterations per align call (BASELINE config 5)."""
        self._check(lib().locgpu_graph_enable(self._h, int(on)))

    # ---- measurement hooks
    def profile_enable(self, on=True):
        self._check(lib().locgpu_profile_enable(self._h, int(on)))

    def profile_read(self, reset=True):
        out = np.zeros(6)
        self._check(lib().locgpu_profile_read(self._h, out.ctypes.data, int(reset)))
        return dict(search_ms=out[0], accum_ms=out[1], solve_ms=out[2], search_n=int(out[3]), accum_n=int(out[4]), solve_n=int(out[5]))

    def visit_count_enable(self, on=True):
        self._check(lib().locgpu_visit_count_enable(self._h, int(on)))

    def visit_count_read(self, reset=True):
        out = np.zeros(4, dtype=np.uint64)
        self._check(lib().locgpu_visit_count_read(self._h, out.ctypes.data, int(reset)))
        return dict(nodes=int(out[0]), leaves=int(out[1]), queries=int(out[2]), distinct_slots=int(out[3]))


    def debug_batch_nn(self, batch, k=5):
        """Neighbour lists of the batch's most recent search stage as target point indices, [n_scans, max_points, k] (-1 = none)."""
        out = np.empty((batch.n_local, batch.max_points, k), dtype=np.int32)
        self._check(lib().locgpu_debug_batch_nn(self._h, batch._h, int(k), out.ctypes.data))
        return out  # rows beyond a scan's point count are stale

    def search_stats_read(self, reset=True):
        out = np.zeros(4, dtype=np.uint64)
        self._check(lib().locgpu_search_stats_read(self._h, out.ctypes.data, int(reset)))
        return dict(searched=int(out[0]), redone=int(out[1]), walked=int(out[2]), replayed=int(out[3]))


def _xyzi(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError("cloud must be [n, 4] float32 (x, y, z, intensity)")
    return a


class Cloud:
    """A cloud resident in HBM (locgpu_cloud): float32 x, y, z, intensity per point + PCL's is_dense flag."""

    def __init__(self, ctx, points=None, is_dense=True, _borrowed=None):
        self.ctx = ctx
        self._owned = _borrowed is None
        if _borrowed is not None:
            self._h = _borrowed
        else:
            self._h = ctypes.c_void_p()
            ctx._check(lib().locgpu_cloud_create(ctx._h, ctypes.byref(self._h)))
            if points is not None:
                self.upload(points, is_dense)

    def upload(self, points, is_dense=True):
        a = _xyzi(points)
        self.ctx._check(lib().locgpu_cloud_upload(self._h, a.ctypes.data, a.shape[0], 16, 12, int(is_dense)))
        return self

    @property
    def info(self):
        n, d = ctypes.c_size_t(0), ctypes.c_int(0)
        self.ctx._check(lib().locgpu_cloud_info(self._h, ctypes.byref(n), ctypes.byref(d)))
        return int(n.value), bool(d.value)

    def __len__(self):
        return self.info[0]

    @property
    def is_dense(self):
        return self.info[1]

    def download(self):
        n = len(self)
        out = np.zeros((n, 4), np.float32)
        self.ctx._check(lib().locgpu_cloud_download(self._h, out.ctypes.data, n, 16, 12))
        return out

    def _out(self, out):
        return out if out is not None else Cloud(self.ctx)

    def copy(self, out=None):
        out = self._out(out)
        self.ctx._check(lib().locgpu_cloud_copy(self._h, out._h))
        return out

    def remove_nan(self, out=None):
        out = self._out(out)
        self.ctx._check(lib().locgpu_cloud_remove_nan(self._h, out._h))
        return out

    def voxel_filter(self, leaf, out=None, with_passthrough=False):
        out = self._out(out)
        pt = ctypes.c_int(0)
        self.ctx._check(lib().locgpu_cloud_voxel_filter(self._h, float(leaf), out._h, ctypes.byref(pt)))
        return (out, bool(pt.value)) if with_passthrough else out

    def crop_box(self, mn, mx, out=None):
        out = self._out(out)
        mn, mx = np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(mx, np.float32)
        self.ctx._check(lib().locgpu_cloud_crop_box(self._h, mn.ctypes.data, mx.ctypes.data, out._h))
        return out

    def transform(self, pose, out=None):
        out = self._out(out)
        self.ctx._check(lib().locgpu_cloud_transform(self._h, _pose(pose).ctypes.data, out._h))
        return out

    def append(self, other):
        self.ctx._check(lib().locgpu_cloud_append(self._h, other._h))
        return self

    def loam_extract(self, ring, num_scan=16):
        """LoamFeatureExtract::Extract on a resident cloud: returns (edge, surf) resident clouds."""
        ring = np.ascontiguousarray(ring, dtype=np.uint8)
        if len(ring) != len(self):
            raise ValueError("one ring byte per point")
        edge, surf = Cloud(self.ctx), Cloud(self.ctx)
        self.ctx._check(lib().locgpu_cloud_loam_extract(self._h, ring.ctypes.data, int(num_scan), edge._h, surf._h))
        return edge, surf

    def close(self):
        if self._owned and getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().locgpu_cloud_destroy(self._h)
        self._h = None

    def __del__(self):
        self.close()


class Submap:
    """Lio::AddCloud's keyframe local map in HBM (locgpu_submap)."""

    def __init__(self, ctx, num_kfs, leaf):
        self.ctx = ctx
        self._h = ctypes.c_void_p()
        ctx._check(lib().locgpu_submap_create(ctx._h, int(num_kfs), float(leaf), ctypes.byref(self._h)))

    def add_keyframe(self, scan, pose=None):
        self.ctx._check(lib().locgpu_submap_add_keyframe(self._h, scan._h, _pose(pose).ctypes.data if pose is not None else None))

    def cloud(self):
        h = ctypes.c_void_p()
        self.ctx._check(lib().locgpu_submap_cloud(self._h, ctypes.byref(h)))
        return Cloud(self.ctx, _borrowed=h)

    def last_keyframe(self):
        h = ctypes.c_void_p()
        self.ctx._check(lib().locgpu_submap_last_keyframe(self._h, ctypes.byref(h)))
        return Cloud(self.ctx, _borrowed=h)

    @property
    def info(self):
        k, n = ctypes.c_int(0), ctypes.c_size_t(0)
        self.ctx._check(lib().locgpu_submap_info(self._h, ctypes.byref(k), ctypes.byref(n)))
        return int(k.value), int(n.value)

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().locgpu_submap_destroy(self._h)
        self._h = None

    def __del__(self):
        self.close()


class MarshalledScans:
    """The (pointer, count) arrays of a list of host scans, built once: what a C/C++ caller of locgpu_batch_upload_async holds
    anyway. Pass it wherever a list of scans is accepted to skip the per-call Python marshalling (≈0.4 ms for 256 scans)."""

    def __init__(self, scans):
        self.scans, self.ptrs, self.cnts, self.stride = Batch._marshal(list(scans))

    def __len__(self):
        return len(self.scans)


class Batch:
    """A batch of scans resident in HBM (locgpu_batch). ``n_scans`` = the scans poses are kept for (all n_total of a sharded batch)."""

    def __init__(self, ctx, scans, first=None, n_total=None, n_scans=None, max_points=None):
        self.ctx = ctx
        self._h = ctypes.c_void_p()
        self._keep = None
        if scans is None:  # capacity only; fill with upload_async
            self.n_local = self.n_scans = int(n_scans)
            self.max_points = int(max_points)
            ctx._check(lib().locgpu_batch_create_empty(ctx._h, self.n_scans, int(max_points), ctypes.byref(self._h)))
            return
        if n_total is not None and len(scans) == 0:  # a rank that holds none of the batch's scans still takes part in its collectives
            self.n_local, self.max_points, self.n_scans = 0, 0, int(n_total)
            ctx._check(lib().locgpu_batch_create_sharded(ctx._h, None, None, 16, 0, int(first or 0), self.n_scans, ctypes.byref(self._h)))
            return
        scans, ptrs, cnts, stride = self._marshal(scans)
        self.n_local = len(scans)
        self.max_points = max(s.shape[0] for s in scans)
        if n_total is None:
            self.n_scans = self.n_local
            ctx._check(lib().locgpu_batch_create(ctx._h, ptrs, cnts, stride, self.n_local, ctypes.byref(self._h)))
        else:  # sharded: this rank holds scans [first, first + len(scans)) of n_total
            self.n_scans = int(n_total)
            ctx._check(lib().locgpu_batch_create_sharded(ctx._h, ptrs, cnts, stride, self.n_local, int(first or 0), self.n_scans, ctypes.byref(self._h)))

    @staticmethod
    def _marshal(scans):
        if isinstance(scans, MarshalledScans):
            return scans.scans, scans.ptrs, scans.cnts, scans.stride
        scans = [_cloud(s) for s in scans]
        if not scans:
            raise ValueError("empty batch")
        stride = scans[0].strides[0]
        if any(s.strides[0] != stride for s in scans):
            raise ValueError("all scans of a batch must share one point stride")
        ptrs = (ctypes.c_void_p * len(scans))(*[s.ctypes.data for s in scans])
        cnts = (ctypes.c_size_t * len(scans))(*[s.shape[0] for s in scans])
        return scans, ptrs, cnts, stride

    def upload_async(self, scans):
        """Replace the batch's scans; returns at once (the copy runs beside the GPU's current work). The arrays are kept
        alive here until upload_wait() / the next align call on this batch."""
        self.upload_wait()
        scans, ptrs, cnts, stride = self._marshal(scans)
        if len(scans) != self.n_local:
            raise ValueError("upload_async needs %d scans" % self.n_local)
        self._keep = (scans, ptrs, cnts)
        self.ctx._check(lib().locgpu_batch_upload_async(self._h, ptrs, cnts, stride))

    def upload_wait(self):
        if self._keep is not None:
            rc = lib().locgpu_batch_upload_wait(self._h)
            self._keep = None
            self.ctx._check(rc)

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().locgpu_batch_destroy(self._h)
        self._h = None
        self._keep = None

    def __del__(self):
        self.close()


class Pool:
    """The open-scan pool (locgpu_pool): jobs of scans submitted at any time, iterated together, collected by ticket."""

    def __init__(self, ctx, slots, max_points, scans_per_job=0, chunk=0, opts=None, ndt=False, prefetch=-1):
        self.ctx = ctx
        self._h = ctypes.c_void_p()
        self._keep, self._n = {}, {}
        o = PoolOpts()
        lib().locgpu_pool_opts_default(ctypes.byref(o))
        o.slots, o.scans_per_job, o.chunk, o.matcher, o.max_points = int(slots), int(scans_per_job), int(chunk), (1 if ndt else 0), int(max_points)
        o.prefetch = int(prefetch)
        if opts is not None:
            o.icp = opts
        ctx._check(lib().locgpu_pool_create(ctx._h, ctypes.byref(o), ctypes.byref(self._h)))

    def submit(self, scans, init_poses, first=0, n_total=None):
        """scans: the scans this rank holds of the job (a list or MarshalledScans; may be empty on a rank of a sharded job);
        init_poses: [n_total, 7]. Returns the ticket."""
        n_total = len(scans) if n_total is None else int(n_total)
        ip = _pose(init_poses).reshape(n_total, 7)
        t = ctypes.c_int64(0)
        if len(scans):
            sc, ptrs, cnts, stride = Batch._marshal(scans)
        else:
            sc, ptrs, cnts, stride = [], None, None, 16
        rc = lib().locgpu_pool_submit(self._h, ptrs, cnts, stride, len(sc), int(first), n_total, ip.ctypes.data, ctypes.byref(t))
        if t.value:  # the job was accepted (a ticket was handed out): whatever rc says, the upload service may be reading the clouds
            self._keep[t.value] = (sc, ptrs, cnts)  # they stay alive until the ticket has been waited for
            self._n[t.value] = n_total
        self.ctx._check(rc)
        return t.value

    def wait(self, ticket):
        n = self._n.pop(ticket, None)
        if n is None:
            raise LocGpuError(-1, "pool.wait: unknown ticket (a ticket is good once)")
        out = np.zeros((n, 7))
        st = (AlignStats * n)()
        rc = lib().locgpu_pool_wait(self._h, ctypes.c_int64(ticket), out.ctypes.data, st)
        self._keep.pop(ticket, None)
        self.ctx._check(rc)
        return out, [_stats_dict(s) for s in st]

    def step(self, block=True):
        """One turn of the pool (look at the chunk in flight, scans out, jobs in, next chunk) without collecting a job."""
        self.ctx._check(lib().locgpu_pool_step(self._h, 1 if block else 0))

    def done(self, ticket):
        d = ctypes.c_int(0)
        self.ctx._check(lib().locgpu_pool_done(self._h, ctypes.c_int64(ticket), ctypes.byref(d)))
        return bool(d.value)

    def info(self):
        out = (ctypes.c_int64 * 8)()
        self.ctx._check(lib().locgpu_pool_info(self._h, out))
        return dict(slots=out[0], free=out[1], jobs=out[2], iterations=out[3], scan_iterations=out[4], open=out[5], regions=out[6], free_regions=out[7])

    def profile_read(self, reset=True):
        out = (ctypes.c_double * 2)()
        self.ctx._check(lib().locgpu_pool_profile_read(self._h, out, 1 if reset else 0))
        return dict(chunk_ms=out[0], chunks=int(out[1]))

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().locgpu_pool_destroy(self._h)
        self._h = None
        self._keep = {}

    def __del__(self):
        self.close()


def gn_update(hb, method, min_effective_pts, eps, pose):
    """Host-side Gauss–Newton update on reduced normal equations (point-sharded multi-GPU mode)."""
    hb = np.ascontiguousarray(hb, dtype=np.float64).reshape(44)
    p = np.array(_pose(pose), copy=True)
    dx = np.zeros(6)
    applied, stop = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib().locgpu_gn_update(hb.ctypes.data, method, min_effective_pts, eps, p.ctypes.data, dx.ctypes.data, ctypes.byref(applied),
                                ctypes.byref(stop))
    if rc != 0:
        raise LocGpuError(rc, "gn_update failed")
    return p, dx, bool(applied.value), bool(stop.value)
