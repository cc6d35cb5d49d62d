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
LIB_PATH = os.path.join(_HERE, "liblocgpu.so")
CSRC = os.path.join(_HERE, "csrc")

P2P, P2LINE, P2PLANE = 0, 1, 2
SEARCH_TREE_FAITHFUL, SEARCH_GRID_EXACT = 0, 1
CENTER, NEARBY6 = 0, 1
DIRECT_NDT, INCREMENTAL_NDT = 1, 2

# every symbol include/locgpu.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "locgpu_icp_opts_default", "locgpu_ndt_opts_default", "locgpu_create", "locgpu_destroy", "locgpu_last_error",
    "locgpu_device_count", "locgpu_icp_set_target", "locgpu_icp_target_info", "locgpu_knn", "locgpu_icp_hb", "locgpu_icp_align",
    "locgpu_transform_cloud", "locgpu_batch_create", "locgpu_batch_destroy", "locgpu_icp_align_batch", "locgpu_ndt_align_batch",
    "locgpu_icp_hb_batch", "locgpu_gn_update", "locgpu_ndt_set_target", "locgpu_ndt_target_info", "locgpu_ndt_dump",
    "locgpu_ndt_align", "locgpu_profile_enable", "locgpu_profile_read", "locgpu_visit_count_enable", "locgpu_visit_count_read",
    "locgpu_search_stats_read", "locgpu_graph_enable",
]


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
            "locgpu_icp_set_target": (i32, [vp, vp, sz, sz]), "locgpu_icp_target_info": (i32, [vp, vp]),
            "locgpu_knn": (i32, [vp, vp, sz, i32, i32, f32, i32, vp, vp]),
            "locgpu_icp_hb": (i32, [vp, vp, sz, sz, vp, vp, vp, vp, vp, vp]),
            "locgpu_icp_align": (i32, [vp, vp, sz, sz, vp, vp, vp, vp]),
            "locgpu_transform_cloud": (i32, [vp, vp, vp, sz, sz, vp, sz]),
            "locgpu_batch_create": (i32, [vp, vp, vp, sz, i32, vp]), "locgpu_batch_destroy": (None, [vp]),
            "locgpu_icp_align_batch": (i32, [vp, vp, vp, vp, vp, vp]), "locgpu_ndt_align_batch": (i32, [vp, vp, vp, vp, vp]),
            "locgpu_icp_hb_batch": (i32, [vp, vp, vp, vp, vp]),
            "locgpu_gn_update": (i32, [vp, i32, i32, dbl, vp, vp, vp, vp]),
            "locgpu_ndt_set_target": (i32, [vp, vp, sz, sz, vp]), "locgpu_ndt_target_info": (i32, [vp, vp]),
            "locgpu_ndt_dump": (i32, [vp, vp, vp, vp, sz, vp]), "locgpu_ndt_align": (i32, [vp, vp, sz, sz, vp, vp, vp]),
            "locgpu_profile_enable": (i32, [vp, i32]), "locgpu_profile_read": (i32, [vp, vp, i32]),
            "locgpu_visit_count_enable": (i32, [vp, i32]), "locgpu_visit_count_read": (i32, [vp, vp, i32]),
            "locgpu_search_stats_read": (i32, [vp, vp, i32]), "locgpu_graph_enable": (i32, [vp, i32]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def device_count():
    return int(lib().locgpu_device_count())


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
    def icp_set_target(self, cloud):
        c = _cloud(cloud)
        self._check(lib().locgpu_icp_set_target(self._h, c.ctypes.data, c.shape[0], c.strides[0]))

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

    # ---- pcl::transformPointCloud(src, out, pose.matrix().cast<float>())
    def transform_cloud(self, pose, src):
        s = _cloud(src)
        out = s.copy()
        self._check(lib().locgpu_transform_cloud(self._h, _pose(pose).ctypes.data, s.ctypes.data, s.shape[0], s.strides[0], out.ctypes.data,
                                                 out.strides[0]))
        return out

    # ---- batches
    def batch(self, scans):
        return Batch(self, scans)

    def icp_align_batch(self, batch, init_poses, opts):
        ip = _pose(init_poses).reshape(batch.n_scans, 7)
        out = np.zeros_like(ip)
        st = (AlignStats * batch.n_scans)()
        self._check(lib().locgpu_icp_align_batch(self._h, batch._h, ip.ctypes.data, ctypes.byref(opts), out.ctypes.data, st))
        return out, [_stats_dict(s) for s in st]

    def ndt_align_batch(self, batch, init_poses):
        ip = _pose(init_poses).reshape(batch.n_scans, 7)
        out = np.zeros_like(ip)
        st = (AlignStats * batch.n_scans)()
        self._check(lib().locgpu_ndt_align_batch(self._h, batch._h, ip.ctypes.data, out.ctypes.data, st))
        return out, [_stats_dict(s) for s in st]

    def icp_hb_batch(self, batch, poses, opts):
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
        out = np.zeros(3, dtype=np.uint64)
        self._check(lib().locgpu_visit_count_read(self._h, out.ctypes.data, int(reset)))
        return dict(nodes=int(out[0]), leaves=int(out[1]), queries=int(out[2]))


    def search_stats_read(self, reset=True):
        out = np.zeros(2, dtype=np.uint64)
        self._check(lib().locgpu_search_stats_read(self._h, out.ctypes.data, int(reset)))
        return dict(searched=int(out[0]), redone=int(out[1]))


class Batch:
    """A batch of scans resident in HBM (locgpu_batch)."""

    def __init__(self, ctx, scans):
        self.ctx = ctx
        scans = [_cloud(s) for s in scans]
        if not scans:
            raise ValueError("empty batch")
        stride = scans[0].strides[0]
        if any(s.strides[0] != stride for s in scans):
            raise ValueError("all scans of a batch must share one point stride")
        self.n_scans = len(scans)
        self.counts = [s.shape[0] for s in scans]
        ptrs = (ctypes.c_void_p * self.n_scans)(*[s.ctypes.data for s in scans])
        cnts = (ctypes.c_size_t * self.n_scans)(*self.counts)
        self._h = ctypes.c_void_p()
        ctx._check(lib().locgpu_batch_create(ctx._h, ptrs, cnts, stride, self.n_scans, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().locgpu_batch_destroy(self._h)
        self._h = None

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
