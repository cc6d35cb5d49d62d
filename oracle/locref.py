"""ctypes binding of the CPU oracle (``oracle/locref.cpp``).

TEST INFRASTRUCTURE ONLY: importable from ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — never from ``loc_lib_amd``. PARITY UNPINNED (see locref.cpp).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblocref.so")

P2P, P2LINE, P2PLANE = 0, 1, 2
DIRECT_NDT, INCREMENTAL_NDT = 1, 2
CENTER, NEARBY6 = 0, 1
TRACE_W = 50  # H36 B6 dx6 eff ok


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("locref.cpp", "locref_math.hpp", "locref_kdtree.hpp", "locref_flat.hpp", "locref_filters.hpp", "locref_loam.hpp")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblocref.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_vp, _sz, _i, _f, _d = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_float, ctypes.c_double


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        sig = {
            "locref_kdtree_create": (_vp, [_vp, _sz, _sz]),
            "locref_kdtree_destroy": (None, [_vp]),
            "locref_kdtree_info": (None, [_vp, _vp]),
            "locref_kdtree_knn": (None, [_vp, _vp, _sz, _i, _i, _f, _vp, _vp]),
            "locref_kdtree_dump": (_sz, [_vp, _vp, _vp, _vp, _sz]),
            "locref_fit_plane": (_i, [_vp, _i, _vp]),
            "locref_fit_line": (_i, [_vp, _i, _d, _vp, _vp]),
            "locref_clamped_info": (None, [_vp, _vp]),
            "locref_lu6": (_d, [_vp, _vp, _vp]),
            "locref_apply_update": (None, [_vp, _vp]),
            "locref_transform_points_f64": (None, [_vp, _vp, _sz, _vp]),
            "locref_icp_create": (_vp, [_i, _vp, _i, _f]),
            "locref_icp_destroy": (None, [_vp]),
            "locref_icp_set_target": (None, [_vp, _vp, _sz, _sz]),
            "locref_icp_tree_info": (None, [_vp, _vp]),
            "locref_icp_hb": (_i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp]),
            "locref_icp_align": (_i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _i, _vp]),
            "locref_icp_align_flat": (None, [_vp, _vp, _vp, _sz, _i, _vp, _vp, _vp, _i]),
            "locref_ndt_create": (_vp, [_vp]),
            "locref_ndt_destroy": (None, [_vp]),
            "locref_ndt_set_target": (None, [_vp, _vp, _sz, _sz]),
            "locref_ndt_num_voxels": (_sz, [_vp]),
            "locref_ndt_dump": (_sz, [_vp, _vp, _vp, _vp, _sz]),
            "locref_ndt_align": (_i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _i, _vp]),
            "locref_transform_cloud_f32": (None, [_vp, _vp, _sz, _sz, _vp, _sz]),
            "locref_remove_nan": (_sz, [_vp, _sz, _i, _vp]),
            "locref_crop_box": (_sz, [_vp, _sz, _i, _vp, _vp, _vp]),
            "locref_box_edges": (None, [_vp, _vp, _vp, _vp]),
            "locref_voxel_grid": (_sz, [_vp, _sz, _i, _f, _i, _vp, _vp]),
            "locref_transform_cloud_f64": (None, [_vp, _vp, _sz, _i, _vp]),
            "locref_loam_extract": (None, [_vp, _vp, _sz, _i, _i, _vp, _vp, _vp, _vp]),
            "locref_localmap_create": (_vp, [_sz, _f, _i]),
            "locref_localmap_destroy": (None, [_vp]),
            "locref_localmap_add_keyframe": (None, [_vp, _vp, _sz, _i]),
            "locref_localmap_size": (_sz, [_vp]),
            "locref_localmap_dense": (_i, [_vp]),
            "locref_localmap_copy": (None, [_vp, _vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] >= 3
    return a


def _pose(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    assert p.shape == (7,)
    return p


class KdTree:
    """Reference KD-tree (kdtree.cpp) over float32 xyz."""

    def __init__(self, xyz):
        xyz = _f32(xyz)
        self._h = lib().locref_kdtree_create(xyz.ctypes.data, xyz.shape[0], xyz.shape[1])
        if not self._h:
            raise ValueError("empty cloud")
        info = np.zeros(3, dtype=np.int64)
        lib().locref_kdtree_info(self._h, info.ctypes.data)
        self.num_leaves, self.num_nodes, self.depth = (int(v) for v in info)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().locref_kdtree_destroy(self._h)
            self._h = None

    def knn(self, queries, k=5, approximate=True, alpha=0.1, with_stats=False):
        q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32)[:, :3])
        out = np.empty((q.shape[0], k), dtype=np.int32)
        st = np.zeros(2, dtype=np.uint64)
        lib().locref_kdtree_knn(self._h, q.ctypes.data, q.shape[0], k, int(approximate), alpha, out.ctypes.data,
                                st.ctypes.data if with_stats else None)
        return (out, st) if with_stats else out

    def dump(self):
        n = self.num_nodes
        axis = np.empty(n, dtype=np.int32)
        th = np.empty(n, dtype=np.float32)
        pidx = np.empty(n, dtype=np.int32)
        m = lib().locref_kdtree_dump(self._h, axis.ctypes.data, th.ctypes.data, pidx.ctypes.data, n)
        assert m == n
        return axis, th, pidx


def fit_plane(pts):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    out = np.zeros(4)
    ok = lib().locref_fit_plane(pts.ctypes.data, pts.shape[0], out.ctypes.data)
    return bool(ok), out


def fit_line(pts, eps=0.5):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    o, d = np.zeros(3), np.zeros(3)
    ok = lib().locref_fit_line(pts.ctypes.data, pts.shape[0], eps, o.ctypes.data, d.ctypes.data)
    return bool(ok), o, d


def clamped_info(sigma):
    s = np.ascontiguousarray(sigma, dtype=np.float64).reshape(9)
    out = np.zeros(9)
    lib().locref_clamped_info(s.ctypes.data, out.ctypes.data)
    return out.reshape(3, 3)


def lu6(H, b):
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(36)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(6)
    det = lib().locref_lu6(H.ctypes.data, b.ctypes.data, x.ctypes.data)
    return det, x


def apply_update(pose, dx):
    p = _pose(pose).copy()
    dx = np.ascontiguousarray(dx, dtype=np.float64)
    lib().locref_apply_update(p.ctypes.data, dx.ctypes.data)
    return p


def transform_points(pose, pts):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    out = np.empty_like(pts)
    lib().locref_transform_points_f64(_pose(pose).ctypes.data, pts.ctypes.data, pts.shape[0], out.ctypes.data)
    return out


def transform_cloud_f32(pose, cloud):
    c = _f32(cloud)
    out = c.copy()
    lib().locref_transform_cloud_f32(_pose(pose).ctypes.data, c.ctypes.data, c.shape[0], c.shape[1], out.ctypes.data, out.shape[1])
    return out


class Icp:
    """IcpRegistration restated (icp_registration.cpp). ``opts`` keys follow IcpOptions (hpp:29-37)."""

    def __init__(self, method=P2PLANE, use_ann=True, alpha=0.1, **opts):
        o = dict(max_iteration=20, max_nn_distance=1.0, max_plane_distance=0.1, max_line_distance=0.5, min_effective_pts=10, eps=1e-2)
        o.update(opts)
        arr = np.array([o["max_iteration"], o["max_nn_distance"], o["max_plane_distance"], o["max_line_distance"],
                        o["min_effective_pts"], o["eps"]], dtype=np.float64)
        self._h = lib().locref_icp_create(method, arr.ctypes.data, int(use_ann), alpha)
        self.method = method

    def __del__(self):
        if getattr(self, "_h", None):
            lib().locref_icp_destroy(self._h)
            self._h = None

    def set_target(self, xyz):
        xyz = _f32(xyz)
        lib().locref_icp_set_target(self._h, xyz.ctypes.data, xyz.shape[0], xyz.shape[1])

    def tree_info(self):
        info = np.zeros(3, dtype=np.int64)
        lib().locref_icp_tree_info(self._h, info.ctypes.data)
        return dict(num_leaves=int(info[0]), num_nodes=int(info[1]), depth=int(info[2]))

    def hb(self, src, pose):
        src = _f32(src)
        H, B, eff = np.zeros(36), np.zeros(6), np.zeros(1)
        ok = lib().locref_icp_hb(self._h, src.ctypes.data, src.shape[0], src.shape[1], _pose(pose).ctypes.data, H.ctypes.data,
                                 B.ctypes.data, eff.ctypes.data)
        return bool(ok), H.reshape(6, 6), B, int(eff[0])

    def align(self, src, init_pose, trace_cap=20, with_stats=False):
        src = _f32(src)
        out = np.zeros(7)
        trace = np.zeros((trace_cap, TRACE_W))
        st = np.zeros(2, dtype=np.uint64)
        iters = lib().locref_icp_align(self._h, src.ctypes.data, src.shape[0], src.shape[1], _pose(init_pose).ctypes.data,
                                       out.ctypes.data, trace.ctypes.data, trace_cap, st.ctypes.data if with_stats else None)
        res = dict(pose=out, iters=iters, trace=trace[:min(iters, trace_cap)])
        if with_stats:
            res["nodes_visited"], res["leaves_visited"] = int(st[0]), int(st[1])
        return res


    def align_flat(self, scans, init_poses, threads=1):
        """BASELINE.md R2 (threads=1) / R3 (threads>1): the point-to-plane path through the flat-array port (locref_flat.hpp),
        whole scans dealt to native threads. Returns (poses [n, 7], iterations [n]); bit-identical to align()."""
        scans = [_f32(s) for s in scans]
        width = scans[0].shape[1]
        assert all(s.shape[1] == width for s in scans)
        n = len(scans)
        ptrs = (ctypes.c_void_p * n)(*[s.ctypes.data for s in scans])
        cnts = (ctypes.c_size_t * n)(*[s.shape[0] for s in scans])
        inits = np.ascontiguousarray(np.asarray(init_poses, dtype=np.float64).reshape(n, 7))
        out = np.zeros((n, 7))
        iters = np.zeros(n, dtype=np.int32)
        lib().locref_icp_align_flat(self._h, ptrs, cnts, width, n, inits.ctypes.data, out.ctypes.data, iters.ctypes.data, int(threads))
        return out, iters


class Ndt:
    """NdtRegistration restated (ndt_registration.cpp). ``opts`` keys follow NdtOptions (hpp:27-42)."""

    def __init__(self, method=DIRECT_NDT, nearby_type=NEARBY6, **opts):
        o = dict(max_iteration=20, voxel_size=1.0, min_effective_pts=10, min_pts_in_voxel=3, eps=1e-2, res_outlier_th=20.0,
                 capacity=100000)
        o.update(opts)
        arr = np.array([o["max_iteration"], o["voxel_size"], o["min_effective_pts"], o["min_pts_in_voxel"], o["eps"],
                        o["res_outlier_th"], o["capacity"], nearby_type, method], dtype=np.float64)
        self._h = lib().locref_ndt_create(arr.ctypes.data)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().locref_ndt_destroy(self._h)
            self._h = None

    def set_target(self, xyz):
        xyz = _f32(xyz)
        lib().locref_ndt_set_target(self._h, xyz.ctypes.data, xyz.shape[0], xyz.shape[1])

    def num_voxels(self):
        return int(lib().locref_ndt_num_voxels(self._h))

    def dump(self):
        n = self.num_voxels()
        keys = np.zeros((n, 3), dtype=np.int32)
        mu = np.zeros((n, 3))
        info = np.zeros((n, 9))
        lib().locref_ndt_dump(self._h, keys.ctypes.data, mu.ctypes.data, info.ctypes.data, n)
        return keys, mu, info.reshape(n, 3, 3)

    def align(self, src, init_pose, trace_cap=20):
        src = _f32(src)
        out = np.array(_pose(init_pose), copy=True)  # status 1 leaves result untouched; caller sees init
        written = np.zeros(7)
        trace = np.zeros((trace_cap, TRACE_W))
        iters = ctypes.c_int(0)
        st = lib().locref_ndt_align(self._h, src.ctypes.data, src.shape[0], src.shape[1], _pose(init_pose).ctypes.data,
                                    written.ctypes.data, trace.ctypes.data, trace_cap, ctypes.byref(iters))
        if st != 1:
            out = written
        return dict(pose=out, status=st, iters=iters.value, trace=trace[:min(iters.value, trace_cap)])


# ---------------------------------------------------------------------------------------------
# Cloud filters either side of the matcher (locref_filters.hpp). Clouds are float32 [n, 4] = x, y, z, intensity.
SORT_STD, SORT_STABLE = 0, 1


def _xyzi(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 4
    return a


def remove_nan(cloud, is_dense):
    """pcl::removeNaNFromPointCloud as RemoveNanPoint calls it (point_cloud_utils.h:13-20)."""
    cloud = _xyzi(cloud)
    out = np.empty_like(cloud)
    m = lib().locref_remove_nan(cloud.ctypes.data, len(cloud), int(is_dense), out.ctypes.data)
    return out[:m].copy()


def box_edges(step, origin):
    """BoxFilter::CalculateEdge (box_filter.cpp:59-66): float32 (min, max) corners."""
    step, origin = np.asarray(step, np.float32), np.asarray(origin, np.float32)
    mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().locref_box_edges(step.ctypes.data, origin.ctypes.data, mn.ctypes.data, mx.ctypes.data)
    return mn, mx


def crop_box(cloud, is_dense, mn, mx):
    """BoxFilter::Filter → pcl::CropBox (box_filter.cpp:25-32)."""
    cloud = _xyzi(cloud)
    mn, mx = np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(mx, np.float32)
    out = np.empty_like(cloud)
    m = lib().locref_crop_box(cloud.ctypes.data, len(cloud), int(is_dense), mn.ctypes.data, mx.ctypes.data, out.ctypes.data)
    return out[:m].copy()


def voxel_grid(cloud, is_dense, leaf, order=SORT_STD, with_info=False):
    """VoxelFilter::Filter → pcl::VoxelGrid (voxel_filter.cpp:19-25). Returns centroids in voxel-index order."""
    cloud = _xyzi(cloud)
    out = np.empty_like(cloud)
    info = np.zeros(7, np.int32)
    m = lib().locref_voxel_grid(cloud.ctypes.data, len(cloud), int(is_dense), float(leaf), int(order), out.ctypes.data, info.ctypes.data)
    res = out[:m].copy()
    if with_info:
        return res, dict(status=int(info[0]), min_b=info[1:4].copy(), div_b=info[4:7].copy())
    return res


def transform_cloud_f64(pose, cloud, is_dense=True):
    """pcl::transformPointCloud with a double 4x4 (lio.cpp:244,279): double arithmetic, float32 result."""
    cloud = _xyzi(cloud)
    out = np.empty_like(cloud)
    lib().locref_transform_cloud_f64(_pose(pose).ctypes.data, cloud.ctypes.data, len(cloud), int(is_dense), out.ctypes.data)
    return out


class LocalMap:
    """Keyframe branch of Lio::AddCloud (lio.cpp:268-306): the matching target after each keyframe."""

    def __init__(self, num_kfs, leaf, order=SORT_STD):
        self._h = lib().locref_localmap_create(int(num_kfs), float(leaf), int(order))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().locref_localmap_destroy(self._h)
            self._h = None

    def add_keyframe(self, cloud, is_dense=True):
        cloud = _xyzi(cloud)
        lib().locref_localmap_add_keyframe(self._h, cloud.ctypes.data, len(cloud), int(is_dense))

    @property
    def is_dense(self):
        return bool(lib().locref_localmap_dense(self._h))

    def cloud(self):
        out = np.empty((lib().locref_localmap_size(self._h), 4), np.float32)
        lib().locref_localmap_copy(self._h, out.ctypes.data)
        return out


def loam_extract(cloud, ring, num_scan=16, order=SORT_STD):
    """LoamFeatureExtract::Extract (loam_feature_extract.cpp:19-91): returns (edge, surf) clouds [m, 4]."""
    cloud = _xyzi(cloud)
    ring = np.ascontiguousarray(ring, dtype=np.uint8)
    assert len(ring) == len(cloud)
    edge, surf = np.empty_like(cloud), np.empty_like(cloud)
    ne, ns = ctypes.c_size_t(0), ctypes.c_size_t(0)
    lib().locref_loam_extract(cloud.ctypes.data, ring.ctypes.data, len(cloud), int(num_scan), int(order), edge.ctypes.data, ctypes.byref(ne),
                              surf.ctypes.data, ctypes.byref(ns))
    return edge[:ne.value].copy(), surf[:ns.value].copy()


def bfnn_knn(cloud, queries, k):
    """BfnnRegistration::FindNearstPoints (LocUtils/src/model/search_point/bfnn/bfnn.cpp:24-50) for many queries: float32 squared
    distances in Eigen's x0 + (x1 + x2) order, sorted ascending, first k indices. std::sort's order among equal distances is
    unspecified; this restatement (and the product) order them by index (numpy's stable sort)."""
    c = np.ascontiguousarray(np.asarray(cloud, dtype=np.float32)[:, :3])
    q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32)[:, :3])
    out = np.empty((len(q), k), np.int32)
    for i in range(len(q)):
        d = c - q[i]
        d2 = (d[:, 0] * d[:, 0]) + ((d[:, 1] * d[:, 1]) + (d[:, 2] * d[:, 2]))
        out[i] = np.argsort(d2, kind="stable")[:k]
    return out
