"""Known-answer tests that pin the CPU oracle independently of itself.

The reference ships no golden vectors ("parity unpinned"), so the oracle is checked against closed forms and
against numpy: brute-force k-NN, numpy.linalg.svd plane/line fits, numpy solve, closed-form SE3 algebra.
"""
import os

import numpy as np
import pytest

from conftest import pose_delta


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_exact_knn_matches_bruteforce(locref):
    rng = np.random.RandomState(1)
    pts = rng.rand(5000, 3).astype(np.float32) * 20
    q = rng.rand(300, 3).astype(np.float32) * 20
    tree = locref.KdTree(pts)
    idx = tree.knn(q, k=5, approximate=False)
    d = ((q[:, None, :].astype(np.float64) - pts[None].astype(np.float64)) ** 2).sum(-1)
    ref = np.argsort(d, axis=1, kind="stable")[:, :5]
    got_d = np.take_along_axis(d, idx.astype(np.int64), 1)
    ref_d = np.take_along_axis(d, ref, 1)
    np.testing.assert_allclose(got_d, ref_d, rtol=1e-6)  # same distances (indices may differ only on exact ties)
    assert (idx == ref).mean() > 0.999


def test_ann_is_pruned_subset_semantics(locref):
    """alpha=0.1 visits fewer nodes, returns sorted distances, 1-NN almost always exact."""
    rng = np.random.RandomState(2)
    pts = rng.rand(20000, 3).astype(np.float32) * 30
    q = rng.rand(500, 3).astype(np.float32) * 30
    tree = locref.KdTree(pts)
    a, sa = tree.knn(q, k=5, approximate=True, alpha=0.1, with_stats=True)
    e, se = tree.knn(q, k=5, approximate=False, with_stats=True)
    assert sa[0] < se[0] and sa[1] < se[1]
    da = ((q[:, None] - pts[a]) ** 2).sum(-1)
    assert np.all(np.diff(da, axis=1) >= 0)
    assert (a[:, 0] == e[:, 0]).mean() > 0.95


def test_k_larger_than_tree_returns_nothing(locref):
    pts = np.random.RandomState(3).rand(3, 3).astype(np.float32)
    tree = locref.KdTree(pts)
    assert np.all(tree.knn(pts, k=5) == -1)  # kdtree.cpp:149-153


def test_degenerate_duplicates_collapse_to_one_leaf(locref):
    pts = np.ones((10, 3), dtype=np.float32)
    tree = locref.KdTree(pts)
    assert tree.num_leaves == 1 and tree.num_nodes == 1  # kdtree.cpp:76-81,118-120: all-equal set keeps points[0]


def test_fit_plane_matches_numpy_svd(locref):
    rng = np.random.RandomState(4)
    for _ in range(200):
        n = rng.randn(3)
        n /= np.linalg.norm(n)
        c = rng.randn(3) * 50
        basis = np.linalg.svd(n[None])[2][1:]
        p = c + (rng.randn(5, 2) * 0.3) @ basis + rng.randn(5, 1) * 0.01 * n
        ok, coef = locref.fit_plane(p)
        A = np.hstack([p, np.ones((5, 1))])
        v = np.linalg.svd(A)[2][3]
        if np.dot(v, coef) < 0:
            v = -v
        np.testing.assert_allclose(coef, v, atol=1e-9)
        assert abs(np.linalg.norm(coef) - 1) < 1e-12          # unit 4-vector: the 3-normal is NOT unit (A13)
        assert ok == bool(np.all((A @ v) ** 2 <= 1e-2))


def test_fit_plane_rejects_non_planar(locref):
    p = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 5.0], [1, 1, -5.0]])
    ok, _ = locref.fit_plane(p)
    assert not ok


def test_fit_line_matches_numpy_svd(locref):
    rng = np.random.RandomState(5)
    for _ in range(200):
        d = rng.randn(3)
        d /= np.linalg.norm(d)
        p = rng.randn(3) * 30 + rng.randn(5, 1) * d + rng.randn(5, 3) * 0.02
        ok, o, dr = locref.fit_line(p, eps=0.5)
        np.testing.assert_allclose(o, p.mean(0), atol=1e-12)
        v = np.linalg.svd(p - p.mean(0))[2][0]
        if np.dot(v, dr) < 0:
            v = -v
        np.testing.assert_allclose(dr, v, atol=1e-9)
        assert ok


def test_lu6_matches_numpy(locref):
    rng = np.random.RandomState(6)
    for _ in range(50):
        J = rng.randn(40, 6)
        H = J.T @ J
        b = rng.randn(6)
        det, x = locref.lu6(H, b)
        np.testing.assert_allclose(x, np.linalg.solve(H, b), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(det, np.linalg.det(H), rtol=1e-9)
    det, _ = locref.lu6(np.zeros((6, 6)), np.ones(6))
    assert det == 0.0  # `H.determinant() == 0` → false (icp cpp:210)


def test_se3_update_closed_form(locref):
    pose = np.array([0, 0, 0, 1.0, 1, 2, 3])
    dx = np.array([0, 0, np.pi / 2, 0.5, -0.5, 0.25])
    out = locref.apply_update(pose, dx)
    np.testing.assert_allclose(out[:4], [0, 0, np.sin(np.pi / 4), np.cos(np.pi / 4)], atol=1e-15)
    np.testing.assert_allclose(out[4:], [1.5, 1.5, 3.25], atol=1e-15)  # translation is added, not rotated (A17)
    tiny = locref.apply_update(pose, np.array([1e-12, 0, 0, 0, 0, 0]))
    np.testing.assert_allclose(tiny[:4], [5e-13, 0, 0, 1], atol=1e-20)  # Taylor branch


def test_transform_points_matches_matrix(locref):
    rng = np.random.RandomState(7)
    q = rng.randn(4)
    q /= np.linalg.norm(q)
    pose = np.concatenate([q, rng.randn(3) * 10])
    p = rng.randn(100, 3) * 20
    out = locref.transform_points(pose, p)
    np.testing.assert_allclose(out, p @ _rot(q).T + pose[4:], atol=1e-12)
    cloud = p.astype(np.float32)
    outc = locref.transform_cloud_f32(pose, cloud)
    np.testing.assert_allclose(outc, cloud.astype(np.float64) @ _rot(q).T + pose[4:], atol=2e-5)


def test_clamped_info_is_clamped_inverse(locref):
    rng = np.random.RandomState(8)
    A = rng.randn(3, 3)
    S = A @ np.diag([4.0, 1.0, 1e-6]) @ A.T
    S = (S + S.T) / 2
    info = locref.clamped_info(S)
    w, V = np.linalg.eigh(S)
    w = np.maximum(w, w.max() * 1e-3)
    np.testing.assert_allclose(info, V @ np.diag(1 / w) @ V.T, rtol=1e-8, atol=1e-10)


def _plane_world(rng, n=40000):
    """Ground + two walls with 1 cm noise, ±15 m (dense, well conditioned)."""
    g = np.c_[rng.uniform(-15, 15, n), rng.uniform(-15, 15, n), rng.randn(n) * 0.01]
    w1 = np.c_[15 + rng.randn(n // 2) * 0.01, rng.uniform(-15, 15, n // 2), rng.uniform(0, 6, n // 2)]
    w2 = np.c_[rng.uniform(-15, 15, n // 2), 15 + rng.randn(n // 2) * 0.01, rng.uniform(0, 6, n // 2)]
    return np.vstack([g, w1, w2]).astype(np.float32)


@pytest.mark.parametrize("method", [0, 1, 2])
def test_identity_alignment_has_tiny_update(locref, method):
    """Scan = subset of the map, identity pose ⇒ residuals ≈ 0 ⇒ first dx ≈ 0 and the loop stops after one iteration."""
    rng = np.random.RandomState(9)
    m = _plane_world(rng)
    scan = m[rng.choice(len(m), 3000, replace=False)]
    icp = locref.Icp(method=method)
    icp.set_target(m)
    r = icp.align(scan, np.array([0, 0, 0, 1.0, 0, 0, 0]))
    assert r["iters"] == 1
    assert np.linalg.norm(r["trace"][0, 42:48]) < 5e-3
    dt, dr = pose_delta(r["pose"], [0, 0, 0, 1, 0, 0, 0])
    assert dt < 5e-3 and dr < 5e-3


def test_p2plane_recovers_translation_along_normal(locref):
    """Ground-only world, scan lifted by 5 cm: one GN step brings it back (only z is observable there)."""
    rng = np.random.RandomState(10)
    n = 60000
    g = np.c_[rng.uniform(-15, 15, n), rng.uniform(-15, 15, n), rng.randn(n) * 0.002].astype(np.float32)
    scan = g[rng.choice(n, 4000, replace=False)].copy()
    icp = locref.Icp(method=2)
    icp.set_target(g)
    ok, H, B, eff = icp.hb(scan, np.array([0, 0, 0, 1.0, 0, 0, 0.05]))
    assert eff == len(scan)
    # normal equations restricted to z: dz = B[5]/H[5,5] = -0.05 (the homogeneous normal's scale cancels)
    assert abs(B[5] / H[5, 5] + 0.05) < 2e-3


def test_known_pose_recovery_ndt(locref, small_world):
    ndt = locref.Ndt()
    ndt.set_target(small_world["map"])
    assert ndt.num_voxels() > 1000
    r = ndt.align(small_world["scan10k"], small_world["init_pose"])
    assert r["status"] == 0
    dt, dr = pose_delta(r["pose"], small_world["true_pose"])
    # the loop stops at |dx| < eps = 1e-2 per step, i.e. still ~0.1 m short of the optimum on weak geometry
    assert dt < 0.15 and dr < 0.01 and dt < 0.5 * np.linalg.norm(small_world["init_pose"][4:] - small_world["true_pose"][4:])


def test_ndt_voxel_stats_match_numpy(locref, small_world):
    m = small_world["map"]
    ndt = locref.Ndt()
    ndt.set_target(m)
    keys, mu, info = ndt.dump()
    k_all = np.trunc(m[:, :3].astype(np.float64) * 1.0).astype(np.int64)  # truncation toward zero (A22)
    for i in range(0, len(keys), max(1, len(keys) // 50)):
        sel = np.all(k_all == keys[i], axis=1)
        p = m[sel, :3].astype(np.float64)
        assert len(p) > 3
        np.testing.assert_allclose(mu[i], p.mean(0), atol=1e-9)
        cov = np.cov(p.T)
        w, V = np.linalg.eigh(cov)
        w = np.maximum(w, w.max() * 1e-3)
        np.testing.assert_allclose(info[i], V @ np.diag(1 / w) @ V.T, rtol=1e-6, atol=1e-6)


def test_ndt_det_zero_leaves_pose_untouched(locref):
    """No voxel survives (too sparse) ⇒ H = 0 ⇒ `return false` before result_pose is written (ndt cpp:435-436)."""
    rng = np.random.RandomState(11)
    m = (rng.rand(200, 3) * 100).astype(np.float32)
    ndt = locref.Ndt()
    ndt.set_target(m)
    init = np.array([0, 0, 0, 1.0, 1, 2, 3])
    r = ndt.align(m[:50], init)
    assert r["status"] == 1 and r["iters"] == 1
    np.testing.assert_array_equal(r["pose"], init)


def test_p2p_sixteenth_quirk(locref):
    """J_rot = R·hat(q)/16 and dx = (H⁻¹/16)·err (icp cpp:84,287): check H's rotation block scale against numpy."""
    rng = np.random.RandomState(12)
    m = _plane_world(rng, 20000)
    scan = m[rng.choice(len(m), 500, replace=False)]
    icp = locref.Icp(method=0)
    icp.set_target(m)
    ok, H, B, eff = icp.hb(scan, np.array([0, 0, 0, 1.0, 0, 0, 0]))
    assert ok and eff == 500
    np.testing.assert_allclose(H[3:, 3:], np.eye(3) * eff, atol=1e-9)
    q = scan.astype(np.float64)
    hat = lambda v: np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    Hrr = sum((hat(v) / 16).T @ (hat(v) / 16) for v in q)
    np.testing.assert_allclose(H[:3, :3], Hrr, rtol=1e-9)


def test_flat_port_equals_reference_style(locref, synth, small_world):
    """the flat-array port of the point-to-plane path (oracle/locref_flat.hpp) and its threaded driver give the
    poses and iteration counts of the reference-style restatement (R1) bit for bit — approximate and exact pruning."""
    m, init = small_world["map"], small_world["init_pose"]
    scans = [small_world["scan2k"], small_world["scan10k"][::4], small_world["scan2k"][::2]]
    inits = np.stack([init, init, init])
    inits[1, 4:] += [0.05, -0.02, 0.01]
    for use_ann in (True, False):
        icp = locref.Icp(method=locref.P2PLANE, use_ann=use_ann)
        icp.set_target(m)
        want = [icp.align(s, p) for s, p in zip(scans, inits)]
        for threads in (1, 3):
            poses, iters = icp.align_flat(scans, inits, threads=threads)
            for i, w in enumerate(want):
                np.testing.assert_array_equal(poses[i], w["pose"])
                assert iters[i] == w["iters"]


def test_oracle_keeps_the_orders_read_off_the_reference_binary(tmp_path):
    """oracle/PINNING.md: `Vector3d` reductions are `(x + y) + z` and `dx.norm()` is `(d0² + (d2² + d4²)) + (d1² + (d3² + d5²))` in the
    reference's own prebuilt binary (P2Plane `dis` 0x5869a, `FitPlane` 0x79e65, P2P `dis2` 0x57945, `AlignP2Plane` 0x5b113). Inputs on
    which the other associations round differently hold the restatement to them (tests/cpp/oracle_orders.cpp, compiled with the
    oracle's flags)."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "oracle_orders")
    r = subprocess.run(["g++", "-std=c++17", "-O3", "-ffp-contract=off", "-I", os.path.join(root, "oracle"),
                        os.path.join(root, "tests", "cpp", "oracle_orders.cpp"), "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "oracle orders ok" in r.stdout, r.stdout + r.stderr
