"""GPU parity tests: the HIP path (through the C ABI, liblocgpu.so) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): index work bit-exact; poses within 1e-4 m / 1e-4 rad of the reference CPU path
(the observed differences are ~1e-10: only FP64 summation order differs).
"""
import numpy as np
import pytest

from conftest import pose_delta

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4    # north_star tolerance, metres
POSE_TOL_RAD = 1e-4  # north_star tolerance, radians
HB_RTOL = 1e-9       # FP64 sums differ by summation order only


def _hb_close(Hg, Bg, Ho, Bo):
    scale = max(np.abs(Ho).max(), 1e-300)
    assert np.abs(Hg - Ho).max() <= HB_RTOL * scale
    scale_b = max(np.abs(Bo).max(), np.abs(Ho).max() * 1e-6, 1e-300)
    assert np.abs(Bg - Bo).max() <= 1e-8 * scale_b


# ----------------------------------------------------------------------------------------------- search
@pytest.mark.parametrize("k", [1, 5])
@pytest.mark.parametrize("approximate", [True, False])
def test_knn_indices_and_visits_bit_exact(gpu_ctx, locref, small_world, k, approximate):
    m = small_world["map"]
    rng = np.random.RandomState(100 + k)
    q = (m[rng.choice(len(m), 4000, replace=False), :3] + rng.randn(4000, 3).astype(np.float32) * 0.05).astype(np.float32)
    gpu_ctx.icp_set_target(m)
    tree = locref.KdTree(m)
    info = gpu_ctx.icp_target_info()
    assert (info["num_leaves"], info["num_nodes"], info["depth"]) == (tree.num_leaves, tree.num_nodes, tree.depth)
    got, vis = gpu_ctx.knn(q, k=k, approximate=approximate, alpha=0.1, with_visits=True)
    ref, st = tree.knn(q, k=k, approximate=approximate, alpha=0.1, with_stats=True)
    np.testing.assert_array_equal(got, ref)
    assert int(vis[:, 0].sum()) == int(st[0]) and int(vis[:, 1].sum()) == int(st[1])


def test_knn_other_k_and_alpha(gpu_ctx, locref, small_world):
    m = small_world["map"]
    q = small_world["scan2k"][:500, :3] * 0.5 + m[:500, :3] * 0.5
    gpu_ctx.icp_set_target(m)
    tree = locref.KdTree(m)
    for k, alpha in ((3, 0.3), (8, 0.05), (2, 1.0)):
        np.testing.assert_array_equal(gpu_ctx.knn(q, k=k, approximate=True, alpha=alpha), tree.knn(q, k=k, approximate=True, alpha=alpha))


def test_knn_duplicates_and_tiny_tree(gpu_ctx, locref, api):
    pts = np.random.RandomState(5).rand(64, 3).astype(np.float32)
    pts[10:20] = pts[10]  # duplicates collapse into one leaf (kdtree.cpp:76-81)
    gpu_ctx.icp_set_target(pts)
    tree = locref.KdTree(pts)
    assert gpu_ctx.icp_target_info()["num_leaves"] == tree.num_leaves == 55
    q = np.random.RandomState(6).rand(200, 3).astype(np.float32)
    np.testing.assert_array_equal(gpu_ctx.knn(q, k=5), tree.knn(q, k=5))
    gpu_ctx.icp_set_target(pts[:3])
    with pytest.raises(api.LocGpuError) as e:  # k > size_: kdtree.cpp:149-153
        gpu_ctx.knn(q, k=5)
    assert e.value.code == -4


def test_knn_ties_follow_libstdcxx_heap(gpu_ctx, locref):
    """Lattice points ⇒ many exactly equal float32 distances: order/eviction must follow std::priority_queue."""
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(6), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    q = (g[::7] + 0.5).astype(np.float32)
    gpu_ctx.icp_set_target(g)
    tree = locref.KdTree(g)
    for approx in (True, False):
        np.testing.assert_array_equal(gpu_ctx.knn(q, k=5, approximate=approx), tree.knn(q, k=5, approximate=approx))


# ----------------------------------------------------------------------------------------------- H, B
@pytest.mark.parametrize("method", [0, 1, 2])
def test_icp_hb_matches_oracle(gpu_ctx, api, locref, small_world, method):
    m, s, pose = small_world["map"], small_world["scan10k"], small_world["init_pose"]
    gpu_ctx.icp_set_target(m)
    icp = locref.Icp(method=method)
    icp.set_target(m)
    ok_o, Ho, Bo, eff_o = icp.hb(s, pose)
    ok_g, Hg, Bg, eff_g = gpu_ctx.icp_hb(s, pose, api.icp_opts(method=method))
    assert ok_g == ok_o and eff_g == eff_o
    _hb_close(Hg, Bg, Ho, Bo)
    np.testing.assert_array_equal(Hg, Hg.T)


def test_icp_hb_unbounded_map_and_far_queries_take_the_exact_kernel(gpu_ctx, api, locref, small_world):
    """The fast search kernel assumes squared distances cannot overflow: a map with an astronomically far point
    (kdtree_build.cpp `bounded`) and source points beyond 1e18 m are answered by the exact kernel, with the same result."""
    m, s, pose = small_world["map"].copy(), small_world["scan2k"].copy(), small_world["init_pose"]
    m[7] = [3e19, -2e19, 1e19]          # (3e19)^2 overflows float32
    for method in (0, 2):
        gpu_ctx.icp_set_target(m)
        icp = locref.Icp(method=method)
        icp.set_target(m)
        ok_o, Ho, Bo, eff_o = icp.hb(s, pose)
        ok_g, Hg, Bg, eff_g = gpu_ctx.icp_hb(s, pose, api.icp_opts(method=method))
        assert ok_g == ok_o and eff_g == eff_o
        _hb_close(Hg, Bg, Ho, Bo)
    # bounded map again, but three source points far outside float32's squared range
    m = small_world["map"]
    s[5] = [5e19, 0, 0]
    s[900] = [0, -7e18, 1e19]
    s[1500] = [1e30, 1e30, 1e30]
    gpu_ctx.icp_set_target(m)
    icp = locref.Icp(method=0)
    icp.set_target(m)
    ok_o, Ho, Bo, eff_o = icp.hb(s, pose)
    ok_g, Hg, Bg, eff_g = gpu_ctx.icp_hb(s, pose, api.icp_opts(method=0))
    assert ok_g == ok_o and eff_g == eff_o
    _hb_close(Hg, Bg, Ho, Bo)


def test_icp_hb_on_exactly_planar_neighbourhoods(gpu_ctx, api, locref):
    """Noise-free planes: the 5×4 [x y z 1] matrix of every neighbourhood is exactly rank 3 (smallest singular value = rounding
    noise). The plane 4-vector must still be the true plane (device: orthogonal complement of the three dominant columns)."""
    rng = np.random.default_rng(5)
    xy = rng.uniform(-30, 30, (60000, 2))
    planes = [(0.3, 0.5, 2.0), (-0.2, 0.1, -1.0), (0.0, 0.0, 5.0)]
    pts = []
    for k, (a, b, c) in enumerate(planes):
        sel = xy[k::3]
        pts.append(np.stack([sel[:, 0], sel[:, 1], a * sel[:, 0] + b * sel[:, 1] + c], 1))
    m = np.concatenate(pts).astype(np.float32)
    q = m[rng.choice(len(m), 4000, replace=False)].astype(np.float64) + rng.normal(0, 0.03, (4000, 3))
    s = q.astype(np.float32)
    pose = np.array([0.001, -0.002, 0.0015, 1.0, 0.02, -0.01, 0.015])
    pose[:4] /= np.linalg.norm(pose[:4])
    gpu_ctx.icp_set_target(m)
    icp = locref.Icp(method=2)
    icp.set_target(m)
    ok_o, Ho, Bo, eff_o = icp.hb(s, pose)
    ok_g, Hg, Bg, eff_g = gpu_ctx.icp_hb(s, pose, api.icp_opts(method=2))
    assert ok_g == ok_o and eff_g == eff_o and eff_o > 3000
    _hb_close(Hg, Bg, Ho, Bo)


def test_icp_hb_exact_search_mode_of_the_tree(gpu_ctx, api, locref, small_world):
    """SetEnableANN(false) (kdtree.cpp:285-288): exact pruning rule through the same tree."""
    m, s, pose = small_world["map"], small_world["scan2k"], small_world["init_pose"]
    gpu_ctx.icp_set_target(m)
    icp = locref.Icp(method=2, use_ann=False)
    icp.set_target(m)
    ok_o, Ho, Bo, eff_o = icp.hb(s, pose)
    ok_g, Hg, Bg, eff_g = gpu_ctx.icp_hb(s, pose, api.icp_opts(method=2, approximate=0))
    assert ok_g == ok_o and eff_g == eff_o
    _hb_close(Hg, Bg, Ho, Bo)


def test_icp_hb_too_few_points_is_false(gpu_ctx, api, locref, small_world):
    m = small_world["map"]
    gpu_ctx.icp_set_target(m)
    s = small_world["scan2k"][:5]
    ok_g, Hg, Bg, eff_g = gpu_ctx.icp_hb(s, small_world["init_pose"], api.icp_opts(method=2))
    icp = locref.Icp(method=2)
    icp.set_target(m)
    ok_o, Ho, Bo, eff_o = icp.hb(s, small_world["init_pose"])
    assert ok_g == ok_o == False and eff_g == eff_o  # noqa: E712  effective_num < min_effective_pts_ (icp cpp:204)


# ----------------------------------------------------------------------------------------------- align
@pytest.mark.parametrize("method", [0, 1, 2])
def test_icp_align_matches_oracle(gpu_ctx, api, locref, small_world, method):
    m, s, init = small_world["map"], small_world["scan10k"], small_world["init_pose"]
    gpu_ctx.icp_set_target(m)
    icp = locref.Icp(method=method)
    icp.set_target(m)
    ro = icp.align(s, init)
    pg, st = gpu_ctx.icp_align(s, init, api.icp_opts(method=method))
    dt, dr = pose_delta(pg, ro["pose"])
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    assert st["iterations"] == ro["iters"]
    assert st["last_effective_num"] == int(ro["trace"][-1, 48])
    assert dt < 1e-8 and dr < 1e-8  # what we actually expect: summation-order noise only


def test_configs0_at_its_exact_size(gpu_ctx, api, locref, synth):
    """BASELINE.json configs[0] as written — a single 10 k-pt synthetic scan vs the 100 k-pt map (synth.make_map(100000): the whole
    ±150 m world at 100 k points, not a dense local crop), reference-default point-to-plane ICP through the single-scan host-pointer
    entry point — for a few scans of the circuit: pose within the north-star tolerance of the oracle, equal iteration counts."""
    m = synth.make_map(100000)
    gpu_ctx.icp_set_target(m)
    icp = locref.Icp(method=locref.P2PLANE)
    icp.set_target(m)
    opts = api.icp_opts(method=api.P2PLANE)
    for sid in (0, 7, 100):
        s = synth.make_scan(sid, subsample=10000)
        _, init = synth.make_pose(sid)
        pg, st = gpu_ctx.icp_align(s, init, opts)
        ro = icp.align(s, init)
        dt, dr = pose_delta(pg, ro["pose"])
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (sid, dt, dr)
        assert st["iterations"] == ro["iters"], (sid, st, ro["iters"])


def test_icp_align_nonconverging_runs_max_iterations(gpu_ctx, api, locref, small_world):
    m, s, init = small_world["map"], small_world["scan2k"], small_world["init_pose"]
    gpu_ctx.icp_set_target(m)
    opts = api.icp_opts(method=2, eps=1e-12, max_iteration=7)
    icp = locref.Icp(method=2, eps=1e-12, max_iteration=7)
    icp.set_target(m)
    ro = icp.align(s, init)
    pg, st = gpu_ctx.icp_align(s, init, opts)
    assert st["iterations"] == ro["iters"] == 7 and not st["converged"]
    dt, dr = pose_delta(pg, ro["pose"])
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD


def test_icp_align_batch_ragged(gpu_ctx, api, locref, synth):
    """Batch of scans of different sizes and poses vs one map: each scan's result equals its own single-scan oracle run."""
    m = synth.make_local_map(80000, 5, half=25.0)
    scans = [synth.make_scan(5, subsample=n, crop_half=22.0) for n in (3000, 1000, 2500, 257)]
    _, init = synth.make_pose(5)
    inits = np.stack([init, init, init, init])
    inits[1, 4:] += [0.05, -0.05, 0.02]
    inits[2, 4:] -= [0.1, 0.0, 0.05]
    gpu_ctx.icp_set_target(m)
    b = gpu_ctx.batch(scans)
    out, stats = gpu_ctx.icp_align_batch(b, inits, api.icp_opts(method=2))
    icp = locref.Icp(method=2)
    icp.set_target(m)
    for i, s in enumerate(scans):
        ro = icp.align(s, inits[i])
        dt, dr = pose_delta(out[i], ro["pose"])
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (i, dt, dr)
        assert stats[i]["iterations"] == ro["iters"]
    b.close()


def test_p2p_skips_nonfinite_source_points(gpu_ctx, api, locref, small_world):
    m, s, init = small_world["map"], small_world["scan2k"].copy(), small_world["init_pose"]
    s[::50, 0] = np.nan  # pcl::isFinite skip exists only in P2P (icp cpp:64)
    gpu_ctx.icp_set_target(m)
    icp = locref.Icp(method=0)
    icp.set_target(m)
    ok_o, Ho, Bo, eff_o = icp.hb(s, init)
    ok_g, Hg, Bg, eff_g = gpu_ctx.icp_hb(s, init, api.icp_opts(method=0))
    assert ok_g == ok_o and eff_g == eff_o
    _hb_close(Hg, Bg, Ho, Bo)


def test_point_sharded_hb_sum_equals_full(gpu_ctx, api, small_world):
    """Point-sharding (multi-GPU mode with a real exchange): per-shard H,B summed = H,B of the whole scan."""
    m, s, init = small_world["map"], small_world["scan10k"], small_world["init_pose"]
    gpu_ctx.icp_set_target(m)
    opts = api.icp_opts(method=2)
    b_full = gpu_ctx.batch([s])
    full = gpu_ctx.icp_hb_batch(b_full, init, opts)[0]
    b_sh = gpu_ctx.batch([s[:6000], s[6000:]])
    parts = gpu_ctx.icp_hb_batch(b_sh, np.stack([init, init]), opts)
    summed = parts[:, :43].sum(0)
    np.testing.assert_allclose(summed[:36], full[:36], rtol=1e-10, atol=1e-12 * np.abs(full[:36]).max())
    np.testing.assert_allclose(summed[36:42], full[36:42], rtol=1e-8, atol=1e-10 * np.abs(full[:36]).max())
    assert summed[42] == full[42]


def test_point_sharded_align_equals_monolithic(gpu_ctx, api, small_world):
    """multi_gpu.point_sharded_align (per-iteration H,B over point shards → sum → host update, the loop that carries the RCCL
    all-reduce when ranks own the shards) reproduces the on-device Gauss–Newton loop: same iterations, same pose."""
    from loc_lib_amd import multi_gpu
    m, s, init = small_world["map"], small_world["scan10k"], small_world["init_pose"]
    gpu_ctx.icp_set_target(m)
    for method in (2, 0):
        opts = api.icp_opts(method=method)
        want, st = gpu_ctx.icp_align(s, init, opts)
        shards = gpu_ctx.batch([s[:3500], s[3500:7000], s[7000:]])  # three "ranks" worth of points

        def hb_fn(pose):
            hb = gpu_ctx.icp_hb_batch(shards, np.stack([pose] * 3), opts)
            out = hb[:, :43].sum(0)
            return np.concatenate([out, [0.0]])

        got, iters = multi_gpu.point_sharded_align(hb_fn, api.gn_update, init, method=method)
        shards.close()
        assert iters == st["iterations"]
        dt, dr = pose_delta(got, want)
        assert dt < 1e-9 and dr < 1e-9, (method, dt, dr)


# ----------------------------------------------------------------------------------------------- output cloud
def test_transform_cloud_bit_exact(gpu_ctx, locref, small_world):
    s = np.zeros((2000, 8), dtype=np.float32)  # pcl::PointXYZI stride (32 bytes)
    s[:, :3] = small_world["scan2k"][:, :3]
    s[:, 4] = np.arange(2000)                  # intensity survives
    pose = small_world["init_pose"]
    out_g = gpu_ctx.transform_cloud(pose, s)
    out_o = locref.transform_cloud_f32(pose, s)
    np.testing.assert_array_equal(out_g.view(np.uint32), out_o.view(np.uint32))
    np.testing.assert_array_equal(out_g[:, 4], s[:, 4])


@pytest.mark.parametrize("method", ["p2p", "p2line", "p2plane", "ndt"])
def test_scan_match_whole_equals_align_plus_transform(gpu_ctx, locref, small_world, method):
    """locgpu_*_scan_match (ScanMatch whole: icp_registration.cpp:216-244, ndt_registration.cpp:238-261): the pose is locgpu_*_align's bit
    for bit, the output cloud is the oracle's float32 transformPointCloud of the source under that pose bit for bit, every other field
    of a point is the source's — as a separate output cloud and in place."""
    from loc_lib_amd import api
    m = small_world["map"]
    s = np.zeros((2000, 8), dtype=np.float32)  # pcl::PointXYZI stride (32 bytes)
    s[:, :3] = small_world["scan2k"][:, :3]
    s[:, 3] = 1.0
    s[:, 4] = np.arange(2000)
    s[:, 5:] = 7.5
    init = small_world["init_pose"]
    if method == "ndt":
        gpu_ctx.ndt_set_target(m)
        want_pose, want_st = gpu_ctx.ndt_align(s, init)
        pose, st, cloud = gpu_ctx.ndt_scan_match(s, init)
    else:
        gpu_ctx.icp_set_target(m)
        opts = api.icp_opts(method=dict(p2p=api.P2P, p2line=api.P2LINE, p2plane=api.P2PLANE)[method])
        want_pose, want_st = gpu_ctx.icp_align(s, init, opts)
        pose, st, cloud = gpu_ctx.icp_scan_match(s, init, opts)
    np.testing.assert_array_equal(pose, want_pose)
    assert st == want_st
    want_cloud = locref.transform_cloud_f32(pose, s)
    np.testing.assert_array_equal(cloud.view(np.uint32), want_cloud.view(np.uint32))
    np.testing.assert_array_equal(cloud[:, 3:], s[:, 3:])
    if method != "ndt":
        s2 = s.copy()
        pose2, _, cloud2 = gpu_ctx.icp_scan_match(s2, init, opts, in_place=True)
        assert cloud2 is s2
        np.testing.assert_array_equal(pose2, want_pose)
        np.testing.assert_array_equal(s2.view(np.uint32), want_cloud.view(np.uint32))


def test_ndt_scan_match_det_zero_keeps_the_callers_pose(gpu_ctx, locref):
    """det(H) == 0: AlignNdt returns before it assigns result_pose (ndt_registration.cpp:435-436) — the caller's value stays, and the
    output cloud is the source under THAT pose (:258)."""
    rng = np.random.RandomState(11)
    m = (rng.rand(200, 3) * 100).astype(np.float32)
    gpu_ctx.ndt_set_target(m)
    init = np.array([0, 0, 0, 1.0, 1, 2, 3])
    mine = np.array([0.0, 0.0, np.sin(0.2), np.cos(0.2), -4.0, 5.0, 6.0])
    pose, st, cloud = gpu_ctx.ndt_scan_match(m[:50], init, result_pose=mine)
    assert st["status"] == 1
    np.testing.assert_array_equal(pose, mine)
    np.testing.assert_array_equal(cloud.view(np.uint32), locref.transform_cloud_f32(mine, m[:50]).view(np.uint32))


# ----------------------------------------------------------------------------------------------- NDT
def test_ndt_voxel_table_matches_oracle(gpu_ctx, locref, small_world):
    m = small_world["map"]
    gpu_ctx.ndt_set_target(m)
    ndt = locref.Ndt()
    ndt.set_target(m)
    kg, mug, ig = gpu_ctx.ndt_dump()
    ko, muo, io = ndt.dump()
    assert len(kg) == len(ko) == gpu_ctx.ndt_target_info()["num_voxels"]
    og, oo = np.lexsort(kg.T[::-1]), np.lexsort(ko.T[::-1])
    np.testing.assert_array_equal(kg[og], ko[oo])
    # round 5: the build sums a voxel's points sequentially in input order (stable sort by key), the reference's own order
    # (math_utils.h:55-72): μ — and Σ behind info — are the oracle's bits; info goes through the same one-sided Jacobi on both sides
    np.testing.assert_array_equal(mug[og], muo[oo])
    scale = np.abs(io[oo]).max(axis=(1, 2), keepdims=True)
    assert (np.abs(ig[og] - io[oo]) / scale).max() < 1e-12
    # and two ingests of the same map give the same bits (rounds 1-4 summed with FP64 atomics: they did not)
    gpu_ctx.ndt_set_target(m)
    kg2, mug2, ig2 = gpu_ctx.ndt_dump()
    og2 = np.lexsort(kg2.T[::-1])
    assert np.array_equal(kg2[og2], kg[og]) and np.array_equal(mug2[og2], mug[og]) and np.array_equal(ig2[og2], ig[og])


@pytest.mark.parametrize("nearby", [1, 0])
def test_ndt_align_matches_oracle(gpu_ctx, api, locref, small_world, nearby):
    m, s, init = small_world["map"], small_world["scan10k"], small_world["init_pose"]
    gpu_ctx.ndt_set_target(m, api.ndt_opts(nearby_type=nearby))
    ndt = locref.Ndt(nearby_type=nearby)
    ndt.set_target(m)
    ro = ndt.align(s, init)
    pg, st = gpu_ctx.ndt_align(s, init)
    assert st["status"] == ro["status"] == 0
    assert st["iterations"] == ro["iters"]
    dt, dr = pose_delta(pg, ro["pose"])
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)


def test_ndt_det_zero_returns_init_pose(gpu_ctx, locref):
    rng = np.random.RandomState(11)
    m = (rng.rand(200, 3) * 100).astype(np.float32)
    gpu_ctx.ndt_set_target(m)
    init = np.array([0, 0, 0, 1.0, 1, 2, 3])
    pg, st = gpu_ctx.ndt_align(m[:50], init)
    assert st["status"] == 1 and st["iterations"] == 1  # ndt cpp:435-436
    np.testing.assert_array_equal(pg, init)


def test_ndt_negative_coordinates_truncate_toward_zero(gpu_ctx, locref):
    """Cells touching 0 are double width (A22): a dense blob across the origin lands in ONE voxel per axis sign pair."""
    rng = np.random.RandomState(12)
    m = (rng.rand(4000, 3) * 1.8 - 0.9).astype(np.float32)
    gpu_ctx.ndt_set_target(m)
    ndt = locref.Ndt()
    ndt.set_target(m)
    kg, _, _ = gpu_ctx.ndt_dump()
    ko, _, _ = ndt.dump()
    assert len(kg) == len(ko) == 1 and tuple(kg[0]) == (0, 0, 0)


# ----------------------------------------------------------------------------------------------- exact grid search
def _knn_equal_up_to_ties(got, ref, q, pts):
    """Same index lists, except where float32 distances tie exactly (then the distances must still agree position by position)."""
    same = np.all(got == ref, axis=1)
    if same.all():
        return 0
    bad = np.where(~same)[0]

    def d2(idx):
        d = q[bad, None, :] - pts[idx[bad]]
        d = d.astype(np.float32)
        return d[..., 0] * d[..., 0] + (d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])
    np.testing.assert_array_equal(d2(got), d2(ref))
    return len(bad)


@pytest.mark.parametrize("k", [1, 5])
def test_grid_knn_equals_exact_tree(gpu_ctx, api, locref, small_world, k):
    m = small_world["map"]
    rng = np.random.RandomState(200 + k)
    near = (m[rng.choice(len(m), 6000, replace=False), :3] + rng.randn(6000, 3).astype(np.float32) * 0.08).astype(np.float32)
    sparse = (m[rng.choice(len(m), 500, replace=False), :3] + rng.randn(500, 3).astype(np.float32) * 3.0).astype(np.float32)  # several rings
    far = (rng.rand(100, 3).astype(np.float32) - 0.5) * 2000                                                                    # tree fallback
    q = np.vstack([near, sparse, far]).astype(np.float32)
    gpu_ctx.icp_set_target(m)
    tree = locref.KdTree(m)
    got = gpu_ctx.knn(q, k=k, search_mode=api.SEARCH_GRID_EXACT)
    ref = tree.knn(q, k=k, approximate=False)
    n_tied = _knn_equal_up_to_ties(got, ref, q, m[:, :3])
    assert n_tied <= 2


def test_grid_knn_duplicates_and_tiny(gpu_ctx, api, locref):
    pts = np.random.RandomState(5).rand(400, 3).astype(np.float32)
    pts[10:60] = pts[10]  # dropped by the tree's degenerate-leaf rule ⇒ absent from the grid too
    gpu_ctx.icp_set_target(pts)
    tree = locref.KdTree(pts)
    q = np.random.RandomState(6).rand(300, 3).astype(np.float32)
    got = gpu_ctx.knn(q, k=5, search_mode=api.SEARCH_GRID_EXACT)
    assert _knn_equal_up_to_ties(got, tree.knn(q, k=5, approximate=False), q, pts) == 0
    gpu_ctx.icp_set_target(pts[:7])
    tree = locref.KdTree(pts[:7])
    np.testing.assert_array_equal(gpu_ctx.knn(q, k=5, search_mode=api.SEARCH_GRID_EXACT), tree.knn(q, k=5, approximate=False))


@pytest.mark.parametrize("method", [2, 0, 1])
def test_icp_grid_mode_matches_exact_oracle(gpu_ctx, api, locref, small_world, method):
    """search_mode = GRID_EXACT ≡ the reference with SetEnableANN(false) (kdtree.cpp:285-288)."""
    m, s, init = small_world["map"], small_world["scan10k"], small_world["init_pose"]
    gpu_ctx.icp_set_target(m)
    icp = locref.Icp(method=method, use_ann=False)
    icp.set_target(m)
    opts = api.icp_opts(method=method, search_mode=api.SEARCH_GRID_EXACT)
    ok_o, Ho, Bo, eff_o = icp.hb(s, init)
    ok_g, Hg, Bg, eff_g = gpu_ctx.icp_hb(s, init, opts)
    assert ok_g == ok_o and eff_g == eff_o
    _hb_close(Hg, Bg, Ho, Bo)
    ro = icp.align(s, init)
    pg, st = gpu_ctx.icp_align(s, init, opts)
    dt, dr = pose_delta(pg, ro["pose"])
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and st["iterations"] == ro["iters"], (dt, dr)
    # and the exact tree kernel (approximate=0) gives the same alignment as the grid
    pt, st2 = gpu_ctx.icp_align(s, init, api.icp_opts(method=method, approximate=0))
    dt, dr = pose_delta(pg, pt)
    assert dt <= 1e-9 and dr <= 1e-9


# ----------------------------------------------------------------------------------------------- incremental NDT
def _world_scan(locref, synth, sid, n):
    """Scan `sid` expressed in the world frame with its true pose (what Lio feeds the matcher after a keyframe)."""
    s = synth.make_scan(sid, subsample=n, crop_half=36.0)
    true_pose, _ = synth.make_pose(sid)
    return locref.transform_cloud_f32(true_pose, s)


@pytest.mark.parametrize("capacity", [100000, 700])
def test_incremental_ndt_matches_oracle(gpu_ctx, api, locref, synth, small_world, capacity):
    """SetIncNdtTargetCloud called three times (voxel set persists, LRU eviction when capacity is small), then AlignIncNdt."""
    clouds = [small_world["map"][::4], _world_scan(locref, synth, 3, 6000), _world_scan(locref, synth, 4, 5000)]
    opts = api.ndt_opts(method=api.INCREMENTAL_NDT, capacity=capacity)
    ndt = locref.Ndt(method=locref.INCREMENTAL_NDT, capacity=capacity)
    gpu_ctx.ndt_set_target(clouds[0][:10], api.ndt_opts())  # a direct call in between resets the incremental set
    for c in clouds:
        gpu_ctx.ndt_set_target(c, opts)
        ndt.set_target(c)
        assert gpu_ctx.ndt_target_info()["num_voxels"] == ndt.num_voxels()
    if capacity == 700:
        assert ndt.num_voxels() < 700  # evictions happened
    kg, mug, ig = gpu_ctx.ndt_dump()
    ko, muo, io = ndt.dump()
    og, oo = np.lexsort(kg.T[::-1]), np.lexsort(ko.T[::-1])
    np.testing.assert_array_equal(kg[og], ko[oo])
    np.testing.assert_allclose(mug[og], muo[oo], rtol=0, atol=1e-9)
    scale = np.abs(io[oo]).max(axis=(1, 2), keepdims=True)
    assert (np.abs(ig[og] - io[oo]) / scale).max() < 1e-7
    s, init = small_world["scan10k"], small_world["init_pose"]
    ro = ndt.align(s, init)
    pg, st = gpu_ctx.ndt_align(s, init)
    assert st["status"] == ro["status"] and st["iterations"] == ro["iters"]
    dt, dr = pose_delta(pg, ro["pose"])
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)


def test_incremental_ndt_too_few_residuals(gpu_ctx, api, locref):
    """effective_num < min_effective_pts ⇒ AlignIncNdt returns false with result = current pose (ndt cpp:349-353)."""
    rng = np.random.RandomState(3)
    m = (rng.rand(50, 3) * 2).astype(np.float32)
    far = (rng.rand(30, 3) * 2 + 500).astype(np.float32)
    opts = api.ndt_opts(method=api.INCREMENTAL_NDT)
    gpu_ctx.ndt_set_target(m[:5], api.ndt_opts())
    gpu_ctx.ndt_set_target(m, opts)
    ndt = locref.Ndt(method=locref.INCREMENTAL_NDT)
    ndt.set_target(m)
    init = np.array([0, 0, 0, 1.0, 0, 0, 0])
    ro = ndt.align(far, init)
    pg, st = gpu_ctx.ndt_align(far, init)
    assert ro["status"] == 2 and st["status"] == 2 and st["iterations"] == ro["iters"] == 1
    np.testing.assert_array_equal(pg, init)


# ----------------------------------------------------------------------------------------------- hipGraph mode
def test_graph_mode_equals_eager(gpu_ctx, api, synth, small_world):
    """Captured hipGraph of all GN iterations (device-side early-outs) == the eager data-dependent loop, bit for bit;
    the instantiated graph is replayed with new initial poses, other options re-capture."""
    m = small_world["map"]
    scans = [small_world["scan10k"], small_world["scan2k"], small_world["scan10k"][::3]]
    init = small_world["init_pose"]
    inits = np.stack([init, init, init])
    inits[1, 4:] += [0.04, -0.03, 0.01]
    gpu_ctx.icp_set_target(m)
    gpu_ctx.ndt_set_target(m)
    b = gpu_ctx.batch(scans)
    try:
        for method in (2, 0):
            opts = api.icp_opts(method=method)
            gpu_ctx.graph_enable(False)
            want, wst = gpu_ctx.icp_align_batch(b, inits, opts)
            gpu_ctx.graph_enable(True)
            got, gst = gpu_ctx.icp_align_batch(b, inits, opts)      # capture + launch
            np.testing.assert_array_equal(got, want)
            assert [s["iterations"] for s in gst] == [s["iterations"] for s in wst]
            inits2 = inits.copy()
            inits2[:, 4:] += 0.02
            got2, _ = gpu_ctx.icp_align_batch(b, inits2, opts)     # replay with other poses
            gpu_ctx.graph_enable(False)
            want2, _ = gpu_ctx.icp_align_batch(b, inits2, opts)
            np.testing.assert_array_equal(got2, want2)
        gpu_ctx.graph_enable(False)
        want, _ = gpu_ctx.ndt_align_batch(b, inits)
        gpu_ctx.graph_enable(True)
        got, _ = gpu_ctx.ndt_align_batch(b, inits)
        # NDT sums are FP64 atomics-free per scan (fixed order) ⇒ also bitwise equal
        np.testing.assert_array_equal(got, want)
    finally:
        gpu_ctx.graph_enable(False)
        b.close()


# ----------------------------------------------------------------------------------------------- C++ façade
@pytest.mark.parametrize("kind,method", [("icp", 2), ("icp", 0), ("icp", 1), ("ndt", 0)])
def test_cpp_facade_scanmatch(locref, small_world, tmp_path, kind, method):
    """LocUtils::IcpRegistration / NdtRegistration (loc_lib_amd/host) driven through MatchingInterface like Loc::Update."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "cpp", "facade_scanmatch")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    m, s, init = small_world["map"], small_world["scan10k"], small_world["init_pose"]
    np.ascontiguousarray(m[:, :3], dtype=np.float32).tofile(tmp_path / "map.bin")
    np.ascontiguousarray(s[:, :3], dtype=np.float32).tofile(tmp_path / "scan.bin")
    np.asarray(init, dtype=np.float64).tofile(tmp_path / "pose.bin")
    r = subprocess.run([exe, kind, str(method), str(tmp_path / "map.bin"), str(tmp_path / "scan.bin"), str(tmp_path / "pose.bin"),
                        str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(tmp_path / "out.bin", dtype=np.uint8)
    pose = raw[:56].view(np.float64)
    cloud = raw[56:].view(np.float32).reshape(-1, 3)
    if kind == "icp":
        ref = locref.Icp(method=method)
        ref.set_target(m)
        want = ref.align(s, init)["pose"]
    else:
        ref = locref.Ndt()
        ref.set_target(m)
        want = ref.align(s, init)["pose"]
    dt, dr = pose_delta(pose, want)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    np.testing.assert_array_equal(cloud.view(np.uint32), locref.transform_cloud_f32(pose, s[:, :3]).view(np.uint32))


def test_cpp_facade_search_plugins(locref, small_world, tmp_path):
    """KdtreeRegistration::SetTargetCloud / FindNearstPoints / SetEnableANN and BfnnRegistration through SearchPointInterface
    (search_point_interface.h:9-24, kdtree.cpp:261-288, bfnn.cpp:24-50; tests/cpp/facade_search.cpp) against the oracle's lists:
    the tree as constructed (ANN, alpha 0.1), exact, ANN with another alpha; brute force — index for index, in order."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "cpp", "facade_search")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    m = np.ascontiguousarray(small_world["map"][:60000, :3], dtype=np.float32)
    rng = np.random.RandomState(5)
    q = (m[rng.randint(0, len(m), 300)] + rng.normal(0, 0.05, (300, 3))).astype(np.float32)
    k = 5
    m.tofile(tmp_path / "map.bin")
    q.tofile(tmp_path / "q.bin")
    r = subprocess.run([exe, str(tmp_path / "map.bin"), str(tmp_path / "q.bin"), str(k), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout + r.stderr)
    got = np.fromfile(tmp_path / "out.bin", dtype=np.int32).reshape(5, len(q), k)
    tree = locref.KdTree(m)
    np.testing.assert_array_equal(got[0], tree.knn(q, k, approximate=True, alpha=0.1))
    np.testing.assert_array_equal(got[1], tree.knn(q, k, approximate=False))
    np.testing.assert_array_equal(got[2], tree.knn(q, k, approximate=True, alpha=0.3))
    want_bf = locref.bfnn_knn(m, q, k)
    np.testing.assert_array_equal(got[3], want_bf)
    np.testing.assert_array_equal(got[4], want_bf)


def test_cpp_facade_loam(locref, small_world, tmp_path):
    """LoamRegistration (loam_registration.cpp:38-99): edge P2Line + surf P2Plane normal equations summed, own GN loop (eps 1e-3)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "cpp", "facade_scanmatch")
    m, s, init = small_world["map"], small_world["scan10k"], small_world["init_pose"]
    edge_map, surf_map = m[::5], m
    edge, surf = s[::7], s[np.arange(len(s)) % 7 != 0]
    for name, arr in (("em", edge_map), ("sm", surf_map), ("e", edge), ("s", surf)):
        np.ascontiguousarray(arr[:, :3], dtype=np.float32).tofile(tmp_path / (name + ".bin"))
    np.asarray(init, dtype=np.float64).tofile(tmp_path / "pose.bin")
    r = subprocess.run([exe, "loam"] + [str(tmp_path / n) for n in ("em.bin", "sm.bin", "e.bin", "s.bin", "pose.bin", "out.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(tmp_path / "out.bin", dtype=np.uint8)
    pose = raw[:56].view(np.float64)
    cloud = raw[56:].view(np.float32).reshape(-1, 3)
    # oracle composition
    ie, isf = locref.Icp(method=locref.P2LINE), locref.Icp(method=locref.P2PLANE)
    ie.set_target(edge_map)
    isf.set_target(surf_map)
    p = np.array(init, dtype=np.float64)
    for _ in range(20):
        ok1, H1, B1, _ = isf.hb(surf, p)
        ok2, H2, B2, _ = ie.hb(edge, p)
        assert ok1 and ok2
        det, dx = locref.lu6(H1 + H2, B1 + B2)
        p = locref.apply_update(p, dx)
        if np.linalg.norm(dx) < 1e-3:
            break
    dt, dr = pose_delta(pose, p)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    both = np.vstack([edge[:, :3], surf[:, :3]])
    np.testing.assert_array_equal(cloud.view(np.uint32), locref.transform_cloud_f32(pose, both).view(np.uint32))


# ----------------------------------------------------------------------------------------------- golden fixtures
def test_golden_fixture_gpu(gpu_ctx, api):
    """Committed golden vectors (tests/golden/make_golden.py, generated with the oracle in the build container)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "icp_small.npz"))
    gpu_ctx.icp_set_target(g["map"])
    np.testing.assert_array_equal(gpu_ctx.knn(g["queries"], k=5, approximate=True), g["knn_ann"])
    np.testing.assert_array_equal(gpu_ctx.knn(g["queries"], k=5, approximate=False), g["knn_exact"])
    for method, name in ((0, "p2p"), (1, "p2line"), (2, "p2plane")):
        pg, st = gpu_ctx.icp_align(g["scan"], g["init_pose"], api.icp_opts(method=method))
        dt, dr = pose_delta(pg, g["pose_" + name])
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD
        assert st["iterations"] == int(g["iters_" + name])
        ok, H, B, eff = gpu_ctx.icp_hb(g["scan"], g["init_pose"], api.icp_opts(method=method))
        _hb_close(H, B, g["H0_" + name], g["B0_" + name])
    gpu_ctx.ndt_set_target(g["map"])
    pg, st = gpu_ctx.ndt_align(g["scan"], g["init_pose"])
    dt, dr = pose_delta(pg, g["pose_ndt"])
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and st["iterations"] == int(g["iters_ndt"])


# ----------------------------------------------------------------------------------------------- error paths
def test_error_codes_and_empty_inputs(api, small_world):
    ctx = api.Context(0)
    try:
        opts = api.icp_opts(method=2)
        s, init = small_world["scan2k"], small_world["init_pose"]
        with pytest.raises(api.LocGpuError) as e:   # ScanMatch before SetInputTarget
            ctx.icp_align(s, init, opts)
        assert e.value.code == -3
        with pytest.raises(api.LocGpuError) as e:
            ctx.ndt_align(s, init)
        assert e.value.code == -3
        with pytest.raises(api.LocGpuError) as e:   # empty target (kdtree.cpp:12-15 returns false)
            ctx.icp_set_target(np.zeros((0, 3), dtype=np.float32))
        assert e.value.code == -1
        ctx.icp_set_target(small_world["map"])
        with pytest.raises(api.LocGpuError) as e:   # empty source
            ctx.icp_align(np.zeros((0, 3), dtype=np.float32), init, opts)
        assert e.value.code == -1
        with pytest.raises(api.LocGpuError) as e:
            ctx.knn(s, k=9)
        assert e.value.code == -1
        bad = api.icp_opts(method=2)
        bad.method = 7
        with pytest.raises(api.LocGpuError):
            ctx.icp_align(s, init, bad)
        # the context is still usable after errors
        pose, st = ctx.icp_align(s, init, opts)
        assert st["iterations"] >= 1 and np.all(np.isfinite(pose))
        # max_iteration = 0: the loop body never runs, pose = predict (icp cpp:358)
        pose0, st0 = ctx.icp_align(s, init, api.icp_opts(method=2, max_iteration=0))
        np.testing.assert_array_equal(pose0, init)
        assert st0["iterations"] == 0
    finally:
        ctx.close()


# ----------------------------------------------------------------------------------------------- bench configuration
def test_bench_config_parity_10m(gpu_ctx, api, locref, synth):
    """BASELINE configs[2] at full size: 115 200-pt scans vs the 10 M-pt map, reference-default P2Plane — GPU vs the oracle."""
    m = synth.make_map(10_000_000)
    gpu_ctx.icp_set_target(m)
    icp = locref.Icp(method=2)
    icp.set_target(m)
    info = gpu_ctx.icp_target_info()
    assert (info["num_leaves"], info["num_nodes"], info["depth"]) == tuple(icp.tree_info()[k] for k in ("num_leaves", "num_nodes", "depth"))
    scans = [synth.make_scan(sid) for sid in (11, 200)]
    inits = np.stack([synth.make_pose(sid)[1] for sid in (11, 200)])
    b = gpu_ctx.batch(scans)
    out, stats = gpu_ctx.icp_align_batch(b, inits, api.icp_opts(method=2))
    for i, s in enumerate(scans):
        ro = icp.align(s, inits[i])
        dt, dr = pose_delta(out[i], ro["pose"])
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and dt < 1e-9, (i, dt, dr)
        assert stats[i]["iterations"] == ro["iters"]
    # and the hot search kernel's neighbour lists themselves, at the poses an alignment starts from and ends at (230 400 queries each)
    tree = locref.KdTree(m)
    for poses in (inits, out):
        gpu_ctx.icp_hb_batch(b, poses, api.icp_opts(method=2))
        got = gpu_ctx.debug_batch_nn(b, 5)
        for i, s in enumerate(scans):
            q = locref.transform_points(poses[i], np.ascontiguousarray(s[:, :3], dtype=np.float64)).astype(np.float32)
            want = tree.knn(q, 5, approximate=True, alpha=0.1)
            assert np.array_equal(got[i, :len(s)], want), (i, int(np.sum(np.any(got[i, :len(s)] != want, axis=1))))
    b.close()


# ----------------------------------------------------------------------------------------------- properties at full size
def test_full_size_properties(gpu_ctx, api, synth):
    """BASELINE config 2 size (115 200-pt scan vs 1 M-pt map): size-independent properties, no oracle needed.
    (a) exact k-NN distances are sorted and the 1-NN of a map point is itself; (b) H is symmetric PSD;
    (c) aligning an already aligned scan moves it by less than the stopping tolerance; (d) determinism."""
    m = synth.make_map(1_000_000)
    s = synth.make_scan(7)
    true_pose, init = synth.make_pose(7)
    assert len(s) == 115200
    gpu_ctx.icp_set_target(m)
    q = m[::997, :3]
    idx = gpu_ctx.knn(q, k=5, approximate=False)
    d = ((q[:, None, :].astype(np.float64) - m[idx, :3].astype(np.float64)) ** 2).sum(-1)
    assert np.all(np.diff(d, axis=1) >= 0) and np.all(d[:, 0] == 0)
    opts = api.icp_opts(method=2)
    ok, H, B, eff = gpu_ctx.icp_hb(s, init, opts)
    assert ok and np.array_equal(H, H.T) and np.linalg.eigvalsh(H).min() > 0
    p1, st1 = gpu_ctx.icp_align(s, init, opts)
    p2, st2 = gpu_ctx.icp_align(s, init, opts)
    np.testing.assert_array_equal(p1, p2)  # fixed reduction order ⇒ bitwise reproducible
    p3, st3 = gpu_ctx.icp_align(s, p1, opts)
    dt, dr = pose_delta(p3, p1)
    # restarting from the result: the loop stops again within a few small steps (each |dx| ≈ eps = 1e-2 or less)
    assert st3["iterations"] <= 4 and dt < 5e-2 and dr < 1e-2
