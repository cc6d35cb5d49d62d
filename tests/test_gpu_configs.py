"""GPU parity tests at the sizes and compositions BASELINE.json's configs name (the HIP path through the C ABI vs the CPU oracle).

configs[2]  115 200-pt scans vs the 10 M-pt map — the direct-NDT half (the P2Plane half is test_gpu_parity.py::test_bench_config_parity_10m)
configs[3]  a 256-scan batch vs the 10 M-pt map: four spread-out scans against the oracle, all 256 finite, deterministic run to run
configs[4]  the composed streaming loop of Lio::AddCloud (lio.cpp:236-306): removeNaN → voxel filter → ScanMatch against the local map →
            keyframe every 3rd scan (transform, submap update, target re-ingest), eager and with the captured hipGraph
plus the degenerate neighbourhoods ring-structured LiDAR maps produce: five collinear neighbours (rank-2 plane-fit matrices).
"""
import numpy as np
import pytest

from conftest import pose_delta

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4    # north_star tolerance, metres
POSE_TOL_RAD = 1e-4  # north_star tolerance, radians


@pytest.fixture(scope="module")
def world10m(synth):
    return synth.make_map(10_000_000)


def xyzi(a):
    out = np.zeros((len(a), 4), np.float32)
    out[:, :3] = a[:, :3]
    out[:, 3] = (np.arange(len(a)) % 251).astype(np.float32)
    return out


# ----------------------------------------------------------------------------------------------- configs[2], NDT half
def test_ndt_10m_parity(gpu_ctx, locref, synth, world10m):
    """Direct NDT (NdtOptions defaults: voxel 1.0, NEARBY6; ndt_registration.cpp:87-148, 374-464), two full scans vs the 10 M-pt map:
    same voxel count, pose Δ < 1e-9 m / rad, equal iteration counts."""
    gpu_ctx.ndt_set_target(world10m)
    ndt = locref.Ndt()
    ndt.set_target(world10m)
    assert gpu_ctx.ndt_target_info()["num_voxels"] == ndt.num_voxels()
    for sid in (5, 130):
        s = synth.make_scan(sid)
        init = synth.make_pose(sid)[1]
        pg, st = gpu_ctx.ndt_align(s, init)
        ro = ndt.align(s, init)
        dt, dr = pose_delta(pg, ro["pose"])
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and dt < 1e-9 and dr < 1e-9, (sid, dt, dr)
        assert st["iterations"] == ro["iters"] and st["status"] == ro["status"] == 0


# ----------------------------------------------------------------------------------------------- configs[3], one GPU's view
def test_batch_256_scans_vs_10m(gpu_ctx, api, locref, synth, world10m):
    """The whole 256-scan batch of configs[3] on one GPU: scans 3, 77, 150 and 241 against the oracle (pose Δ < 1e-9 m, equal
    iterations), every pose finite and within the perturbation of its truth, and two runs bitwise equal (fixed reduction order)."""
    gpu_ctx.icp_set_target(world10m)
    sids = list(range(256))
    scans = [synth.make_scan(sid) for sid in sids]
    inits = np.stack([synth.make_pose(sid)[1] for sid in sids])
    truth = np.stack([synth.make_pose(sid)[0] for sid in sids])
    opts = api.icp_opts(method=api.P2PLANE)
    b = gpu_ctx.batch(scans)
    try:
        out1, st1 = gpu_ctx.icp_align_batch(b, inits, opts)
        out2, st2 = gpu_ctx.icp_align_batch(b, inits, opts)
    finally:
        b.close()
    np.testing.assert_array_equal(out1, out2)
    assert [s["iterations"] for s in st1] == [s["iterations"] for s in st2]
    assert np.all(np.isfinite(out1))
    assert np.all(np.linalg.norm(out1[:, 4:] - truth[:, 4:], axis=1) < 0.3)  # started ≤ 0.52 m away, ends within the stopping tolerance's reach
    assert all(1 <= s["iterations"] <= 20 for s in st1)
    icp = locref.Icp(method=locref.P2PLANE)
    icp.set_target(world10m)
    for sid in (3, 77, 150, 241):
        ro = icp.align(scans[sid], inits[sid])
        dt, dr = pose_delta(out1[sid], ro["pose"])
        assert dt < 1e-9 and dr < 1e-9, (sid, dt, dr)
        assert st1[sid]["iterations"] == ro["iters"]


# ----------------------------------------------------------------------------------------------- configs[4]
@pytest.mark.parametrize("graph,async_target", [(False, False), (True, False), (False, True)], ids=["eager", "hipgraph", "async_target"])
def test_streaming_loop_matches_oracle(api, locref, synth, graph, async_target):
    """Lio::AddCloud's loop (lio.cpp:236-306) composed from the resident entry points: 10 scans, keyframe every 3rd, at most 3
    keyframes in the local map (so the drop-oldest-and-rebuild branch runs), every pose and every local map equal to the oracle's."""
    ctx = api.Context(0)
    try:
        ctx.graph_enable(graph)
        opts = api.icp_opts(api.P2PLANE)
        sub = api.Submap(ctx, 3, 0.5)
        lm = locref.LocalMap(3, 0.5, order=locref.SORT_STABLE)
        icp_ref = locref.Icp(method=locref.P2PLANE)
        raw, filt = api.Cloud(ctx), api.Cloud(ctx)
        for s in range(10):
            scan = xyzi(synth.make_scan(s))
            scan[s::997, 1] = np.nan  # a few missing returns: removeNaN has something to do
            truth, init = synth.make_pose(s)
            raw.upload(scan, is_dense=False)
            raw.remove_nan(out=filt)
            filt.voxel_filter(0.5, out=filt)
            want_filt = locref.voxel_grid(locref.remove_nan(scan, False), True, 0.5, order=locref.SORT_STABLE)
            assert np.array_equal(filt.download(), want_filt)
            if s == 0:
                pose, kf_src, kf_dense = truth, filt, True       # first frame seeds the map with the FILTERED scan (lio.cpp:238-256)
            else:
                pose, st = ctx.icp_align_cloud(filt, init, opts)
                ro = icp_ref.align(want_filt, init)
                dt, dr = pose_delta(pose, ro["pose"])
                assert dt < 1e-8 and dr < 1e-8, (s, dt, dr)
                assert st["iterations"] == ro["iters"]
                kf_src, kf_dense = raw, False                     # later keyframes keep the RAW scan (lio.cpp:279)
            if s % 3 == 0:
                sub.add_keyframe(kf_src, pose)
                ctx.icp_set_target_cloud(sub.cloud(), wait=not async_target)  # async: the tree is built under the next scan's upload + filter
                lm.add_keyframe(locref.transform_cloud_f64(pose, kf_src.download(), is_dense=kf_dense), is_dense=kf_dense)
                icp_ref.set_target(lm.cloud()[:, :3])
                assert np.array_equal(sub.cloud().download(), lm.cloud(), equal_nan=True)
        assert sub.info[0] == 3  # four keyframes were added, the oldest was dropped
    finally:
        ctx.graph_enable(False)
        ctx.close()


def test_streaming_loop_as_a_two_stage_pipeline(api, locref, synth):
    """Round 4: the same loop as a two-stage front-end — one thread uploads and filters scan i+1 on context A while the caller matches
    scan i (and keeps the keyframe map) on context B of the same GPU; A's clouds are read by B (locgpu.h, "Two contexts on one GPU").
    Every pose equals the sequential loop's bit for bit and the oracle's to 1e-8."""
    import queue
    import threading
    n_scans, kf_every = 9, 3

    def prepared(s):
        scan = xyzi(synth.make_scan(s))
        scan[s::997, 1] = np.nan
        return scan

    def sequential():
        ctx = api.Context(0)
        opts = api.icp_opts(api.P2PLANE)
        sub = api.Submap(ctx, 3, 0.5)
        raw, filt = api.Cloud(ctx), api.Cloud(ctx)
        poses = []
        for s in range(n_scans):
            truth, init = synth.make_pose(s)
            raw.upload(prepared(s), is_dense=False)
            raw.voxel_filter(0.5, out=filt)
            if s == 0:
                pose, kf_src = truth, filt
            else:
                pose, _ = ctx.icp_align_cloud(filt, init, opts)
                kf_src = raw
            if s % kf_every == 0:
                sub.add_keyframe(kf_src, pose)
                ctx.icp_set_target_cloud(sub.cloud())
            poses.append(pose)
        ctx.close()
        return np.stack(poses)

    def pipelined():
        ctx_f, ctx_m = api.Context(0), api.Context(0)
        opts = api.icp_opts(api.P2PLANE)
        sub = api.Submap(ctx_m, 3, 0.5)
        pairs = [(api.Cloud(ctx_f), api.Cloud(ctx_f)) for _ in range(3)]  # the filter stage runs at most two scans ahead
        free, ready = queue.Queue(), queue.Queue()
        for p in pairs:
            free.put(p)
        errors = []

        def stage_filter():
            try:
                for s in range(n_scans):
                    raw, filt = free.get()
                    raw.upload(prepared(s), is_dense=False)
                    raw.voxel_filter(0.5, out=filt)
                    ready.put((s, raw, filt))
            except Exception as e:  # surfaced by the consumer
                errors.append(e)
                ready.put(None)

        th = threading.Thread(target=stage_filter)
        th.start()
        poses = []
        for _ in range(n_scans):
            item = ready.get()
            assert item is not None, errors
            s, raw, filt = item
            truth, init = synth.make_pose(s)
            if s == 0:
                pose, kf_src = truth, filt
            else:
                pose, _ = ctx_m.icp_align_cloud(filt, init, opts)   # a cloud of ctx_f as the source of ctx_m's matcher
                kf_src = raw
            if s % kf_every == 0:
                sub.add_keyframe(kf_src, pose)                       # … and as the keyframe of ctx_m's local map
                ctx_m.icp_set_target_cloud(sub.cloud())
            poses.append(pose)
            free.put((raw, filt))
        th.join()
        for raw, filt in pairs:
            raw.close(); filt.close()
        ctx_f.close(); ctx_m.close()
        return np.stack(poses)

    a, b = sequential(), pipelined()
    assert np.array_equal(a, b)
    # and the oracle, on the last scan (the whole chain of keyframes feeds it)
    lm = locref.LocalMap(3, 0.5, order=locref.SORT_STABLE)
    icp_ref = locref.Icp(method=locref.P2PLANE)
    for s in range(n_scans):
        scan = prepared(s)
        want_filt = locref.voxel_grid(scan, False, 0.5, order=locref.SORT_STABLE)
        truth, init = synth.make_pose(s)
        pose = truth if s == 0 else icp_ref.align(want_filt, init)["pose"]
        if s > 0:
            dt, dr = pose_delta(a[s], pose)
            assert dt < 1e-8 and dr < 1e-8, (s, dt, dr)
        if s % kf_every == 0:
            src, dense = (want_filt, True) if s == 0 else (scan, False)
            lm.add_keyframe(locref.transform_cloud_f64(a[s], src, is_dense=dense), is_dense=dense)
            icp_ref.set_target(lm.cloud()[:, :3])


def test_cpp_streaming_loop_sequential_and_pipelined(api, synth, tmp_path):
    """configs[4] from C++ through the C ABI (tests/cpp/stream_pipeline.cpp): the sequential loop and the two-stage pipeline give the
    same poses bit for bit, and they are the poses of the Python-driven loop above (same entry points, same order)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "perf"))
    import pipeline_microbench as pm
    assert os.path.exists(os.path.join(root, "tests", "cpp", "stream_pipeline")), "run __graft_entry__.build() first"
    res, poses = pm.cpp_stream_section(9, 3, 3, 0.5, passes=1, workdir=str(tmp_path))
    assert res["poses_identical"], res
    ctx = api.Context(0)
    try:
        opts = api.icp_opts(api.P2PLANE)
        sub = api.Submap(ctx, 3, 0.5)
        raw, filt = api.Cloud(ctx), api.Cloud(ctx)
        for s in range(9):
            truth, init = synth.make_pose(s)
            raw.upload(xyzi(synth.make_scan(s)), is_dense=False)
            raw.voxel_filter(0.5, out=filt)
            pose = truth if s == 0 else ctx.icp_align_cloud(filt, init, opts)[0]
            if s % 3 == 0:
                sub.add_keyframe(filt if s == 0 else raw, pose)
                ctx.icp_set_target_cloud(sub.cloud())
            assert np.array_equal(pose, poses[s]), s
    finally:
        ctx.close()


def test_graph_mode_with_grid_search(gpu_ctx, api, small_world):
    """hipGraph capture of the exact grid search (its second work list must exist before the capture starts)."""
    m, s, init = small_world["map"], small_world["scan10k"], small_world["init_pose"]
    gpu_ctx.icp_set_target(m)
    b = gpu_ctx.batch([s, small_world["scan2k"]])
    inits = np.stack([init, init])
    opts = api.icp_opts(method=api.P2PLANE, search_mode=api.SEARCH_GRID_EXACT)
    try:
        gpu_ctx.graph_enable(True)
        got, gst = gpu_ctx.icp_align_batch(b, inits, opts)   # first use of grid mode on this batch happens under capture
        gpu_ctx.graph_enable(False)
        want, wst = gpu_ctx.icp_align_batch(b, inits, opts)
        np.testing.assert_array_equal(got, want)
        assert [x["iterations"] for x in gst] == [x["iterations"] for x in wst]
    finally:
        gpu_ctx.graph_enable(False)
        b.close()


# ----------------------------------------------------------------------------------------------- degenerate neighbourhoods
def _lines_map(rng, exact):
    """A map made of straight lines only: every 5-neighbourhood is collinear, so FitPlane's 5×4 matrix has rank 2.
    exact=True: axis-parallel lines on a 2^-4 m lattice (float32 holds them exactly ⇒ exactly rank 2);
    exact=False: oblique lines rounded to float32 (rank 2 up to 1e-7 relative rounding of the inputs)."""
    pts = []
    t = np.arange(-20, 20, 0.0625)
    for i in range(60):
        o = np.round(rng.uniform(-20, 20, 3) * 16) / 16
        if exact:
            d = np.eye(3)[i % 3]
        else:
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
        pts.append(o[None, :] + t[:, None] * d[None, :])
    return np.concatenate(pts).astype(np.float32)


def _rings_map():
    """Scan-ring arcs on a ground plane (what a single LiDAR sweep leaves in a map): points dense along a ring, rings far apart,
    so five nearest neighbours sit on one arc — nearly collinear, curvature-limited rank 3."""
    pts = []
    for r in np.arange(4.0, 40.0, 1.5):
        az = np.arange(0, 2 * np.pi, 0.002)
        pts.append(np.stack([r * np.cos(az), r * np.sin(az), np.zeros_like(az)], 1))
    return np.concatenate(pts).astype(np.float32)


@pytest.mark.parametrize("world", ["lines_exact", "lines_oblique", "rings"])
@pytest.mark.parametrize("method", [1, 2], ids=["p2line", "p2plane"])
def test_collinear_neighbourhoods(gpu_ctx, api, locref, world, method):
    """Rank-deficient plane fits. For five collinear neighbours the smallest right singular vector of [x y z 1] is not unique
    (a two-dimensional null space): Eigen's JacobiSVD (math_utils.h:124-125) returns whichever its rotation sequence leaves, which
    cannot be known without Eigen. What IS defined by the reference: every such neighbourhood passes the fit test (any null vector
    has zero residual), so effective_num counts it (icp cpp:184). The oracle and the device use the same documented rule for the
    vector ("rank-deficient neighbourhoods"): the null-space direction closest to (0,0,0,1)."""
    rng = np.random.default_rng(11)
    m = _lines_map(rng, world == "lines_exact") if world.startswith("lines") else _rings_map()
    q = m[rng.choice(len(m), 3000, replace=False)].astype(np.float64) + rng.normal(0, 0.02, (3000, 3))
    s = q.astype(np.float32)
    pose = np.array([0.002, -0.001, 0.003, 1.0, 0.01, -0.02, 0.005])
    pose[:4] /= np.linalg.norm(pose[:4])
    gpu_ctx.icp_set_target(m)
    icp = locref.Icp(method=method)
    icp.set_target(m)
    ok_o, Ho, Bo, eff_o = icp.hb(s, pose)
    ok_g, Hg, Bg, eff_g = gpu_ctx.icp_hb(s, pose, api.icp_opts(method=method))
    assert eff_g == eff_o and ok_g == ok_o
    assert eff_o > 2000
    scale = max(np.abs(Ho).max(), 1e-300)
    assert np.abs(Hg - Ho).max() <= 1e-6 * scale, np.abs(Hg - Ho).max() / scale
    assert np.abs(Bg - Bo).max() <= 1e-6 * max(np.abs(Bo).max(), scale * 1e-6)


# ----------------------------------------------------------------------------------------------- configs[3], the exchange step
def test_sharded_batch_over_rccl_equals_plain_batch(api, small_world):
    """The sharded mode's data path on one GPU: an RCCL communicator of ONE rank (ncclCommInitRank + ncclAllReduce really run),
    per-scan sums → all-reduce → solve on every rank. A batch this rank holds completely must reproduce the plain batch bit for
    bit; held partially (scans [1, 3) of 3) the scans it holds are unchanged and the one it does not hold gets no residuals
    (effective_num = 0 ⇒ 20 no-op iterations, pose = initial guess: icp_registration.cpp:204-207,358-376)."""
    from loc_lib_amd import multi_gpu
    m = small_world["map"]
    scans = [small_world["scan10k"], small_world["scan2k"], small_world["scan10k"][::3]]
    init = small_world["init_pose"]
    inits = np.stack([init, init, init])
    inits[1, 4:] += [0.04, -0.03, 0.01]
    ctx = api.Context(0)
    try:
        assert multi_gpu.init_comm(ctx, None) == (0, 1) and ctx.comm_info() == (0, 1)
        ctx.icp_set_target_bcast(m, root=0)          # collective SetInputTarget (root builds, ncclBroadcast of the packed tree)
        ctx.ndt_set_target(m)
        plain = ctx.batch(scans)
        full = multi_gpu.scan_sharded_batch(ctx, scans, 3, 0, 1)
        part = ctx.batch(scans[1:], first=1, n_total=3)
        try:
            for method in (api.P2PLANE, api.P2P):
                opts = api.icp_opts(method=method)
                want, wst = ctx.icp_align_batch(plain, inits, opts)
                got, gst = ctx.icp_align_batch(full, inits, opts)
                np.testing.assert_array_equal(got, want)
                assert [s["iterations"] for s in gst] == [s["iterations"] for s in wst]
                got, gst = ctx.icp_align_batch(part, inits, opts)
                np.testing.assert_array_equal(got[1:], want[1:])
                np.testing.assert_array_equal(got[0], inits[0])
                assert gst[0]["iterations"] == 20 and gst[0]["last_effective_num"] == 0
                hb_w = ctx.icp_hb_batch(plain, inits, opts)
                hb_g = ctx.icp_hb_batch(full, inits, opts)
                np.testing.assert_array_equal(hb_g, hb_w)
            want, _ = ctx.ndt_align_batch(plain, inits)
            got, _ = ctx.ndt_align_batch(full, inits)
            np.testing.assert_array_equal(got, want)
            # point sharding degenerates to the same thing with one rank
            pts = multi_gpu.point_sharded_batch(ctx, scans, 0, 1)
            got, _ = ctx.icp_align_batch(pts, inits, api.icp_opts(method=api.P2PLANE))
            want, _ = ctx.icp_align_batch(plain, inits, api.icp_opts(method=api.P2PLANE))
            np.testing.assert_array_equal(got, want)
            pts.close()
        finally:
            plain.close(); full.close(); part.close()
    finally:
        ctx.close()


def test_async_upload_double_buffer(gpu_ctx, api, small_world):
    """locgpu_batch_upload_async: two batches alternating as a double buffer give the poses of freshly created batches; ragged
    counts, a re-upload with other scans, and the error for a scan larger than the reserved capacity."""
    m = small_world["map"]
    a, b2, c = small_world["scan10k"], small_world["scan2k"], small_world["scan10k"][::3]
    init = small_world["init_pose"]
    inits = np.stack([init, init])
    gpu_ctx.icp_set_target(m)
    opts = api.icp_opts(method=api.P2PLANE)
    sets = [[a, b2], [c, a], [b2, c], [a[:5000], np.ascontiguousarray(a[::2])]]
    want = []
    for s in sets:
        fresh = gpu_ctx.batch(s)
        want.append(gpu_ctx.icp_align_batch(fresh, inits, opts)[0])
        fresh.close()
    bufs = [gpu_ctx.batch_empty(2, len(a)), gpu_ctx.batch_empty(2, len(a))]
    try:
        bufs[0].upload_async(sets[0])
        for i, s in enumerate(sets):
            if i + 1 < len(sets):
                bufs[(i + 1) % 2].upload_async(sets[i + 1])   # runs under the align call below
            got, _ = gpu_ctx.icp_align_batch(bufs[i % 2], inits, opts)
            np.testing.assert_array_equal(got, want[i])
        # strided host layout (pcl::PointXYZI: 32 bytes per point)
        wide = [np.zeros((len(x), 8), np.float32) for x in sets[0]]
        for w, x in zip(wide, sets[0]):
            w[:, :3] = x[:, :3]
            w[:, 4] = 7.0
        bufs[0].upload_async(wide)
        got, _ = gpu_ctx.icp_align_batch(bufs[0], inits, opts)
        np.testing.assert_array_equal(got, want[0])
        big = np.zeros((len(a) + 1, 3), np.float32)
        with pytest.raises(api.LocGpuError) as e:
            bufs[0].upload_async([big, a])
        assert e.value.code == -1
    finally:
        for x in bufs:
            x.close()


# ----------------------------------------------------------------------------------------------- exact-search grid ingest (device)
@pytest.mark.parametrize("case", ["city", "duplicates"])
def test_device_grid_invariants(api, locref, synth, case):
    """Grid ingest on the GPU (csrc/grid_build.hip): holds exactly the tree's leaves, sorted by (tile, cell inside the tile); every
    occupied tile has a record (first leaf, 65 prefix sums of its cells' counts) reachable through the tile hash; the cell edge
    is near the target occupancy."""
    import ctypes
    rng = np.random.RandomState(11)
    if case == "city":
        pts = synth.make_local_map(80000, 3, half=30.0)
    else:
        pts = (rng.rand(5000, 3) * 8).astype(np.float32)
        pts[50:300] = pts[50]
    ctx = api.Context(0)
    try:
        ctx.icp_set_target(pts)
        L = api.lib()
        L.locgpu_debug_grid_dump.restype = ctypes.c_size_t
        L.locgpu_debug_grid_dump.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.c_void_p, ctypes.c_size_t]
        info = np.zeros(9, np.int64)
        prm = np.zeros(6, np.float32)
        n = L.locgpu_debug_grid_dump(ctx._h, info.ctypes.data, prm.ctypes.data, None, 0, None, 0, None, 0)
        tree = locref.KdTree(pts)
        assert n == tree.num_leaves
        dims, n_occ, n_tocc, cap, tdims = info[:3], int(info[3]), int(info[4]), int(info[5]), info[6:9]
        gp = np.zeros((n, 4), np.float32)
        rec = np.zeros(n_tocc, dtype=np.dtype([("pt_start", "<u4"), ("tile_lin", "<u4"), ("cstart", "<u2", (66,))]))
        assert rec.dtype.itemsize == 140
        hsh = np.zeros((cap, 2), np.uint32)
        L.locgpu_debug_grid_dump(ctx._h, info.ctypes.data, prm.ctypes.data, gp.ctypes.data, gp.size, rec.ctypes.data, rec.nbytes, hsh.ctypes.data, hsh.size)
        c = np.floor((gp[:, :3] - prm[:3]) * prm[4]).astype(np.int64)
        assert np.all(c >= 0) and np.all(c < dims)
        t = c // 4
        tile_lin = (t[:, 2] * tdims[1] + t[:, 1]) * tdims[0] + t[:, 0]
        in_tile = ((c[:, 2] % 4) * 4 + (c[:, 1] % 4)) * 4 + (c[:, 0] % 4)
        key = tile_lin * 64 + in_tile
        assert np.all(np.diff(key) >= 0)                                        # (tile, cell) order
        cells_per_pt = len(np.unique(key))
        assert cells_per_pt == n_occ and 1.5 <= n / n_occ <= 16.0               # the edge targets ≈4 leaves per occupied cell
        tiles, starts, counts = np.unique(tile_lin, return_index=True, return_counts=True)
        assert len(tiles) == n_tocc and cap >= 2 * n_tocc
        np.testing.assert_array_equal(rec["tile_lin"].astype(np.int64), tiles)
        np.testing.assert_array_equal(rec["pt_start"].astype(np.int64), starts)
        np.testing.assert_array_equal(rec["cstart"][:, 64].astype(np.int64), counts)
        hist = np.zeros((n_tocc, 64), np.int64)
        np.add.at(hist, (np.searchsorted(tiles, tile_lin), in_tile), 1)
        want_cstart = np.concatenate([np.zeros((n_tocc, 1), np.int64), np.cumsum(hist, axis=1)], axis=1)
        np.testing.assert_array_equal(rec["cstart"][:, :65].astype(np.int64), want_cstart)
        used = hsh[hsh[:, 0] != 0xFFFFFFFF]
        order = np.argsort(used[:, 0])
        np.testing.assert_array_equal(used[order, 0].astype(np.int64), tiles)
        np.testing.assert_array_equal(used[order, 1].astype(np.int64), np.arange(n_tocc))
        # the points are exactly the tree's leaf points (multiset)
        _, _, pidx = tree.dump()
        xyz = np.ascontiguousarray(pts[:, :3], dtype=np.float32)
        leaf_pts = xyz[pidx[pidx >= 0]]
        a = np.sort(leaf_pts.view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel(), order=["x", "y", "z"])
        b = np.sort(np.ascontiguousarray(gp[:, :3]).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel(), order=["x", "y", "z"])
        assert np.array_equal(a, b)
    finally:
        ctx.close()


# ----------------------------------------------------------------------------------------------- BfnnRegistration
@pytest.mark.parametrize("k", [1, 5, 8])
def test_bfnn_matches_brute_force(gpu_ctx, api, locref, small_world, k):
    """BfnnRegistration::FindNearstPoints (bfnn.cpp:24-50): indices of the k smallest float32 distances, ascending, ties by index;
    lattice points give many exact ties. k larger than the cloud is an error (the reference reads past the end of its array)."""
    m = small_world["map"][:50000]
    rng = np.random.RandomState(7 + k)
    q = (m[rng.choice(len(m), 64, replace=False), :3] + rng.randn(64, 3).astype(np.float32) * 0.1).astype(np.float32)
    gpu_ctx.bfnn_set_target(m)
    np.testing.assert_array_equal(gpu_ctx.bfnn_knn(q, k=k), locref.bfnn_knn(m, q, k))
    g = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(5), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    ql = (g[::11] + 0.5).astype(np.float32)
    gpu_ctx.bfnn_set_target(g)
    np.testing.assert_array_equal(gpu_ctx.bfnn_knn(ql, k=k), locref.bfnn_knn(g, ql, k))
    gpu_ctx.bfnn_set_target(g[:3])
    if k > 3:
        with pytest.raises(api.LocGpuError) as e:
            gpu_ctx.bfnn_knn(ql, k=k)
        assert e.value.code == -4


# ----------------------------------------------------------------------------------------------- bench.py contract
@pytest.mark.parametrize("mode", ["weak_streaming", "strong"])  # --resident differs by one flag; each mode is a fresh interpreter that imports torch
def test_bench_line_contract(mode):
    """bench.py on a reduced workload (1 M-pt map, 8 scans): exactly one JSON line with the keys the driver and the judge read;
    the streaming default really has the scan copy inside the timed region; the strong mode runs the sharded batch over RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--map-points", "1000000", "--no-cpu-baseline",
           "--traffic", "none"]
    if mode == "strong":
        cmd += ["--scaling", "strong", "--total-scans", "8"]
    else:
        cmd += ["--scans-per-gpu", "8", "--pool-slots", "0"] + (["--resident"] if mode == "resident" else [])  # plain batches; the strong mode takes the default: the pool
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["unit"] == "scans/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["vs_baseline"] is None
    assert d["scaling"] == ("strong" if mode == "strong" else "weak")
    assert d["config"]["scan_h2d_in_timed_region"] == (mode != "resident")
    assert d["h2d_bytes_per_step"] == (0 if mode == "resident" else 8 * 115200 * 16)
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac", "nominal_bytes_frac"):
        assert key in r, key
    # round 5: the fraction is the bound that binds (vector-instruction issue; None here: --traffic none collects no counters) and can
    # never pass 1; SURVEY 8(d)'s byte model is an effective rate beside it (most node reads are cache hits, so THAT can pass 1)
    assert r["bound"] == "valu_issue" and (r["frac"] is None or 0 < r["frac"] <= 1) and 0 < r["nominal_bytes_frac"] < 2
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-3 * d["value"]
    assert d["median_translation_error_to_truth_m"] < 0.1
    # the workload label follows the arguments (a 1 M-pt map is configs[1], the sharded mode configs[3]); small shards run three in flight
    assert ("configs[3]" if mode == "strong" else "configs[1]") in d["config"]["workload"]
    assert "roofline_k2" in d
    if mode == "strong":  # a small step goes through the open-scan pool: room for 256 scans of this rank = 32 steps in flight
        assert d["pool"]["slots"] == 256 and d["config"]["pipeline_depth"] == 32 and "pool" in d["config"]["workload"]
    else:
        assert d["config"]["pipeline_depth"] == 3 and "pool" not in d
    assert "lane_efficiency" in r and 0 < r["compulsory_bytes"] < r["algorithmic_bytes_per_launch"]
    assert d["roofline_k2"]["bound"].startswith("valu_issue") and "fp64_issue_frac" in d["roofline_k2"]
    assert d["scans_per_rank"] == 8 and d["rccl_ranks"] == 1
    assert "further steps" in d["kernel_ms_source"] and d["kernel_ms_per_step"]["search"] > 0


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` brings up N ranks itself — and on a box with fewer GPUs it fails loudly instead of
    running one rank and printing n_gpus 1 (what a SCALE run would otherwise record eight times)."""
    import os
    import subprocess
    import sys
    from loc_lib_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = api.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and ("--gpus %d but this node has %d GPU" % (n, n - 1)) in out.stderr, out.stderr[-1500:]
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


# ----------------------------------------------------------------------------------------------- other LDS stack depths
@pytest.mark.parametrize("rows", ["12", "15", "24"])
def test_search_parity_at_other_stack_depths(rows):
    """The fast traversal stores `LOCGPU_FAST_STACK` stack rows in LDS and keeps the levels above them as candidates (direct
    expansion, replay) — with 12 rows the replay and overflow paths run ~100× more often than at the default 15, with 24 hardly
    ever. The index-list / H,B / alignment parity tests must pass unchanged either way (the setting is read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LOCGPU_FAST_STACK=rows)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), os.path.join(root, "tests", "test_gpu_configs.py"),
                          "-q", "-m", "gpu", "-x", "-k", "knn or hb or (align and not sharded and not ndt) or golden or hot_search"], env=env, capture_output=True, text=True,
                         timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-1000:]
    assert " passed" in out.stdout


# ----------------------------------------------------------------------------------------------- straggler hand-over
def test_profile_modes_and_marshalled_scans():
    """locgpu_profile_enable: 1 times all three stages, 2 only the search stage (what bench.py uses inside its timed region); a
    pre-marshalled scan list uploads the same bytes as the list itself."""
    from loc_lib_amd import api, synth
    ctx = api.Context(0)
    m = synth.make_map(200_000)
    ctx.icp_set_target(m)
    scans = [synth.make_scan(i, subsample=20000) for i in range(4)]
    inits = np.stack([synth.make_pose(i)[1] for i in range(4)])
    opts = api.icp_opts(method=api.P2PLANE)
    b = ctx.batch(scans)
    ref, _ = ctx.icp_align_batch(b, inits, opts)
    for mode, stages in ((1, ("search", "accum", "solve")), (2, ("search",))):
        ctx.profile_read(reset=True)
        ctx.profile_enable(mode)
        poses, _ = ctx.icp_align_batch(b, inits, opts)
        ctx.profile_enable(False)
        p = ctx.profile_read(reset=True)
        assert np.array_equal(poses, ref)
        for st in ("search", "accum", "solve"):
            if st in stages:
                assert p[st + "_n"] > 0 and p[st + "_ms"] > 0
            else:
                assert p[st + "_n"] == 0 and p[st + "_ms"] == 0
    b2 = ctx.batch_empty(4, max(len(s) for s in scans)) if hasattr(ctx, "batch_empty") else ctx.batch(scans)
    b2.upload_async(api.MarshalledScans(scans))
    b2.upload_wait()
    poses2, _ = ctx.icp_align_batch(b2, inits, opts)
    assert np.array_equal(poses2, ref)


# ----------------------------------------------------------------------------------------------- the hot search kernel's own lists
def _oracle_lists(locref, tree, scan, pose, k, approximate):
    q = locref.transform_points(pose, np.ascontiguousarray(scan[:, :3], dtype=np.float64)).astype(np.float32)  # SE3·p in f64, then ToPointType
    return tree.knn(q, k, approximate=approximate, alpha=0.1)


@pytest.mark.parametrize("approximate", [True, False])
def test_hot_search_kernel_lists_equal_oracle(approximate):
    """The neighbour lists of the search stage that the alignments actually run (icp_search_fast_kernel: un-stored top levels,
    direct expansion of the candidates, in-wave exact traversal, redo list) — read back through locgpu_debug_batch_nn — equal the
    oracle's KdTree::GetClosestPoint lists index for index, in order: ragged batch, poses from far off to converged, ANN and exact."""
    from loc_lib_amd import api, synth
    from oracle import locref
    ctx = api.Context(0)
    m = synth.make_map(1_000_000)
    ctx.icp_set_target(m)
    tree = locref.KdTree(m)
    scans = [synth.make_scan(0), synth.make_scan(5, subsample=30000), synth.make_scan(9, subsample=777)]
    truth = [synth.make_pose(s)[0] for s in (0, 5, 9)]
    inits = [synth.make_pose(s)[1] for s in (0, 5, 9)]
    far = [synth.make_pose(s, trans_amp=3.0, rot_amp_deg=20.0)[1] for s in (0, 5, 9)]
    b = ctx.batch(scans)
    opts = api.icp_opts(method=api.P2PLANE)
    opts.approximate = 1 if approximate else 0
    for poses in (inits, truth, far):
        ctx.icp_hb_batch(b, np.stack(poses), opts)
        got = ctx.debug_batch_nn(b, 5)
        for i, (scan, pose) in enumerate(zip(scans, poses)):
            want = _oracle_lists(locref, tree, scan, pose, 5, approximate)
            assert np.array_equal(got[i, :len(scan)], want), (i, int(np.sum(np.any(got[i, :len(scan)] != want, axis=1))))


def test_hot_search_kernel_lists_on_ties_duplicates_and_p2p():
    """Same check where the fast traversal cannot finish on its own: a lattice map (every distance ties → exact traversal in the
    wave or through the redo list), a map of duplicated points, and the k = 1 search of the point-to-point method."""
    from loc_lib_amd import api
    from oracle import locref
    rng = np.random.default_rng(5)
    g = np.arange(-12, 13, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, g[:6], indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float32)
    dup = np.repeat(rng.uniform(-20, 20, size=(4000, 3)).astype(np.float32), 3, axis=0)
    pose = np.array([0.01, -0.02, 0.03, 0.0, 0.11, -0.07, 0.05])
    pose[3] = np.sqrt(1.0 - np.sum(pose[:3] ** 2))
    for cloud in (lattice, dup):
        ctx = api.Context(0)
        ctx.icp_set_target(cloud)
        tree = locref.KdTree(cloud)
        scan = (cloud[rng.choice(len(cloud), 5000)] + rng.normal(0, 0.05, size=(5000, 3))).astype(np.float32)
        scan_exact = cloud[rng.choice(len(cloud), 3000)].copy()  # queries ON map points: zero distances, ties everywhere
        for s in (scan, scan_exact):
            b = ctx.batch([s])
            for method, k in ((api.P2PLANE, 5), (api.P2P, 1)):
                opts = api.icp_opts(method=method)
                ctx.icp_hb_batch(b, pose[None], opts)
                got = ctx.debug_batch_nn(b, k)[0, :len(s)]
                want = _oracle_lists(locref, tree, s, pose, k, True)
                assert np.array_equal(got, want), (len(cloud), method, int(np.sum(np.any(got != want, axis=1))))
            b.close()
        ctx.close() if hasattr(ctx, "close") else None


def _lines_cloud(rng, n):
    """Eight straight lines through a 60 m box: a deep, unbalanced tree whose split planes pass within centimetres of most queries."""
    t = rng.uniform(-100, 100, size=(n, 1))
    d = rng.normal(size=(8, 3))
    o = rng.uniform(-30, 30, size=(8, 3))
    i = rng.integers(0, 8, n)
    return (o[i] + t * d[i] / np.linalg.norm(d[i], axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("seed,n", [(5, 100_000), (3, 1_000_000)])
def test_hot_search_kernel_lists_when_a_candidate_descent_outgrows_the_stack(seed, n):
    """Found by tools/fuzz_search.py on the batch kernel (round 3): the descent from a candidate of the un-stored levels starts above
    the stored levels and, on a map of straight lines, can push more surviving rows than the stack has — pushes beyond the last row
    fall off the LDS allocation. Such a query must go to the deep pass (every level stored), not keep a list that misses a
    neighbour: 20 000 queries 3 m (σ) off the lines, eight copies so that the batch kernel runs (> 2048 waves), ANN lists index for
    index. (n = 1 M trips it at the default 15 rows, n = 100 k at the 12 rows of test_search_parity_at_other_stack_depths.)"""
    from loc_lib_amd import api
    from oracle import locref
    rng = np.random.default_rng(seed)
    cloud = _lines_cloud(rng, n)
    nq = 20000
    scan = (cloud[rng.integers(0, n, nq)].astype(np.float64) + rng.normal(0, 3.0, size=(nq, 3))).astype(np.float32)
    pose = np.array([0, 0, 0, 1, 0.3, -0.2, 0.1])
    ctx = api.Context(0)
    try:
        ctx.icp_set_target(cloud)
        tree = locref.KdTree(cloud)
        b = ctx.batch([scan] * 8)
        for method, k in ((api.P2PLANE, 5), (api.P2P, 1)):
            opts = api.icp_opts(method=method)
            ctx.icp_hb_batch(b, np.stack([pose] * 8), opts)
            got = ctx.debug_batch_nn(b, k)
            want = _oracle_lists(locref, tree, scan, pose, k, True)
            for i in (0, 7):
                assert np.array_equal(got[i, :nq], want), (k, i, int(np.sum(np.any(got[i, :nq] != want, axis=1))))
        b.close()
    finally:
        ctx.close()


def test_nothing_leaks_over_create_use_destroy_cycles():
    """Everything the C ABI hands out — contexts, targets (ICP, NDT, asynchronous), batches, a captured graph, clouds, a submap — created,
    used and destroyed 80 times after a warm-up: the device memory comes back to the byte-ish (hipMemGetInfo) and the process's resident
    set does not grow by more than a few MB (tools/leak_check.py; 300 cycles: 0.0 MB of device memory, +0.3 MB of RSS)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "leak_check.py"), "--reps", "80"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r"device memory not returned ([-0-9.]+) MB .* RSS growth ([-0-9.]+) MB .* threads ([-+0-9]+)", r.stdout)
    assert m, r.stdout[-500:]
    assert float(m.group(1)) < 2.0 and float(m.group(2)) < 8.0 and int(m.group(3)) <= 0, r.stdout[-500:]  # round 6: no helper thread left behind either


def test_two_matchers_on_one_gpu_concurrently():
    """Two threads of one process, each with its own context, map and scans, running SetInputTarget, single-scan and batch alignments
    and direct NDT at the same time (what two robots' matchers sharing one GPU would do): the process-wide pieces — the host build's
    thread pool and scratch cache, the library's statics, the HIP runtime's streams — are shared. Every pose and iteration count must be
    the one the same calls give when run alone, bit for bit (tests/gpu_two_matchers_case.py)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_two_matchers_case.py"), "10"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["differing"] == 0, out


def test_one_scan_calls_do_not_depend_on_the_call_before(api, synth):
    """Round 4: a one-scan alignment sizes its first chunk of iterations by the call before it (a streaming front-end converges in 4-5
    iterations and used to pay for eight), reads a resident cloud's points in place, and skips the zeroing of the search stage's
    work-list counters when the previous alignment ran to its end. None of that may show in a result: calls with far, exact and
    near initial poses — 10+, 1-2 and a handful of iterations — in every order, through the resident-cloud entry point, the host
    pointer entry point and a one-scan batch, with an H/B evaluation and an NDT call in between, give the poses, iteration counts
    and dx norms of the same call made first on a fresh context, bit for bit."""
    m = synth.make_local_map(300000, 5, half=40.0)
    scan = synth.make_scan(5, crop_half=36.0)[::3].copy()
    xyzi = np.concatenate([scan[:, :3], np.zeros((len(scan), 1), np.float32)], axis=1).astype(np.float32)
    truth, init = synth.make_pose(5)
    far = init.copy(); far[4:] += [0.6, -0.5, 0.1]
    opts = api.icp_opts(method=api.P2PLANE)
    inits = dict(far=far, exact=truth, near=init)

    def fresh(name):
        ctx = api.Context(0)
        ctx.icp_set_target(m)
        c = api.Cloud(ctx, xyzi)
        r = ctx.icp_align_cloud(c, inits[name], opts)
        c.close(); ctx.close()
        return r

    want = {k: fresh(k) for k in inits}
    assert want["far"][1]["iterations"] > 8 and want["exact"][1]["iterations"] <= 3, (want["far"][1], want["exact"][1])

    def same(got, name):
        assert np.array_equal(got[0], want[name][0]), name
        for key in ("iterations", "converged", "last_effective_num", "last_dx_norm"):
            assert got[1][key] == want[name][1][key], (name, key)

    ctx = api.Context(0)
    ctx.icp_set_target(m)
    ctx.ndt_set_target(m)
    c = api.Cloud(ctx, xyzi)
    b = ctx.batch([scan])
    order = ["exact", "far", "near", "near", "far", "exact", "exact", "near", "far", "far"]
    for i, name in enumerate(order):
        same(ctx.icp_align_cloud(c, inits[name], opts), name)
        if i % 3 == 1:
            p, st = ctx.icp_align(scan, inits[name], opts)            # host pointer entry: the same reusable one-scan batch, its own buffer
            same((p, st), name)
        if i % 3 == 2:
            ctx.icp_hb_batch(b, inits[name][None], opts)              # leaves through the solve kernel without an update
            ctx.ndt_align_batch(b, inits[name][None])
        pb, stb = ctx.icp_align_batch(b, inits[name][None], opts)
        same((pb[0], stb[0]), name)
    b.close(); c.close(); ctx.close()


def test_secular_plane_fit_agrees_with_the_four_column_fit(tmp_path):
    """Round 4: the P2Plane fit kernel finds FitPlane's 4-vector (math_utils.h:112-136) from the 3×3 eigen-decomposition of the centred
    neighbours plus the secular equation of the homogeneous column (device_math.hpp plane_null_vector_secular; LOCGPU_PLANE_FIT=1, the
    default) instead of a 4-column one-sided Jacobi SVD (LOCGPU_PLANE_FIT=0). Both compute the same singular vector to rounding: on a
    ragged eight-scan batch the iteration counts are equal, poses agree to 1e-10, and one H/B evaluation (each at its own poses, which
    differ by ~1e-13) to 1e-9 of its scale with equal effective_num / ok. (Each against the CPU checker: test_gpu_parity.py and
    tools/fuzz_hb.py run with the default.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, val in (("secular", "1"), ("jacobi4", "0")):
        f = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_plane_fit_case.py"), f],
                           env=dict(os.environ, LOCGPU_PLANE_FIT=val), capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[tag] = np.load(f)
    a, b = outs["secular"], outs["jacobi4"]
    assert np.array_equal(a["it"], b["it"]), (a["it"], b["it"])
    assert np.abs(a["pose"] - b["pose"]).max() < 1e-10
    hb_a, hb_b = np.asarray(a["hb"]), np.asarray(b["hb"])
    assert not np.array_equal(hb_a, hb_b)  # two different routes to the vector: equal bits would mean the switch did nothing
    scale = np.abs(hb_b[:, :36]).max(axis=1, keepdims=True)
    assert (np.abs(hb_a[:, :42] - hb_b[:, :42]) <= 1e-9 * scale).all()
    assert np.array_equal(hb_a[:, 42:44], hb_b[:, 42:44])  # effective_num, ok


def test_paced_one_scan_alignment_equals_the_chunked_one(tmp_path):
    """A one-scan eager alignment is paced from the host (locgpu_api.hip: the solve kernel posts an iteration word — and the finished
    scan's result under a checksum — to pinned host memory; the host keeps one iteration queued ahead) instead of running in chunks
    sized by the call before (LOCGPU_PACE_AHEAD=0). Same kernels, same data, same order: poses, iteration counts and stats are equal
    bit for bit — P2Plane / P2P / P2Line / direct NDT, a start that converges at once, far starts, runs cut short by max_iteration, a
    scan that never has enough points, new uploads right behind a call, the host-pointer call. Reference loop: icp_registration.cpp:358-376."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, val in (("paced", "1"), ("chunked", "0")):
        f = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_paced_case.py"), f],
                           env=dict(os.environ, LOCGPU_PACE_AHEAD=val), capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[tag] = np.load(f)
    a, b = outs["paced"], outs["chunked"]
    assert len(a["it"]) >= 20 and a["it"].max() >= 9 and a["it"].min() <= 2, a["it"]  # short and long runs both present
    assert np.array_equal(a["it"], b["it"]), (a["it"], b["it"])
    assert np.array_equal(a["pose"], b["pose"])
    assert np.array_equal(a["stats"], b["stats"], equal_nan=True)


def test_source_overwritten_or_freed_right_behind_a_paced_call(api, small_world):
    """a paced one-scan call returns the moment the solve kernel's post says the scan is done — up to one idle iteration
    is still queued behind it, and its kernels must test `done` before they touch the counts or the points. So: a resident cloud is
    OVERWRITTEN (another scan uploaded into it) or DESTROYED right behind the call, and a host array is scribbled over right behind
    the host-pointer call, forty times each with no pause; every following call must give the bits of the same call on a fresh
    context, and nothing may fault."""
    m = small_world["map"]
    s = small_world["scan10k"]
    pose = small_world["init_pose"]
    scans = [np.ascontiguousarray(s[i::3][:3000]) for i in range(3)]
    opts = api.icp_opts(method=api.P2PLANE)
    fresh = api.Context(0)
    fresh.icp_set_target(m)
    want = [fresh.icp_align(x, pose, opts) for x in scans]
    fresh.close()
    ctx = api.Context(0)
    ctx.icp_set_target(m)

    def xyzi(x):
        a = np.zeros((len(x), 4), np.float32)
        a[:, :3] = x[:, :3]
        return a

    cloud = api.Cloud(ctx, xyzi(scans[0]))
    for rep in range(40):
        k = rep % 3
        got = ctx.icp_align_cloud(cloud, pose, opts)
        assert np.array_equal(got[0], want[k][0]) and got[1] == want[k][1], (rep, "resident")
        if rep % 2:
            cloud.close()                                  # freed right behind the call …
            cloud = api.Cloud(ctx, xyzi(scans[(k + 1) % 3]))
        else:
            cloud.upload(xyzi(scans[(k + 1) % 3]))          # … or overwritten in place
    cloud.close()
    for rep in range(40):
        k = rep % 3
        buf = scans[k].copy()
        got = ctx.icp_align(buf, pose, opts)
        buf[:] = np.nan                                    # the caller's array is the caller's again
        assert np.array_equal(got[0], want[k][0]) and got[1] == want[k][1], (rep, "host pointer")
    ctx.close()


def test_align_begin_end_two_batches_in_flight(gpu_ctx, api, small_world):
    """locgpu_*_align_batch_begin / locgpu_align_batch_end: two batches (different scans, ragged counts) begun back to back and ended
    in order give bit for bit the poses, iteration counts and stats of the blocking calls — ICP and direct NDT — also when the
    uploads of the next round are started while both alignments are in flight; a second begin on a pending batch and an end
    without a begin are refused."""
    m = small_world["map"]
    s = small_world["scan10k"]
    pose = small_world["init_pose"]
    gpu_ctx.icp_set_target(m)
    gpu_ctx.ndt_set_target(m)
    sets = [[s, s[:7000], s[100:4100]], [s[::2], s[:9000], s[50:8050]]]
    poses = [np.stack([pose, pose, pose]), np.stack([pose, pose, pose])]
    for p in poses:
        p[1, 4] += 0.05
        p[2, 5] -= 0.04
    opts = api.icp_opts(method=api.P2PLANE)
    ref = []
    for scans, ip in zip(sets, poses):
        b = gpu_ctx.batch(scans)
        ref.append((gpu_ctx.icp_align_batch(b, ip, opts), gpu_ctx.ndt_align_batch(b, ip)))
        b.close()
    ba, bb = gpu_ctx.batch_empty(3, 10000), gpu_ctx.batch_empty(3, 10000)
    for rnd in range(2):
        ba.upload_async(api.MarshalledScans(sets[0]))
        ba.upload_wait()
        bb.upload_async(api.MarshalledScans(sets[1]))
        gpu_ctx.icp_align_batch_begin(ba, poses[0], opts)
        gpu_ctx.icp_align_batch_begin(bb, poses[1], opts)
        with pytest.raises(RuntimeError):
            gpu_ctx.icp_align_batch_begin(ba, poses[0], opts)
        with pytest.raises(RuntimeError):  # no upload into a batch between begin and end (later chunks would read the new scans)
            ba.upload_async(api.MarshalledScans(sets[1]))
        pa, sa = gpu_ctx.align_batch_end(ba)
        pb, sb = gpu_ctx.align_batch_end(bb)
        with pytest.raises(RuntimeError):
            gpu_ctx.align_batch_end(ba)
        assert np.array_equal(pa, ref[0][0][0]) and np.array_equal(pb, ref[1][0][0])
        assert [x["iterations"] for x in sa] == [x["iterations"] for x in ref[0][0][1]]
        assert [x["iterations"] for x in sb] == [x["iterations"] for x in ref[1][0][1]]
        gpu_ctx.ndt_align_batch_begin(ba, poses[0])
        gpu_ctx.ndt_align_batch_begin(bb, poses[1])
        pb, sb = gpu_ctx.align_batch_end(bb)  # ended out of order
        pa, sa = gpu_ctx.align_batch_end(ba)
        assert np.array_equal(pa, ref[0][1][0]) and np.array_equal(pb, ref[1][1][0])
        assert [x["iterations"] for x in sb] == [x["iterations"] for x in ref[1][1][1]]
    ba.close()
    bb.close()


@pytest.mark.parametrize("search", ["tree", "grid_exact"])
def test_four_alignments_in_flight_over_three_streams(gpu_ctx, api, small_world, search):
    """Batches are dealt to the context's three compute streams in turn; a fourth shares a stream with the first. Four alignments
    begun back to back (ICP, ragged scans, different poses) and ended in a scrambled order give the blocking calls' poses bit for bit
    — with the reference-default tree search and with the exact grid search, whose per-iteration binning scratch was once the
    context's (round 5: `bench.py --search grid`, three alignments in flight, died of a memory fault; it is the batch's now)."""
    m = small_world["map"]
    s = small_world["scan10k"]
    pose = small_world["init_pose"]
    gpu_ctx.icp_set_target(m)
    opts = api.icp_opts(method=api.P2PLANE)
    if search == "grid_exact":
        opts.search_mode = api.SEARCH_GRID_EXACT
    sets = [[s, s[:7000]], [s[::2], s[:9000]], [s[100:4100], s[::3]], [s[5:8005], s[:2500]]]
    inits = []
    for i in range(4):
        ip = np.stack([pose, pose])
        ip[0, 4] += 0.01 * i
        ip[1, 5] -= 0.015 * i
        inits.append(ip)
    batches = [gpu_ctx.batch(x) for x in sets]
    want = [gpu_ctx.icp_align_batch(b, ip, opts) for b, ip in zip(batches, inits)]
    for order in ((0, 1, 2, 3), (3, 1, 0, 2), (2, 3, 1, 0)):
        for b, ip in zip(batches, inits):
            gpu_ctx.icp_align_batch_begin(b, ip, opts)
        for i in order:
            got, st = gpu_ctx.align_batch_end(batches[i])
            assert np.array_equal(got, want[i][0])
            assert [x["iterations"] for x in st] == [x["iterations"] for x in want[i][1]]
    for b in batches:
        b.close()


def test_upload_error_and_destroy_while_pending(gpu_ctx, api, small_world):
    """the uploader is per context; a refused upload (scan larger than the batch) leaves it usable, a batch destroyed
    while its upload is still being packed is waited for, and many small batches do not pin 64 MB each."""
    m = small_world["map"]
    s = small_world["scan10k"]
    pose = small_world["init_pose"]
    gpu_ctx.icp_set_target(m)
    opts = api.icp_opts(method=api.P2PLANE)
    want, _ = gpu_ctx.icp_align(s, pose, opts)
    b = gpu_ctx.batch_empty(2, 5000)
    with pytest.raises(RuntimeError):
        b.upload_async(api.MarshalledScans([s, s[:100]]))  # 10 000 points into a 5 000-point batch
    ok = api.MarshalledScans([s[:5000], s[:100]])
    b.upload_async(ok)
    b.close()  # destroy while the worker may still be packing
    many = [gpu_ctx.batch([s]) for _ in range(24)]  # 24 x 64 MB of pinned slots would be 1.5 GB
    got, _ = gpu_ctx.icp_align_batch(many[-1], pose[None], opts)
    assert np.array_equal(got[0], want)
    for x in many:
        x.close()


def test_sharded_batches_with_the_owner_solving_ahead_of_the_exchange():
    """With several ranks a scan-sharded batch lets the rank that holds a scan solve it at once and uses the all-reduce — on the
    communication stream — only to replicate the other ranks' scans. One GPU cannot form a two-rank communicator, but
    LOCGPU_SHARD_DECOUPLED=1 forces that path on the one-rank communicator: the sharded parity test above must pass unchanged (a
    batch held completely, one held partially whose foreign scan is solved from the reduced zeros), and two partially held batches
    in flight at once must give the poses of the blocking calls."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LOCGPU_SHARD_DECOUPLED="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_configs.py"), "-q", "-m", "gpu", "-x", "-k",
                          "sharded_batch_over_rccl or decoupled_two_in_flight"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2500:] + out.stderr[-1000:]


def test_decoupled_two_in_flight(api, small_world):
    """Helper of the test above (also valid on the default path): two sharded batches begun back to back."""
    from loc_lib_amd import multi_gpu
    m = small_world["map"]
    scans = [small_world["scan10k"], small_world["scan2k"], small_world["scan10k"][::3], small_world["scan10k"][::2]]
    init = small_world["init_pose"]
    inits = np.stack([init] * 4)
    inits[2, 4:] += [0.03, 0.02, -0.01]
    ctx = api.Context(0)
    try:
        multi_gpu.init_comm(ctx, None)
        ctx.icp_set_target_bcast(m, root=0)
        opts = api.icp_opts(method=api.P2PLANE)
        plain = ctx.batch(scans)
        want, wst = ctx.icp_align_batch(plain, inits, opts)
        a = ctx.batch(scans[1:3], first=1, n_total=4)   # holds scans 1, 2
        b = ctx.batch(scans[0:2], first=0, n_total=4)   # holds scans 0, 1
        ctx.icp_align_batch_begin(a, inits, opts)
        ctx.icp_align_batch_begin(b, inits, opts)
        pa, sa = ctx.align_batch_end(a)
        pb, sb = ctx.align_batch_end(b)
        np.testing.assert_array_equal(pa[1:3], want[1:3])
        np.testing.assert_array_equal(pb[0:2], want[0:2])
        np.testing.assert_array_equal(pa[[0, 3]], inits[[0, 3]])  # nobody holds them in this one-rank world: no residuals
        assert [s["iterations"] for s in sa[1:3]] == [s["iterations"] for s in wst[1:3]] and sa[0]["iterations"] == 20
        plain.close(); a.close(); b.close()
    finally:
        ctx.close()
