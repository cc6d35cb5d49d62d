"""Known-answer tests of the oracle's cloud filters (oracle/locref_filters.hpp) — PCL 1.8's VoxelGrid / CropBox /
removeNaNFromPointCloud / transformPointCloud as the reference's VoxelFilter, BoxFilter, RemoveNanPoint and Lio::AddCloud
call them. The reference has no tests or golden vectors for these (parity unpinned); the checks below are closed-form or
independent numpy restatements."""
import numpy as np
import pytest


def _rand_cloud(n, seed, scale=10.0):
    rng = np.random.default_rng(seed)
    c = (rng.normal(size=(n, 4)) * scale).astype(np.float32)
    c[:, 3] = rng.uniform(0, 255, n).astype(np.float32)
    return c


def _numpy_voxel(c, leaf):
    """Independent restatement: float32 index arithmetic of voxel_grid.hpp, float64 centroids."""
    inv = np.float32(1.0) / np.float32(leaf)
    fl = np.floor(c[:, :3] * inv)
    min_b = np.floor(c[:, :3].min(0) * inv).astype(np.int64)
    max_b = np.floor(c[:, :3].max(0) * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (fl - min_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    u, inv_idx, cnt = np.unique(idx, return_inverse=True, return_counts=True)
    s = np.zeros((len(u), 4))
    np.add.at(s, inv_idx, c.astype(np.float64))
    return s / cnt[:, None], cnt, min_b, div


def test_voxel_grid_matches_numpy(locref):
    c = _rand_cloud(50000, 1)
    for leaf in (0.5, 1.3, 4.0):
        out, info = locref.voxel_grid(c, True, leaf, with_info=True)
        ref, cnt, min_b, div = _numpy_voxel(c, leaf)
        assert info["status"] == 0
        assert np.array_equal(info["min_b"], min_b) and np.array_equal(info["div_b"], div)
        assert out.shape == (len(ref), 4)
        # float32 running sums of up to cnt.max() terms of magnitude ≤ ~50 (xyz) / 255 (intensity)
        tol = 4 * cnt.max() * np.finfo(np.float32).eps * np.abs(c).max(0)
        assert np.all(np.abs(out - ref) <= tol)


def test_voxel_grid_order_variants_agree_within_rounding(locref):
    c = _rand_cloud(30000, 2, scale=3.0)
    a = locref.voxel_grid(c, True, 1.0, order=locref.SORT_STD)
    b = locref.voxel_grid(c, True, 1.0, order=locref.SORT_STABLE)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= 64 * np.finfo(np.float32).eps * np.abs(c).max()
    # voxels with one or two points do not depend on the order at all (float addition is commutative)
    _, cnt, _, _ = _numpy_voxel(c, 1.0)
    assert np.array_equal(a[cnt <= 2], b[cnt <= 2])


def test_voxel_grid_single_voxel_and_sequential_sum(locref):
    # all points in one voxel: the centroid is the float32 running sum, in input order, divided once
    c = np.array([[0.1, 0.2, 0.3, 1.0], [0.7, 0.1, 0.9, 2.0], [0.33, 0.9, 0.5, 4.0], [0.25, 0.5, 0.125, 8.0]], np.float32)
    out = locref.voxel_grid(c, True, 1.0, order=locref.SORT_STABLE)
    s = np.zeros(4, np.float32)
    for p in c:
        s = (s + p).astype(np.float32)
    assert out.shape == (1, 4)
    assert np.array_equal(out[0], (s / np.float32(4.0)).astype(np.float32))


def test_voxel_grid_output_order_is_ascending_voxel_index(locref):
    # three voxels along z, two along x: index = ix + iy*div_x + iz*div_x*div_y, so x varies fastest
    c = np.array([[1.5, 0.5, 2.5, 0], [0.5, 0.5, 0.5, 1], [1.5, 0.5, 0.5, 2], [0.5, 0.5, 2.5, 3], [0.5, 0.5, 1.5, 4]], np.float32)
    out = locref.voxel_grid(c, True, 1.0)
    assert [int(v) for v in out[:, 3]] == [1, 2, 4, 3, 0]


def test_voxel_grid_negative_coordinates_use_floor(locref):
    c = np.array([[-0.1, 0, 0, 1], [-0.9, 0, 0, 3], [0.1, 0, 0, 5], [-1.1, 0, 0, 7]], np.float32)
    out, info = locref.voxel_grid(c, True, 1.0, with_info=True)
    assert info["min_b"][0] == -2 and info["div_b"][0] == 3
    assert np.allclose(out[:, 3], [7, 2, 5])  # voxels [-2,-1), [-1,0), [0,1)


def test_voxel_grid_non_dense_skips_nonfinite(locref):
    c = _rand_cloud(2000, 3)
    d = c.copy()
    d[::7, 0] = np.nan
    d[3::11, 2] = np.inf
    keep = np.isfinite(d[:, :3]).all(1)
    a = locref.voxel_grid(d, False, 2.0, order=locref.SORT_STABLE)
    b = locref.voxel_grid(np.ascontiguousarray(d[keep]), True, 2.0, order=locref.SORT_STABLE)
    assert np.array_equal(a, b)
    out, info = locref.voxel_grid(np.full((5, 4), np.nan, np.float32), False, 1.0, with_info=True)
    assert len(out) == 0 and info["status"] == 2


def test_voxel_grid_leaf_too_small_passes_input_through(locref):
    # (extent / leaf)^3 > INT32_MAX → PCL warns and copies the input
    c = _rand_cloud(1000, 4, scale=100.0)
    out, info = locref.voxel_grid(c, True, 0.01, with_info=True)
    assert info["status"] == 1 and np.array_equal(out, c)


def test_crop_box_inclusive_and_nan_rules(locref):
    c = np.array([[1, 1, 1, 0], [2, 0, 0, 1], [-2, 0, 0, 2], [2.0001, 0, 0, 3], [0, 0, np.nan, 4], [np.inf, 0, 0, 5]], np.float32)
    mn, mx = [-2, -2, -2], [2, 2, 2]
    dense = locref.crop_box(c, True, mn, mx)
    # dense flag: NaN fails every comparison → kept; +inf is > max → dropped
    assert [int(v) for v in dense[:, 3]] == [0, 1, 2, 4]
    sparse = locref.crop_box(c, False, mn, mx)
    assert [int(v) for v in sparse[:, 3]] == [0, 1, 2]
    big = _rand_cloud(20000, 5)
    got = locref.crop_box(big, True, [-5, -4, -3], [5, 4, 3])
    m = (big[:, 0] >= -5) & (big[:, 0] <= 5) & (big[:, 1] >= -4) & (big[:, 1] <= 4) & (big[:, 2] >= -3) & (big[:, 2] <= 3)
    assert np.array_equal(got, big[m])


def test_box_edges_are_float32_sums(locref):
    mn, mx = locref.box_edges([150, 150, 150], [1234.567, -0.001, 3.25])
    o = np.array([1234.567, -0.001, 3.25], np.float32)
    assert np.array_equal(mn, (np.float32(-150) + o).astype(np.float32)) and np.array_equal(mx, (np.float32(150) + o).astype(np.float32))


def test_remove_nan_trusts_dense_flag(locref):
    c = _rand_cloud(100, 6)
    c[5, 1] = np.nan
    c[9, 0] = -np.inf
    assert np.array_equal(locref.remove_nan(c, True), c, equal_nan=True)
    out = locref.remove_nan(c, False)
    assert len(out) == 98 and np.isfinite(out[:, :3]).all()
    assert np.array_equal(out, c[np.isfinite(c[:, :3]).all(1)])


def test_transform_f64_matches_numpy_double(locref):
    c = _rand_cloud(5000, 7, scale=50.0)
    q = np.array([0.1, -0.2, 0.3, 0.9])
    q /= np.linalg.norm(q)
    pose = np.concatenate([q, [10.5, -3.25, 0.75]])
    out = locref.transform_cloud_f64(pose, c)
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    ref = (c[:, :3].astype(np.float64) @ R.T + pose[4:]).astype(np.float32)
    assert np.abs(out[:, :3] - ref).max() <= 2 * np.finfo(np.float32).eps * 200
    assert np.array_equal(out[:, 3], c[:, 3])
    d = c.copy()
    d[0, 0] = np.nan
    o2 = locref.transform_cloud_f64(pose, d, is_dense=False)
    assert np.isnan(o2[0, 0]) and np.array_equal(o2[0, 1:], d[0, 1:])  # skipped: stays as copied


def test_local_map_follows_lio_bookkeeping(locref):
    """lio.cpp:283-300: append + re-filter while the queue is short; drop the oldest and rebuild once it overflows."""
    kfs = [_rand_cloud(3000, 10 + i, scale=4.0) + np.array([i, 0, 0, 0], np.float32) for i in range(5)]
    lm = locref.LocalMap(3, 0.8, order=locref.SORT_STABLE)
    cur = np.zeros((0, 4), np.float32)
    for i, kf in enumerate(kfs):
        lm.add_keyframe(kf)
        if i < 3:
            cur = locref.voxel_grid(np.concatenate([cur, kf]), True, 0.8, order=locref.SORT_STABLE)
        else:
            cur = locref.voxel_grid(np.concatenate(kfs[i - 2:i + 1]), True, 0.8, order=locref.SORT_STABLE)
        assert np.array_equal(lm.cloud(), cur)
        assert lm.is_dense
    # filtering an already filtered map is not idempotent in general (centroids move), but the voxel count cannot grow
    again = locref.voxel_grid(cur, True, 0.8)
    assert len(again) <= len(cur)


@pytest.mark.parametrize("n", [1, 2, 17])
def test_tiny_clouds(locref, n):
    c = np.abs(_rand_cloud(n, 20 + n))  # one octant: a single 1000 m voxel holds everything
    out = locref.voxel_grid(c, True, 1000.0)
    assert len(out) == 1
    assert np.allclose(out[0], c.astype(np.float64).mean(0), rtol=1e-5, atol=1e-4)
    assert len(locref.voxel_grid(np.zeros((0, 4), np.float32), True, 1.0)) == 0


# ---- LOAM feature picker (oracle/locref_loam.hpp) -------------------------------------------------------------------------
def _py_loam(cloud, ring, num_scan):
    """Independent, literal Python restatement of loam_feature_extract.cpp:19-151 (float32 sums via numpy scalars)."""
    f32 = np.float32
    edge, surf = [], []
    for r in range(num_scan):
        L = cloud[ring == r]
        if len(L) < 131:
            continue
        curv = []
        for j in range(5, len(L) - 5):
            d = []
            for a in range(3):
                s = f32(L[j - 5, a])
                for k in (-4, -3, -2, -1):
                    s = f32(s + L[j + k, a])
                s = f32(s - f32(f32(10) * L[j, a]))
                for k in (1, 2, 3, 4, 5):
                    s = f32(s + L[j + k, a])
                d.append(float(s))
            curv.append((j, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))
        total = len(L) - 10
        for sec in range(6):
            ln = total // 6
            start, end = ln * sec, (ln * (sec + 1) - 1 if sec < 5 else total - 1)
            sub = sorted(curv[start:end], key=lambda t: (t[1], t[0]))
            picked, n_pick = [], 0
            for i in range(len(sub) - 1, -1, -1):
                ind = sub[i][0]
                if ind in picked:
                    continue
                if sub[i][1] <= 0.1:
                    break
                n_pick += 1
                picked.append(ind)
                if n_pick <= 20:
                    edge.append(L[ind])
                else:
                    break
                for sgn in (1, -1):
                    for k in range(1, 6):
                        a, b = L[ind + sgn * k], L[ind + sgn * (k - 1)]
                        dd = [float(f32(a[c] - b[c])) for c in range(3)]
                        if dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2] > 0.05:
                            break
                        picked.append(ind + sgn * k)
            for ind, _ in sub:
                if ind not in picked:
                    surf.append(L[ind])
    as_arr = lambda x: np.array(x, np.float32).reshape(-1, 4)
    return as_arr(edge), as_arr(surf)


def _ring_cloud(synth, scan_id, rings, per_ring=1800):
    s = synth.make_scan(scan_id)
    idx = np.concatenate([np.arange(r * 1800, r * 1800 + per_ring) for r in rings])
    c = np.zeros((len(idx), 4), np.float32)
    c[:, :3] = s[idx, :3]
    c[:, 3] = (idx % 256).astype(np.float32)
    ring = np.repeat(np.arange(len(rings)), per_ring).astype(np.uint8)
    return c, ring


def test_loam_extract_matches_python_restatement(locref, synth):
    c, ring = _ring_cloud(synth, 2, [3, 40, 60], per_ring=400)
    # interleave the rings: bucketing must keep input order inside each ring
    perm = np.argsort(np.arange(len(c)) % 400, kind="stable")
    c, ring = c[perm], ring[perm]
    e_ref, s_ref = _py_loam(c, ring, 3)
    for order in (locref.SORT_STD, locref.SORT_STABLE):
        e, s = locref.loam_extract(c, ring, 3, order=order)
        assert np.array_equal(e, e_ref) and np.array_equal(s, s_ref)
    assert len(e_ref) > 0 and len(s_ref) > 0


def test_loam_extract_quirks(locref, synth):
    c, ring = _ring_cloud(synth, 4, [10], per_ring=1800)
    e, s = locref.loam_extract(c, ring, 16)
    # at most 20 edges per sector, 6 sectors; each sector drops its last element; the rest is edge, surface or marked
    assert len(e) <= 120
    assert len(e) + len(s) <= 1790 - 6
    # rings shorter than 131 points give nothing; points of rings >= num_scan are ignored
    short, rs = _ring_cloud(synth, 4, [10], per_ring=130)
    e2, s2 = locref.loam_extract(short, rs, 16)
    assert len(e2) == 0 and len(s2) == 0
    e3, s3 = locref.loam_extract(c, ring + 20, 16)
    assert len(e3) == 0 and len(s3) == 0
    # a flat, evenly sampled ring has no edges: every point of every sector (minus its last) is a surface point
    th = np.linspace(0, 2 * np.pi, 600, endpoint=False)
    flat = np.stack([10 * np.cos(th), 10 * np.sin(th), np.zeros_like(th), np.arange(600)], 1).astype(np.float32)
    e4, s4 = locref.loam_extract(flat, np.zeros(600, np.uint8), 1)
    assert len(e4) == 0 and len(s4) == 590 - 6
