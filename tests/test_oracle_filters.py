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
