"""GPU parity of the LOAM feature picker (loam_feature_extract.cpp:19-151) against
oracle/locref_loam.hpp through the C ABI (locgpu_cloud_loam_extract, locgpu_loam_extract). Bar: the same points in the same
order, bit for bit (the oracle ordering equal curvatures by ring position, which is the one choice std::sort leaves open)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FULL_POINT = np.dtype({"names": ["x", "y", "z", "w", "range", "radius", "intensity", "ring", "angle", "time_span", "time_intervel", "height"],
                       "formats": ["<f4", "<f4", "<f4", "<f4", "<f4", "<f4", "u1", "u1", "u1", "<f8", "<f8", "<f4"],
                       "offsets": [0, 4, 8, 12, 16, 20, 24, 25, 26, 32, 40, 48], "itemsize": 64})  # LocUtils::FullPointType, point_types.h:65-78


def _scan(synth, scan_id):
    s = synth.make_scan(scan_id)
    c = np.zeros((len(s), 4), np.float32)
    c[:, :3] = s[:, :3]
    c[:, 3] = (np.arange(len(s)) % 256).astype(np.float32)
    ring = (np.arange(len(s)) // 1800).astype(np.uint8)
    return c, ring


@pytest.mark.parametrize("scan_id", [0, 17])
def test_loam_extract_full_scan(gpu_ctx, locref, synth, api, scan_id):
    c, ring = _scan(synth, scan_id)
    e_ref, s_ref = locref.loam_extract(c, ring, 64, order=locref.SORT_STABLE)
    edge, surf = api.Cloud(gpu_ctx, c).loam_extract(ring, 64)
    assert np.array_equal(edge.download(), e_ref) and np.array_equal(surf.download(), s_ref)
    assert 0 < len(e_ref) <= 64 * 6 * 20 and len(s_ref) > 10000
    # this scan has no equal curvatures inside a sector: the reference's own (unstable) order gives the same clouds
    e_std, s_std = locref.loam_extract(c, ring, 64, order=locref.SORT_STD)
    assert np.array_equal(e_std, e_ref) and np.array_equal(s_std, s_ref)


def test_loam_extract_interleaved_input_and_partial_rings(gpu_ctx, locref, synth, api):
    c, ring = _scan(synth, 9)
    # column-major delivery (all rings of azimuth 0, then azimuth 1, …) with some returns missing and two rings nearly empty
    perm = np.argsort(np.arange(len(c)) % 1800, kind="stable")
    c, ring = c[perm], ring[perm]
    keep = np.ones(len(c), bool)
    keep[::13] = False
    keep[(ring == 5) & (np.arange(len(c)) % 16 != 0)] = False   # ring 5 keeps 1 in 16 → < 131 points → skipped
    keep[ring == 6] = False
    c, ring = np.ascontiguousarray(c[keep]), np.ascontiguousarray(ring[keep])
    e_ref, s_ref = locref.loam_extract(c, ring, 64, order=locref.SORT_STABLE)
    edge, surf = api.Cloud(gpu_ctx, c).loam_extract(ring, 64)
    assert np.array_equal(edge.download(), e_ref) and np.array_equal(surf.download(), s_ref)
    # num_scan smaller than the rings present: the other rings are ignored
    e16, s16 = locref.loam_extract(c, ring, 16, order=locref.SORT_STABLE)
    edge, surf = api.Cloud(gpu_ctx, c).loam_extract(ring, 16)
    assert np.array_equal(edge.download(), e16) and np.array_equal(surf.download(), s16)
    assert len(e16) < len(e_ref)


def test_loam_extract_ties_and_flat_rings(gpu_ctx, locref, api):
    # exact ties: a regular polygon ring repeated — equal curvatures everywhere; ties are ordered by ring position on both sides
    th = np.linspace(0, 2 * np.pi, 720, endpoint=False)
    sq = np.stack([np.clip(8 * np.cos(th), -5, 5), np.clip(8 * np.sin(th), -5, 5), np.zeros_like(th), np.arange(720) % 200], 1).astype(np.float32)
    c = np.concatenate([sq, sq + np.array([0, 0, 1, 0], np.float32)])
    ring = np.repeat(np.arange(2), 720).astype(np.uint8)
    e_ref, s_ref = locref.loam_extract(c, ring, 2, order=locref.SORT_STABLE)
    edge, surf = api.Cloud(gpu_ctx, c).loam_extract(ring, 2)
    assert np.array_equal(edge.download(), e_ref) and np.array_equal(surf.download(), s_ref)
    assert len(e_ref) > 0
    flat = np.stack([10 * np.cos(th), 10 * np.sin(th), np.zeros_like(th), np.arange(720)], 1).astype(np.float32)
    edge, surf = api.Cloud(gpu_ctx, flat).loam_extract(np.zeros(720, np.uint8), 1)
    assert len(edge) == 0 and len(surf) == 710 - 6


def test_loam_extract_one_shot_full_point_records(gpu_ctx, locref, synth):
    c, ring = _scan(synth, 3)
    rec = np.zeros(len(c), FULL_POINT)
    rec["x"], rec["y"], rec["z"], rec["w"] = c[:, 0], c[:, 1], c[:, 2], 1.0
    rec["intensity"] = c[:, 3].astype(np.uint8)
    rec["ring"] = ring
    rec["time_span"] = 0.1
    edge, surf = gpu_ctx.loam_extract_full(rec, 64)
    e_ref, s_ref = locref.loam_extract(c, ring, 64, order=locref.SORT_STABLE)
    assert np.array_equal(edge, e_ref) and np.array_equal(surf, s_ref)


def test_loam_extract_edge_cases(gpu_ctx, api):
    empty = api.Cloud(gpu_ctx)
    edge, surf = empty.loam_extract(np.zeros(0, np.uint8), 16)
    assert len(edge) == 0 and len(surf) == 0
    small = api.Cloud(gpu_ctx, np.random.default_rng(0).normal(size=(100, 4)).astype(np.float32))
    edge, surf = small.loam_extract(np.zeros(100, np.uint8), 16)   # < 131 points in the ring
    assert len(edge) == 0 and len(surf) == 0
    with pytest.raises(api.LocGpuError):
        small.loam_extract(np.zeros(100, np.uint8), 0)
    with pytest.raises(api.LocGpuError):
        small.loam_extract(np.zeros(100, np.uint8), 300)
    # a ring longer than 6 x 2048 points is refused, loudly
    long_ring = api.Cloud(gpu_ctx, np.random.default_rng(1).normal(size=(13000, 4)).astype(np.float32))
    with pytest.raises(api.LocGpuError):
        long_ring.loam_extract(np.zeros(13000, np.uint8), 1)


def test_cpp_facade_loam_feature_extract(locref, synth, tmp_path):
    """LocUtils::LoamFeatureExtract (loc_lib_amd/host) driven like Lio::AddCloud(FullCloudPtr) (lio.cpp:321-323)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "cpp", "facade_filters")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    c, ring = _scan(synth, 12)
    c.tofile(tmp_path / "in.bin")
    ring.tofile(tmp_path / "ring.bin")
    r = subprocess.run([exe, "loam", str(tmp_path / "in.bin"), str(tmp_path / "ring.bin"), "64", str(tmp_path / "out")], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    edge = np.fromfile(str(tmp_path / "out.edge.bin"), np.float32).reshape(-1, 4)
    surf = np.fromfile(str(tmp_path / "out.surf.bin"), np.float32).reshape(-1, 4)
    e_ref, s_ref = locref.loam_extract(c, ring, 64, order=locref.SORT_STABLE)
    assert [int(t) for t in r.stdout.split()] == [len(e_ref), len(s_ref)]
    assert np.array_equal(edge, e_ref) and np.array_equal(surf, s_ref)
