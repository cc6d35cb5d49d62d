"""Generates tests/golden/filters_small.npz with the CPU oracle (oracle/locref_filters.hpp) in the build container.

The reference holds no golden vectors for its filter wrappers and PCL cannot be built here, so these vectors pin the
ORACLE's behaviour (regression) and let the GPU box check the HIP filters without regenerating anything: a 20 k-pt scan
with NaN returns and intensities, its removeNaN / VoxelGrid (both summation orders) / CropBox outputs, and the local map
after each of 5 keyframes with num_kfs = 3 (Lio::AddCloud's keyframe branch), and a 6-ring scan with its LOAM edge / surface
features (LoamFeatureExtract::Extract).

    python tests/golden/make_golden_filters.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loc_lib_amd import synth  # noqa: E402
from oracle import locref  # noqa: E402


def xyzi(a, seed):
    out = np.zeros((len(a), 4), np.float32)
    out[:, :3] = a[:, :3]
    out[:, 3] = np.random.RandomState(seed).randint(0, 256, len(a)).astype(np.float32)
    return out


def main():
    scan = xyzi(synth.make_scan(11, subsample=20000), 1)
    scan[::41, 0] = np.nan
    scan[7::97, 2] = np.inf
    out = dict(scan=scan)
    out["no_nan"] = locref.remove_nan(scan, False)
    out["voxel_std"] = locref.voxel_grid(out["no_nan"], True, 0.4, order=locref.SORT_STD)
    out["voxel_stable"] = locref.voxel_grid(out["no_nan"], True, 0.4, order=locref.SORT_STABLE)
    out["voxel_nondense_stable"] = locref.voxel_grid(scan, False, 0.4, order=locref.SORT_STABLE)
    mn, mx = locref.box_edges([15, 12, 2], [2.5, -1.0, 0.25])
    out["box_min"], out["box_max"] = mn, mx
    out["crop_dense"] = locref.crop_box(scan, True, mn, mx)
    out["crop_nondense"] = locref.crop_box(scan, False, mn, mx)
    lm = locref.LocalMap(3, 0.6, order=locref.SORT_STABLE)
    for s in range(5):
        kf_scan = xyzi(synth.make_scan(20 + 3 * s, subsample=5000), 10 + s)
        pose, _ = synth.make_pose(20 + 3 * s)
        out["kf_scan_%d" % s], out["kf_pose_%d" % s] = kf_scan, pose
        kf = locref.transform_cloud_f64(pose, kf_scan)
        if s == 0:
            out["kf_world_0"] = kf
        lm.add_keyframe(kf)
        out["local_map_%d" % s] = lm.cloud()
    # LOAM feature picker: 6 rings x 900 azimuths of scan 31, delivered column by column (rings interleaved)
    full = synth.make_scan(31)
    idx = np.concatenate([np.arange(r * 1800, r * 1800 + 900) for r in (2, 12, 25, 38, 50, 61)])
    lo = xyzi(full[idx], 99)
    ring = np.repeat(np.arange(6), 900).astype(np.uint8)
    perm = np.argsort(np.arange(len(lo)) % 900, kind="stable")
    lo, ring = np.ascontiguousarray(lo[perm]), np.ascontiguousarray(ring[perm])
    out["loam_cloud"], out["loam_ring"] = lo, ring
    out["loam_edge"], out["loam_surf"] = locref.loam_extract(lo, ring, 16, order=locref.SORT_STABLE)
    e_std, s_std = locref.loam_extract(lo, ring, 16, order=locref.SORT_STD)
    assert np.array_equal(e_std, out["loam_edge"]) and np.array_equal(s_std, out["loam_surf"])  # no equal curvatures in this input
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "filters_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: len(v) for k, v in out.items() if k.startswith(("voxel", "crop", "local", "no_nan"))})


if __name__ == "__main__":
    main()
