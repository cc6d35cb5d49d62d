"""Generates tests/golden/icp_small.npz with the CPU oracle (oracle/locref.cpp) in the build container.

The reference holds no golden vectors for this path and cannot be built here, so these vectors
pin the ORACLE's behaviour (regression) and let the GPU box check the HIP path without regenerating anything:
inputs (25 k-pt local map, 2 k-pt scan, 2 k queries) and expected outputs (k-NN index lists in the reference's
approximate and exact modes, first-iteration H/B, final poses and iteration counts of P2P / P2Line / P2Plane / NDT).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loc_lib_amd import synth  # noqa: E402
from oracle import locref  # noqa: E402


def main():
    m = synth.make_local_map(25000, 9, half=25.0)
    scan = synth.make_scan(9, subsample=2000, crop_half=22.0)
    true_pose, init_pose = synth.make_pose(9, trans_amp=0.15, rot_amp_deg=1.0)
    rng = np.random.RandomState(20240901)
    queries = (m[rng.choice(len(m), 2000, replace=False)] + rng.randn(2000, 3).astype(np.float32) * 0.05).astype(np.float32)
    out = dict(map=m, scan=scan, queries=queries, true_pose=true_pose, init_pose=init_pose)
    tree = locref.KdTree(m)
    ann, st_ann = tree.knn(queries, k=5, approximate=True, alpha=0.1, with_stats=True)
    exact, st_exact = tree.knn(queries, k=5, approximate=False, with_stats=True)
    out.update(knn_ann=ann, knn_exact=exact, visits_ann=st_ann, visits_exact=st_exact,
               tree_info=np.array([tree.num_leaves, tree.num_nodes, tree.depth]))
    for method, name in ((locref.P2P, "p2p"), (locref.P2LINE, "p2line"), (locref.P2PLANE, "p2plane")):
        icp = locref.Icp(method=method)
        icp.set_target(m)
        ok, H, B, eff = icp.hb(scan, init_pose)
        r = icp.align(scan, init_pose)
        out["H0_" + name], out["B0_" + name], out["eff0_" + name], out["ok0_" + name] = H, B, eff, ok
        out["pose_" + name], out["iters_" + name], out["trace_" + name] = r["pose"], r["iters"], r["trace"]
    ndt = locref.Ndt()
    ndt.set_target(m)
    r = ndt.align(scan, init_pose)
    out["pose_ndt"], out["iters_ndt"], out["status_ndt"], out["trace_ndt"] = r["pose"], r["iters"], r["status"], r["trace"]
    out["ndt_num_voxels"] = ndt.num_voxels()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "icp_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;",
          {k: (int(out["iters_" + k]), np.round(out["pose_" + k][4:], 4).tolist()) for k in ("p2p", "p2line", "p2plane", "ndt")})


if __name__ == "__main__":
    main()
