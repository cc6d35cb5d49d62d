#!/usr/bin/env python3
"""Replay one dumped fuzz_search case (tools/fuzz_search.py --only N --dump file.npz) and print the queries whose lists differ from the oracle's."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loc_lib_amd import api  # noqa: E402
from oracle import locref  # noqa: E402

d = np.load(sys.argv[1])
cloud, scan, pose = d["cloud"], d["scan"], d["pose"]
ctx = api.Context(0)
ctx.icp_set_target(cloud)
tree = locref.KdTree(cloud)
b = ctx.batch([scan])
nq = len(scan)
qq = locref.transform_points(pose, np.ascontiguousarray(scan[:, :3], dtype=np.float64)).astype(np.float32)
print("leaves", tree.num_leaves, "depth", tree.depth, "nq", nq, "env", {k: v for k, v in os.environ.items() if k.startswith("LOCGPU_")})
for method, k in ((api.P2PLANE, 5), (api.P2P, 1)):
    for approximate in (1, 0):
        opts = api.icp_opts(method=method)
        opts.approximate = approximate
        ctx.icp_hb_batch(b, pose[None], opts)
        got = ctx.debug_batch_nn(b, k)[0, :nq]
        want = tree.knn(qq, k, approximate=bool(approximate), alpha=0.1)
        bad = np.nonzero(np.any(got != want, axis=1))[0]
        print("k", k, "approx", approximate, "differing lists:", len(bad))
        for i in bad[:6]:
            dg = np.linalg.norm(cloud[got[i]].astype(np.float64) - qq[i], axis=1) if np.all(got[i] < len(cloud)) else None
            dw = np.linalg.norm(cloud[want[i]].astype(np.float64) - qq[i], axis=1)
            print("  q", i, qq[i], "\n    got ", got[i], dg, "\n    want", want[i], dw)
