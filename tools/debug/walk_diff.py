#!/usr/bin/env python3
"""Debug aid: compare the hot search kernel's lists with the oracle's on a small case and classify the differences."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loc_lib_amd import api, synth
from oracle import locref

n_map = int(os.environ.get("MAP", 200000))
ctx = api.Context(0)
m = synth.make_map(n_map)
ctx.icp_set_target(m)
print("target", ctx.icp_target_info())
tree = locref.KdTree(m)
if os.environ.get("ONE"):
    scans = [synth.make_scan(5, subsample=10000)]
    poses = [synth.make_pose(5)[1]]
else:
    scans = [synth.make_scan(0), synth.make_scan(5, subsample=30000)]
    poses = [synth.make_pose(s)[1] for s in (0, 5)]
b = ctx.batch(scans)
opts = api.icp_opts(method=api.P2PLANE)
ctx.search_stats_read(reset=True)
ctx.icp_hb_batch(b, np.stack(poses), opts)
print("stats", ctx.search_stats_read(reset=True))
got = ctx.debug_batch_nn(b, 5)
for i, (scan, pose) in enumerate(zip(scans, poses)):
    q = locref.transform_points(pose, np.ascontiguousarray(scan[:, :3], dtype=np.float64)).astype(np.float32)
    want = tree.knn(q, 5, approximate=True, alpha=0.1)
    g = got[i, :len(scan)]
    bad = np.any(g != want, axis=1)
    print("scan", i, "queries", len(scan), "mismatching", int(bad.sum()), "with -1:", int(np.any(g < 0, axis=1).sum()))
    idx = np.nonzero(bad)[0]
    for j in idx[:12]:
        dg = np.sum((m[np.maximum(g[j], 0)] - q[j]) ** 2, axis=1)
        dw = np.sum((m[want[j]] - q[j]) ** 2, axis=1)
        print("  q", j, "lane", j % 64, "got", g[j], np.round(dg, 4), "want", want[j], np.round(dw, 4))
    if len(idx):
        print("  mismatch lanes histogram (mod 64):", np.bincount(idx % 64, minlength=64))
