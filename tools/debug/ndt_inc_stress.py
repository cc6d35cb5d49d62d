#!/usr/bin/env python3
"""Stress for a rare mismatch seen once in tools/fuzz_ndt.py (incremental NDT, capacity 3000, after direct-NDT cases in the same process):
alternate a direct-NDT context and an incremental one many times and compare every incremental alignment with the oracle's (computed once)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loc_lib_amd import api, synth  # noqa: E402
from oracle import locref  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sid = 17
m = synth.make_local_map(120000, sid, half=40.0)
m2 = synth.make_local_map(60000, sid + 1, half=40.0)
scan = synth.make_scan(sid, subsample=9000, crop_half=36.0)
_, init = synth.make_pose(sid, trans_amp=0.4, rot_amp_deg=3.0, seed=5)
kw = dict(voxel_size=2.1311215983306657, nearby_type=1, min_pts_in_voxel=3, res_outlier_th=100.0, min_effective_pts=200, max_iteration=30)
cap = 3000
ref = locref.Ndt(method=api.INCREMENTAL_NDT, capacity=cap, **kw)
ref.set_target(m)
ref.set_target(m2)
want = ref.align(scan, init)
print("oracle: iterations", want["iters"], "status", want["status"], "voxels", ref.num_voxels())
bad = 0
for r in range(reps):
    c1 = api.Context(0)
    c1.ndt_set_target(m, api.ndt_opts(voxel_size=0.5 + 0.01 * (r % 50), min_pts_in_voxel=3))
    c1.ndt_align(scan, init)
    c1.icp_set_target(m)
    c1.icp_align(scan, init, api.icp_opts(method=api.P2PLANE))
    del c1
    c2 = api.Context(0)
    c2.ndt_set_target(m, api.ndt_opts(method=api.INCREMENTAL_NDT, capacity=cap, **kw))
    c2.ndt_set_target(m2, api.ndt_opts(method=api.INCREMENTAL_NDT, capacity=cap, **kw))
    got, st = c2.ndt_align(scan, init)
    if st["iterations"] != want["iters"] or st["status"] != want["status"] or np.abs(got - want["pose"]).max() > 1e-9:
        bad += 1
        print("rep", r, "MISMATCH", st, np.abs(got - want["pose"]).max(), flush=True)
    del c2
print("reps", reps, "mismatches", bad)
