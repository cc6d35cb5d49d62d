"""tools/latency_detail.py — where a single ScanMatch's time goes, scan by scan: Gauss–Newton iterations until convergence and wall time
per iteration for twelve synthetic scans against the 10 M-point map (reference-default point-to-plane ICP, one scan per call — BASELINE
configs[1]). Round 5: 3–14 iterations, 57–111 µs per iteration — the iteration's cost follows the scan's LONGEST traversal, not a
fixed launch cost (profiles/experiments.md).        gpurun -- 'python tools/latency_detail.py'"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from loc_lib_amd import api, synth
ctx = api.Context(0)
m = synth.make_map(10_000_000)
ctx.icp_set_target(m)
opts = api.icp_opts(method=api.P2PLANE)
for sid in range(12):
    scan = synth.make_scan(sid); _, init = synth.make_pose(sid)
    b = ctx.batch([scan])
    ctx.icp_align_batch(b, init, opts)
    t0 = time.perf_counter()
    for _ in range(20):
        p, st = ctx.icp_align_batch(b, init, opts)
    dt = (time.perf_counter() - t0) / 20
    print(sid, "iters", st[0]["iterations"], "ms %.3f" % (1e3 * dt), "us/iter %.1f" % (1e6 * dt / st[0]["iterations"]))
    b.close()
