#!/usr/bin/env python3
"""Single-scan latency (BASELINE config 5 shape): one 115200-pt scan per call vs the 10M-pt map, eager vs hipGraph mode."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loc_lib_amd import api, synth  # noqa: E402

ctx = api.Context(0)
m = synth.make_map(int(os.environ.get("MAP_POINTS", 10_000_000)))
ctx.icp_set_target(m)
ctx.ndt_set_target(m)
out = {}
only = os.environ.get("LAT_ONLY")  # e.g. p2plane_eager: one leg only (for a kernel trace)
sids = [int(x) for x in os.environ.get("LAT_SCANS", "0,1,2,3,4,5,6,7,8,9,10,11").split(",")]  # which synthetic scans (a trace of one of them)
for name in ("p2plane", "ndt"):
    for graph in (False, True):
        if only and only != "%s_%s" % (name, "graph" if graph else "eager"):
            continue
        ctx.graph_enable(graph)
        ts, its = [], []
        for sid in sids:
            scan = synth.make_scan(sid)
            _, init = synth.make_pose(sid)
            b = ctx.batch([scan])
            opts = api.icp_opts(method=api.P2PLANE)
            fn = (lambda: ctx.icp_align_batch(b, init, opts)) if name == "p2plane" else (lambda: ctx.ndt_align_batch(b, init))
            fn()  # warm (captures the graph in graph mode)
            t0 = time.perf_counter()
            for _ in range(5):
                p, st = fn()
            ts.append((time.perf_counter() - t0) / 5)
            its.append(st[0]["iterations"])
            b.close()
        out["%s_%s" % (name, "graph" if graph else "eager")] = dict(ms_per_scan=round(1e3 * float(np.mean(ts)), 4),
                                                                     scans_per_s=round(1.0 / float(np.mean(ts)), 1), mean_iters=float(np.mean(its)))
# host-pointer path, what the façade's ScanMatch calls per scan: pack + H2D + align + D2H of the pose
for graph in (False, True):
    if only:
        break
    ctx.graph_enable(graph)
    opts = api.icp_opts(method=api.P2PLANE)
    ts = []
    for sid in sids:
        scan = synth.make_scan(sid)
        _, init = synth.make_pose(sid)
        ctx.icp_align(scan, init, opts)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.icp_align(scan, init, opts)
        ts.append((time.perf_counter() - t0) / 5)
    out["p2plane_hostptr_%s" % ("graph" if graph else "eager")] = dict(ms_per_scan=round(1e3 * float(np.mean(ts)), 4), scans_per_s=round(1.0 / float(np.mean(ts)), 1))
ctx.graph_enable(False)
print(json.dumps(out))
