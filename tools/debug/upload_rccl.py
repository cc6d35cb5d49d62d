#!/usr/bin/env python3
"""Does an RCCL communicator on the context slow the batch uploader down? (strong + streaming bench: every second upload's worker ran 12 ms late)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loc_lib_amd import api, multi_gpu, synth

mode = sys.argv[1]  # plain | comm | comm_sharded
ctx = api.Context(0)
if mode != "plain":
    multi_gpu.init_comm(ctx, None)
ctx.icp_set_target(synth.make_map(1_000_000))
scans = [synth.make_scan(i % 8) for i in range(256)]
inits = np.stack([synth.make_pose(i % 8)[1] for i in range(256)])
mk = (lambda: ctx.batch(scans, first=0, n_total=256)) if mode == "comm_sharded" else (lambda: ctx.batch(scans))
nb = int(os.environ.get('NB', 2))
bufs = [mk() for _ in range(nb)]
sc = api.MarshalledScans(scans)
opts = api.icp_opts(method=api.P2PLANE)
for g in range(12):
    t0 = time.perf_counter()
    bufs[(g + 1) % nb].upload_async(sc)
    t1 = time.perf_counter()
    ctx.icp_align_batch(bufs[g % nb], inits, opts)
    t2 = time.perf_counter()
    print(mode, "step", g, "upload_async %.2f ms, align %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
