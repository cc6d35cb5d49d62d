#!/usr/bin/env python3
"""What the sharded mode's machinery costs with more than one rank, as far as ONE GPU can tell: two ranks as two threads (own contexts,
loopback communicator — tests/cpp/loopback_rccl.hip — instead of RCCL over xGMI) each hold half of N scans and run bench.py's step loop
(three alignments in flight, owner solving ahead of the exchange); compared with ONE context holding all N scans as a plain batch. Both
ranks share the one GPU, so the ratio says how much the exchange path serialises or adds — not how the job scales.

    LOCGPU_RCCL_LIB=tests/cpp/libloopback_rccl.so python tools/world2_loopback_bench.py [--scans 64] [--steps 30]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loc_lib_amd import api, multi_gpu, synth  # noqa: E402


def run_steps(ctx, bufs, inits, opts, n, depth, barrier=None):
    inflight, begun, g = [], 0, 0
    if barrier:
        barrier.wait()
    t0 = time.perf_counter()
    while begun < n or inflight:
        while begun < n and len(inflight) < depth:
            b = bufs[g % len(bufs)]
            g += 1
            ctx.icp_align_batch_begin(b, inits, opts)
            inflight.append(b)
            begun += 1
        res = ctx.align_batch_end(inflight.pop(0))
    return time.perf_counter() - t0, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=64)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--map-points", type=int, default=10_000_000)
    a = ap.parse_args()
    assert os.environ.get("LOCGPU_RCCL_LIB"), "set LOCGPU_RCCL_LIB to the loopback communicator"
    m = synth.make_map(a.map_points)
    scans = [synth.make_scan(i) for i in range(a.scans)]
    inits = np.stack([synth.make_pose(i)[1] for i in range(a.scans)])
    opts = api.icp_opts(method=api.P2PLANE)
    depth = 3
    # ---- one context, plain batch
    ctx = api.Context(0)
    ctx.icp_set_target(m)
    bufs = [ctx.batch(scans) for _ in range(depth)]
    run_steps(ctx, bufs, inits, opts, 4, depth)
    t_plain, (want, _) = run_steps(ctx, bufs, inits, opts, a.steps, depth)
    for b in bufs:
        b.close()
    ctx.close()
    # ---- two ranks, two threads
    uid = api.comm_unique_id()
    barrier = threading.Barrier(2)
    out = {}

    def rank_main(rank):
        c = api.Context(0)
        c.comm_init(rank, 2, uid)
        c.icp_set_target_bcast(m if rank == 0 else None, root=0)
        lo, hi = multi_gpu.shard_range(a.scans, rank, 2)
        bs = [c.batch(scans[lo:hi], first=lo, n_total=a.scans) for _ in range(depth)]
        run_steps(c, bs, inits, opts, 4, depth, barrier)
        t, (poses, _) = run_steps(c, bs, inits, opts, a.steps, depth, barrier)
        out[rank] = (t, poses)
        for b in bs:
            b.close()
        c.close()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    t_two = max(out[0][0], out[1][0])
    same = bool(np.array_equal(out[0][1], want) and np.array_equal(out[1][1], want))
    print(json.dumps(dict(scans=a.scans, steps=a.steps, alignments_in_flight=depth, one_context_plain_scans_s=a.scans * a.steps / t_plain,
                          two_ranks_loopback_scans_s=a.scans * a.steps / t_two, ratio=t_plain / t_two, poses_bit_identical_on_both_ranks=same,
                          note="both ranks share ONE GPU; the communicator is the test double, not RCCL")))


if __name__ == "__main__":
    main()
