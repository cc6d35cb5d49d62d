"""Helper of tests/test_gpu_configs.py::test_two_matchers_on_one_gpu_concurrently: two threads, each with its own context, map and scans,
run SetInputTarget + single-scan and batch alignments at the same time (the process-wide pieces — host build pool, ingest scratch,
library statics — are shared); every result must equal the one the same calls give alone. Prints one JSON line."""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from loc_lib_amd import api, synth  # noqa: E402


def work(seed, reps, out):
    res = []
    for r in range(reps):
        sid = seed * 100 + r
        m = synth.make_local_map(40000 + 7000 * (r % 5), sid, half=40.0)
        scan = synth.make_scan(sid, crop_half=36.0)[:: 3 + (r % 4)].copy()
        _, init = synth.make_pose(sid)
        ctx = api.Context(0)
        ctx.icp_set_target(m)
        p1, s1 = ctx.icp_align(scan, init, api.icp_opts(method=api.P2PLANE))
        b = ctx.batch([scan, scan[::2].copy(), scan[1::3].copy()])
        pb, sb = ctx.icp_align_batch(b, np.stack([init] * 3), api.icp_opts(method=api.P2PLANE))
        ctx.ndt_set_target(m)
        p2, s2 = ctx.ndt_align(scan, init)
        b.close()
        ctx.close()
        res.append((p1.tobytes(), s1["iterations"], pb.tobytes(), tuple(s["iterations"] for s in sb), np.asarray(p2).tobytes(), s2["iterations"]))
    out.append(res)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    alone = {}
    for seed in (1, 2):
        o = []
        work(seed, reps, o)
        alone[seed] = o[0]
    outs = {1: [], 2: []}
    ts = [threading.Thread(target=work, args=(seed, reps, outs[seed])) for seed in (1, 2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    bad = sum(1 for seed in (1, 2) for a, c in zip(alone[seed], outs[seed][0]) if a != c)
    print(json.dumps(dict(reps=reps, differing=bad)))


if __name__ == "__main__":
    main()
