"""Helper of tests/test_gpu_pool.py::test_sharded_pool_over_rccl_one_rank: a scan pool on a context that has a one-rank RCCL
communicator (so that every pooled iteration runs sum_partials → ncclAllReduce → solve) against plain batches of the same jobs.
LOCGPU_SHARD_DECOUPLED (read once per process by the library) selects whether the exchange runs on the pool's stream (0) or, as with
several ranks, behind the owner's solve on the communication stream (1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from loc_lib_amd import api, multi_gpu, synth  # noqa: E402


def main():
    m = synth.make_local_map(200000, 3, half=40.0)
    s = synth.make_scan(3, subsample=10000, crop_half=36.0)
    truth, pose = synth.make_pose(3)
    rng = np.random.default_rng(17)
    jobs = []
    for j in range(6):
        scans, inits = [], []
        for i in range(3):
            scans.append(np.ascontiguousarray(s[int(rng.integers(0, 2000))::int(rng.integers(1, 3))][: int(rng.integers(2000, 5000))]))
            ip = np.array(truth if (i + j) % 4 == 0 else pose)
            ip[4:] += rng.uniform(-0.2, 0.2, 3)
            inits.append(ip)
        jobs.append((scans, np.stack(inits)))
    ctx = api.Context(0)
    assert multi_gpu.init_comm(ctx, None) == (0, 1)
    ctx.icp_set_target_bcast(m, root=0)
    ctx.ndt_set_target(m)
    for kind in ("p2plane", "ndt"):
        opts = api.icp_opts(method=api.P2PLANE)
        want = []
        for scans, inits in jobs:
            b = ctx.batch(scans)
            want.append(ctx.ndt_align_batch(b, inits) if kind == "ndt" else ctx.icp_align_batch(b, inits, opts))
            b.close()
        pool = api.Pool(ctx, slots=7, max_points=5000, scans_per_job=3, chunk=2, opts=opts, ndt=(kind == "ndt"))
        tickets = [pool.submit(scans, inits, first=0, n_total=len(scans)) for scans, inits in jobs]
        for t, w in zip(tickets, want):
            got, st = pool.wait(t)
            assert np.array_equal(got, w[0]), (kind, np.abs(got - w[0]).max())
            assert st == w[1]
        pool.close()
    ctx.close()
    print("pool over rccl ok")


if __name__ == "__main__":
    main()
