"""EIGHT ranks of BASELINE configs[3] as eight THREADS on one GPU over the loopback communicator (tests/cpp/loopback_rccl.hip), driven
the way `bench.py --gpus 8` drives a rank: collective SetInputTarget (rank 0 builds the tree, broadcast), 256 scans
in all, 32 per rank, every step a pool job sharded over the ranks, TWO pools (lanes) per rank kept full by bench.py's run_steps loop
(a step is submitted to the emptier lane as soon as it has source regions; collected in order; `step` in between), one all-reduce of
[slots][32] doubles per pooled iteration, owner solves ahead. Every rank must end every step with ALL 256 poses, bit-identical to the
plain 256-scan batch of a context without a communicator; the ranks must have entered the same number of collectives; no rendezvous
may time out. Scans are cut to every 16th point (7 200 points) so that sixteen pools fit one GPU and the run takes seconds.

Run by tests/test_gpu_world8.py with LOCGPU_RCCL_LIB pointing at the double."""
import ctypes
import os
import sys
import threading
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from loc_lib_amd import api, multi_gpu, synth  # noqa: E402

WORLD = 8
N_TOTAL = 256
N_STEPS = 5
MAP_POINTS = 1_000_000


def rank_main(rank, uid, m, scans, inits, want, log):
    ctx = api.Context(0)
    try:
        ctx.comm_init(rank, WORLD, uid)
        assert ctx.comm_info() == (rank, WORLD)
        ctx.icp_set_target_bcast(m if rank == 0 else None, root=0)  # ONE host tree build, the packed tree broadcast to seven ranks
        opts = api.icp_opts(method=api.P2PLANE)
        lo, hi = multi_gpu.shard_range(N_TOTAL, rank, WORLD)
        mine = api.MarshalledScans(scans[lo:hi])
        max_pts = max(len(s) for s in scans)
        # bench.py's sizes for this case: room for 256 scans of the rank = 8 steps = 2 048 slots in two lanes — here a quarter of that
        # (512 slots, two steps in flight per lane), the same code path
        lanes_n, pool_slots = 2, 2 * N_TOTAL
        depth = pool_slots // N_TOTAL
        pools = [api.Pool(ctx, slots=pool_slots // lanes_n, max_points=max_pts, scans_per_job=N_TOTAL, opts=opts) for _ in range(lanes_n)]
        inflight, begun, results = [], 0, []
        while begun < N_STEPS or inflight:  # bench.py run_steps(), pool branch
            while begun < N_STEPS and len(inflight) < 4 * depth:
                p = max(pools, key=lambda q: q.info()["free_regions"])
                if inflight and p.info()["free_regions"] < N_TOTAL:
                    break
                inflight.append((p, p.submit(mine, inits, first=lo, n_total=N_TOTAL)))
                begun += 1
            p, t = inflight[0]
            if p.done(t):
                results.append(p.wait(t))
                inflight.pop(0)
            else:
                for q in pools:
                    q.step(True)
        assert len(results) == N_STEPS
        for got, st in results:
            np.testing.assert_array_equal(got, want[0], err_msg="pool lanes, rank %d" % rank)
            assert [s["iterations"] for s in st] == [s["iterations"] for s in want[1]], rank
        for q in pools:
            q.close()
        log.append((rank, "pool lanes"))
        # configs[3] the round-4 way as well: one sharded batch of the 256 scans, the rank holds 32 of them
        b = ctx.batch(scans[lo:hi], first=lo, n_total=N_TOTAL)
        got, st = ctx.icp_align_batch(b, inits, opts)
        np.testing.assert_array_equal(got, want[0], err_msg="sharded batch, rank %d" % rank)
        b.close()
        log.append((rank, "sharded batch"))
    finally:
        ctx.close()


def main():
    stub = os.environ.get("LOCGPU_RCCL_LIB")
    assert stub, "run through tests/test_gpu_world8.py (needs the loopback communicator)"
    m = synth.make_map(MAP_POINTS)
    ids = [i % 256 for i in range(N_TOTAL)]
    scan_of = {}
    for sid in sorted(set(ids)):
        scan_of[sid] = np.ascontiguousarray(synth.make_scan(sid)[::16])
    scans = [scan_of[s] for s in ids]
    inits = np.stack([synth.make_pose(s)[1] for s in ids])
    ctx = api.Context(0)  # the plain batch on a context without a communicator
    ctx.icp_set_target(m)
    plain = ctx.batch(scans)
    want = ctx.icp_align_batch(plain, inits, api.icp_opts(method=api.P2PLANE))
    plain.close()
    ctx.close()
    its = sorted({s["iterations"] for s in want[1]})
    assert len(its) >= 3, its  # the scans really leave the pool at different times
    uid = api.comm_unique_id()
    log, errors = [], []

    def guarded(rank):
        try:
            rank_main(rank, uid, m, scans, inits, want, log)
        except BaseException:  # noqa: BLE001 — reported below; the other ranks then time out in their next collective
            errors.append("rank %d:\n%s" % (rank, traceback.format_exc()))

    threads = [threading.Thread(target=guarded, args=(r,)) for r in range(WORLD)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(900)
    if any(t.is_alive() for t in threads):
        print("HUNG; progress:", sorted(log))
        os._exit(3)
    if errors:
        print("\n".join(errors))
        print("progress:", sorted(log))
        sys.exit(1)
    counts = (ctypes.c_ulonglong * 8)()
    ctypes.CDLL(stub).loopback_rccl_counts(counts)
    counts = list(counts)
    assert len(set(counts)) == 1 and counts[0] > 0, counts
    print("WORLD8 OK: %d collectives on every rank; %s" % (counts[0], sorted(log)))


if __name__ == "__main__":
    main()
