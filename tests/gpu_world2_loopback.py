"""Two ranks of a sharded job as two THREADS on one GPU (run by tests/test_gpu_world2.py with LOCGPU_RCCL_LIB pointing at the loopback
double, tests/cpp/loopback_rccl.hip). Each thread owns a context and a communicator rank and runs what a rank process runs: collective
SetInputTarget (tree broadcast + status exchange), scan-sharded batches (even, ragged, and one rank holding nothing), two alignments in
flight, H/B evaluation, NDT, and a point-sharded batch. Every rank must end with ALL poses of the batch, bit-identical to the plain
one-context batch for the scan-sharded cases (a scan's sums come from one rank; everybody else adds zeros)."""
import os
import sys
import threading
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from loc_lib_amd import api, multi_gpu, synth  # noqa: E402

WORLD = 2


def world():
    m = synth.make_local_map(200000, 3, half=40.0)
    s10 = synth.make_scan(3, subsample=10000, crop_half=36.0)
    s2 = synth.make_scan(3, subsample=2000, crop_half=36.0)
    _, init = synth.make_pose(3)
    scans = [s10, s2, s10[::3], s10[::2], s2[::2]]
    inits = np.stack([init] * len(scans))
    inits[1, 4:] += [0.04, -0.03, 0.01]
    inits[3, 4:] += [-0.02, 0.03, 0.0]
    return m, scans, inits


def iters(stats):
    return [s["iterations"] for s in stats]


def reference(m, scans, inits):
    """The plain batch on a context without a communicator."""
    ctx = api.Context(0)
    ctx.icp_set_target(m)
    ctx.ndt_set_target(m)
    plain = ctx.batch(scans)
    want = {}
    for name, method in (("plane", api.P2PLANE), ("point", api.P2P)):
        opts = api.icp_opts(method=method)
        want[name] = ctx.icp_align_batch(plain, inits, opts)
        want[name + "_hb"] = ctx.icp_hb_batch(plain, inits, opts)
    want["ndt"] = ctx.ndt_align_batch(plain, inits)
    plain.close()
    ctx.close()
    return want


SPLITS = {"even": [(0, 3), (3, 5)], "ragged": [(0, 4), (4, 5)], "rank1_empty": [(0, 5), (5, 5)], "rank0_empty": [(0, 0), (0, 5)]}


def rank_main(rank, uid, m, scans, inits, want, log):
    n = len(scans)
    ctx = api.Context(0)
    try:
        ctx.comm_init(rank, WORLD, uid)
        assert ctx.comm_info() == (rank, WORLD)
        ctx.icp_set_target_bcast(m if rank == 0 else None, root=0)   # rank 1 gets the tree through the broadcast only
        ctx.ndt_set_target(m)
        for split, ranges in SPLITS.items():
            lo, hi = ranges[rank]
            b = ctx.batch(scans[lo:hi], first=lo, n_total=n)
            for name, method in (("plane", api.P2PLANE), ("point", api.P2P)):
                opts = api.icp_opts(method=method)
                got, st = ctx.icp_align_batch(b, inits, opts)
                np.testing.assert_array_equal(got, want[name][0], err_msg="%s %s rank %d" % (split, name, rank))
                assert iters(st) == iters(want[name][1]), (split, name, rank)
                hb = ctx.icp_hb_batch(b, inits, opts)
                np.testing.assert_array_equal(hb, want[name + "_hb"], err_msg="hb %s %s rank %d" % (split, name, rank))
            got, st = ctx.ndt_align_batch(b, inits)
            np.testing.assert_array_equal(got, want["ndt"][0], err_msg="%s ndt rank %d" % (split, rank))
            b.close()
            log.append((rank, split))
        # two alignments in flight, differently split, ended in the order they were begun — and once in the other order
        opts = api.icp_opts(method=api.P2PLANE)
        a = ctx.batch(scans[slice(*SPLITS["even"][rank])], first=SPLITS["even"][rank][0], n_total=n)
        b = ctx.batch(scans[slice(*SPLITS["ragged"][rank])], first=SPLITS["ragged"][rank][0], n_total=n)
        for order in ((a, b), (b, a)):
            for _ in range(3):
                ctx.icp_align_batch_begin(a, inits, opts)
                ctx.icp_align_batch_begin(b, inits, opts)
                for x in order:
                    got, st = ctx.align_batch_end(x)
                    np.testing.assert_array_equal(got, want["plane"][0], err_msg="in flight, rank %d" % rank)
                    assert iters(st) == iters(want["plane"][1])
        a.close()
        b.close()
        log.append((rank, "two in flight"))
        # the open-scan pool, sharded: five jobs of five scans through twelve slots (jobs wait for slots; the slot bookkeeping must
        # come out the same on both ranks), each split differently over the ranks, one exchange per pooled iteration
        for kind in ("plane", "ndt"):
            pool = api.Pool(ctx, slots=12, max_points=10000, scans_per_job=n, chunk=2, opts=opts, ndt=(kind == "ndt"))
            tickets = []
            for split in ("even", "ragged", "rank1_empty", "rank0_empty", "even"):
                lo, hi = SPLITS[split][rank]
                tickets.append(pool.submit(scans[lo:hi], inits, first=lo, n_total=n))
            for t in tickets:
                got, st = pool.wait(t)
                np.testing.assert_array_equal(got, want[kind][0], err_msg="pool %s, rank %d" % (kind, rank))
                assert iters(st) == iters(want[kind][1])
            pool.close()
        log.append((rank, "pool"))
        # ... and the way bench.py drives it with several ranks: TWO pools (lanes) per rank, the steps dealt to the emptier one as soon as
        # it has source regions for them, collected in order, `step` in between — every decision taken on numbers all ranks see
        # (locgpu_pool_info / locgpu_pool_done), so both ranks make the same calls in the same order
        lanes = [api.Pool(ctx, slots=6, max_points=10000, scans_per_job=n, opts=opts) for _ in range(2)]
        lo, hi = SPLITS["even"][rank]
        inflight, begun, got_all = [], 0, []
        n_steps = 7
        while begun < n_steps or inflight:
            while begun < n_steps and len(inflight) < 6:
                p = max(lanes, key=lambda q: q.info()["free_regions"])
                if inflight and p.info()["free_regions"] < n:
                    break
                inflight.append((p, p.submit(scans[lo:hi], inits, first=lo, n_total=n)))
                begun += 1
            p, t = inflight[0]
            if p.done(t):
                got_all.append(p.wait(t))
                inflight.pop(0)
            else:
                for q in lanes:
                    q.step(True)
        assert len(got_all) == n_steps
        for got, st in got_all:
            np.testing.assert_array_equal(got, want["plane"][0], err_msg="pool lanes, rank %d" % rank)
            assert iters(st) == iters(want["plane"][1])
        for q in lanes:
            q.close()
        log.append((rank, "pool lanes"))
        # point sharding: every rank holds a slice of every scan, so the sums really are sums of two parts — equal on both ranks,
        # and equal to the plain batch up to the order of the additions
        pts = multi_gpu.point_sharded_batch(ctx, scans, rank, WORLD)
        got, st = ctx.icp_align_batch(pts, inits, opts)
        assert np.abs(got - want["plane"][0]).max() < 1e-9 and iters(st) == iters(want["plane"][1])
        pts.close()
        log.append((rank, "points", got))
    finally:
        ctx.close()


def main():
    assert os.environ.get("LOCGPU_RCCL_LIB"), "run through tests/test_gpu_world2.py (needs the loopback communicator)"
    m, scans, inits = world()
    want = reference(m, scans, inits)
    uid = api.comm_unique_id()
    log, errors = [], []

    def guarded(rank):
        try:
            rank_main(rank, uid, m, scans, inits, want, log)
        except BaseException:  # noqa: BLE001 — reported below; the other rank then times out in its next collective
            errors.append("rank %d:\n%s" % (rank, traceback.format_exc()))

    threads = [threading.Thread(target=guarded, args=(r,)) for r in range(WORLD)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    if any(t.is_alive() for t in threads):
        print("HUNG; progress:", [x[:2] for x in log])
        os._exit(3)
    if errors:
        print("\n".join(errors))
        print("progress:", [x[:2] for x in log])
        sys.exit(1)
    pt = [x[2] for x in log if x[1] == "points"]
    np.testing.assert_array_equal(pt[0], pt[1])
    print("WORLD2 OK:", sorted(x[:2] for x in log))


if __name__ == "__main__":
    main()
