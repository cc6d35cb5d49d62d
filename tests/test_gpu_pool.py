"""The open-scan pool (locgpu_pool, loc_lib_amd/csrc/scan_pool.hip): jobs submitted at any time, iterated together over the union
of their open scans, collected by ticket — against the plain batch calls (bit for bit) and, through them, the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _jobs(small_world, n_jobs, per_job):
    """Ragged jobs cut from the 10 k-pt scan, with initial poses that converge after different numbers of iterations (one at
    once, some far)."""
    s = small_world["scan10k"]
    pose, truth = small_world["init_pose"], small_world["true_pose"]
    rng = np.random.default_rng(5)
    jobs = []
    for j in range(n_jobs):
        scans, inits = [], []
        for i in range(per_job):
            lo = int(rng.integers(0, 3000))
            step = int(rng.integers(1, 4))
            scans.append(np.ascontiguousarray(s[lo::step][: int(rng.integers(1500, 6000))]))
            ip = np.array(pose)
            ip[4:] += rng.uniform(-0.25, 0.25, 3) * (2.0 if (i + j) % 5 == 0 else 1.0)
            if (i + 3 * j) % 7 == 0:
                ip = np.array(truth)
            inits.append(ip)
        jobs.append((scans, np.stack(inits)))
    return jobs


@pytest.mark.parametrize("method", ["p2plane", "p2p", "p2line"])
def test_pool_jobs_get_the_plain_batch_bits(gpu_ctx, api, small_world, method):
    """Nine four-scan jobs through a pool of eight slots and twelve source regions (so submits wait for regions, scans wait for slots
    and enter one by one, scans of different jobs share launches, slots and regions are reused) give poses, iteration counts and
    stats equal to locgpu_icp_align_batch on a plain batch of each job, bit for bit."""
    gpu_ctx.icp_set_target(small_world["map"])
    opts = api.icp_opts(method=dict(p2plane=api.P2PLANE, p2p=api.P2P, p2line=api.P2LINE)[method])
    jobs = _jobs(small_world, 9, 4)
    want = []
    for scans, inits in jobs:
        b = gpu_ctx.batch(scans)
        want.append(gpu_ctx.icp_align_batch(b, inits, opts))
        b.close()
    its = sorted({st["iterations"] for w in want for st in w[1]})
    assert len(its) >= 3, its  # the jobs really finish at different times
    for chunk in (1, 3):
        pool = api.Pool(gpu_ctx, slots=8, max_points=6000, scans_per_job=4, chunk=chunk, opts=opts, prefetch=4)
        tickets = [pool.submit(scans, inits) for scans, inits in jobs[:3]]  # the arena is full now
        for scans, inits in jobs[3:]:
            tickets.append(pool.submit(scans, inits))  # lets running scans finish until the job has regions
        for t, w in reversed(list(zip(tickets, want))):  # collected in another order than submitted
            got, st = pool.wait(t)
            assert np.array_equal(got, w[0])
            assert st == w[1]
        info = pool.info()
        assert info["free"] == 8 and info["free_regions"] == 12 and info["jobs"] == 0 and info["open"] == 0
        with pytest.raises(RuntimeError):
            pool.wait(tickets[0])  # a ticket is good once
        pool.close()


def test_pool_with_direct_ndt(gpu_ctx, api, small_world):
    gpu_ctx.ndt_set_target(small_world["map"])
    jobs = _jobs(small_world, 5, 3)
    want = []
    for scans, inits in jobs:
        b = gpu_ctx.batch(scans)
        want.append(gpu_ctx.ndt_align_batch(b, inits))
        b.close()
    pool = api.Pool(gpu_ctx, slots=7, max_points=6000, scans_per_job=3, chunk=2, ndt=True)
    tickets = [pool.submit(scans, inits) for scans, inits in jobs]
    for t, w in zip(tickets, want):
        got, st = pool.wait(t)
        assert np.array_equal(got, w[0])
        assert st == w[1]
    pool.close()


def test_pool_refusals(gpu_ctx, api, small_world):
    gpu_ctx.icp_set_target(small_world["map"])
    opts = api.icp_opts(method=api.P2PLANE)
    s = small_world["scan10k"]
    pose = small_world["init_pose"]
    pool = api.Pool(gpu_ctx, slots=2, max_points=4000, scans_per_job=2, opts=opts)
    with pytest.raises(RuntimeError):
        pool.submit([s[:5000]], pose[None])           # more points than a slot holds
    with pytest.raises(RuntimeError):
        pool.submit([s[:100]] * 5, np.stack([pose] * 5))  # more scans than source regions (slots + prefetch = 4)
    with pytest.raises(RuntimeError):
        pool.submit([s[:100]], np.stack([pose] * 2), first=0, n_total=2)  # part of a job without a communicator
    t = pool.submit([s[:3000], s[:10]], np.stack([pose, pose]))  # the pool still works
    got, st = pool.wait(t)
    b = gpu_ctx.batch([s[:3000], s[:10]])
    want, wst = gpu_ctx.icp_align_batch(b, np.stack([pose, pose]), opts)
    assert np.array_equal(got, want) and st == wst
    b.close()
    pool.close()
    with pytest.raises(RuntimeError):
        api.Pool(gpu_ctx, slots=4, max_points=100, opts=api.icp_opts(method=api.P2PLANE, search_mode=api.SEARCH_GRID_EXACT))


def test_sharded_pool_over_rccl_one_rank(api, small_world):
    """A context with a (one-rank) RCCL communicator: every pooled iteration goes through sum_partials → ncclAllReduce → solve; the
    jobs' poses are the plain batches', bit for bit — with the collective on the pool's stream and, forced through the same switch
    the sharded batches use, with the owner solving ahead and the exchange on the communication stream."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for decoupled in ("0", "1"):
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_pool_rccl_case.py")], env=dict(os.environ, LOCGPU_SHARD_DECOUPLED=decoupled),
                           capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "pool over rccl ok" in r.stdout


@pytest.mark.parametrize("approx", [1, 0])
def test_pooled_scans_on_the_64_lane_kernels_equal_the_plain_batch_on_the_16_lane_one(gpu_ctx, api, small_world, approx):
    """launch_walk_kd picks the search kernel by the number of waves of the LAUNCH — the 16-lane kernel (every level stored,
    ties answered in the wave) below 2 048 full waves, the 64-lane walk + deep pass + redo chain above — so a pooled scan can run
    another kernel than in its plain batch. Forty slots x ~6 000 points are ≈ 3 800 waves (the chain), a plain four-scan batch ≈ 380
    (the 16-lane kernel): the jobs' poses, iteration counts and stats must still be the plain batches' bit for bit, with the
    reference's alpha-pruned search and with the exact one."""
    gpu_ctx.icp_set_target(small_world["map"])
    opts = api.icp_opts(method=api.P2PLANE)
    opts.approximate = approx
    jobs = _jobs(small_world, 16, 4)  # the kernel is chosen by ceil(max_points / 64) x open slots: 94 x 40 at first (the chain), the 16-lane kernel once ≤ 21 slots are open
    want = []
    for scans, inits in jobs:
        b = gpu_ctx.batch(scans)
        want.append(gpu_ctx.icp_align_batch(b, inits, opts))
        b.close()
    pool = api.Pool(gpu_ctx, slots=40, max_points=6000, scans_per_job=4, chunk=2, opts=opts, prefetch=8)
    tickets = [pool.submit(scans, inits) for scans, inits in jobs]
    assert pool.info()["slots"] == 40
    for t, w in zip(tickets, want):
        got, st = pool.wait(t)
        assert np.array_equal(got, w[0])
        assert st == w[1]
    pool.close()


_FAILED_COPY_CASE = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from loc_lib_amd import api, synth
ctx = api.Context(0)
ctx.icp_set_target(synth.make_local_map(200000, 3, half=40.0))
opts = api.icp_opts(method=api.P2PLANE)
s, pose = synth.make_scan(3, subsample=10000, crop_half=36.0), synth.make_pose(3)[1]
jobs = [([np.ascontiguousarray(s[j::3][:3000 + 100 * i]) for i in range(3)], np.stack([pose] * 3)) for j in range(3)]
want = []
for scans, inits in jobs:
    b = ctx.batch(scans)
    want.append(ctx.icp_align_batch(b, inits, opts))
    b.close()
pool = api.Pool(ctx, slots=4, max_points=4000, scans_per_job=3, chunk=2, opts=opts, prefetch=8)
tickets = [pool.submit(scans, inits) for scans, inits in jobs]   # every submit returns its ticket: the second job's copy fails LATER, on the service thread
for k in (0, 2):
    got, st = pool.wait(tickets[k])
    assert np.array_equal(got, want[k][0]) and st == want[k][1], k
try:
    pool.wait(tickets[1])
    raise SystemExit("the failed job's wait did not fail")
except RuntimeError as e:
    assert "injected failure" in str(e), str(e)
info = pool.info()
assert info["free"] == 4 and info["free_regions"] == 12 and info["jobs"] == 0 and info["open"] == 0, info
t = pool.submit(*jobs[0])   # and the pool goes on
got, st = pool.wait(t)
assert np.array_equal(got, want[0][0])
pool.close()
ctx.close()
print("failed copy case ok")
"""


def test_a_failed_copy_fails_its_job_and_not_the_pool(tmp_path):
    """a copy that fails on the upload service thread used to leave its job at the head of the waiting list for good —
    every later submit / step / wait returned its error and no slot or region came back. Now the job is FAILED: its scans pass through
    the pool empty, locgpu_pool_wait(ticket) reports the failure, the jobs before and behind it get the plain batches' bits, every slot
    and region is free at the end and the pool takes further jobs. (LOCGPU_TEST_FAIL_UPLOAD=2: the second region copy of the process.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FAILED_COPY_CASE], env=dict(os.environ, LOCGPU_TEST_FAIL_UPLOAD="2"), capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "failed copy case ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
