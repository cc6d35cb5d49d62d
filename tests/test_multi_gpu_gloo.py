"""world_size-2 `gloo` tests of the multi-GPU host logic (loc_lib_amd/multi_gpu.py) on CPU.

No GPU here, so the per-rank H,B evaluation is done by the oracle (a checker standing in for locgpu_icp_hb_batch);
what is under test is the product's sharding, the all-reduce of the 44-double normal equations, the host-side
Gauss–Newton update (locgpu_gn_update in liblocgpu.so) and the pose gather."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from loc_lib_amd import api, multi_gpu, synth
    from oracle import locref

    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = synth.make_local_map(60000, 3, half=30.0)
    scan = synth.make_scan(3, subsample=4000, crop_half=27.0)
    _, init = synth.make_pose(3)
    icp = locref.Icp(method=2)
    icp.set_target(m)

    # (1) point sharding: each rank sums over its slice; one all-reduce per iteration; identical update everywhere
    lo, hi = multi_gpu.shard_range(len(scan), rank, world)

    def hb_fn(pose):
        ok, H, B, eff = icp.hb(scan[lo:hi], pose)
        return np.concatenate([H.reshape(-1), B, [eff, float(ok)]])

    pose, iters = multi_gpu.point_sharded_align(hb_fn, api.gn_update, init, method=2, dist=dist)

    # (2) scan sharding: ranks own disjoint scans; poses gathered in order
    scans = [synth.make_scan(3, subsample=n, crop_half=27.0) for n in (1500, 800, 1200)]
    s_lo, s_hi = multi_gpu.shard_range(len(scans), rank, world)
    local = np.stack([icp.align(scans[i], init)["pose"] for i in range(s_lo, s_hi)]) if s_hi > s_lo else np.zeros((0, 7))
    gathered = multi_gpu.gather_poses(local, len(scans), dist)
    out_q.put((rank, pose, iters, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from loc_lib_amd import multi_gpu
    for n, w in ((256, 8), (10, 3), (2, 4), (0, 2)):
        parts = [multi_gpu.shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_world2_point_and_scan_sharding(locref, synth):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # reference: single-process oracle
    m = synth.make_local_map(60000, 3, half=30.0)
    scan = synth.make_scan(3, subsample=4000, crop_half=27.0)
    _, init = synth.make_pose(3)
    icp = locref.Icp(method=2)
    icp.set_target(m)
    ro = icp.align(scan, init)
    for rank, pose, iters, gathered in res:
        assert iters == ro["iters"]
        assert np.linalg.norm(pose[4:] - ro["pose"][4:]) < 1e-9 and np.abs(pose[:4] - ro["pose"][:4]).max() < 1e-9
    np.testing.assert_array_equal(res[0][1], res[1][1])  # both ranks hold the same pose bit for bit
    scans = [synth.make_scan(3, subsample=n, crop_half=27.0) for n in (1500, 800, 1200)]
    want = np.stack([icp.align(s, init)["pose"] for s in scans])
    for _, _, _, gathered in res:
        np.testing.assert_allclose(gathered, want, atol=1e-12)
