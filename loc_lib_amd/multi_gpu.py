"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).

Two ways the path shards:

* scan sharding (BASELINE config 4, the bench's mode): scans are independent given the read-only map, so each rank
  aligns its own contiguous slice of the batch and no collective sits on the data path; poses are gathered once at the end.
* point sharding (one large alignment split over ranks): every rank evaluates H, B over its slice of the source points
  (``locgpu_icp_hb_batch``), the 44-double normal equations are summed with ONE all-reduce per Gauss–Newton iteration,
  and every rank applies the same update (``locgpu_gn_update``) — the reference loop of
  IcpRegistration::AlignP2Plane (icp_registration.cpp:345-381) with the sum over points distributed.
"""
import numpy as np


def init_comm(ctx, dist=None):
    """Join this rank's context to the RCCL communicator of the job (native collectives inside liblocgpu.so: the per-iteration
    all-reduce of sharded batches and the tree broadcast). The 128-byte RCCL id is made on rank 0 and handed round through
    torch.distributed's store (works on gloo and nccl process groups alike). Without a process group: a one-rank communicator."""
    from . import api
    if dist is None or not dist.is_initialized():
        ctx.comm_init(0, 1, api.comm_unique_id())
        return 0, 1
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [api.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init(rank, world, box[0])
    return rank, world


def scan_sharded_batch(ctx, scans, n_total, rank, world):
    """BASELINE configs[3]: a batch of n_total scans split contiguously over the ranks; `scans` = this rank's slice
    (shard_range(n_total, rank, world)). Aligning the batch is collective and returns all n_total poses on every rank."""
    lo, hi = shard_range(n_total, rank, world)
    if len(scans) != hi - lo:
        raise ValueError("rank %d holds scans [%d, %d) of %d" % (rank, lo, hi, n_total))
    return ctx.batch(scans, first=lo, n_total=n_total)


def point_sharded_batch(ctx, scans, rank, world):
    """One large alignment split by points: every rank holds the slice shard_range(len(scan), rank, world) of every scan."""
    parts = []
    for s in scans:
        lo, hi = shard_range(len(s), rank, world)
        parts.append(s[lo:hi])
    return ctx.batch(parts, first=0, n_total=len(scans))


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slice of n_items for `rank` of `world` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_poses(local_poses, n_total, dist=None):
    """All ranks' [n_local, 7] poses in rank order → [n_total, 7] on every rank."""
    local_poses = np.ascontiguousarray(local_poses, dtype=np.float64).reshape(-1, 7)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_poses
    import torch
    world = dist.get_world_size()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = torch.zeros((pad, 7), dtype=torch.float64, device=dev)
    buf[: local_poses.shape[0]] = torch.from_numpy(local_poses).to(dev)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return np.concatenate([out[r][: hi - lo].cpu().numpy() for r, (lo, hi) in enumerate(sizes)], axis=0)


def allreduce_hb(hb, dist=None):
    """Sum the [n, 44] normal equations (H36, B6, effective_num, ok) over ranks; `ok` is recomputed by gn_update."""
    hb = np.ascontiguousarray(hb, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return hb
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.from_numpy(hb.copy()).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def point_sharded_align(hb_fn, gn_update_fn, init_pose, method, max_iteration=20, min_effective_pts=10, eps=1e-2, dist=None):
    """Gauss–Newton loop with the per-point sums distributed over ranks.

    hb_fn(pose) -> 44 doubles over THIS rank's points; gn_update_fn = loc_lib_amd.api.gn_update.
    Returns (pose, iterations). Every rank ends with the same pose (same reduced numbers, same host arithmetic).
    """
    pose = np.array(init_pose, dtype=np.float64)
    iters = 0
    for _ in range(max_iteration):
        hb = allreduce_hb(np.asarray(hb_fn(pose)).reshape(1, 44), dist)[0]
        iters += 1
        pose, dx, applied, stop = gn_update_fn(hb, method, min_effective_pts, eps, pose)
        if stop:
            break
    return pose, iters
