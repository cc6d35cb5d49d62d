#!/usr/bin/env python3
"""Per-iteration dumps of the bench workload for offline studies (run ON the GPU box; LOCGPU_STAMP diagnostic build, timing meaningless):
for `--scans` scans vs the 10 M-pt map, one Gauss–Newton iteration at a time (poses chained like the real loop), the main-loop rounds
every query needed in the search kernel and whether its five neighbour indices changed since the previous iteration. Feeds tools/sim_wave_binning.py (cost-binned waves: does
iteration i's cost predict iteration i+1's?) and the plane-cache study (how many 5-lists are unchanged from one iteration to the next).

    LOCGPU_STAMP=1 python tools/iter_dump.py --scans 16 --iters 9 --out gpurun_out/iter_dump.npz
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LOCGPU_STAMP", "1")
from loc_lib_amd import api, synth  # noqa: E402


def dump_rounds(ctx, b):
    out = np.zeros(b.n_local * b.max_points, dtype=np.uint32)
    fn = api.lib().locgpu_debug_stamp_trips
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert fn(ctx._h, b._h, out.ctypes.data) == 0
    return out.reshape(b.n_local, b.max_points)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=16)
    ap.add_argument("--iters", type=int, default=9)
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    ctx = api.Context(0)
    ctx.icp_set_target(synth.make_map(a.map_points))
    ids = [int(i * 256 / a.scans) % 256 for i in range(a.scans)]
    scans = [synth.make_scan(i) for i in ids]
    poses = np.stack([synth.make_pose(i)[1] for i in ids])
    b = ctx.batch(scans)
    opts = api.icp_opts(method=api.P2PLANE, max_iteration=1)
    rounds, same_bits, dxn, eff = [], [], [], []
    prev = None
    for it in range(a.iters):
        ctx.search_stats_read(reset=True)
        poses, st = ctx.icp_align_batch(b, poses, opts)
        rounds.append(dump_rounds(ctx, b).astype(np.uint16))
        nn = ctx.debug_batch_nn(b, 5)
        same = np.all(nn == prev, axis=2) if prev is not None else np.zeros(nn.shape[:2], dtype=bool)
        same_bits.append(np.packbits(same, axis=1))
        prev = nn
        dxn.append([s["last_dx_norm"] for s in st])
        s = ctx.search_stats_read(reset=True)
        eff.append(s["walked"] / max(s["replayed"], 1))
        sys.stderr.write("iteration %d: lane efficiency %.3f, mean |dx| %.4f, 5-lists unchanged since the previous iteration %.3f (whole 64-point waves %.3f, 256-point blocks %.3f)\n"
                         % (it, eff[-1], float(np.mean(dxn[-1])), same.mean(), same.reshape(len(ids), -1, 64).all(axis=2).mean(), same.reshape(len(ids), -1, 256).all(axis=2).mean()))
    np.savez_compressed(a.out, rounds=np.stack(rounds), same=np.stack(same_bits), dx_norm=np.array(dxn), lane_eff=np.array(eff), scan_ids=np.array(ids))
    print("wrote", a.out)


if __name__ == "__main__":
    main()
