#!/usr/bin/env python3
"""Main-loop trip histograms of the hot search kernel (LOCGPU_STAMP diagnostic build; timing meaningless).

    LOCGPU_STAMP=1 python tools/trip_hist.py [--scans 64] [--map-points 10000000]
Prints (stderr, from liblocgpu.so) the per-lane and per-wave-maximum histograms in bins of two trips: first for ONE H/B evaluation
at the initial poses, then for a whole alignment.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LOCGPU_STAMP", "1")
from loc_lib_amd import api, synth  # noqa: E402


def dump_trips(ctx, b):
    import ctypes
    out = np.zeros(b.n_local * b.max_points, dtype=np.uint32)
    fn = api.lib().locgpu_debug_stamp_trips
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert fn(ctx._h, b._h, out.ctypes.data) == 0
    return out.reshape(b.n_local, b.max_points)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=64)
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--dump", default=None, help="npz file for the per-query trip counts of the two H/B evaluations")
    args = ap.parse_args()
    ctx = api.Context(0)
    ctx.icp_set_target(synth.make_map(args.map_points))
    scans = [synth.make_scan(i % 256) for i in range(args.scans)]
    inits = np.stack([synth.make_pose(i % 256)[1] for i in range(args.scans)])
    b = ctx.batch(scans)
    opts = api.icp_opts(method=api.P2PLANE)
    ctx.search_stats_read(reset=True)
    ctx.icp_hb_batch(b, inits, opts)
    t_init = dump_trips(ctx, b) if args.dump else None
    sys.stderr.write("== one H/B evaluation at the initial poses\n")
    sys.stderr.flush()
    ctx.search_stats_read(reset=True)
    poses, st = ctx.icp_align_batch(b, inits, opts)
    sys.stderr.write("== whole alignment (%.2f iterations per scan)\n" % np.mean([s["iterations"] for s in st]))
    sys.stderr.flush()
    ctx.search_stats_read(reset=True)
    # converged poses: what the late iterations look like
    ctx.icp_hb_batch(b, poses, opts)
    if args.dump:
        np.savez_compressed(args.dump, init=t_init, conv=dump_trips(ctx, b))
    sys.stderr.write("== one H/B evaluation at the converged poses\n")
    sys.stderr.flush()
    ctx.search_stats_read(reset=True)


if __name__ == "__main__":
    main()
