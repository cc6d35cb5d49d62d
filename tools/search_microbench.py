#!/usr/bin/env python3
"""Times ONE Gauss–Newton H/B evaluation (search + fit/accumulate + solve) with every scan of the batch active.

    python tools/search_microbench.py [--scans 32] [--map-points 10000000] [--reps 10]
Reports per-kernel HIP-event times and the algorithmic GB/s of the search kernel (same formula as bench.py).
LOCGPU_SEARCH_VARIANT selects experimental kernel variants (see icp_kernels.hip).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loc_lib_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=32)
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--method", type=int, default=2)
    ap.add_argument("--search", choices=["tree", "tree_exact", "grid"], default="tree")
    args = ap.parse_args()
    ctx = api.Context(0)
    m = synth.make_map(args.map_points)
    ctx.icp_set_target(m)
    scans = [synth.make_scan(i % 256) for i in range(args.scans)]
    inits = np.stack([synth.make_pose(i % 256)[1] for i in range(args.scans)])
    b = ctx.batch(scans)
    opts = api.icp_opts(method=args.method)
    if args.search == "grid":
        opts.search_mode = api.SEARCH_GRID_EXACT
    elif args.search == "tree_exact":
        opts.approximate = 0
    ctx.visit_count_enable(True)
    hb_ref = ctx.icp_hb_batch(b, inits, opts)
    vc = ctx.visit_count_read(reset=True)
    ctx.visit_count_enable(False)
    hb = ctx.icp_hb_batch(b, inits, opts)  # warm-up + correctness of the variant vs the instrumented default path
    same = bool(np.array_equal(hb, hb_ref))
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    for _ in range(args.reps):
        ctx.icp_hb_batch(b, inits, opts)
    p = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    # bookkeeping counters LAST: once requested they stay on, and their atomics distort the timing above
    ctx.search_stats_read(reset=True)
    ctx.icp_hb_batch(b, inits, opts)
    ss = ctx.search_stats_read(reset=True)
    q = vc["queries"]
    k = 1 if args.method == 0 else 5
    sbytes = q * 16 + vc["nodes"] * 16 + q * 4 * k
    print(json.dumps(dict(variant=os.environ.get("LOCGPU_SEARCH_VARIANT", "0"), search=args.search, scans=args.scans, queries=q, hb_identical_to_default=same, redo_frac=round(ss["redone"] / max(ss["searched"], 1), 6), walk_frac=round(ss["walked"] / max(ss["searched"], 1), 5), replay_frac=round(ss["replayed"] / max(ss["searched"], 1), 5),
                          search_ms=round(p["search_ms"], 4), accum_ms=round(p["accum_ms"], 4), solve_ms=round(p["solve_ms"], 4),
                          search_us_per_scan=round(1e3 * p["search_ms"] / args.scans, 2),
                          search_alg_GBs=round(sbytes / 1e9 / (p["search_ms"] / 1e3), 1),
                          nodes_per_query=round(vc["nodes"] / q, 2), leaves_per_query=round(vc["leaves"] / q, 2))))


if __name__ == "__main__":
    main()
