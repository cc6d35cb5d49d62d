#!/usr/bin/env python3
"""Occupancy curve of the hot search kernel (icp_search_walk_kernel<5,15,64>) on the bench workload.

The kernel's waves per CU are set by its dynamic LDS (15 rows x 64 lanes x 8 B = 7 680 B -> 21 one-wave workgroups in 160 KiB).
LOCGPU_K1_LDS_BYTES asks for a LARGER allocation (the extra bytes sit above the stack, unused), i.e. FEWER waves per CU; this script
sweeps it in one process on one box: same map, same 256 scans, same batch, `--steps` profiled steps per point, HIP events around every
stage (the library's own profile counters). Prints a Markdown table -> profiles/r06_k1_occupancy.md.

    gpurun -- python tools/k1_occupancy.py [--scans 256] [--steps 3]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LOCGPU_K1_LDS_BYTES", "0")  # present at the first launch: the library re-reads it at every launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--bytes", type=str, default="7680,8960,10240,11520,12800,15360,20480,7680")
    args = ap.parse_args()
    import torch  # noqa: F401  (first HIP runtime in the process, as in bench.py)
    from loc_lib_amd import api, synth

    ctx = api.Context(0)
    ctx.icp_set_target(synth.make_map(args.map_points))
    ids = [i % 256 for i in range(args.scans)]
    scan_of = {sid: synth.make_scan(sid) for sid in sorted(set(ids))}
    scans = [scan_of[s] for s in ids]
    inits = np.stack([synth.make_pose(s)[1] for s in ids])
    opts = api.icp_opts(method=api.P2PLANE)
    b = ctx.batch(scans)
    ctx.icp_align_batch(b, inits, opts)  # warm-up
    rows = []
    for nbytes in [int(x) for x in args.bytes.split(",")]:
        os.environ["LOCGPU_K1_LDS_BYTES"] = str(nbytes)
        alloc = -(-nbytes // 1280) * 1280  # LDS allocation granule on gfx950: 1 280 B (tests/cpp/lds_semantics.hip)
        waves = min(32, (160 * 1024) // alloc)
        ctx.profile_read(reset=True)
        ctx.profile_enable(1)
        for _ in range(args.steps):
            ctx.icp_align_batch(b, inits, opts)
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        t_search = prof["search_ms"] * prof["search_n"] / args.steps
        t_accum = prof["accum_ms"] * prof["accum_n"] / args.steps
        rows.append((nbytes, waves, t_search, prof["search_ms"], t_accum))
        print("lds %6d B  %2d waves/CU  search %.3f ms/step (%.4f ms/launch)  fit %.3f ms/step" % rows[-1], flush=True)
    print()
    print("| dynamic LDS per wave (B) | waves per CU | search ms per %d-scan step | per launch (ms) | vs 21 waves | fit/accumulate ms per step |" % args.scans)
    print("|---|---|---|---|---|---|")
    base = rows[0][2]
    for r in rows:
        print("| %d | %d | %.3f | %.4f | %.3f | %.3f |" % (r[0], r[1], r[2], r[3], r[2] / base, r[4]))
    b.close()
    ctx.close()


if __name__ == "__main__":
    main()
