#!/usr/bin/env python3
"""Collect HBM traffic of the bench's kernels from rocprofv3 PMC counters (run ON the GPU box).

Two separate passes as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2, and
PMC runs must not be combined with tracing): `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, program directly after `--`.
Units/corrections: both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide reads, so
traffic = (2·FETCH_SIZE + WRITE_SIZE)·1024 bytes per dispatch. (The 2× calibration was made on 16-B-per-lane streaming
loads; the search kernel's 16-B gathers are the same instruction but not streamed — treat the figure as ±2×.)

    python3 tools/collect_traffic.py --out profiles/r01_traffic.json [--scans-per-gpu 256]
"""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counter, outdir, bench_args):
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", outdir, "--", "python3", os.path.join(ROOT, "bench.py")] + bench_args
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"))
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(outdir, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("locgpu::", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--scans-per-gpu", type=int, default=256)
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--method", default="p2plane")
    ap.add_argument("--search", default="tree", choices=["tree", "tree_exact", "grid"])
    ap.add_argument("--workdir", default=os.path.join(ROOT, "gpurun_out", "traffic"))
    a = ap.parse_args()
    bench_args = ["--steps", "2", "--warmup", "0", "--no-cpu-baseline", "--traffic", "none", "--scans-per-gpu", str(a.scans_per_gpu), "--map-points",
                  str(a.map_points), "--method", a.method, "--search", a.search]
    fetch = run_pass("FETCH_SIZE", os.path.join(a.workdir, "fetch"), bench_args)
    write = run_pass("WRITE_SIZE", os.path.join(a.workdir, "write"), bench_args)
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        kernels[k] = dict(dispatches=nf, FETCH_SIZE_KiB=round(f, 2), WRITE_SIZE_KiB=round(w, 2), traffic_bytes_per_launch=int((2 * f + w) * 1024))
    out = dict(config=dict(scans_per_gpu=a.scans_per_gpu, map_points=a.map_points, method=a.method, search=a.search),
               formula="(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per dispatch (gfx950 FETCH_SIZE half-count correction)", kernels=kernels)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in kernels.items() if "icp_" in k or "ndt_" in k or "gn_" in k or "grid_" in k}))


if __name__ == "__main__":
    main()
