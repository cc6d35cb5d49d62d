#!/usr/bin/env python3
"""Fills BASELINE.md's table (configs 1-5 of BASELINE.json) on the GPU box: runs bench.py / the microbenches per config and
collects CPU R1 (the oracle is written in the reference's style: per-query std::vector + std::priority_queue, pointer nodes —
1 thread), CPU R2 (the flat-array port, 1 thread), CPU R3 (R2 scan-parallel over native threads on all host cores), GPU scans/s
(scan H2D inside the timed region), ms per GN iteration, roofline fractions, pose delta.

    python3 tests/perf/baseline_table.py --out gpurun_out/baseline_table.json
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run_bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + list(extra)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    for ln in reversed(out.stdout.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise RuntimeError("bench produced no JSON line: " + out.stderr[-400:])


def row_from_bench(name, j):
    cb = j.get("cpu_baseline") or {}
    cfg = j.get("config", {})
    r2, r3 = cb.get("r2_flat_port") or {}, cb.get("r3_flat_port_all_cores") or {}
    return dict(config=name, workload=cfg.get("workload"), cpu_r1_scans_s=cb.get("value"), cpu_r2_scans_s=r2.get("value"), cpu_r3_scans_s=r3.get("value"),
                cpu_cores=r3.get("cores"), gpu_scans_s=j["value"], ms_per_step=j["ms_per_step"], scan_h2d_in_timed_region=cfg.get("scan_h2d_in_timed_region"),
                search_hbm_frac=(j.get("roofline") or {}).get("hbm_frac"),
                gpu_iter_ms_per_scan=j.get("icp_iter_ms_per_scan"), gn_iterations_per_scan=j.get("gn_iterations_per_scan"),
                search_roofline_frac=(j.get("roofline") or {}).get("frac"), nominal_bytes_frac=(j.get("roofline") or {}).get("nominal_bytes_frac"), traffic_bytes=(j.get("roofline") or {}).get("traffic"),
                pose_delta_m=cb.get("max_pose_delta_gpu_vs_cpu_m"), gpu_over_cpu=cb.get("gpu_over_cpu"))


def config1():
    """10 k-pt scan vs 100 k-pt map, one scan at a time through the host-pointer entry point (the plumbing case)."""
    from loc_lib_amd import api, synth
    from oracle import locref
    m = synth.make_map(100_000)
    rows = []
    ctx = api.Context(0)
    ctx.icp_set_target(m)
    icp = locref.Icp(method=locref.P2PLANE)
    icp.set_target(m)
    opts = api.icp_opts(api.P2PLANE)
    t_cpu = t_gpu = 0.0
    worst = 0.0
    n = 16
    for s in range(n):
        scan = synth.make_scan(s, subsample=10000)
        _, init = synth.make_pose(s)
        ctx.icp_align(scan, init, opts)  # warm
        t0 = time.perf_counter()
        pose, st = ctx.icp_align(scan, init, opts)
        t_gpu += time.perf_counter() - t0
        t0 = time.perf_counter()
        r = icp.align(scan, init)
        t_cpu += time.perf_counter() - t0
        worst = max(worst, float(np.linalg.norm(pose[4:] - r["pose"][4:])))
        rows.append(st["iterations"])
    ctx.close()
    return dict(config="1: 10k scan vs 100k map, P2Plane, single-scan host-pointer calls", cpu_r1_scans_s=n / t_cpu, gpu_scans_s=n / t_gpu,
                gn_iterations_per_scan=float(np.mean(rows)), pose_delta_m=worst, gpu_over_cpu=round(t_cpu / t_gpu, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "baseline_table.json"))
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    steps = ["--steps", "3", "--warmup", "1"] if a.quick else ["--steps", "10", "--warmup", "2"]
    rows = [config1()]
    rows.append(row_from_bench("2: 115200-pt scans vs 1M-pt map, P2Plane (64 scans per step)", run_bench("--map-points", "1000000", "--scans-per-gpu", "64", "--traffic", "none", *steps)))
    rows.append(row_from_bench("3a: vs 10M-pt map, P2Plane (64 scans per step)", run_bench("--scans-per-gpu", "64", "--traffic", "none", *steps)))
    rows.append(row_from_bench("3b: vs 10M-pt map, direct NDT (64 scans per step)", run_bench("--scans-per-gpu", "64", "--method", "ndt", "--traffic", "none", *steps)))
    small = ["--steps", "12", "--warmup", "2"] if a.quick else ["--steps", "80", "--warmup", "8"]
    rows.append(row_from_bench("3c: vs 10M-pt map, P2Plane (32 scans per step: what one of eight ranks holds of configs[3]) — through the open-scan pool (the default for small steps)",
                               run_bench("--scans-per-gpu", "32", "--traffic", "none", *small)))
    rows.append(row_from_bench("3c': the same as plain batches, three alignments in flight (rounds 1-4)", run_bench("--scans-per-gpu", "32", "--pool-slots", "0", "--no-cpu-baseline", "--traffic", "none", *small)))
    rows.append(row_from_bench("4: 256 scans vs 10M-pt map, P2Plane, one GPU (the default bench line: three alignments in flight)", run_bench(*steps)))
    fast = ["--no-cpu-baseline", "--traffic", "none"]
    rows.append(row_from_bench("4p: the same with one alignment at a time (--pipeline 1)", run_bench("--pipeline", "1", *fast, *steps)))
    rows.append(row_from_bench("4q: the same with two alignments in flight (--pipeline 2)", run_bench("--pipeline", "2", *fast, *steps)))
    for n in (256, 64, 32):
        rows.append(row_from_bench("4s: configs[3] as written on ONE rank: %d scans in all, RCCL all-reduce every iteration%s" % (n, "" if n == 256 else " — open-scan pool"),
                                   run_bench("--scaling", "strong", "--total-scans", str(n), *fast, "--steps", str(max(10, 2560 // n)), "--warmup", "4")))
        if n != 256:
            rows.append(row_from_bench("4s': %d scans in all as plain sharded batches, three alignments in flight (rounds 1-4)" % n,
                                       run_bench("--scaling", "strong", "--total-scans", str(n), "--pool-slots", "0", *fast, "--steps", str(max(10, 2560 // n)), "--warmup", "4")))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "perf", "pipeline_microbench.py"), "--only", "stream"], capture_output=True, text=True, timeout=900)
    st = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])["stream"]
    rows.append(dict(config="5: streaming loop (upload, removeNaN, voxel filter, P2Plane vs local map, keyframe every 5th: submap + re-ingest with the tree built on a worker thread)",
                     gpu_scans_s=st["scans_per_s"], ms_per_scan=st["ms_per_scan"], pose_delta_m=st["max_pose_abs_diff"], local_map_points=st["local_map_points"],
                     gpu_scans_s_blocking_target=st.get("blocking_target", {}).get("scans_per_s")))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(rows=rows, host_threads=os.cpu_count(), usable_cores=len(os.sched_getaffinity(0))), open(a.out, "w"), indent=1)
    print("| config | CPU R1 scans/s (1 thread) | CPU R2 scans/s (1 thread) | CPU R3 scans/s (cores) | GPU×1 scans/s | GPU ms per scan-iteration | dominant kernel: VALU issue frac / nominal-bytes frac / HBM frac | pose Δ vs oracle [m] |")
    print("|---|---|---|---|---|---|---|---|")
    f = lambda v, p="%.3g": "—" if v is None else p % v
    for r in rows:
        print("| %s | %s | %s | %s | %s | %s | %s / %s | %s |" % (r["config"], f(r.get("cpu_r1_scans_s")), f(r.get("cpu_r2_scans_s")),
              ("%s (%s)" % (f(r.get("cpu_r3_scans_s")), r.get("cpu_cores"))) if r.get("cpu_r3_scans_s") else "—", f(r.get("gpu_scans_s"), "%.4g"),
              f(r.get("gpu_iter_ms_per_scan")), f(r.get("search_roofline_frac")) + " / " + f(r.get("nominal_bytes_frac")), f(r.get("search_hbm_frac")), f(r.get("pose_delta_m"), "%.1e")))


if __name__ == "__main__":
    main()
