#!/usr/bin/env python3
"""What slam_demo runs BY DEFAULT, measured. slam_demo/config/slam.yaml selects, per node:

  lio_mapping   :18,41   matching_method 1 (ICP) with icp_option.method 0 = P2P — k = 1 search, icp_registration.cpp:57-103,267-303;
                :52-53   its NDT option block: incremental voxels, nearby CENTER (ndt_registration.cpp:150-236,262-372)
  localisation  :69,111-112  direct NDT, NEARBY6, voxel 1.2 (ndt_registration.cpp:87-148,374-464)

Rounds 1-4 timed the reference's P2Plane and direct NDT at voxel 1.0 only. Rows, each with the CPU restatement R1 (one thread, the
reference's style) beside the GPU figure and the pose delta between the two:

  D1a  P2P, one 115 200-pt scan per call vs the 10 M-pt map                 (single ScanMatch latency)
  D1b  P2P, 256 scans per step vs the 10 M-pt map                           (bench.py --method p2p)
  D1c  the streaming loop of configs[4] with P2P as the matcher
  D2a  incremental NDT / CENTER, one filtered scan per call vs its local map
  D2b  incremental NDT / CENTER, 64 scans per step vs the local map
  D2c  the streaming loop with incremental NDT / CENTER as the matcher
  D3a  direct NDT / NEARBY6 / voxel 1.2, one scan per call vs the 10 M-pt map
  D3b  direct NDT / NEARBY6 / voxel 1.2, 256 scans per step                  (bench.py --method ndt --ndt-voxel 1.2)

    python3 tests/perf/defaults_table.py --out gpurun_out/defaults_table.json
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
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))


def run_bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + list(extra)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    for ln in reversed(out.stdout.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise RuntimeError("bench produced no JSON line: " + out.stderr[-400:])


def bench_row(name, j):
    cb = j.get("cpu_baseline") or {}
    return dict(row=name, workload=j["config"]["workload"], gpu_scans_s=j["value"], ms_per_step=j["ms_per_step"], gpu_iter_ms_per_scan=j.get("icp_iter_ms_per_scan"),
                gn_iterations_per_scan=j.get("gn_iterations_per_scan"), kernel_ms_per_step=j.get("kernel_ms_per_step"), cpu_r1_scans_s=cb.get("value"),
                pose_delta_m=cb.get("max_pose_delta_gpu_vs_cpu_m"), gpu_over_cpu=cb.get("gpu_over_cpu"))


def latency(align_gpu, align_cpu, cases, reps=5, cpu_cases=2):
    """Mean wall time of one blocking call (ms), iterations, CPU R1 ms over the first `cpu_cases` cases, worst pose delta."""
    ts, its, tc, worst = [], [], [], 0.0
    for i, case in enumerate(cases):
        align_gpu(case)  # warm
        t0 = time.perf_counter()
        for _ in range(reps):
            pose, st = align_gpu(case)
        ts.append((time.perf_counter() - t0) / reps)
        its.append(st["iterations"])
        if i < cpu_cases:
            t0 = time.perf_counter()
            r = align_cpu(case)
            tc.append(time.perf_counter() - t0)
            worst = max(worst, float(np.abs(np.asarray(pose) - r["pose"]).max()))
            assert st["iterations"] == r["iters"], (st, r["iters"])
    return dict(gpu_ms_per_scan=round(1e3 * float(np.mean(ts)), 4), gpu_scans_s=round(1.0 / float(np.mean(ts)), 1), gn_iterations_per_scan=float(np.mean(its)),
                cpu_r1_ms_per_scan=round(1e3 * float(np.mean(tc)), 2), cpu_r1_scans_s=round(1.0 / float(np.mean(tc)), 3), pose_delta=worst,
                gpu_over_cpu=round(float(np.mean(tc)) / float(np.mean(ts)), 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "defaults_table.json"))
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    from loc_lib_amd import api, synth
    from oracle import locref  # the checker and the CPU baseline beside every row
    import pipeline_microbench as pm

    steps = ["--steps", "4", "--warmup", "1"] if a.quick else ["--steps", "20", "--warmup", "3"]
    rows = []
    ctx = api.Context(0)
    big = synth.make_map(10_000_000)
    n_lat = 4 if a.quick else 12
    full = [(synth.make_scan(s), synth.make_pose(s)[1]) for s in range(n_lat)]

    # ---- D1a / D3a: one full scan per call vs the 10 M-pt map
    ctx.icp_set_target(big)
    p2p = api.icp_opts(method=api.P2P)
    ref = locref.Icp(method=locref.P2P)
    ref.set_target(big)
    rows.append(dict(row="D1a: ICP P2P (lio_mapping default), one 115200-pt scan per call vs the 10M-pt map", **latency(lambda c: ctx.icp_align(c[0], c[1], p2p), lambda c: ref.align(c[0], c[1]), full)))
    del ref
    nd = api.ndt_opts(voxel_size=1.2, nearby_type=api.NEARBY6)
    t0 = time.perf_counter()
    ctx.ndt_set_target(big, nd)
    t_ing = time.perf_counter() - t0
    ref = locref.Ndt(voxel_size=1.2, nearby_type=locref.NEARBY6)
    t0 = time.perf_counter()
    ref.set_target(big)
    t_ing_cpu = time.perf_counter() - t0
    r = latency(lambda c: ctx.ndt_align(c[0], c[1]), lambda c: ref.align(c[0], c[1]), full)
    r.update(target_ingest_ms=round(1e3 * t_ing, 1), cpu_target_ingest_ms=round(1e3 * t_ing_cpu, 1), voxels=ctx.ndt_target_info()["num_voxels"])
    rows.append(dict(row="D3a: direct NDT / NEARBY6 / voxel 1.2 (localisation default), one 115200-pt scan per call vs the 10M-pt map", **r))
    del ref

    # ---- D2a / D2b: incremental NDT / CENTER vs a local map (the mapping node matches filtered scans against its keyframe map)
    sid = 3
    local = synth.make_local_map(400000, sid, half=40.0)
    scan = synth.make_scan(sid, crop_half=36.0)
    xyzi = np.zeros((len(scan), 4), np.float32)
    xyzi[:, :3] = scan[:, :3]
    filt = np.ascontiguousarray(api.Cloud(ctx, xyzi).voxel_filter(0.5).download()[:, :3])
    _, init = synth.make_pose(sid)
    rng = np.random.default_rng(1)
    cases = []
    for i in range(n_lat):
        ip = np.array(init)
        ip[4:] += rng.uniform(-0.1, 0.1, 3)
        cases.append((filt, ip))
    inc = api.ndt_opts(method=api.INCREMENTAL_NDT, nearby_type=api.CENTER)
    ctx.ndt_set_target(local, inc)
    ref = locref.Ndt(method=locref.INCREMENTAL_NDT, nearby_type=locref.CENTER)
    ref.set_target(local)
    r = latency(lambda c: ctx.ndt_align(c[0], c[1]), lambda c: ref.align(c[0], c[1]), cases, cpu_cases=4)
    r.update(scan_points=len(filt), local_map_points=len(local), voxels=ctx.ndt_target_info()["num_voxels"])
    rows.append(dict(row="D2a: incremental NDT / CENTER (lio_mapping's NDT option), one voxel-filtered scan per call vs a 400k-pt local map", **r))
    nb = 16 if a.quick else 64
    scans_b = [scan[i % 3::3] for i in range(nb)]  # thirds of the cropped scan: ~20 k points each
    inits_b = np.stack([init] * nb)
    inits_b[:, 4:] += rng.uniform(-0.1, 0.1, (nb, 3))
    b = ctx.batch(scans_b)
    ctx.ndt_align_batch(b, inits_b)
    reps = 3 if a.quick else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        poses, st = ctx.ndt_align_batch(b, inits_b)
    t_gpu = (time.perf_counter() - t0) / reps
    b.close()
    t0 = time.perf_counter()
    n_cpu = 4
    worst = 0.0
    for i in range(n_cpu):
        rr = ref.align(scans_b[i], inits_b[i])
        worst = max(worst, float(np.abs(rr["pose"] - poses[i]).max()))
        assert rr["iters"] == st[i]["iterations"]
    t_cpu = (time.perf_counter() - t0) / n_cpu
    rows.append(dict(row="D2b: incremental NDT / CENTER, %d scans per step (%d points each) vs the 400k-pt local map" % (nb, len(scans_b[0])), gpu_scans_s=round(nb / t_gpu, 1),
                     ms_per_step=round(1e3 * t_gpu, 3), gn_iterations_per_scan=float(np.mean([s["iterations"] for s in st])), cpu_r1_scans_s=round(1.0 / t_cpu, 3),
                     pose_delta=worst, gpu_over_cpu=round(nb / t_gpu * t_cpu, 1)))
    del ref

    # ---- D1c / D2c: the streaming loop (configs[4]) with the default matchers; the P2Plane loop beside them for scale
    n_stream = 20 if a.quick else 40
    for matcher, label in (("p2p", "D1c: streaming loop (upload, filters, keyframes, target re-ingest) with ICP P2P as the matcher"),
                           ("ndt_inc_center", "D2c: streaming loop with incremental NDT / CENTER as the matcher"),
                           ("p2plane", "(for scale) the same loop with P2Plane, the matcher of BASELINE.md row 5")):
        # a fresh context per pass: the incremental voxel set persists across SetInputTarget calls (ndt cpp:150-183), and every pass
        # (like the oracle beside the checked one) must start from an empty set
        res = {}
        for tag, n, check in (("warm", 10, False), ("timed", n_stream, False), ("checked", n_stream, True)):
            c2 = api.Context(0)
            if tag != "warm":
                pm.stream_section(c2, locref, 3, 5, 10, 0.5, 0.5, check=False, matcher="p2plane")  # buffers and kernels warm, no NDT state
            res[tag] = pm.stream_section(c2, locref, n, 5, 10, 0.5, 0.5, check=check, async_target=(matcher != "ndt_inc_center" and not check), matcher=matcher)
            c2.close()
        timed, checked = res["timed"], res["checked"]
        rows.append(dict(row=label, gpu_scans_s=round(timed["scans_per_s"], 1), ms_per_scan={k: round(v, 4) for k, v in timed["ms_per_scan"].items()},
                         local_map_points=timed["local_map_points"], pose_delta=checked["max_pose_abs_diff"]))
    ctx.close()
    del big

    # ---- D1b / D3b: 256 scans per step vs the 10 M-pt map through bench.py (CPU R1 inside)
    rows.append(bench_row("D1b: ICP P2P, 256 scans per step vs the 10M-pt map", run_bench("--method", "p2p", "--traffic", "none", *steps)))
    rows.append(bench_row("D3b: direct NDT / NEARBY6 / voxel 1.2, 256 scans per step vs the 10M-pt map", run_bench("--method", "ndt", "--ndt-voxel", "1.2", "--traffic", "none", *steps)))
    rows.append(bench_row("(for scale) direct NDT / NEARBY6 / voxel 1.0 (BASELINE.md row 3b at 256 scans per step)", run_bench("--method", "ndt", "--traffic", "none", *steps)))

    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(rows=rows, usable_cores=len(os.sched_getaffinity(0))), open(a.out, "w"), indent=1)
    f = lambda v, p="%.4g": "—" if v is None else p % v
    print("| row | GPU scans/s | GPU ms per call or step | GN iterations per scan | CPU R1 scans/s (1 thread) | GPU / CPU | pose Δ vs oracle |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        ms = r.get("gpu_ms_per_scan", r.get("ms_per_step"))
        if ms is None and "ms_per_scan" in r:
            ms = sum(r["ms_per_scan"].values())
        print("| %s | %s | %s | %s | %s | %s | %s |" % (r["row"], f(r.get("gpu_scans_s")), f(ms), f(r.get("gn_iterations_per_scan")), f(r.get("cpu_r1_scans_s")), f(r.get("gpu_over_cpu")),
                                                       f(r.get("pose_delta", r.get("pose_delta_m")), "%.1e")))


if __name__ == "__main__":
    main()
