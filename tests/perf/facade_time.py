#!/usr/bin/env python3
"""The call slam_demo makes, timed end to end: `match_ptr_->ScanMatch(host CloudPtr, predict, fresh output CloudPtr,
pose)` through the C++ façade (tests/cpp/facade_scanmatch `time` mode: loc.cpp:215,229) beside the C ABI's host-pointer alignment
without the output cloud, and the CPU oracle (R1, single thread, alignment only) on the same scans.

Two workloads: the bench's (115 200-pt scans vs the 10 M-pt map) and the streaming loop's (a voxel-filtered ≈12 k-pt scan vs a ≈400 k-pt
local map), each with IcpRegistration (P2Plane) and NdtRegistration (direct NDT, NEARBY6, voxel 1.0).

    python3 tests/perf/facade_time.py --out gpurun_out/facade_time.json
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "facade_time.json"))
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--scans", type=int, default=12)
    ap.add_argument("--cpu-scans", type=int, default=3)
    a = ap.parse_args()
    from loc_lib_amd import synth
    from oracle import locref  # reported baseline only
    exe = os.path.join(ROOT, "tests", "cpp", "facade_scanmatch")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "loc_lib_amd", "host"), "../../tests/cpp/facade_scanmatch"], stdout=subprocess.DEVNULL)
    rows = []
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        def workload(name, m, scans, inits):
            mp, sp, pp = (os.path.join(d, x) for x in ("map.bin", "scans.bin", "poses.bin"))
            np.ascontiguousarray(m[:, :3], np.float32).tofile(mp)
            np.concatenate([np.ascontiguousarray(s[:, :3], np.float32) for s in scans]).tofile(sp)
            np.ascontiguousarray(inits, np.float64).tofile(pp)
            for kind, method in (("icp", 2), ("ndt", 0)):
                out = subprocess.run([exe, "time", kind, str(method), mp, sp, str(len(scans)), pp, str(a.reps)], capture_output=True, text=True, timeout=900)
                if out.returncode != 0:
                    raise RuntimeError("facade_scanmatch time failed (%d): %s" % (out.returncode, out.stderr[-400:]))
                r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
                r["workload"] = name
                r["output_cloud_cost_ms"] = round(r["facade_scanmatch_ms_median"] - r["abi_align_host_pointer_ms_median"], 4)
                # CPU R1: the oracle's alignment alone (no output cloud), one thread
                cpu = locref.Icp(method=locref.P2PLANE) if kind == "icp" else locref.Ndt()
                cpu.set_target(m)
                t = 0.0
                for s, ip in list(zip(scans, inits))[:a.cpu_scans]:
                    t0 = time.perf_counter()
                    cpu.align(s, ip)
                    t += time.perf_counter() - t0
                r["cpu_r1_align_ms"] = round(1e3 * t / a.cpu_scans, 2)
                rows.append(r)
                print(json.dumps(r), flush=True)

        n = a.scans
        scans = [synth.make_scan(s) for s in range(n)]
        inits = np.stack([synth.make_pose(s)[1] for s in range(n)])
        workload("115200-pt scan vs 10M-pt map", synth.make_map(10_000_000), scans, inits)
        # the streaming loop's sizes: a 0.5 m voxel-filtered scan against a local map of ten keyframes
        m = synth.make_local_map(400_000, 3, half=60.0)
        def filtered(sid):
            sc = synth.make_scan(sid, crop_half=55.0)
            xyzi = np.zeros((len(sc), 4), np.float32)
            xyzi[:, :3] = sc[:, :3]
            return locref.voxel_grid(xyzi, True, 0.5, order=locref.SORT_STABLE)[:, :3]

        small = [filtered(s) for s in range(n)]
        k = min(len(s) for s in small)
        small = [np.ascontiguousarray(s[:k]) for s in small]  # equal sizes: the driver reads the scans back to back
        workload("%d-pt voxel-filtered scan vs 400k-pt local map" % k, m, small, inits)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(rows=rows), open(a.out, "w"), indent=1)
    print("| workload | matcher | façade ScanMatch, host cloud in → host cloud + pose out (ms, median) | C ABI host-pointer align, no cloud (ms) | output cloud costs (ms) | CPU R1 align (ms) |")
    print("|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %.3f | %.3f | %.3f | %.1f |" % (r["workload"], "IcpRegistration P2Plane" if r["kind"] == "icp" else "NdtRegistration direct", r["facade_scanmatch_ms_median"],
                                                         r["abi_align_host_pointer_ms_median"], r["output_cloud_cost_ms"], r["cpu_r1_align_ms"]))


if __name__ == "__main__":
    main()
