#!/usr/bin/env python3
"""Kernel timeline of the C++ streaming loop (run ON the GPU box): tests/cpp/stream_pipeline (BASELINE configs[4] through the C ABI,
mode 0 = sequential, 1 = two-stage) under `rocprofv3 --kernel-trace`; prints the dispatches of a window in the middle of the last pass
with their durations and the gaps in front of them, and per kernel the count and time of the whole last pass — what one scan of the
loop costs on the GPU and what the GPU waits for.

    python3 tools/stream_trace.py [--mode 0] [--scans 40] [--window-us 2500] [--out profiles/r04_stream_trace_mode0.txt]
"""
import argparse
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--scans", type=int, default=40)
    ap.add_argument("--kf-every", type=int, default=5)
    ap.add_argument("--num-kfs", type=int, default=10)
    ap.add_argument("--leaf", type=float, default=0.5)
    ap.add_argument("--window-us", type=float, default=2500.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from loc_lib_amd import synth
    d = tempfile.mkdtemp(prefix="locgpu_stream_trace_", dir="/tmp")
    try:
        scans = []
        for s in range(a.scans):
            p = synth.make_scan(s)
            scans.append(np.concatenate([p[:, :3], np.zeros((len(p), 1), np.float32)], axis=1))
        scans = np.stack(scans).astype(np.float32)
        poses = np.stack([np.concatenate(synth.make_pose(s)) for s in range(a.scans)]).astype(np.float64)
        f_scans, f_poses, f_out = os.path.join(d, "scans.bin"), os.path.join(d, "poses.bin"), os.path.join(d, "out.bin")
        scans.tofile(f_scans)
        poses.tofile(f_poses)
        exe = os.path.join(ROOT, "tests", "cpp", "stream_pipeline")
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", exe, f_scans, f_poses, str(a.scans), str(scans.shape[1]),
                            str(a.kf_every), str(a.num_kfs), str(a.leaf), str(a.mode), "1", f_out],
                           capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp", STREAM_PIPELINE_TIMES="1"), cwd="/tmp", timeout=900)
        if r.returncode != 0:
            raise SystemExit("stream_pipeline under rocprofv3 failed: " + r.stderr[-800:])
        host = [ln for ln in r.stderr.splitlines() if ln.startswith("per scan [ms]")]
        f = glob.glob(os.path.join(d, "**", "t_kernel_trace.csv"), recursive=True)[0]
        rows = sorted(csv.DictReader(open(f)), key=lambda q: int(q["Start_Timestamp"]))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    # the last pass = the second half of the dispatches (pass 0 is the untimed one, same work)
    rows = rows[len(rows) // 2:]
    t_first, t_last = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
    mid = (t_first + t_last) // 2
    out = ["# tools/stream_trace.py --mode %d: %d scans, last pass %.1f us on the GPU clock (%.1f us per scan)" % (a.mode, a.scans, (t_last - t_first) / 1e3, (t_last - t_first) / 1e3 / a.scans)]
    if host:
        out.append("# host side, " + host[-1])
    def short(kn):
        return kn.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("locgpu::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rocprim::")

    per, busy = {}, 0.0
    for q in rows:
        name = short(q["Kernel_Name"])
        dur = (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e3
        k = name.split("<")[0]
        per[k] = (per.get(k, (0, 0.0))[0] + 1, per.get(k, (0, 0.0))[1] + dur)
        busy += dur
    out.append("# kernels %.1f us of the pass (sum of durations; two streams may overlap), per scan %.1f us in %.1f dispatches" % (busy, busy / a.scans, len(rows) / a.scans))
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        out.append("#   %-58s x%-5d %9.1f us  (%.1f per scan)" % (k[:58], n, t, t / a.scans))
    prev_end = None
    for q in rows:
        st = int(q["Start_Timestamp"])
        if st < mid or st > mid + a.window_us * 1e3:
            continue
        name = short(q["Kernel_Name"])[:64]
        dur = (int(q["End_Timestamp"]) - st) / 1e3
        gap = 0.0 if prev_end is None else (st - prev_end) / 1e3
        prev_end = max(prev_end or 0, int(q["End_Timestamp"]))
        out.append("%9.1f us  gap %6.1f  run %7.1f  q%-3s %s" % ((st - mid) / 1e3, gap, dur, q.get("Queue_Id", "?"), name))
    text = "\n".join(out)
    print(text)
    if a.out:
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
