#!/usr/bin/env python3
"""Kernel timeline of ONE single-scan ScanMatch (run ON the GPU box): tools/latency_microbench.py (one leg: LAT_ONLY, default
p2plane_eager) under `rocprofv3 --kernel-trace`, then the dispatches of the last alignment with their durations and the gaps in
front of them — where the 0.6-0.8 ms of a one-scan call go (kernels vs launch gaps vs the host's read-backs between chunks).

    python3 tools/single_scan_trace.py [--leg p2plane_eager] [--out profiles/r04_single_scan_trace.txt]
"""
import argparse
import csv
import glob
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", default="p2plane_eager")
    ap.add_argument("--out", default=None)
    ap.add_argument("--scan", type=int, default=11, help="which synthetic scan (their traversals differ: tools/latency_detail.py)")
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="locgpu_strace_", dir="/tmp")
    try:
        subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", "python3", os.path.join(ROOT, "tools", "latency_microbench.py")],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp", LAT_ONLY=a.leg, LAT_SCANS=str(a.scan)), cwd="/tmp", timeout=900, check=True)
        f = glob.glob(os.path.join(d, "**", "t_kernel_trace.csv"), recursive=True)[0]
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    # the last alignment = the dispatches behind the last gap of more than 150 us (the host works between calls)
    start = 0
    for i in range(1, len(rows)):
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 150_000:
            start = i
    sel = rows[start:]
    t0 = int(sel[0]["Start_Timestamp"])
    out, busy, gaps, prev_end = [], 0.0, 0.0, None
    per = {}
    for r in sel:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("locgpu::", "")[:70]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gap = 0.0 if prev_end is None else (int(r["Start_Timestamp"]) - prev_end) / 1e3
        prev_end = int(r["End_Timestamp"])
        busy += dur
        gaps += max(gap, 0.0)
        k = name.split("<")[0]
        per[k] = (per.get(k, (0, 0.0))[0] + 1, per.get(k, (0, 0.0))[1] + dur)
        out.append("%9.1f us  gap %6.1f  run %7.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, gap, dur, name))
    out.append("one call: %d dispatches over %.1f us: kernels %.1f us, gaps %.1f us" % (len(sel), (prev_end - t0) / 1e3, busy, gaps))
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        out.append("   %-60s x%-3d %8.1f us" % (k, n, t))
    text = "\n".join(out)
    print(text)
    if a.out:
        open(a.out, "w").write("# tools/single_scan_trace.py --leg %s: the last single-scan call of tools/latency_microbench.py (115 200-pt scan %d vs the 10 M-pt map)\n%s\n" % (a.leg, a.scan, text))


if __name__ == "__main__":
    main()
