#!/usr/bin/env python3
"""Per-iteration kernel durations of one 256-scan step (run ON the GPU box): a short `bench.py` run under `rocprofv3 --kernel-trace`
(one alignment at a time, scans resident), then the dispatches of the last step grouped by Gauss–Newton iteration (a solve kernel
ends one). Environment variables of the library (LOCGPU_*) pass through, so two settings are two calls.

    python3 tools/iter_trace.py [--scans 256] [--out profiles/r04_iteration_trace.txt]
"""
import argparse
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=256)
    ap.add_argument("--out", default=None)
    ap.add_argument("--search", default="tree", help="bench.py --search (tree | tree_exact | grid)")
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="locgpu_itrace_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", "python3", os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
               "--resident", "--pipeline", "1", "--extra-steps", "1", "--no-cpu-baseline", "--traffic", "none", "--scans-per-gpu", str(a.scans), "--search", a.search]
        subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=900, check=True)
        f = glob.glob(os.path.join(d, "**", "t_kernel_trace.csv"), recursive=True)[0]
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    finally:
        shutil.rmtree(d, ignore_errors=True)

    def key(n):
        for k, pat in (("ball", "grid_ball_search"), ("walk", "icp_search_walk_kernel"), ("deep", "icp_search_walk_list"), ("redo", "icp_search_redo"),
                       ("refit", "plane_refit"), ("accum", "accum_kernel"), ("solve", "gn_solve")):
            if pat in n:
                return k
        return "other"

    solves = [i for i, r in enumerate(rows) if "gn_solve" in r["Kernel_Name"]]
    # the last step = the dispatches behind the last solve that follows a gap of "other" work... simply: the last 16 iterations that form one alignment
    # (an alignment's iterations are consecutive; a step runs 8 + 4 + 4 of them at most)
    ends = solves[-16:]
    start = solves[-17] + 1 if len(solves) > 16 else 0
    out = []
    cur, it, t0, prev_end = {}, 0, int(rows[start]["Start_Timestamp"]), None
    tot = {}
    for r in rows[start:ends[-1] + 1]:
        k = key(r["Kernel_Name"])
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        cur[k] = cur.get(k, 0.0) + dur
        tot[k] = tot.get(k, 0.0) + dur
        if k == "solve":
            wall = (int(r["End_Timestamp"]) - t0) / 1e3
            out.append("iter %2d: %s | wall %.0f us" % (it, "  ".join("%s %.0f" % kv for kv in cur.items()), wall))
            it += 1
            cur, t0 = {}, int(r["End_Timestamp"])
    out.append("last 16 iterations, us per kernel class: " + "  ".join("%s %.0f" % kv for kv in tot.items()) + "  | sum %.0f" % sum(tot.values()))
    text = "\n".join(out)
    print(text)
    if a.out:
        open(a.out, "w").write("# tools/iter_trace.py: kernel durations (us) of the last 16 Gauss-Newton iterations of a bench run, %d scans vs the 10 M-pt map; env: %s\n%s\n"
                               % (a.scans, " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("LOCGPU_")), text))


if __name__ == "__main__":
    main()
