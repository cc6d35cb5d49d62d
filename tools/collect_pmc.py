#!/usr/bin/env python3
"""Per-kernel SQ / cache counters of the bench's hot kernels (run ON the GPU box): several `rocprofv3 --pmc` passes (≤ 8 SQ counters
each, nothing but --pmc), summarised per kernel and per wave. Writes a Markdown table.

    python3 tools/collect_pmc.py --out profiles/r02_pmc_kernels.md [--scans 64]
"""
import argparse
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
    ["SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    ["SQ_WAVES", "SQ_THREAD_CYCLES_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INST_LEVEL_VMEM"],
    ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_HIT_sum", "TCC_MISS_sum"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--scans", type=int, default=64)
    ap.add_argument("--method", default="p2plane", help="bench.py --method (ndt: the direct-NDT accumulate kernel)")
    a = ap.parse_args()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    cmd_tail = [os.path.realpath(sys.executable), os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--resident", "--no-cpu-baseline", "--traffic", "none",
                "--scans-per-gpu", str(a.scans), "--method", a.method]
    for counters in PASSES:
        d = tempfile.mkdtemp(prefix="locgpu_pmc_", dir="/tmp")
        try:
            subprocess.run(["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--"] + cmd_tail, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=600, check=False)
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("locgpu::", "")
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    keep = [k for k in agg if any(s in k for s in ("icp_search_walk", "icp_plane_accum_kernel", "icp_point_accum_kernel", "icp_search_redo_kernel", "gn_solve_kernel", "ndt_accum_kernel"))]
    lines = ["# PMC counters of the hot kernels (mean per dispatch; `bench.py --steps 2 --resident --scans-per-gpu %d --method %s`, 10 M-pt map)" % (a.scans, a.method), "",
             "Collected by `tools/collect_pmc.py`: one `rocprofv3 --pmc` pass per counter group, nothing else enabled. SQ_*_CYCLES and SQ_WAIT_* count quad-cycles",
             "(MI355X_MICROARCH.md); per-wave values = counter / SQ_WAVES of the same pass.", ""]
    for k in sorted(keep):
        m = {c: sum(v) / len(v) for c, v in agg[k].items()}
        waves = m.get("SQ_WAVES", 0.0)
        lines += ["## `%s`" % k, "", "| counter | per dispatch | per wave |", "|---|---|---|"]
        for c in sorted(m):
            per_wave = ("%.1f" % (m[c] / waves)) if waves and c.startswith("SQ_") and c != "SQ_WAVES" else ""
            lines.append("| %s | %.4g | %s |" % (c, m[c], per_wave))
        if m.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            lines.append("| L1 hit rate (1 − TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES) | %.3f | |" % (1.0 - m.get("TCP_TCC_READ_REQ_sum", 0.0) / m["TCP_TOTAL_CACHE_ACCESSES_sum"]))
        if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) > 0:
            lines.append("| L2 hit rate (TCC_HIT / (TCC_HIT + TCC_MISS)) | %.3f | |" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
        lines.append("")
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main()
