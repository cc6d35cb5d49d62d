#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel mean of each counter (and per-wave values)."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("locgpu::", "")[:48]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if pat and pat not in k:
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    waves = m.get("SQ_WAVES", 0)
    print(k)
    for c in sorted(m):
        extra = "  (%.1f per wave)" % (m[c] / waves) if waves and c.startswith("SQ_") and c != "SQ_WAVES" else ""
        print("    %-32s %14.4g%s" % (c, m[c], extra))
