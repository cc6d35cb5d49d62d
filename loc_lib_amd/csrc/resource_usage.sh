#!/bin/bash
# Prints name / VGPRs / scratch / LDS / occupancy of every kernel in a .hip file (compiler view).
f=${1:-icp_kernels.hip}
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize $( [ "$(basename "$f")" = icp_kernels.hip ] && echo -mllvm -amdgpu-sched-strategy=max-ilp ) -fPIC -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | python3 -c '
import sys,re
cur={}
for line in sys.stdin:
    m=re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        cur={"name":t.split(":",1)[1].strip()}
    elif ":" in t:
        k,v=t.split(":",1); cur[k.strip()]=v.strip()
        if k.strip().startswith("LDS Size"):
            import subprocess
            n=subprocess.run(["c++filt",cur["name"]],capture_output=True,text=True).stdout.strip().split("(")[0]
            print("%-60s VGPR %4s AGPR %3s SGPR %3s scratch %5s occ %2s LDS %6s"%(n[:60],cur.get("VGPRs"),cur.get("AGPRs"),cur.get("TotalSGPRs"),cur.get("ScratchSize [bytes/lane]"),cur.get("Occupancy [waves/SIMD]"),cur.get("LDS Size [bytes/block]")))
'
