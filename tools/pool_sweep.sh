# A/B of the open-scan pool against plain batches on one GPU (rows 3c / 4s of BASELINE.md): bash tools/pool_sweep.sh on the GPU box.
set -x
B="python bench.py --traffic none --no-cpu-baseline"
$B --scaling strong --total-scans 32 --steps 160 > gpurun_out/s32_pool2.json 2> gpurun_out/s32_pool2.err
$B --scaling strong --total-scans 32 --steps 160 --pool-lanes 1 > gpurun_out/s32_pool1.json 2> gpurun_out/s32_pool1.err
$B --scaling strong --total-scans 32 --steps 160 --pool-lanes 2 --pool-slots 512 > gpurun_out/s32_pool2_512.json 2> gpurun_out/s32_pool2_512.err
$B --scaling strong --total-scans 64 --steps 80 > gpurun_out/s64_pool2.json 2> gpurun_out/s64_pool2.err
$B --steps 40 > gpurun_out/w256_plain.json 2> gpurun_out/w256_plain.err
$B --steps 40 --pool-slots 768 > gpurun_out/w256_pool768_2.json 2> gpurun_out/w256_pool768_2.err
$B --steps 40 --pool-slots 1024 > gpurun_out/w256_pool1024_2.json 2> gpurun_out/w256_pool1024_2.err
$B --steps 40 --pool-slots 1024 --pool-lanes 3 > gpurun_out/w256_pool1024_3.json 2> gpurun_out/w256_pool1024_3.err
for f in s32_pool2 s32_pool1 s32_pool2_512 s64_pool2 w256_plain w256_pool768_2 w256_pool1024_2 w256_pool1024_3; do echo $f; python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/$f.json") if l.startswith("{")][-1])
    print(d["value"], d["ms_per_step"], d["icp_iter_ms_per_scan"], d["kernel_ms_per_step"], d.get("pool"), d["gn_iterations_per_scan"])
except Exception as e:
    print("ERR", e); print(open("gpurun_out/$f.err").read()[-1500:])
PY
done
