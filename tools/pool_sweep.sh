set -x
python -m pytest tests/test_gpu_world2.py tests/test_gpu_pool.py -x -q -m gpu 2>&1 | tail -45
B="python bench.py --traffic none --no-cpu-baseline"
$B --scaling strong --total-scans 32 --steps 160 > gpurun_out/s32_pool.json 2> gpurun_out/s32_pool.err
$B --scaling strong --total-scans 32 --steps 160 --pool-slots 0 > gpurun_out/s32_plain.json 2> gpurun_out/s32_plain.err
$B --scaling strong --total-scans 64 --steps 80 > gpurun_out/s64_pool.json 2> gpurun_out/s64_pool.err
$B --steps 40 > gpurun_out/w256_plain.json 2> gpurun_out/w256_plain.err
$B --steps 40 --pool-slots 768 > gpurun_out/w256_pool768.json 2> gpurun_out/w256_pool768.err
$B --steps 40 --pool-slots 512 > gpurun_out/w256_pool512.json 2> gpurun_out/w256_pool512.err
for f in gpurun_out/s32_pool gpurun_out/s32_plain gpurun_out/s64_pool gpurun_out/w256_plain gpurun_out/w256_pool768 gpurun_out/w256_pool512; do echo $f; python - <<PY
import json
try:
    d = json.loads([l for l in open("$f.json") if l.startswith("{")][-1])
    print(d["value"], d["ms_per_step"], d["icp_iter_ms_per_scan"], d["kernel_ms_per_step"], d.get("pool"), d["gn_iterations_per_scan"])
except Exception as e:
    print("ERR", e); print(open("$f.err").read()[-1500:])
PY
done
