B="python bench.py --traffic none --no-cpu-baseline"
for args in "--steps 40" "--steps 40 --pool-slots 768" "--steps 40 --pool-slots 1024" "--scaling strong --total-scans 32 --steps 160" "--scaling strong --total-scans 64 --steps 80"; do
  $B $args > gpurun_out/sw.json 2> gpurun_out/sw.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/sw.json") if l.startswith("{")][-1])
    print("$args:", d["value"], d["ms_per_step"], d["icp_iter_ms_per_scan"], d["kernel_ms_per_step"], d.get("pool"))
except Exception as e:
    print("ERR $args", e); print(open("gpurun_out/sw.err").read()[-800:])
PY
done
python -m pytest tests/test_gpu_pool.py tests/test_gpu_world2.py -q -m gpu 2>&1 | grep -E "passed|failed"
