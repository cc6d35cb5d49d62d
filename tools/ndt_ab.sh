# A/B of two builds of the library on the direct-NDT batch workload: bash tools/ndt_ab.sh <other liblocgpu.so>
for lib in "" "$1"; do
  LOCGPU_LIB=$lib python bench.py --method ndt --traffic none --no-cpu-baseline --steps 20 --pool-slots 0 > gpurun_out/ndt_ab.json 2> gpurun_out/ndt_ab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/ndt_ab.json") if l.startswith("{")][-1])
print("lib=${lib:-default}", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["setup_s"], d["gn_iterations_per_scan"])
PY
done
