# A/B of builds of the library on the headline workload (kernel times from the profiled steps, throughput with three in flight):
#   bash tools/ab_bench.sh "<lib1.so> <lib2.so> ..." [extra bench args]      ("" = the in-tree build)
libs=$1; shift
for rep in 1 2; do for lib in "" $libs; do
  LOCGPU_LIB=$lib python bench.py --traffic none --no-cpu-baseline --steps 20 --pool-slots 0 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/ab.json") if l.startswith("{")][-1])
print("lib=${lib:-default}", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["icp_iter_ms_per_scan"])
PY
done; done
