mkdir -p gpurun_out/r3b
export LOCGPU_WALK=1
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "hot_search or stack_depths" > gpurun_out/r3b/pytest_walk.txt 2>&1
tail -5 gpurun_out/r3b/pytest_walk.txt
for w in 0 1; do
LOCGPU_WALK=$w python tools/search_microbench.py --scans 64 2>/dev/null | tail -1 > gpurun_out/r3b/micro_w$w.json
LOCGPU_WALK=$w python bench.py --steps 10 --warmup 2 --resident --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' > gpurun_out/r3b/bench_w$w.json
done
cat gpurun_out/r3b/micro_w*.json
python - <<'PY'
import json
for w in (0,1):
    d=json.load(open('gpurun_out/r3b/bench_w%d.json'%w)); print(w, d['value'], d['kernel_ms_per_step'], d['median_translation_error_to_truth_m'])
PY
