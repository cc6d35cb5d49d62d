for rep in 1 2; do
for m in 2 12 13 14; do
LOCGPU_WALK_MODE=$m python bench.py --steps 10 --warmup 2 --resident --no-cpu-baseline --traffic none --pipeline 1 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode $m', d['value'], d['kernel_ms_per_step'], d['median_translation_error_to_truth_m'])"
done
done
LOCGPU_WALK_MODE=13 timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "hot_search" 2>&1 | tail -2
