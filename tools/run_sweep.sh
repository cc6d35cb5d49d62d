export LOCGPU_WALK_MODE=${MODES:-2}
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "hot_search or stack_depths" 2>&1 | tail -3
for m in 2; do
echo "== mode $m"
LOCGPU_WALK_MODE=$m python tools/search_microbench.py --scans 64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('micro', d['search_ms'], 'deep', d['walk_frac'], 'redo', d['redo_frac'], 'same', d['hb_identical_to_default'])"
LOCGPU_WALK_MODE=$m python bench.py --steps 10 --warmup 2 --resident --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['kernel_ms_per_step'])"
done
