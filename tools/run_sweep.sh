export LOCGPU_WALK=1
for pad in 0 1280 2560 5120; do
echo "== pad $pad"
LOCGPU_LDS_PAD=$pad python tools/search_microbench.py --scans 64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('micro', d['search_ms'], 'deep', d['walk_frac'], 'redo', d['redo_frac'], 'same', d['hb_identical_to_default'])"
LOCGPU_LDS_PAD=$pad python bench.py --steps 10 --warmup 2 --resident --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['kernel_ms_per_step'])"
done
