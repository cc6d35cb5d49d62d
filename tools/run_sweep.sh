for rep in 1 2; do
for v in 0 1; do
LOCGPU_ACTIVE_LIST=$v python bench.py --steps 10 --warmup 2 --resident --no-cpu-baseline --traffic none --pipeline 1 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('active list $v: 256 resident', d['value'], d['kernel_ms_per_step'])"
LOCGPU_ACTIVE_LIST=$v python bench.py --scans-per-gpu 32 --steps 40 --warmup 4 --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('active list $v: 32 h2d depth 3', d['value'])"
LOCGPU_ACTIVE_LIST=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('active list $v: 256 h2d', d['value'])"
done; done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
