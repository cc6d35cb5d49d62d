for rep in 1 2; do
for g in 1024 2048 2304 4608; do
LOCGPU_DEEP_GRID=$g python bench.py --steps 10 --warmup 2 --resident --no-cpu-baseline --traffic none --pipeline 1 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deep grid $g', d['value'], d['kernel_ms_per_step'])"
done
done
