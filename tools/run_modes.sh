timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for m in tree_exact grid; do
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --traffic live --search $m --resident 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$m 256', d['value'], d['kernel_ms_per_step'], 'frac', r['frac'], 'hbm_frac', r['hbm_frac'], 'traffic', r['traffic'], r['kernel'])"
python bench.py --steps 10 --warmup 1 --no-cpu-baseline --traffic none --search $m --resident --scans-per-gpu 64 --pipeline 1 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m 64', d['value'], d['kernel_ms_per_step'])"
done
