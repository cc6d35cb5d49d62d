timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for n in 32 64 256; do
for pl in 1 2; do
python bench.py --steps $((2560/n)) --warmup 3 --no-cpu-baseline --traffic none --scans-per-gpu $n --pipeline $pl 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stream   n=$n pl=$pl', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
python bench.py --steps $((2560/n)) --warmup 3 --no-cpu-baseline --traffic none --resident --scans-per-gpu $n --pipeline $pl 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resident n=$n pl=$pl', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
done
done
python bench.py --steps 40 --warmup 3 --no-cpu-baseline --traffic none --scaling strong --total-scans 32 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('strong n=32', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config']['workload'][:60])"
