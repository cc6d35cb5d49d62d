for n in 256 128 64 32; do
for pl in 1 2; do
python bench.py --steps $((2560/n)) --warmup 3 --no-cpu-baseline --traffic none --scans-per-gpu $n --pipeline $pl 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stream   n=$n pl=$pl', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
done
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --resident --pipeline 2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resident n=256 pl=2', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
for n in 32 64 256; do
python bench.py --steps $((1280/n)) --warmup 3 --no-cpu-baseline --traffic none --scaling strong --total-scans $n 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('strong n=$n', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config']['pipeline_depth'])"
done
