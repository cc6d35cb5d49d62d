for d in 0 1; do
for pl in 1 2; do
if [ $d = 1 ]; then export LOCGPU_COMM_DIRECT=1; else unset LOCGPU_COMM_DIRECT; fi
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --scaling strong --total-scans 256 --pipeline $pl --resident 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('strong 256 direct=$d pl=$pl', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
python bench.py --steps 40 --warmup 3 --no-cpu-baseline --traffic none --scaling strong --total-scans 32 --pipeline $pl --resident 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('strong 32 direct=$d pl=$pl', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
done
