for rep in 1 2; do
for cfg in "A 2" "B 3"; do set -- $cfg
for sc in 32 64; do
LOCGPU_LIB=build_variants/liblocgpu_$1.so python bench.py --total-scans $sc --scaling strong --steps 40 --warmup 4 --no-cpu-baseline --traffic none --pipeline $2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 depth $2 total $sc strong', d['value'])"
done; done
done
