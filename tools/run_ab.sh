# A/B of two builds of the library in one session on one box: build_variants/liblocgpu_{A,B}.so
for rep in 1 2; do
for v in ${VARIANTS:-A B}; do
LOCGPU_LIB=build_variants/liblocgpu_$v.so python bench.py --steps 10 --warmup 2 --resident --no-cpu-baseline --traffic none --pipeline 1 ${BENCH_ARGS} 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['kernel_ms_per_step'])"
done
done
