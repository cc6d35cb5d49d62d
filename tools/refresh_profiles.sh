# Refreshes every profiles/r03_* artefact on ONE box after the last kernel change (run through gpurun; results land in gpurun_out/final).
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 3 2>/dev/null | grep '^{' > $O/bench_line.json
python bench.py --steps 20 --warmup 3 --resident --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' > $O/bench_resident_line.json
python bench.py --steps 20 --warmup 3 --pipeline 2 --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' > $O/bench_pipeline2_line.json
python bench.py --scaling strong --steps 10 --warmup 2 --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' > $O/bench_strong_line.json
python tools/latency_microbench.py 2>/dev/null | grep '^{' > $O/latency.json
python3 tools/collect_pmc.py --out $O/pmc_kernels.md --scans 64 > /dev/null 2>&1
python3 tools/collect_traffic.py --out $O/traffic.json > /dev/null 2>&1
python tests/perf/pipeline_microbench.py 2>/dev/null | grep '^{' > $O/pipeline.json
python3 tests/perf/baseline_table.py --out $O/baseline_table.json > $O/baseline_table.md 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --traffic none > /dev/null 2>&1
cp $O/prof/bench_kernel_stats.csv $O/bench_kernel_stats.csv; rm -rf $O/prof
ls -la $O
