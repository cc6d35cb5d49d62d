# Refreshes every profiles/r03_* artefact on ONE box after the last kernel change (run through gpurun; results land in gpurun_out/final).
# Every command has its own timeout: a hung kernel must not eat the box.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | grep '^{' > $O/bench_line.json
timeout 200 python bench.py --steps 20 --warmup 3 --resident --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' > $O/bench_resident_line.json
timeout 200 python bench.py --steps 20 --warmup 3 --pipeline 2 --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' > $O/bench_pipeline2_line.json
timeout 200 python bench.py --steps 20 --warmup 3 --pipeline 3 --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' > $O/bench_pipeline3_line.json
timeout 200 python bench.py --scaling strong --steps 10 --warmup 2 --no-cpu-baseline --traffic none 2>/dev/null | grep '^{' > $O/bench_strong_line.json
timeout 200 python tools/latency_microbench.py 2>/dev/null | grep '^{' > $O/latency.json
timeout 400 python3 tools/collect_pmc.py --out $O/pmc_kernels.md --scans 64 > /dev/null 2>&1
timeout 400 python3 tools/collect_traffic.py --out $O/traffic.json > /dev/null 2>&1
timeout 300 python tests/perf/pipeline_microbench.py 2>/dev/null | grep '^{' > $O/pipeline.json
timeout 1200 python3 tests/perf/baseline_table.py --out $O/baseline_table.json > $O/baseline_table.md 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --traffic none > /dev/null 2>&1
cp $O/prof/bench_kernel_stats.csv $O/bench_kernel_stats.csv; rm -rf $O/prof
cd $R && timeout 200 bash tools/debug/iter_trace.sh > $O/iteration_trace.txt 2>&1
ls -la $O
