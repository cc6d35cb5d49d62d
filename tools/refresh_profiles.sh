# The ONE entry point that refreshes every profiles/r06_* artefact on one box after the last kernel change:
#     gpurun --timeout 3600 -- 'bash tools/refresh_profiles.sh'
# Results land in gpurun_out/final (copy what is to be judged into profiles/ as r06_<name>). Every command has its own timeout: a hung
# kernel must not eat the box. Under rocprofv3 the program itself follows `--` (never a shell or env wrapper).
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
export TMPDIR=/tmp
line() { grep '^{' | tail -1; }
# ---- the bench line as the driver runs it (live counters, CPU baseline), and its variants
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | line > $O/bench_line.json
timeout 300 python bench.py --no-cpu-baseline --traffic none 2>/dev/null | line > $O/bench_default_80steps_line.json
timeout 200 python bench.py --steps 20 --warmup 5 --resident --no-cpu-baseline --traffic none 2>/dev/null | line > $O/bench_resident_line.json
timeout 200 python bench.py --steps 20 --warmup 5 --pipeline 1 --no-cpu-baseline --traffic none 2>/dev/null | line > $O/bench_pipeline1_line.json
# ---- configs[3] on ONE rank (RCCL communicator of one rank): the open-scan pool (default for small steps) against plain sharded batches
for n in 32 64; do
  timeout 300 python bench.py --scaling strong --total-scans $n --steps $((5120 / n)) --warmup 8 --no-cpu-baseline --traffic none 2>/dev/null | line > $O/bench_strong_1rank_${n}_pool_line.json
  timeout 300 python bench.py --scaling strong --total-scans $n --steps $((5120 / n)) --warmup 8 --pool-slots 0 --no-cpu-baseline --traffic none 2>/dev/null | line > $O/bench_strong_1rank_${n}_plain_line.json
done
timeout 300 python bench.py --scaling strong --total-scans 32 --steps 20 --warmup 3 --no-cpu-baseline --traffic none 2>/dev/null | line > $O/bench_strong_1rank_32_pool_20steps_line.json
timeout 300 python bench.py --scaling strong --total-scans 256 --steps 20 --warmup 5 --no-cpu-baseline --traffic none 2>/dev/null | line > $O/bench_strong_1rank_256_line.json
timeout 600 python bench.py --scaling strong --total-scans 32 --steps 160 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/bench_strong_1rank_32_pool_counters_line.json
# ---- SetEnableANN(false): the exact search through the tree and through the cell grid (K1b, frozen), with live counters
timeout 600 python bench.py --search tree_exact --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line > $O/bench_tree_exact_line.json
timeout 600 python bench.py --search grid --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line > $O/bench_grid_line.json
# ---- direct NDT: the line with live counters (row 3b), kernel stats below
timeout 600 python bench.py --method ndt --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line > $O/bench_ndt_line.json
timeout 400 python3 tools/collect_pmc.py --out $O/pmc_ndt_kernel.md --scans 64 --method ndt > /dev/null 2>&1
# ---- the call slam_demo makes (facade ScanMatch: host cloud in, host cloud + pose out), the K1 occupancy point of the shipped shape and of round 5's
timeout 900 python3 tests/perf/facade_time.py --out $O/facade_time.json > $O/facade_time.md 2>/dev/null
(timeout 200 python tools/k1_occupancy.py --bytes 5120,5120; LOCGPU_FAST_STACK=15 timeout 200 python tools/k1_occupancy.py --bytes 7680,7680) > $O/k1_shape_ab.txt 2>&1
# ---- latency, counters, tables, traces
timeout 200 python tools/latency_microbench.py 2>/dev/null | line > $O/latency.json
timeout 400 python3 tools/collect_pmc.py --out $O/pmc_kernels.md --scans 64 > /dev/null 2>&1
timeout 600 python tests/perf/pipeline_microbench.py 2>/dev/null | grep '^{' > $O/pipeline.json   # four JSON lines: filters, inc_ndt, cpp_stream, stream
timeout 1800 python3 tests/perf/baseline_table.py --out $O/baseline_table.json > $O/baseline_table.md 2>/dev/null
timeout 1500 python3 tests/perf/defaults_table.py --out $O/defaults_table.json > $O/defaults_table.md 2>$O/defaults_table.err
timeout 300 python3 tools/iter_trace.py --out $O/iteration_trace.txt > /dev/null 2>&1
timeout 300 python3 tools/single_scan_trace.py --out $O/single_scan_trace.txt > /dev/null 2>&1
timeout 300 python3 tools/single_scan_trace.py --scan 2 --out $O/single_scan_trace_scan2.txt > /dev/null 2>&1   # a scan whose longest traversal is twice scan 11's
timeout 200 python tools/latency_detail.py > $O/latency_detail.txt 2>/dev/null
timeout 300 python3 tools/stream_trace.py --mode 0 --window-us 1500 --out $O/stream_trace_sequential.txt > /dev/null 2>&1
timeout 300 python3 tools/stream_trace.py --mode 1 --window-us 1500 --out $O/stream_trace_two_stage.txt > /dev/null 2>&1
(timeout 120 ./tools/ubench/tree_build_bench 35133 300; LOCGPU_BUILD_TIMES=1 timeout 60 ./tools/ubench/tree_build_bench 35133 3 2>&1 | tail -4; LOCGPU_BUILD_THREADS=1 timeout 120 ./tools/ubench/tree_build_bench 35133 100; timeout 200 ./tools/ubench/tree_build_bench 10000000 3) > $O/tree_build.txt 2>&1
timeout 120 ./tools/ubench/plane_fit_accuracy > $O/plane_fit_accuracy.txt 2>&1
# ---- rocprofv3 kernel statistics of the same commands (the program itself after `--`)
cd /tmp
stats() {  # stats <name> <bench args…>
  name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py "$@" --no-cpu-baseline --traffic none > /dev/null 2>&1
  cp $O/prof/*/bench_kernel_stats.csv $O/${name}_kernel_stats.csv 2>/dev/null || cp $O/prof/bench_kernel_stats.csv $O/${name}_kernel_stats.csv; rm -rf $O/prof
}
stats bench --steps 20 --warmup 5
stats bench_pipeline1 --steps 20 --warmup 5 --pipeline 1   # ONE alignment in flight: launch durations that do not overlap — what roofline.avg_launch_ms must agree with
stats bench_ndt --method ndt --steps 20 --warmup 5 --pipeline 1
stats bench_grid --search grid --steps 10 --warmup 3 --pipeline 1
stats bench_strong_1rank_32_pool --scaling strong --total-scans 32 --steps 160 --warmup 8 --pool-lanes 1
cd $R
# ---- the GPU suite, then parity at length
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 600 python tools/ndt_determinism.py --reps 5000 > $O/ndt_determinism.log 2>&1
timeout 500 python tools/fuzz_search.py --cases 300 --seed 4 > $O/fuzz_search.log 2>&1
timeout 600 python tools/fuzz_ndt.py --cases 900 > $O/fuzz_ndt.log 2>&1
timeout 600 python tools/fuzz_align.py --cases 120 > $O/fuzz_align.log 2>&1
timeout 200 python tools/fuzz_hb.py > $O/fuzz_hb.log 2>&1
timeout 900 python tools/fuzz_batch.py --cases 80 --seed 6 > $O/fuzz_batch.log 2>&1
tail -2 $O/ndt_determinism.log $O/fuzz_search.log $O/fuzz_ndt.log $O/fuzz_align.log $O/fuzz_hb.log $O/fuzz_batch.log
ls -la $O
