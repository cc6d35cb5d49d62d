# A longer parity campaign than tools/refresh_profiles.sh runs, with other seeds (≈ 25 GPU-minutes):
#     gpurun --timeout 2700 -- 'bash tools/fuzz_long.sh'        → gpurun_out/fuzz_long/*.log, summary in fuzz_long.log
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fuzz_long; mkdir -p $O; cd $R
timeout 600 python tools/fuzz_search.py --cases 600 --seed 31 --grid > $O/fuzz_search.log 2>&1
for rows in 12 15; do LOCGPU_FAST_STACK=$rows timeout 400 python tools/fuzz_search.py --cases 300 --seed 37 > $O/fuzz_search_rows$rows.log 2>&1; done
timeout 500 python tools/fuzz_ndt.py --cases 600 --seed 41 > $O/fuzz_ndt.log 2>&1
timeout 500 python tools/fuzz_align.py --cases 240 > $O/fuzz_align.log 2>&1
timeout 600 python tools/fuzz_batch.py --cases 120 --seed 29 > $O/fuzz_batch.log 2>&1
timeout 300 python tools/fuzz_filters.py --cases 2000 --seed 9 > $O/fuzz_filters.log 2>&1
timeout 300 python tools/fuzz_loam.py --cases 600 --seed 9 > $O/fuzz_loam.log 2>&1
for f in fuzz_search fuzz_search_rows12 fuzz_search_rows15 fuzz_ndt fuzz_align fuzz_batch fuzz_filters fuzz_loam; do echo "== $f"; tail -n 3 $O/$f.log; done > $O/fuzz_long.log
cat $O/fuzz_long.log
