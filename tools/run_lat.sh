mkdir -p gpurun_out/r3d
python tools/latency_microbench.py 2>/dev/null | tail -1 > gpurun_out/r3d/lat.json; cat gpurun_out/r3d/lat.json
LOCGPU_WALK=0 python tools/latency_microbench.py 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
LAT_ONLY=p2plane_eager rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3d/prof -o lat -- python3 $GRAFT_REPO_ROOT/tools/latency_microbench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/r3d/prof | head; head -12 gpurun_out/r3d/prof/lat_kernel_stats.csv | cut -c1-200
