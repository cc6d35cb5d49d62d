timeout 600 python tools/fuzz_search.py --cases 240 --seed 21 2>&1 | tail -4
LOCGPU_FAST_STACK=12 timeout 600 python tools/fuzz_search.py --cases 240 --seed 22 2>&1 | tail -4
LOCGPU_FAST_STACK=12 LOCGPU_WALK_MODE=0 timeout 600 python tools/fuzz_search.py --cases 120 --seed 23 2>&1 | tail -4
LOCGPU_FAST_STACK=12 LOCGPU_WALK_MODE=2 timeout 600 python tools/fuzz_search.py --cases 120 --seed 24 2>&1 | tail -4
LOCGPU_FAST_STACK=12 timeout 600 python tools/debug/lines_overflow_scan.py 2>&1 | grep -v "differing 0" | tail -5
timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -x -k "hot_search or other_stack" 2>&1 | tail -3
VARIANTS="A B" bash tools/run_ab.sh
