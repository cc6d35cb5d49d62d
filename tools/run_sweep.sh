timeout 900 python tools/fuzz_search.py --cases 270 --seed 31 2>&1 | tail -12
LOCGPU_FAST_STACK=12 timeout 900 python tools/fuzz_search.py --cases 270 --seed 32 2>&1 | tail -12
