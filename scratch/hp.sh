P2GPU_HOSTPROF=1 python bench.py --steps 6 --warmup 3 --in-flight 1 --timed-only --clock-warmup-ms 0 --no-cpu-baseline 2>&1 | grep hostprof | tail -3
