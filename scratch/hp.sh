P2GPU_HOSTPROF=1 python bench.py --steps 6 --warmup 3 --in-flight 1 --timed-only --no-cpu-baseline 2>&1 | grep hostprof | tail -3
