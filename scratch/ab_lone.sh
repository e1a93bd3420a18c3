#!/bin/bash
# lone-proof latency A/B of library variants on ONE box, alternating: scratch/ab_lone.sh <variant> [...]  ("default" = in-tree build)
for rep in 1 2 3; do
for v in "$@"; do
  if [ "$v" != default ]; then export P2GPU_LIBRARY=$PWD/acvm-backend-plonky2_amd/csrc/build_alt/libp2gpu_$v.so; else unset P2GPU_LIBRARY; fi
  P2GPU_HOSTPROF=1 python bench.py --steps 8 --warmup 3 --in-flight 1 --timed-only --clock-warmup-ms 0 --no-cpu-baseline 2>&1 | grep hostprof | tail -4 | awk -v v=$v '{print v, $NF, $(NF-1)}' | tr '\n' ';'
  echo
done
done
