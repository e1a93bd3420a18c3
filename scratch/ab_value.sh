#!/bin/bash
# throughput (`value`, four proofs in flight) and lone latency of library variants on one box, alternating:
#   scratch/ab_value.sh <variant>[:ENV=V,...] ...      (variant `default` = the shipped library)
for rep in 1 2 3; do
for spec in "$@"; do
  v=${spec%%:*}; envs=""
  if [ "$spec" != "$v" ]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  if [ "$v" != default ]; then export P2GPU_LIBRARY=$PWD/acvm-backend-plonky2_amd/csrc/build_alt/libp2gpu_$v.so; else unset P2GPU_LIBRARY; fi
  a=$(env $envs python bench.py --timed-only --steps 192 --warmup 16 $BENCH_FLAGS 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1))")
  b=$(env $envs python bench.py --timed-only --in-flight 1 --steps 24 --warmup 6 $BENCH_FLAGS 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],3))")
  echo "$spec: $a proofs/s (4 in flight), $b ms lone"
done
done
