#!/bin/bash
# A/B of an environment knob on one box, alternating: scratch/ab_env.sh NAME v1 v2 ...   (bench flags in P2GPU_BENCH_FLAGS)
name=$1; shift
for rep in 1 2; do
for v in "$@"; do
  env $name=$v python bench.py --steps 48 --warmup 8 --no-cpu-baseline --pipelined 0 --profile-steps 3 $P2GPU_BENCH_FLAGS 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); k=d['kernel_ms_per_proof']
print('$name=$v', round(d['value'],1), 'proofs/s;', round(d['latency_ms_single_proof'],3), 'ms lone;', {n: v for n, v in k.items() if 'hash_lde_leaves_kf_kernel<true>' in n or 'ntt_pass_kernel<1, false, 12>' in n or 'quotient_kernel' in n}, 'sum', round(sum(k.values()),3))"
done
done
