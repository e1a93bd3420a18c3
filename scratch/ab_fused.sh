#!/bin/bash
# A/B of the fused two-pass transform on one box: scratch/ab_fused.sh [bench flags]
# P2GPU_NTT_FUSED 0 = one launch per pass, 1 = fused, 2 = fused + agent-scope fences; P2GPU_NTT_LAG = units the second pass trails by
run() {
  env "$@" python bench.py --steps 48 --warmup 8 --no-cpu-baseline --pipelined 0 --profile-steps 3 $P2GPU_BENCH_FLAGS 2>gpurun_out/ab_err.log | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); k=d['kernel_ms_per_proof']
print('$*', round(d['value'],1), 'proofs/s;', round(d['latency_ms_single_proof'],3), 'ms lone;', {n: round(v,3) for n, v in k.items() if 'hash_lde_leaves_kf_kernel<true>' in n or 'ntt_' in n or 'quotient_kernel' in n}, 'sum', round(sum(k.values()),3))" || tail -5 gpurun_out/ab_err.log
}
for rep in 1 2; do
  run P2GPU_NTT_FUSED=0
  for lag in $LAGS; do run P2GPU_NTT_FUSED=1 P2GPU_NTT_LAG=$lag; done
  for x in $EXTRA; do run P2GPU_NTT_FUSED=1 $x; done
done
