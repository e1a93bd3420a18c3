#!/bin/bash
# quotient kernel time vs the gate-group balance (P2GPU_PERM_COST = weight of the permutation argument in group 0)
for pc in default 400 1200 2000 3000 5000 100000; do
  if [ $pc = default ]; then unset P2GPU_PERM_COST; else export P2GPU_PERM_COST=$pc; fi
  python bench.py --steps 6 --warmup 2 --in-flight 1 --no-cpu-baseline --pipelined 0 --profile-steps 3 --mix ecdsa "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); k=d['kernel_ms_per_proof']; print('perm_cost $pc:', {n: v for n, v in k.items() if 'quotient_kernel' in n}, round(d['latency_ms_single_proof'],3))"
done
