#!/bin/bash
# sweep of the Merkle-tail switch-over (nodes per coset) on the bench workload
for m in 4096 2048 1024 512 256 128; do
  echo -n "P2GPU_TAIL_NODES=$m: "
  P2GPU_TAIL_NODES=$m python bench.py --steps 10 --warmup 3 --in-flight 1 --no-cpu-baseline --pipelined 0 --profile-steps 1 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],3), 'ms/proof')"
done
