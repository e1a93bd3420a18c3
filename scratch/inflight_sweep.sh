#!/bin/bash
# throughput of the timed region for several numbers of proofs in flight: bash scratch/inflight_sweep.sh [bench flags]
for s in 4 5 6 8 4 3 2 1; do
  python bench.py --in-flight $s --steps 96 --warmup 8 --timed-only "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('in flight', $s, round(d['value'],1), 'proofs/s', round(d['ms_per_step'],3), 'ms/proof')"
done
