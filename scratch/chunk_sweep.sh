#!/bin/bash
# host-witness entry (p2gpu_prove): lone proof and throughput vs the number of rate blocks per upload chunk
for b in 2 3 4 5 2 3 5; do
  export P2GPU_CHUNK_BLOCKS=$b
  python bench.py --steps 24 --warmup 4 --no-cpu-baseline --pipelined 0 --profile-steps 1 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); h=d['host_witness']
print('blocks per chunk $b: host lone', round(h['ms_per_proof'],2), 'ms, in flight', round(h['proofs_per_sec_in_flight'],1), 'proofs/s; sparse', round(h['sparse']['ms_per_proof'],2), round(h['sparse']['proofs_per_sec_in_flight'],1))"
done
