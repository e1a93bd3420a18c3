#!/bin/bash
# A/B of library variants on the bench workload + a quick parity check of each: scratch/ab.sh <variant> [...]  ("" = default build)
for v in "$@"; do
  if [ -n "$v" ] && [ "$v" != default ]; then export P2GPU_LIBRARY=$PWD/acvm-backend-plonky2_amd/csrc/build_alt/libp2gpu_$v.so; else unset P2GPU_LIBRARY; fi
  echo "== variant '${v:-default}'"
  python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "field_prim or ifft or deep or roundtrip or proof_bytes" 2>&1 | tail -2
  python bench.py --steps 10 --warmup 3 --in-flight 1 --no-cpu-baseline --pipelined 0 --profile-steps 3 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); k=d['kernel_ms_per_proof']; print(round(d['ms_per_step'],3), 'ms/proof; LDE', k.get('ntt_pass_kernel<1, false, 12>'), 'iNTT', k.get('ntt_pass_kernel<0, true, 12>'), 'host', d['host_witness'] and round(d['host_witness']['ms_per_proof'],2))"
done
