#!/bin/bash
# A/B of library variants on the bench workload (throughput with 4 proofs in flight, lone proof, leaf-hash / fill / LDE kernel
# times): scratch/ab_leaf.sh <variant> [...]   ("default" = the in-tree build)
for v in "$@"; do
  if [ -n "$v" ] && [ "$v" != default ]; then export P2GPU_LIBRARY=$PWD/acvm-backend-plonky2_amd/csrc/build_alt/libp2gpu_$v.so; else unset P2GPU_LIBRARY; fi
  echo "== variant '${v:-default}' ${P2GPU_BENCH_FLAGS}"
  python bench.py --steps 32 --warmup 6 --no-cpu-baseline --pipelined 0 --profile-steps 3 $P2GPU_BENCH_FLAGS 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); k=d['kernel_ms_per_proof']
print(round(d['value'],1), 'proofs/s;', round(d['latency_ms_single_proof'],3), 'ms lone; host', d['host_witness'] and round(d['host_witness']['ms_per_proof'],2), d['host_witness'] and round(d['host_witness']['proofs_per_sec_in_flight'],1))
sp=(d['host_witness'] or {}).get('sparse')
if sp: print('sparse entry:', round(sp['ms_per_proof'],2), 'ms lone,', sp['proofs_per_sec_in_flight'] and round(sp['proofs_per_sec_in_flight'],1), 'proofs/s in flight,', sp['dense_columns'], 'columns over PCIe')
print({n: v for n, v in k.items() if 'hash_lde' in n or 'fill' in n or 'ntt_pass_kernel<1, false, 12>' in n or 'quotient_kernel' in n})"
done
