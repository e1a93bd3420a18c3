#!/bin/bash
# A/B of environment switches on the bench workload: for each "NAME=VALUE" argument (and the default) print value, lone latency and
# the top lone kernel times.   bash scratch/ab_env.sh [--args "bench args"] VAR=val [VAR=val ...]
ARGS=""
if [ "$1" == "--args" ]; then ARGS="$2"; shift 2; fi
run() {
  env $1 python bench.py --no-cpu-baseline --no-cold-process --pipelined 0 --detail /tmp/ab_detail.json $ARGS > /dev/null 2>&1
  python - "$1" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_detail.json"))
ks = d["kernel_ms_per_proof_lone"]
lde = sum(v for k, v in ks.items() if k.startswith(("ntt_dit", "ntt_pass_kernel<1")))
intt = sum(v for k, v in ks.items() if k.startswith(("ntt_dif", "ntt_pass_kernel<0")))
print(f"{sys.argv[1]:28s} value {d['value']:7.1f}  lone {d['latency_ms_single_proof']:6.3f} ms  host {d['value_host_witness']:6.1f}  LDE {lde:.3f} iNTT {intt:.3f}  " +
      "  ".join(f"{k.split('(')[0][:34]} {v:.3f}" for k, v in list(ks.items())[:6]))
PY
}
run "P2GPU_NOP=1"
for kv in "$@"; do run "$kv"; done
