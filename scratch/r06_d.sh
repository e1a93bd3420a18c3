#!/bin/bash
O=gpurun_out/r06_d; mkdir -p $O
b() { name=$1; shift; python bench.py --detail $O/bench$name.json "$@" 2> $O/bench$name.err | grep "^{" | tail -1 > $O/line$name.json; tail -2 $O/bench$name.err; }
P2GPU_SUMS_GROUPS=1 b _g1 --mix ecdsa --no-cpu-baseline --no-cold-process
b _g4 --mix ecdsa --no-cpu-baseline --no-cold-process
python - <<'PY'
import json
for n in ("_g1","_g4"):
    d=json.load(open(f"gpurun_out/r06_d/line{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["latency_ms_single_proof"])
    b=json.load(open(f"gpurun_out/r06_d/bench{n}.json"))
    for k,v in sorted((b.get("kernel_ms_per_proof_lone") or {}).items(), key=lambda x:-x[1])[:6]: print("    %-50s %.3f"%(k[:50],v))
PY
