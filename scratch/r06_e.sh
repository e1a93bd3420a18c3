#!/bin/bash
O=gpurun_out/r06_e; mkdir -p $O
b() { name=$1; shift; python bench.py --detail $O/bench$name.json "$@" 2> $O/bench$name.err | grep "^{" | tail -1 > $O/line$name.json; tail -1 $O/bench$name.err; }
for f in 4 6 8; do b _if$f --in-flight $f --no-cpu-baseline --no-cold-process; done
python - <<'PY'
import json
for n in ("_if4","_if6","_if8"):
    d=json.load(open(f"gpurun_out/r06_e/line{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["latency_ms_single_proof"], d["latency_ms_single_proof_host_witness"], d["value_host_witness"])
PY
