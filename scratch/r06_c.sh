#!/bin/bash
# round 6: half-domain gate evaluation -- parity, then the heavy gate mixes
T=r06; O=gpurun_out/${T}_c; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "proof_bytes_match or gate_vectors or baseline_size or half_domain or fuzz or measurement_switches" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
b() { name=$1; shift; python bench.py --detail $O/bench$name.json "$@" 2> $O/bench$name.err | grep "^{" | tail -1 > $O/line$name.json; tail -2 $O/bench$name.err; }
b _d17_ecdsa --mix ecdsa --no-cpu-baseline --no-cold-process
b _d19_ecdsa --degree-bits 19 --mix ecdsa --no-cpu-baseline --no-cold-process --steps 8 --warmup 4 --pipelined 0
python - <<'PY'
import json
for n in ("_d17_ecdsa","_d19_ecdsa"):
    d=json.load(open(f"gpurun_out/r06_c/line{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["latency_ms_single_proof"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"])
    b=json.load(open(f"gpurun_out/r06_c/bench{n}.json"))
    for k,v in sorted(b["kernel_ms_per_proof_lone"].items(), key=lambda x:-x[1])[:8]: print("    %-50s %.3f"%(k[:50],v))
PY
