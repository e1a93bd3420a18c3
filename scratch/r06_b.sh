#!/bin/bash
# round 6: fused Poseidon layers -- parity, then the two workloads that use them
T=r06; O=gpurun_out/${T}_b; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "poseidon or public_inputs or gate_vectors or reference_proofs or translate or verifier or baseline_size" 2>&1 | tail -5
b() { name=$1; shift; python bench.py --detail $O/bench$name.json "$@" 2> $O/bench$name.err | grep "^{" | tail -1 > $O/line$name.json; tail -2 $O/bench$name.err; }
b _pi4 --public-inputs 4 --no-cpu-baseline --no-cold-process
b _poseidon --hasher poseidon --no-cpu-baseline --no-cold-process --steps 12 --warmup 3
python - <<'PY'
import json
for n in ("_pi4","_poseidon"):
    d=json.load(open(f"gpurun_out/r06_b/line{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["latency_ms_single_proof"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"])
    b=json.load(open(f"gpurun_out/r06_b/bench{n}.json"))
    ks=b.get("kernels") or b.get("kernel_stats") or {}
    print([ (k,v) for k,v in list(ks.items())[:0]])
PY
