#!/bin/bash
# round 6, last pass: the whole GPU suite, then the configurations touched since scratch/configs.sh ran
O=gpurun_out/r06_cfg; mkdir -p $O gpurun_out/r06_full
python -m pytest tests -m gpu -x -q > gpurun_out/r06_full/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r06_full/pytest.log | tail -1
b() { name=$1; shift; python bench.py --detail $O/bench$name.json "$@" 2> $O/bench$name.err | grep "^{" | tail -1 > $O/line$name.json; }
b ""
b _pi4 --public-inputs 4 --no-cpu-baseline --no-cold-process
b _d17_ecdsa --mix ecdsa --no-cpu-baseline --no-cold-process
b _d13_arith --degree-bits 13 --mix arith --no-cpu-baseline --no-cold-process --steps 64
python - <<'PY'
import json
for n in ("","_pi4","_d17_ecdsa","_d13_arith"):
    d=json.load(open(f"gpurun_out/r06_cfg/line{n}.json")); print(n, d["value"], d["ms_per_step"], d["latency_ms_single_proof"], d.get("value_host_witness"))
PY
