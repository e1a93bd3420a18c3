#!/bin/bash
# round 6, second session: the whole GPU suite + smoke + the default bench + the multi-rank flows after the bench.py split
O=gpurun_out/r06_final2; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
b() { name=$1; shift; python bench.py --detail $O/bench$name.json "$@" 2> $O/bench$name.err | grep "^{" | tail -1 > $O/line$name.json; }
( time python bench.py --detail $O/bench.json > $O/line.json 2> $O/bench.err ) 2> $O/time.txt; tail -3 $O/time.txt
b _gloo2 --gpus 2 --backend gloo --steps 16 --warmup 4
b _group2_same_gpu --group 0,0 --steps 8 --warmup 2 --no-cpu-baseline
python - <<'PY'
import json
for n in ("","_gloo2","_group2_same_gpu"):
    try:
        d=json.load(open(f"gpurun_out/r06_final2/line{n}.json")); print(n or "default", d["value"], d["ms_per_step"], d.get("latency_ms_single_proof"), d.get("value_host_witness"), d.get("latency_ms_sharded"), d.get("latency_ms_sharded_group"), len(json.dumps(d)))
    except Exception as e: print(n, "ERR", e)
PY
