#!/bin/bash
# the other BASELINE configs on one GPU + variants of the headline config; JSON lines under gpurun_out/r03c_cfg/
mkdir -p gpurun_out/r03c_cfg
python bench.py > gpurun_out/r03c_cfg/bench.json 2> gpurun_out/r03c_cfg/bench.err
python bench.py --public-inputs 4 --no-cpu-baseline > gpurun_out/r03c_cfg/bench_pi4.json 2>/dev/null
python bench.py --mix ecdsa --no-cpu-baseline > gpurun_out/r03c_cfg/bench_d17_ecdsa.json 2>/dev/null
python bench.py --degree-bits 13 --mix arith --no-cpu-baseline --steps 30 > gpurun_out/r03c_cfg/bench_d13_arith.json 2>/dev/null
python bench.py --degree-bits 19 --mix ecdsa --no-cpu-baseline --steps 8 --warmup 4 --pipelined 0 > gpurun_out/r03c_cfg/bench_d19_ecdsa.json 2>/dev/null
python bench.py --degree-bits 21 --mix sha --no-cpu-baseline --steps 4 --warmup 1 --pipelined 0 --profile-steps 2 > gpurun_out/r03c_cfg/bench_d21_sha.json 2>/dev/null
python bench.py --degree-bits 21 --mix grammar --no-cpu-baseline --steps 4 --warmup 1 --pipelined 0 --profile-steps 2 > gpurun_out/r03c_cfg/bench_d21_grammar.json 2>/dev/null
python bench.py --hasher poseidon --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/r03c_cfg/bench_poseidon.json 2>/dev/null
python bench.py --group 0,0 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r03c_cfg/bench_group2_same_gpu.json 2>/dev/null
python bench.py --workload sha256 --no-cpu-baseline > gpurun_out/r03c_cfg/bench_sha256x4.json 2>/dev/null
# two ranks over gloo sharing the one GPU: the multi-rank flow of bench.py (replicas, then one sharded proof)
P2GPU_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 8 --warmup 2 --in-flight 2 > gpurun_out/r03c_cfg/bench_gloo2.json 2>gpurun_out/r03c_cfg/bench_gloo2.err
P2GPU_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 8 --warmup 2 --mode sharded > gpurun_out/r03c_cfg/bench_gloo2_sharded.json 2>gpurun_out/r03c_cfg/bench_gloo2_sharded.err
for f in gpurun_out/r03c_cfg/*.json; do python - "$f" <<'PY'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
    print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s with', d['in_flight_per_gpu'], 'in flight;', round(d['latency_ms_single_proof'],3), 'ms lone proof; host witness', d['host_witness'] and round(d['host_witness']['ms_per_proof'],2), 'ms')
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
