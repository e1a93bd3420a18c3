#!/bin/bash
# every configuration of DESIGN.md / profiles/NUMBERS.md on one box -> gpurun_out/r04_cfg/*.json   (copied to profiles/r04_bench*.json)
O=gpurun_out/r04_cfg; mkdir -p $O
j() { grep "^{" | tail -1; }
python bench.py 2> $O/bench.err | j > $O/bench.json
python bench.py --public-inputs 4 --no-cpu-baseline --no-cold-process 2>/dev/null | j > $O/bench_pi4.json
python bench.py --mix ecdsa --no-cpu-baseline --no-cold-process 2>/dev/null | j > $O/bench_d17_ecdsa.json
python bench.py --degree-bits 13 --mix arith --no-cpu-baseline --no-cold-process --steps 64 2>/dev/null | j > $O/bench_d13_arith.json
python bench.py --degree-bits 19 --mix ecdsa --no-cpu-baseline --no-cold-process --steps 8 --warmup 4 --pipelined 0 2>/dev/null | j > $O/bench_d19_ecdsa.json
python bench.py --degree-bits 21 --mix sha --no-cpu-baseline --steps 4 --warmup 1 --pipelined 0 --profile-steps 2 2>/dev/null | j > $O/bench_d21_sha.json
python bench.py --degree-bits 21 --mix grammar --no-cpu-baseline --steps 4 --warmup 1 --pipelined 0 --profile-steps 2 2>/dev/null | j > $O/bench_d21_grammar.json
python bench.py --hasher poseidon --no-cpu-baseline --no-cold-process --steps 12 --warmup 3 2>/dev/null | j > $O/bench_poseidon.json
python bench.py --group 0,0 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | j > $O/bench_group2_same_gpu.json
python bench.py --workload sha256 --no-cpu-baseline --no-cold-process 2>/dev/null | j > $O/bench_sha256x4.json
# two ranks sharing the one GPU (gloo): the whole multi-rank flow -- replicas, one sharded proof over the ranks, the device-group probe
python bench.py --gpus 2 --backend gloo --steps 16 --warmup 4 2>/dev/null | j > $O/bench_gloo2.json
ls -la $O
