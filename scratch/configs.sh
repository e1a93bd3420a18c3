#!/bin/bash
# every configuration of DESIGN.md / profiles/NUMBERS.md on one box -> gpurun_out/<tag>_cfg/: the full results (bench.py --detail; copied to
# profiles/<tag>_bench*.json) and the stdout lines (copied to profiles/<tag>_line*.json)          usage: bash scratch/configs.sh r05
T=${1:-r05}; O=gpurun_out/${T}_cfg; mkdir -p $O
b() { name=$1; shift; python bench.py --detail $O/bench$name.json "$@" 2> $O/bench$name.err | grep "^{" | tail -1 > $O/line$name.json; }
b ""
b _pi4 --public-inputs 4 --no-cpu-baseline --no-cold-process
b _d17_ecdsa --mix ecdsa --no-cpu-baseline --no-cold-process
b _d13_arith --degree-bits 13 --mix arith --no-cpu-baseline --no-cold-process --steps 64
b _d19_ecdsa --degree-bits 19 --mix ecdsa --no-cpu-baseline --no-cold-process --steps 8 --warmup 4 --pipelined 0
b _d21_sha --degree-bits 21 --mix sha --no-cpu-baseline --steps 4 --warmup 1 --pipelined 0 --profile-steps 2
b _d21_grammar --degree-bits 21 --mix grammar --no-cpu-baseline --steps 4 --warmup 1 --pipelined 0 --profile-steps 2
b _poseidon --hasher poseidon --no-cpu-baseline --no-cold-process --steps 12 --warmup 3
b _group2_same_gpu --group 0,0 --steps 8 --warmup 2 --no-cpu-baseline
b _sha256x4 --workload sha256 --no-cpu-baseline --no-cold-process
# two ranks sharing the one GPU (gloo): the whole multi-rank flow -- replicas, one sharded proof over the ranks, the device-group probe
b _gloo2 --gpus 2 --backend gloo --steps 16 --warmup 4
rm -f $O/*.err
ls -la $O
