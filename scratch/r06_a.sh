#!/bin/bash
# round 6, first measurement pass: host bulk upload (value_host_witness), 2^24 rows with several proofs in flight, the multi-rank flow
T=r06; O=gpurun_out/${T}_a; mkdir -p $O
b() { name=$1; shift; python bench.py --detail $O/bench$name.json "$@" 2> $O/bench$name.err | grep "^{" | tail -1 > $O/line$name.json; tail -3 $O/bench$name.err; }
b "" --no-cpu-baseline --no-cold-process
P2GPU_HOST_BULK=0 b _nobulk --no-cpu-baseline --no-cold-process
b _gloo2 --gpus 2 --backend gloo --steps 16 --warmup 4 --no-cpu-baseline
b _d21_grammar --degree-bits 21 --mix grammar --no-cpu-baseline --steps 8 --warmup 2 --pipelined 0 --profile-steps 2
b _d21_sha --degree-bits 21 --mix sha --no-cpu-baseline --steps 8 --warmup 2 --pipelined 0 --profile-steps 2
ls -la $O
