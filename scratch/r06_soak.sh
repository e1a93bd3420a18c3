#!/bin/bash
O=gpurun_out/r06_soak; mkdir -p $O
python scratch/fuzz_parity.py 61 120 > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
python scratch/stress_group.py 40 4 > $O/stress_group4.log 2>&1; tail -1 $O/stress_group4.log
python scratch/stress_group.py 20 8 > $O/stress_group8.log 2>&1; tail -1 $O/stress_group8.log
python scratch/stress.py > $O/stress.log 2>&1; tail -1 $O/stress.log
