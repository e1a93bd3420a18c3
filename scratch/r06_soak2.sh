#!/bin/bash
O=gpurun_out/r06_soak2; mkdir -p $O
python scratch/fuzz_parity.py 77 400 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
python scratch/soak.py 12000 > $O/soak.log 2>&1; tail -1 $O/soak.log
python scratch/stress_group.py 60 8 > $O/stress_group8.log 2>&1; tail -1 $O/stress_group8.log
python scratch/stress_group.py 60 2 > $O/stress_group2.log 2>&1; tail -1 $O/stress_group2.log
