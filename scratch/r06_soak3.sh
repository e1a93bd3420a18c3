#!/bin/bash
# round 6, second session: new seeds on the final code
O=gpurun_out/r06_soak3; mkdir -p $O
python scratch/fuzz_parity.py 101 500 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
python scratch/fuzz_parity.py 202 300 > $O/fuzz2.log 2>&1; tail -1 $O/fuzz2.log
python scratch/stress.py > $O/stress.log 2>&1; tail -1 $O/stress.log
python scratch/stress_group.py 60 4 > $O/stress_group4.log 2>&1; tail -1 $O/stress_group4.log
python scratch/leak.py sha > $O/leak_sha.log 2>&1; tail -1 $O/leak_sha.log
python scratch/leak.py ecdsa > $O/leak_ecdsa.log 2>&1; tail -1 $O/leak_ecdsa.log
