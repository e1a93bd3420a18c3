#!/bin/bash
# round-4 counter passes, keyed by (mix, degree_bits) as bench.py's newest() expects: profiles/r04_<mix><d>_*
# (separate rocprofv3 passes per scratch/prof.sh; never --pmc together with a trace domain gpurun refuses)
bash scratch/prof.sh r04_sha17 > /dev/null 2>&1
bash scratch/prof.sh r04_ecdsa17 --mix ecdsa > /dev/null 2>&1
bash scratch/prof.sh r04_ecdsa19 --mix ecdsa --degree-bits 19 > /dev/null 2>&1
bash scratch/prof.sh r04_grammar21 --mix grammar --degree-bits 21 --steps 3 --warmup 1 > /dev/null 2>&1
# the timed command with the default four proofs in flight: what rocprofv3 sees at the operating point of `value`
( REPO=$(pwd); export TMPDIR=/tmp; cd /tmp; OUT=$REPO/gpurun_out/prof_r04_sha17_inflight4; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py --steps 48 --warmup 8 --timed-only --clock-warmup-ms 0 > $OUT/stats.log 2>&1
  find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete )
bash scratch/clock_pmc.sh > gpurun_out/r04_clock_sha.txt 2>&1
bash scratch/clock_pmc.sh --mix ecdsa > gpurun_out/r04_clock_ecdsa.txt 2>&1
du -sh gpurun_out/prof_r04_*
