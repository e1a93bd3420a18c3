#!/bin/bash
# round 6 counter passes: the bench line (sha17), the heavy gate mix (ecdsa17) and the Poseidon mode; condensed by profiles/summarize.py
bash scratch/prof.sh r06_sha17 > /dev/null 2>&1
bash scratch/prof.sh r06_ecdsa17 --mix ecdsa > /dev/null 2>&1
bash scratch/prof.sh r06_poseidon17 --hasher poseidon > /dev/null 2>&1
bash scratch/prof.sh r06_grammar17 --mix grammar > /dev/null 2>&1
for t in r06_sha17 r06_ecdsa17 r06_poseidon17 r06_grammar17; do python profiles/summarize.py $t > gpurun_out/prof_$t/summary.log 2>&1; tail -2 gpurun_out/prof_$t/summary.log; done
mkdir -p gpurun_out/r06_prof_out; cp profiles/r06_*_kernel_stats.csv profiles/r06_*_summary.json gpurun_out/r06_prof_out/ 2>/dev/null
du -sh gpurun_out/prof_r06_*; ls gpurun_out/r06_prof_out
