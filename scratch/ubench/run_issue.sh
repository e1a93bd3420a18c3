#!/bin/bash
# VALU issue ceiling, round 3: un-profiled HIP-event pass, then ONE rocprofv3 --pmc pass of the same binary
# (counters only + kernel trace for the timestamps; no other trace domain), joined into profiles/r03_ubench.txt
#   gpurun -- bash scratch/ubench/run_issue.sh [iters] [issue|issue_survey]
REPO=$(pwd); export TMPDIR=/tmp
IT=${1:-100000}
BIN=${2:-issue}
OUT=$REPO/gpurun_out/ubench_r03_$BIN; rm -rf $OUT; mkdir -p $OUT
cd /tmp
$REPO/scratch/ubench/$BIN $IT > $OUT/events.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM \
  --kernel-trace --output-format csv -d $OUT/pmc -- $REPO/scratch/ubench/$BIN $IT > $OUT/pmc_events.txt 2>&1
cd $REPO
find $OUT -name "*_agent_info.csv" -delete
python scratch/ubench/join_issue.py $OUT > $OUT/r03_ubench.txt
tail -5 $OUT/r03_ubench.txt
