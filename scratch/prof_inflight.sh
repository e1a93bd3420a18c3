#!/bin/bash
# The rocprofv3 passes of scratch/prof.sh at the OPERATING POINT OF `value`: the default number of proofs in flight (4) instead of one
# (VERDICT r04 item 5).  Same separation of passes: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; SQ counters.
#   bash scratch/prof_inflight.sh <tag> [extra bench.py args]   -> gpurun_out/prof_<tag>/{stats,fetch,write,sq}
set -x
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
TAG=${1:-run}; shift
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python $REPO/bench.py --steps 24 --warmup 8 --timed-only --clock-warmup-ms 0 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
cd $REPO
find $OUT -name "*_agent_info.csv" -delete
du -sh $OUT
