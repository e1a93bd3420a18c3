#!/bin/bash
# rocprofv3 recipe (guide: cd /tmp && export TMPDIR=/tmp first); outputs under gpurun_out/prof_*
set -x
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
OUT=$REPO/gpurun_out
CMD="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $CMD > $OUT/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -- $CMD > $OUT/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -- $CMD > $OUT/prof_write.log 2>&1
cd $REPO
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -type f | head -30
du -sh gpurun_out/prof_*
