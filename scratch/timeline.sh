#!/bin/bash
# rocprofv3 kernel trace of a lone proof -> scratch/timeline.py: busy time, idle gaps by the kernel that precedes them
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
OUT=$REPO/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $REPO/bench.py --steps 6 --warmup 3 --in-flight 1 --timed-only --clock-warmup-ms 0 "$@" > $OUT/log 2>&1
f=$(find $OUT -name "*_kernel_trace.csv" | head -1)
python3 $REPO/scratch/timeline.py "$f" 1
find $OUT -name "*.csv" -delete
