#!/bin/bash
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
OUT=$REPO/gpurun_out/prof_ecdsa; rm -rf $OUT; mkdir -p $OUT
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --mix ecdsa"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
cd $REPO
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda:[0.0,0])
for p in glob.glob('gpurun_out/prof_ecdsa/fetch/**/*_counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('p2::','')
        acc[k][0]+=float(r['Counter_Value']); acc[k][1]+=1
for k,(v,n) in sorted(acc.items(), key=lambda kv:-kv[1][0])[:5]: print(k, n, 'avg read MB (x2 corrected)', round(2*v/n*1024/1e6,1))
PY
head -4 gpurun_out/prof_ecdsa/stats/*/*_kernel_stats.csv | cut -c1-160
