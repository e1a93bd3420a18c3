#!/bin/bash
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
OUT=$REPO/gpurun_out
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/prof_sq -- $CMD > $OUT/prof_sq.log 2>&1
cd $REPO
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for p in glob.glob('gpurun_out/prof_sq/**/*_counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('p2::','')
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[k]+=1
for k,v in sorted(acc.items(), key=lambda kv:-kv[1].get('SQ_BUSY_CYCLES',0))[:8]:
    n=cnt[k] or 1
    print(k, 'launches',n, {c: round(x/n) for c,x in v.items()})
PY
