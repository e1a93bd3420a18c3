#!/bin/bash
# one rocprofv3 --pmc pass (SQ counters) of the timed bench command: bash scratch/sq_only.sh <tag> [bench flags]
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
TAG=${1:-run}; shift
OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT; rm -rf $OUT/sq
CMD="python $REPO/bench.py --steps 5 --warmup 2 --in-flight 1 --timed-only $@"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
cd $REPO; find $OUT -name "*_agent_info.csv" -delete
python - "$OUT" <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for p in glob.glob(sys.argv[1]+'/sq/**/*_counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('p2::','')
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[k]+=1
for k,v in sorted(acc.items(), key=lambda kv:-kv[1].get('SQ_BUSY_CYCLES',0))[:6]:
    n=cnt[k] or 1
    print(k, 'launches',n, {c: round(x/n) for c,x in v.items()}, 'lds_conflict_frac', round(v.get('SQ_LDS_BANK_CONFLICT',0)/max(1,v.get('SQ_ACTIVE_INST_LDS',1)),3))
PY
