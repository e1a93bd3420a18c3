#!/bin/bash
# shader clock per kernel of the timed bench command: GRBM_GUI_ACTIVE / 8 XCDs / dispatch time (rocprofv3 --pmc, counters only)
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
OUT=$REPO/gpurun_out/prof_clock; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT -- python $REPO/bench.py --steps 5 --warmup 2 --in-flight 1 --timed-only --clock-warmup-ms 0 "$@" > $OUT/log.txt 2>&1
cd $REPO
python - "$OUT" <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(lambda: [0.0,0.0,0,0.0])
for p in glob.glob(sys.argv[1]+'/**/*_counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('p2::','')
        ns=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
        if r['Counter_Name']=='GRBM_GUI_ACTIVE':
            a=acc[k]; a[0]+=float(r['Counter_Value'])/8; a[1]+=ns; a[2]+=1
        if r['Counter_Name']=='SQ_INSTS_VALU': acc[k][3]+=float(r['Counter_Value'])
for k,(gui,ns,n,iv) in sorted(acc.items(), key=lambda kv:-kv[1][1])[:8]:
    print('%-40s launches %3d avg %8.1f us  clock %.3f GHz  cycles/VALU wave-instr/SIMD %.2f' % (k,n,ns/n/1e3,gui/ns, gui/(iv/1024) if iv else 0))
PY
