#!/bin/bash
# instruction-cache counters of one kernel: scratch/icache.sh "<bench flags>" <kernel substring>
REPO=$(pwd); export TMPDIR=/tmp
FLAGS=$1; KERN=$2
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQC?_[A-Z_]*(ICACHE|IFETCH|INST_CACHE|DCACHE)[A-Z_]*" | sort -u | tr '\n' ' '; echo
OUT=/tmp/icache; rm -rf $OUT
(cd /tmp && rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT -- python $REPO/bench.py --steps 3 --warmup 1 --in-flight 1 --timed-only $FLAGS > $OUT.log 2>&1; tail -3 $OUT.log)
python - "$OUT" "$KERN" <<'PY'
import csv,glob,collections,sys
out,kern=sys.argv[1:3]
acc=collections.defaultdict(lambda:collections.defaultdict(float))
for p in glob.glob(out+'/**/*_counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        if kern in r['Kernel_Name']:
            acc[r['Counter_Name']][r['Dispatch_Id']]=float(r['Counter_Value'])
print({c: round(sum(d.values())/max(1,len(d))/1e6,3) for c,d in acc.items()}, '(millions per launch)')
PY
