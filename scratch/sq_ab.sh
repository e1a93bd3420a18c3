#!/bin/bash
# SQ-counter A/B of library variants for one kernel: scratch/sq_ab.sh "<bench flags>" <kernel substring> <variant> [...]
REPO=$(pwd); export TMPDIR=/tmp
FLAGS=$1; KERN=$2; shift 2
for v in "$@"; do
  if [ "$v" != default ]; then export P2GPU_LIBRARY=$REPO/acvm-backend-plonky2_amd/csrc/build_alt/libp2gpu_$v.so; else unset P2GPU_LIBRARY; fi
  OUT=/tmp/sqab_$v; rm -rf $OUT
  (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT -- python $REPO/bench.py --steps 3 --warmup 1 --in-flight 1 --timed-only $FLAGS > $OUT.log 2>&1)
  python - "$OUT" "$KERN" "$v" <<'PY'
import csv,glob,collections,sys
out,kern,v=sys.argv[1:4]
acc=collections.defaultdict(lambda:collections.defaultdict(float)); n=collections.Counter()
for p in glob.glob(out+'/**/*_counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        if kern in r['Kernel_Name']:
            acc[r['Counter_Name']][r['Dispatch_Id']]=float(r['Counter_Value'])
print(v, {c: round(sum(d.values())/max(1,len(d))/1e6,2) for c,d in acc.items()}, '(millions per launch)')
PY
done
