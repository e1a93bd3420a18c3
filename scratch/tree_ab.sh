#!/bin/bash
# A/B of the leaf-hash launch that also builds the first two tree levels (P2GPU_LEAF_LEVELS=1, default) against leaves only + one
# launch per level (=0): rocprofv3 kernel trace of a lone proof, every hash / tree dispatch of ONE proof in order with its duration.
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
for v in 1 0; do
  OUT=$REPO/gpurun_out/tree_ab_$v; rm -rf $OUT; mkdir -p $OUT
  P2GPU_LEAF_LEVELS=$v rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $REPO/bench.py --steps 6 --warmup 3 --in-flight 1 --timed-only --clock-warmup-ms 0 "$@" > $OUT/log 2>&1
  f=$(find $OUT -name "*_kernel_trace.csv" | head -1)
  echo "== P2GPU_LEAF_LEVELS=$v"
  python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.split("(")[0].replace("void ", "").replace("p2::", "")
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
ends = [i for i, e in enumerate(ev) if e[2].startswith("gather")]
pe = ev[ends[-2] + 1: ends[-1] + 1]
t0 = pe[0][0]
print(f"proof span {(pe[-1][1]-t0)/1e3:.1f} us, busy {sum(e[1]-e[0] for e in pe)/1e3:.1f} us, {len(pe)} dispatches")
tot = 0
for i, (s, e, n) in enumerate(pe):
    if n.startswith(("hash_", "merkle")):
        gap = (s - pe[i-1][1]) / 1e3 if i else 0
        tot += e - s
        print(f"  +{(s-t0)/1e3:8.1f} us  {(e-s)/1e3:8.1f} us  (gap before {gap:5.1f})  {n}")
print(f"hash + tree kernels: {tot/1e3:.1f} us")
PY
  find $OUT -name "*.csv" -delete
done
