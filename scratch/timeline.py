#!/usr/bin/env python3
"""Timeline of ONE proof from a rocprofv3 --kernel-trace CSV: busy time, idle gaps between consecutive
dispatches (grouped by what precedes them), and the biggest gaps.  usage: timeline.py <kernel_trace.csv> [proof-index-from-end]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.split("(")[0].replace("void ", "").replace("p2::", "")
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
# a proof ends with gather_u64_kernel
ends = [i for i, e in enumerate(ev) if e[2].startswith("gather")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
i1 = ends[-k]; i0 = ends[-k - 1] + 1
pe = ev[i0:i1 + 1]
t0, t1 = pe[0][0], pe[-1][1]
busy = sum(e[1] - e[0] for e in pe)
print(f"proof: {len(pe)} dispatches, span {(t1-t0)/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(t1-t0-busy)/1e6:.3f} ms")
gaps = []
for a, b in zip(pe, pe[1:]):
    gaps.append((b[0] - a[1], a[2], b[2], (a[1]-t0)/1e6))
by = collections.defaultdict(lambda: [0, 0])
for g, a, b, _ in gaps:
    by[a][0] += g; by[a][1] += 1
print("idle after kernel (total us, count):")
for a, (g, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {a:40s} {g/1e3:9.1f} us  x{n}")
print("largest gaps:")
for g, a, b, t in sorted(gaps, reverse=True)[:16]:
    print(f"  {g/1e3:8.1f} us at +{t:.3f} ms  {a} -> {b}")
kt = collections.defaultdict(lambda: [0, 0])
for s, e, n in pe:
    kt[n][0] += e - s; kt[n][1] += 1
print("kernel time:")
for n, (t, c) in sorted(kt.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"  {n:44s} {t/1e3:9.1f} us  x{c}")
