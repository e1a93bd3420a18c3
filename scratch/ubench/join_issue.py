#!/usr/bin/env python3
"""Join issue's HIP-event table (un-profiled pass) with the per-dispatch counters of the rocprofv3 --pmc pass.
Rows are matched by order: the binary launches, per row, one 64-trip warm-up and one timed dispatch of the same kernel
and grid; the timed one is the longer of each consecutive pair."""
import csv, glob, os, sys
from collections import OrderedDict

out = sys.argv[1]
ev = [l.rstrip("\n") for l in open(os.path.join(out, "events.txt"))]
head = [l for l in ev if l.startswith("#")]
rows = [l.split() for l in ev if l and not l.startswith("#") and not l.startswith("instruction")]
disp = OrderedDict()
for p in sorted(glob.glob(os.path.join(out, "pmc", "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
    for r in csv.DictReader(open(p)):
        if not r["Kernel_Name"].startswith("k"):
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]),
                                                    "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
ds = [disp[k] for k in sorted(disp)]
timed = [b for a, b in zip(ds[0::2], ds[1::2])]
for h in head:
    print(h)
print("# columns: HIP-event wall of the un-profiled pass | --pmc pass: dispatch ns, clock = GRBM_GUI_ACTIVE/8/ns, cycles per wave-instr per SIMD")
print("#          = GRBM_GUI_ACTIVE/8 / (SQ_INSTS_VALU / 1024), SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES*4 / (waves * GRBM_GUI_ACTIVE) (1 = resident whole kernel)")
print("%-24s %6s %-9s %2s %9s %10s | %9s %7s %9s %10s %9s %9s" % ("instruction", "chains", "banks", "K", "ms", "T lane-i/s", "pmc ms", "GHz", "cyc/instr", "T lane-i/s", "sqbusy/gui", "resid"))
for i, r in enumerate(rows):
    name, chains, bank, K, ms, rate, winstr = r[0], r[1], r[2], r[3], float(r[4]), float(r[5]), float(r[6])
    line = "%-24s %6s %-9s %2s %9.3f %10.2f |" % (name, chains, bank, K, ms, rate)
    if i < len(timed):
        t = timed[i]
        gui = t.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # the counter is summed over the 8 XCDs (raw value / wall = 19.1 GHz = 8 x 2.39)
        iv = t.get("SQ_INSTS_VALU", 0.0)
        ns = t["ns"]
        waves = t.get("SQ_WAVES", 0.0) or 1.0
        line += " %9.3f %7.3f %9.3f %10.2f %9.2f %9.2f" % (ns / 1e6, gui / ns if ns else 0, gui / (iv / 1024.0) if iv else 0, iv * 64 / (ns * 1e-9) / 1e12 if ns else 0,
                                                       t.get("SQ_BUSY_CYCLES", 0.0) / gui if gui else 0, t.get("SQ_WAVE_CYCLES", 0.0) * 4 / (waves * gui) if gui else 0)
    print(line)
