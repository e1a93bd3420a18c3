#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_{stats,fetch,write}) into small tracked files:
  profiles/<tag>_kernel_stats.csv   -- `rocprofv3 --kernel-trace --stats` per-kernel summary
  profiles/<tag>_pmc_summary.json   -- per kernel: launches, avg FETCH_SIZE / WRITE_SIZE and the
                                       HBM traffic per launch derived from them
Units / corrections follow /opt/skills/guides (MI355X_MICROARCH.md "HBM"): the counters are in
KiB (bytes = value * 1024) and on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
coalesced stream, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as reported.
FETCH_SIZE and WRITE_SIZE come from SEPARATE --pmc passes (TCC slot limit).
Usage: bash scratch/prof.sh r01b (on the GPU box, via gpurun); python profiles/summarize.py r01b
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("p2::", "")
    return n.strip()


def newest(paths):
    return sorted(paths, key=os.path.getmtime)[-1:]


def counter_avgs(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    # gpurun MERGES a call's files into gpurun_out/: a tag profiled twice leaves both runs' CSVs side by side
    # (the pid is in the file name) -- only the newest run counts
    for path in newest(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                a = acc[short(row["Kernel_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def main():
    tag = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else os.path.join("gpurun_out", "prof_" + tag)
    here = os.path.dirname(os.path.abspath(__file__))
    stats = glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(newest(stats)[0], os.path.join(here, f"{tag}_kernel_stats.csv"))
    fetch = counter_avgs(os.path.join(src, "fetch"), "FETCH_SIZE")
    write = counter_avgs(os.path.join(src, "write"), "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out[k] = {
            "launches": max(nf, nw),
            "avg_FETCH_SIZE_KiB": f,
            "avg_WRITE_SIZE_KiB": w,
            "hbm_read_bytes_per_launch": 2.0 * f * 1024.0,
            "hbm_write_bytes_per_launch": w * 1024.0,
            "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
        }
    with open(os.path.join(here, f"{tag}_pmc_summary.json"), "w") as fo:
        json.dump({"note": "bytes = KiB * 1024; read side doubled (gfx950 FETCH_SIZE correction); separate --pmc passes",
                   "command": "python bench.py --steps 5 --warmup 2 --timed-only [+ workload flags] (scratch/prof.sh): 7 proofs + one circuit creation", "kernels": out}, fo, indent=1)
    # SQ counters (one more --pmc pass): where the waves' cycles go -- VALU issue vs waiting
    sq_names = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES"]
    sq = {}
    for cn in sq_names:
        for k, (avg, nl) in counter_avgs(os.path.join(src, "sq"), cn).items():
            sq.setdefault(k, {"launches": nl})[cn] = avg
    if sq:
        for k, v in sq.items():
            wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            v["frac_valu_active"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
            v["frac_wait_memory_or_barrier"] = v.get("SQ_WAIT_ANY", 0.0) / wc
            v["frac_wait_issue"] = v.get("SQ_WAIT_INST_ANY", 0.0) / wc
        with open(os.path.join(here, f"{tag}_sq_summary.json"), "w") as fo:
            json.dump({"note": "per-launch averages of SQ counters (quad-cycle units summed over waves); fractions are of SQ_WAVE_CYCLES",
                       "kernels": sq}, fo, indent=1)
    # whole-run totals: 7 proofs (2 warm-up + 5 timed) + one circuit creation in the profiled process
    tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values())
    print(f"HBM traffic of the whole process: {tot / 1e9:.2f} GB = {tot / 7 / 1e9:.2f} GB per proof (7 proofs + circuit creation)")
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k:32s} n={v['launches']:4d} read {v['hbm_read_bytes_per_launch']/1e6:10.1f} MB  write {v['hbm_write_bytes_per_launch']/1e6:10.1f} MB")


if __name__ == "__main__":
    main()
