#!/usr/bin/env python3
"""Static VALU instruction mix of the product's kernels by ISSUE-RATE CLASS (profiles/r03_ubench.txt): which share of a
kernel's vector instructions are opcodes a gfx950 SIMD retires at the full rate (2.2 cycles per wave64 instruction:
v_add_u32, v_sub_u32, v_and/or/xor_b32, v_lshrrev_b32, v_mov_b32, v_bitop3_b32, v_mul_lo_u16, v_fma/mul_f32 -- without an
SGPR source) and which at half rate (4.2 cycles: everything else).  From that the ceiling of the kernel's own mix:
  additive  = N / (N_full / 71.0 T + N_half / 37.7 T)            (measured single-kind rates, 8 waves per SIMD)
Static counts over the whole kernel body (the kernels are dominated by straight-line unrolled loops, so static ~ dynamic;
for the Keccak-f loop it is exact).  Usage: python profiles/isa_mix.py > profiles/r03_isa_mix.json   (needs hipcc)"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "acvm-backend-plonky2_amd", "csrc")
FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32", "v_bitop3_b32",
        "v_mul_lo_u16", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_not_b32"}
FULL_T, HALF_T = 71.0e12, 37.7e12


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return [o.split("(")[0].replace("void ", "").replace("p2::", "") for o in out]


def main():
    res = {}
    for unit in ("merkle", "ntt", "plonk", "fri", "witness"):
        with tempfile.NamedTemporaryFile(suffix=".s") as f:
            subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-pass-failed", "--cuda-device-only", "-S",
                            os.path.join(CSRC, unit + ".hip"), "-o", f.name], check=True, capture_output=True)
            lines = [l.strip() for l in open(f.name)]
        cur, body = None, {}
        for l in lines:
            m = re.match(r"^(_Z\w+):", l)
            if m:
                cur = m.group(1)
                body[cur] = []
            elif l.startswith(".Lfunc_end"):   # (not the first s_endpgm: a kernel with an early exit has several)
                cur = None
            elif cur and l and not l.startswith((";", ".", "/")) and not l.endswith(":"):
                body[cur].append(l)
        names = list(body)
        for mangled, nice in zip(names, demangle(names)):
            ops = collections.Counter()
            full = half = 0
            for ins in body[mangled]:
                op = ins.split()[0]
                if not op.startswith("v_") or op.startswith(("v_readfirstlane", "v_readlane", "v_writelane")):
                    continue
                base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
                ops[base] += 1
                operands = ins[len(op):].split("bitop3:")[0]
                has_sgpr_src = bool(re.search(r",\s*(s\d+|s\[\d+:\d+\]|vcc|exec)\b", operands)) and not base.startswith(("v_cmp", "v_add_co", "v_sub_co", "v_addc", "v_subb", "v_cndmask", "v_mad_u64"))
                if base in FULL and not has_sgpr_src and not op.endswith(("_dpp", "_sdwa")):
                    full += 1
                else:
                    half += 1
            n = full + half
            if n < 50:
                continue
            res[nice] = {"valu": n, "full_rate": full, "half_rate": half, "full_fraction": round(full / n, 4),
                         "mix_ceiling_lane_instr_per_s": n / (full / FULL_T + half / HALF_T),
                         "top": dict(ops.most_common(8))}
    json.dump({"note": "static VALU counts per kernel by issue-rate class; ceilings from profiles/r03_ubench.txt (full 71.0 T, half 37.7 T "
                       "lane-instr/s measured at 8 waves per SIMD, 2.38 GHz); additive model, transitions between classes cost a little more "
                       "(bitop3,bitop3,alignbit interleaved: 49.9 T measured vs 55 T additive)",
               "kernels": res}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
