#!/usr/bin/env python3
"""Safety net of the fixed-register Keccak kernels (keccak.hpp P2_KF_*): the sponge state lives in v[P2_KF_BASE ...) for the
whole kernel, so NOTHING hipcc generates outside the asm blocks may name a register at or above P2_KF_BASE, and the kernels
must not spill.  Compiles merkle.hip to assembly and checks every *_kf_kernel.  `make kf-check` (part of `make`)."""
import os
import re
import subprocess
import sys
import tempfile

here = os.path.dirname(os.path.abspath(__file__))
base = int(re.search(r"#define P2_KF_BASE (\d+)", open(os.path.join(here, "keccak_fixed.inc")).read()).group(1))
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-pass-failed", "-Wno-inline-asm", "--cuda-device-only", "-S",
                    os.path.join(here, "merkle.hip"), "-o", f.name], check=True, capture_output=True)
    lines = [l.strip() for l in open(f.name)]
bad = 0
names = [l[:-1].split(":")[0] for l in lines if re.match(r"^_Z\w*kf_kernel\w*:", l)]
assert names, "no *_kf_kernel found"
for name in names:
    i = lines.index(name + ":") if name + ":" in lines else [k for k, l in enumerate(lines) if l.startswith(name + ":")][0]
    inasm, mx = False, 0
    for l in lines[i:]:
        if l.startswith("s_endpgm"):
            break
        if "ASMSTART" in l:
            inasm = True
        elif "ASMEND" in l:
            inasm = False
        elif not inasm and l and not l.startswith((";", ".")) and not l.endswith(":"):
            regs = [int(x) for x in re.findall(r"\bv(\d+)\b", l)] + [int(b) for _, b in re.findall(r"v\[(\d+):(\d+)\]", l)]
            if regs:
                mx = max(mx, max(regs))
    scratch = [int(m.group(1)) for l in lines if (m := re.match(r"\.set " + re.escape(name) + r"\.private_seg_size, (\d+)", l))]
    ok = mx < base and (not scratch or scratch[0] == 0)
    print(f"{name[:60]:60s} compiler registers up to v{mx} (state from v{base}), scratch {scratch[0] if scratch else '?'} B: {'ok' if ok else 'VIOLATION'}")
    bad += not ok
sys.exit(1 if bad else 0)
