#!/usr/bin/env python3
"""Safety net of the fixed-register Keccak kernels (keccak.hpp P2_KF_*): the sponge state lives in v[P2_KF_BASE ... P2_KF_TOP)
for the whole kernel, so NOTHING hipcc generates outside the asm blocks may name a register at or above P2_KF_BASE, the
kernels must not spill, and the kernel descriptor must reserve the registers up to P2_KF_TOP.

    kf_check.py [HIPCC [ARCH [HIPFLAGS...]]]

`make` passes its own $(HIPCC), $(ARCH) and $(HIPFLAGS), so the assembly checked here is the assembly of the objects that were
just linked (the flags that only make sense for an object file are dropped).  A function's body runs to its `.Lfunc_end` /
`.size` line, not to the first `s_endpgm`: an early-exit `s_endpgm` must not leave the rest of the body unchecked."""
import os
import re
import subprocess
import sys
import tempfile

here = os.path.dirname(os.path.abspath(__file__))
inc = open(os.path.join(here, "keccak_fixed.inc")).read()
base = int(re.search(r"#define P2_KF_BASE (\d+)", inc).group(1))
top = int(re.search(r"#define P2_KF_TOP (\d+)", inc).group(1))
hipcc = sys.argv[1] if len(sys.argv) > 1 else "hipcc"
arch = sys.argv[2] if len(sys.argv) > 2 else "gfx950"
flags = [f for f in sys.argv[3:] if f not in ("-fPIC", "-c") and not f.startswith("--offload-arch")] or ["-O3", "-std=c++17", "-Wno-pass-failed", "-Wno-inline-asm"]
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.run([hipcc] + flags + [f"--offload-arch={arch}", "--cuda-device-only", "-S", os.path.join(here, "merkle.hip"), "-o", f.name],
                   check=True, capture_output=True)
    lines = [l.strip() for l in open(f.name)]
bad = 0
names = [l[:-1].split(":")[0] for l in lines if re.match(r"^_Z\w*kf_kernel\w*:", l)]
assert names, "no *_kf_kernel found"
for name in names:
    i = [k for k, l in enumerate(lines) if l.startswith(name + ":")][0]
    ends = [k for k in range(i, len(lines)) if re.match(r"^\.Lfunc_end\d+:", lines[k]) or lines[k].startswith(".size\t" + name + ",") or lines[k].startswith(".size " + name + ",")]
    assert ends, f"no end marker for {name}"
    inasm, mx, endpgm, barriers = False, 0, 0, 0
    for l in lines[i:ends[0]]:
        if l.startswith("s_endpgm"):
            endpgm += 1
        if l.startswith("s_barrier"):
            barriers += 1
        if "ASMSTART" in l:
            inasm = True
        elif "ASMEND" in l:
            inasm = False
        elif not inasm and l and not l.startswith((";", ".")) and not l.endswith(":"):
            regs = [int(x) for x in re.findall(r"\bv(\d+)\b", l)] + [int(b) for _, b in re.findall(r"v\[(\d+):(\d+)\]", l)]
            if regs:
                mx = max(mx, max(regs))
    scratch = [int(m.group(1)) for l in lines if (m := re.match(r"\.set " + re.escape(name) + r"\.private_seg_size, (\d+)", l))]
    # the descriptor's register count: the block clobbers v[BASE, TOP) -- the allocation must cover them
    nfree = None
    k0 = [k for k, l in enumerate(lines) if l.startswith(".amdhsa_kernel " + name)]
    if k0:
        for l in lines[k0[0]:k0[0] + 80]:
            m = re.match(r"\.amdhsa_next_free_vgpr\s+(\S+)", l)
            if m:
                try:
                    nfree = int(m.group(1))
                except ValueError:   # symbolic: resolve through the .set lines
                    sym = [int(mm.group(1)) for ll in lines if (mm := re.match(r"\.set " + re.escape(name) + r"\.num_vgpr, (\d+)", ll))]
                    nfree = sym[0] if sym else None
                break
    # hash_lde_leaves_kf_kernel<V, 2>: waves whose digests have been handed on LEAVE the loop while the others still meet at the
    # second exchange -- correct on gfx950 (s_barrier counts the waves that are still alive) as long as the loop keeps exactly its
    # two barriers: one per exchange, not duplicated or merged by a restructuring of the loop
    want_barriers = 2 if re.search(r"hash_lde_leaves_kf_kernelILb[01]ELi2E", name) else None
    ok = mx < base and (not scratch or scratch[0] == 0) and endpgm >= 1 and nfree is not None and nfree >= top and \
        (want_barriers is None or barriers == want_barriers)
    print(f"{name[:60]:60s} compiler registers up to v{mx} (state v{base}..v{top - 1}), scratch {scratch[0] if scratch else '?'} B, "
          f"next_free_vgpr {nfree}, s_endpgm x{endpgm}, s_barrier x{barriers}: {'ok' if ok else 'VIOLATION'}")
    bad += not ok
sys.exit(1 if bad else 0)
