#!/bin/bash
# per-kernel HIP-event times of library variants on one box (timing experiments: results may be wrong): scratch/ab_kern.sh <variant> [...]
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" != default ]; then export P2GPU_LIBRARY=$PWD/acvm-backend-plonky2_amd/csrc/build_alt/libp2gpu_$v.so; else unset P2GPU_LIBRARY; fi
  python - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
import numpy as np, torch
pkg = ge.load_package()
blob, wires = pkg.make_circuit(17, os.environ.get("MIX", "sha"), 1)
cd = pkg.CircuitData(blob)
cd.set("self_check", 0)
wd = torch.from_numpy(wires.view(np.int64)).cuda()
for _ in range(3): cd.prove(wd)
cd.set("profile", 1)
for _ in range(4): cd.prove(wd)
st = cd.kernel_stats()
print("$v", {k: round(v["ms"] / 4, 4) for k, v in st.items() if "ntt_pass" in k or "quotient_kernel" in k or "hash_lde" in k})
PY
done
done
