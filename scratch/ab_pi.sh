#!/bin/bash
# per-kernel times of library variants on a public-input circuit (poseidon_gate_kernel): scratch/ab_pi.sh <variant> ...
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" != default ]; then export P2GPU_LIBRARY=$PWD/acvm-backend-plonky2_amd/csrc/build_alt/libp2gpu_$v.so; else unset P2GPU_LIBRARY; fi
  python - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
import numpy as np, torch, time
pkg = ge.load_package()
blob, wires, pis = pkg.make_circuit(17, "sha", 1, num_public_inputs=4)
cd = pkg.CircuitData(blob)
wd = torch.from_numpy(wires.view(np.int64)).cuda()
for _ in range(3): cd.prove(wd, public_inputs=pis)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(8): cd.prove(wd, public_inputs=pis)
lone = (time.time() - t0) / 8 * 1e3
cd.set("profile", 2)
for _ in range(4): cd.prove(wd, public_inputs=pis)
st = cd.kernel_stats()
print("$v", "lone %.3f ms" % lone, {k: round(v["ms"] / 4, 4) for k, v in st.items() if "poseidon" in k or "quotient_kernel" in k})
PY
done
done
