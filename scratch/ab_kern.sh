#!/bin/bash
# per-kernel HIP-event times of library variants on one box (timing experiments: results may be wrong):
#   [MIX=sha] [ENVS="A=1 B=2"] scratch/ab_kern.sh <variant>[:ENV=V[,ENV=V]] ...     (variant `default` = the shipped library)
for rep in 1 2; do
for spec in "$@"; do
  v=${spec%%:*}; envs=""
  if [ "$spec" != "$v" ]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  if [ "$v" != default ]; then export P2GPU_LIBRARY=$PWD/acvm-backend-plonky2_amd/csrc/build_alt/libp2gpu_$v.so; else unset P2GPU_LIBRARY; fi
  env $envs python - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
import numpy as np, torch
pkg = ge.load_package()
d = int(os.environ.get("DBITS", "17"))
blob, wires = pkg.make_circuit(d, os.environ.get("MIX", "sha"), 1)
cd = pkg.CircuitData(blob)
cd.set("self_check", 0)
wd = torch.from_numpy(wires.view(np.int64)).cuda()
for _ in range(3): cd.prove(wd)
import time
torch.cuda.synchronize(); t0 = time.time()
for _ in range(8): cd.prove(wd)
lone = (time.time() - t0) / 8 * 1e3
cd.set("profile", 1)
for _ in range(4): cd.prove(wd)
st = cd.kernel_stats()
print("$spec", "lone %.3f ms" % lone, {k: round(v["ms"] / 4, 4) for k, v in st.items() if "ntt_" in k or "quotient_kernel" in k or "hash_lde" in k or "gate_sums" in k or "poseidon" in k})
PY
done
done
