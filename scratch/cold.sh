#!/bin/bash
# cold-process anatomy on one box: scratch/cold.sh   (writes the d = 17 sha circuit + witness to /dev/shm first)
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
import numpy as np
pkg = ge.load_package()
blob, wires = pkg.make_circuit(17, "sha", 1)
np.asarray(blob).tofile("/dev/shm/c.blob"); np.asarray(wires).tofile("/dev/shm/w.bin")
PY
T=acvm-backend-plonky2_amd/p2gpu-prove
echo "== floor"; for i in 1 2 3; do acvm-backend-plonky2_amd/p2gpu-cold-floor; done
echo "== p2gpu-prove --timing"; for i in 1 2 3; do $T /dev/shm/c.blob /dev/shm/w.bin /dev/shm/p.bin --timing 2>/dev/null; done
echo "== with P2GPU_TRACE=1"; P2GPU_TRACE=1 $T /dev/shm/c.blob /dev/shm/w.bin /dev/shm/p.bin --timing 2>&1 | grep -v "^\[p2gpu\] [a-z_ ]*: ok" | head -60
for e in "$@"; do echo "== env $e"; for i in 1 2; do env $e $T /dev/shm/c.blob /dev/shm/w.bin /dev/shm/p.bin --timing 2>/dev/null; done; done
rm -f /dev/shm/c.blob /dev/shm/w.bin /dev/shm/p.bin
