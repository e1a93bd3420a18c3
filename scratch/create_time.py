import sys, time, os
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as entry
import torch
pkg = entry.load_package()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 17
blob, wires = pkg.make_circuit(d, "sha", 1)
cd0 = pkg.CircuitData(blob)  # first handle: runtime/code-object warm-up included
for i in range(3):
    t0 = time.perf_counter()
    cd = pkg.CircuitData(blob)
    torch.cuda.synchronize()
    print("circuit_create #%d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
    t0 = time.perf_counter(); cd.close(); print("  destroy %.1f ms" % ((time.perf_counter() - t0) * 1e3))
