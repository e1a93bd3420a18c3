"""Create / prove / destroy in a loop: device memory must come back."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as entry
import torch
pkg = entry.load_package()
blob, wires = pkg.make_circuit(15, "ecdsa", 1)
free0 = None
for it in range(25):
    cd = pkg.CircuitData(blob)
    cd.prove(wires)
    vd = cd.verifier_data(); vd.verify(cd.prove(wires)); vd.close()
    cd.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 2: free0 = free
    if it in (2, 12, 24): print("iter", it, "free MiB", free >> 20)
print("delta MiB after warm-up:", (free0 - free) >> 20)
