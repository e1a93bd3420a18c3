"""Create / prove / destroy in a loop: device memory must come back."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as entry
import torch
pkg = entry.load_package()
mix = sys.argv[1] if len(sys.argv) > 1 else "sha"
blob, wires, pis = pkg.make_circuit(15, mix, 1, num_public_inputs=9)
free0 = None
for it in range(25):
    cd = pkg.CircuitData(blob)
    cd.prove(wires, public_inputs=pis)
    if mix != "ecdsa":   # (the heavy mix uses every wire: no structured suffix to leave out)
        cd.prove_sparse(wires, 135, int(np.nonzero(wires.reshape(234, -1)[233])[0][0]), public_inputs=pis)
    else:
        cd.set("half_gates", 2)   # the half-domain buffers belong to the handle: they must come back too
        cd.prove(wires, public_inputs=pis)
    vd = cd.verifier_data(); vd.verify(cd.prove(wires, public_inputs=pis)); vd.close()
    cd.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 2: free0 = free
    if it in (2, 12, 24): print("iter", it, "free MiB", free >> 20)
print("delta MiB after warm-up:", (free0 - free) >> 20)
