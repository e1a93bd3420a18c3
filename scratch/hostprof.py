import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e, numpy as np, torch
pkg = e.load_package()
blob, wires = pkg.make_circuit(17, "sha", 1)
cd = pkg.CircuitData(blob)
wd = torch.from_numpy(wires.view(np.int64)).cuda()
for i in range(6): cd.prove(wd)
