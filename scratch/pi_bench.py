"""Cost of public inputs: the PoseidonGate joins the gate set and is evaluated on every LDE row."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as entry
import torch
pkg = entry.load_package()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 17
mix = sys.argv[2] if len(sys.argv) > 2 else "sha"
for npi in (0, 4):
    out = pkg.make_circuit(d, mix, 1, num_public_inputs=npi)
    blob, wires = out[0], out[1]
    pis = out[2] if npi else ()
    cd = pkg.CircuitData(blob)
    wd = torch.from_numpy(wires.view(np.int64)).cuda()
    for _ in range(3): cd.prove(wd, public_inputs=pis)
    cd.set("profile", 1)
    n = 10
    t0 = time.perf_counter()
    for _ in range(n): p = cd.prove(wd, public_inputs=pis)
    wall = (time.perf_counter() - t0) / n * 1e3
    st = cd.kernel_stats()
    q = {k: round(v["ms"] / n, 3) for k, v in st.items() if "quotient_kernel" in k}
    print("public inputs", npi, "ms/proof %.3f" % wall, q, {k: round(v, 2) for k, v in p.timings.items() if k.endswith("_ms")})
    cd.close()
