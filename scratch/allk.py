"""Every kernel launch of one proof (profile=2), sorted by time; sum vs wall clock."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as entry
import torch
pkg = entry.load_package()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 17
mix = sys.argv[2] if len(sys.argv) > 2 else "sha"
blob, wires = pkg.make_circuit(d, mix, 1)
t0 = time.perf_counter()
cd = pkg.CircuitData(blob)
torch.cuda.synchronize()
print("circuit_create (build precompute on GPU incl. blob upload): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
wd = torch.from_numpy(wires.view(np.int64)).cuda()
for _ in range(3): cd.prove(wd)
cd.set("profile", 2)
n = 10
t0 = time.perf_counter()
for _ in range(n): p = cd.prove(wd)
wall = (time.perf_counter() - t0) / n * 1e3
st = cd.kernel_stats()
tot = 0
for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]):
    print("%-44s %8.4f ms/proof %6.1f launches/proof" % (k, v["ms"] / n, v["launches"] / n))
    tot += v["ms"] / n
print("sum of kernels %.3f ms, wall %.3f ms (profile=2 adds event overhead)" % (tot, wall))
print("phases", {k: round(v, 3) for k, v in p.timings.items()})
