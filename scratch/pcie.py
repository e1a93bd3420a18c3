import os, sys, time
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
P = entry.load_package()
blob, w = P.make_circuit(17, 'sha', 1)
cd = P.CircuitData(blob)
wd = torch.from_numpy(w.view(np.int64)).cuda()
routed = np.ascontiguousarray(w[:80])
for name, fn in (('prove_dev (witness resident)', lambda: cd.prove(wd)), ('prove (245 MB host witness)', lambda: cd.prove(w)), ('prove_routed (84 MB host, GPU fill)', lambda: cd.prove_routed(routed))):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(5): p = fn()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt*1e3:.2f} ms/proof (h2d {p.timings['h2d_ms']:.2f} ms)")
# fill kernel alone
cd.set('profile', 1); cd.prove_routed(routed); print({k: v for k, v in cd.kernel_stats().items() if 'fill' in k})
