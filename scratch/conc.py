import os, sys, time, threading
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
P = entry.load_package()
d, mix = 17, 'sha'
blob, w = P.make_circuit(d, mix, 1)
wd = torch.from_numpy(w.view(np.int64)).cuda()
for T in (1, 2, 3):
    cds = [P.CircuitData(blob) for _ in range(T)]
    for cd in cds: cd.prove(wd)
    K = 12
    def work(cd, k):
        for _ in range(k): cd.prove(wd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(cds[i], K // T)) for i in range(T)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"threads={T}: {K/dt:.1f} proofs/s, {dt/K*1e3:.2f} ms/proof", flush=True)
    for cd in cds: cd.close()
