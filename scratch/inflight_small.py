"""Proofs/s of small circuits against the number of proofs in flight (handles / threads) on one GPU.
usage: python scratch/inflight_small.py [d] [mix] [blocking]"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as e
P = e.load_package()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 13
mix = sys.argv[2] if len(sys.argv) > 2 else "arith"
blocking = int(sys.argv[3]) if len(sys.argv) > 3 else -1
blob, w = P.make_circuit(d, mix, 1)
wd = torch.from_numpy(w.view(np.int64)).cuda()
for T in (1, 2, 4, 8, 12, 16, 24, 32):
    cds = [P.CircuitData(blob) for _ in range(T)]
    if blocking >= 0:
        for cd in cds: cd.set("blocking_sync", blocking)
    for cd in cds: cd.prove(wd)
    K = 64 * T
    def work(cd, k):
        for _ in range(k): cd.prove(wd)
    best = 0
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(cds[i], K // T)) for i in range(T)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = max(best, K / dt)
    print(f"d={d} {mix} blocking={blocking} in flight {T:2d}: {best:8.1f} proofs/s, {1e3/best:.3f} ms/proof", flush=True)
    for cd in cds: cd.close()
