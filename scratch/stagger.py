"""Do four proofs in flight run better in lockstep or staggered?  Threads start together, or thread i starts i * step ms late.
usage: python scratch/stagger.py [in_flight] [proofs per thread]"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as e
P = e.load_package()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
blob, w = P.make_circuit(17, "sha", 1)
wd = torch.from_numpy(w.view(np.int64)).cuda()
cds = [P.CircuitData(blob) for _ in range(S)]
for cd in cds:
    for _ in range(3): cd.prove(wd)
def run(step_ms):
    done = [0.0] * S
    def work(i):
        if step_ms: time.sleep(i * step_ms * 1e-3)
        for _ in range(K): cds[i].prove(wd)
        done[i] = time.perf_counter()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = max(done) - t0 - (S - 1) * step_ms * 1e-3 * 0  # (the late starters' delay is inside the region: conservative)
    return S * K / dt
for rep in range(3):
    for step in (0.0, 1.1, 2.2, 4.4):
        print(f"rep {rep} stagger {step:.1f} ms: {run(step):.1f} proofs/s", flush=True)
