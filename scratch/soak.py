"""Soak at the benchmark size: N proofs of synth(17, sha) on 4 handles / threads, every proof's SHA-256 must equal the first one's.
python scratch/soak.py [N]"""
import hashlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
import torch
pkg = entry.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
blob, wires = pkg.make_circuit(17, "sha", 1)
wd = torch.from_numpy(wires.view(np.int64)).cuda()
hs = [pkg.CircuitData(blob) for _ in range(4)]
ref = hashlib.sha256(hs[0].prove(wd).to_bytes()).hexdigest()
bad = []
def work(i):
    for it in range(i, N, 4):
        src = wd if it % 3 else wires          # resident and host-witness entries interleaved
        h = hashlib.sha256(hs[i].prove(src).to_bytes()).hexdigest()
        if h != ref:
            bad.append((it, h))
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
[t.start() for t in th]; [t.join() for t in th]
dt = time.perf_counter() - t0
print(f"{N} proofs at 2^20 LDE rows in {dt:.1f} s ({N / dt:.1f} proofs/s, a third through the host-witness entry), mismatches {len(bad)} {bad[:3]}, digest {ref[:16]}")
