"""Repeatability stress of the single-process device group (p2gpu_init with several ids): several group handles proved
from several caller threads at once -- every proof fans out over its own rank threads -- must always give the oracle's
bytes; host witness (column blocks + peer all-gather, the host scan is off for sharded proofs), resident witness,
sparse and routed entries interleaved."""
import sys, threading, hashlib, os
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
import torch
pkg = entry.load_package()
orc = entry.load_oracle()
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pkg.init([0] * world)
cases = [(9, "ecdsa", 1, 3), (12, "sha", 2, 0), (13, "grammar", 3, 0), (11, "ecdsa", 4, 0)]
handles = []
for d, mix, seed, npi in cases:
    out = pkg.make_circuit(d, mix, seed, num_public_inputs=npi, pi_row_routed_only=True)
    blob, wires = out[0], out[1]
    pis = out[2] if npi else ()
    expect, _ = orc.OracleCircuit(blob).prove(wires, public_inputs=pis)
    for rep in range(2):
        handles.append((pkg.CircuitData(blob), wires, torch.from_numpy(wires.view(np.int64)).cuda(), pis, expect))
bad = []
def work(i, n):
    cd, wh, wd, pis, expect = handles[i]
    for it in range(n):
        if it % 7 == 0:
            cd.set("shard_intt", (it // 7) % 2)   # round 6: column-sharded inverse transforms + in-place all-gather of the blocks, on and off
            cd.set("shard_reduce", (it // 14) % 2)   # ... and the column-sharded FRI batch reduction
            cd.set("shard_zs", (it // 7 + it // 14) % 2)   # ... and the row-sharded chunk quotients of the permutation argument
        if it % 5 == 3:
            p = cd.prove_routed(np.ascontiguousarray(wh[:80]), public_inputs=pis).to_bytes()
        else:
            p = cd.prove(wd if it % 2 else wh, public_inputs=pis).to_bytes()
        if p != expect:
            bad.append((i, it, hashlib.sha256(p).hexdigest()[:12]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
th = [threading.Thread(target=work, args=(i, n)) for i in range(len(handles))]
[t.start() for t in th]; [t.join() for t in th]
print("group world", world, "proofs", n * len(handles), "mismatches", len(bad), bad[:5])
