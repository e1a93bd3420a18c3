"""Randomised GPU-vs-oracle differential run: random (degree, gate mix, seed, public inputs, width)."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); orc = entry.load_oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for it in range(N):
    d = int(rng.integers(5, 14)); mix = ["arith", "sha", "ecdsa"][int(rng.integers(0, 3))]
    seed = int(rng.integers(1, 1 << 30)); npi = int(rng.choice([0, 0, 1, 4, 9, 17])); nw = int(rng.choice([234, 234, 135]))
    if nw == 135 and mix == "ecdsa" and npi == 0 and d < 6: d = 6
    try:
        out = pkg.make_circuit(d, mix, seed, num_public_inputs=npi, num_wires=nw)
    except Exception as e:
        print("skip", d, mix, npi, nw, e); continue
    blob, wires = out[0], out[1]; pis = out[2] if npi else ()
    oc = orc.OracleCircuit(blob); cd = pkg.CircuitData(blob)
    expect, _ = oc.prove(wires, public_inputs=pis)
    got = [cd.prove(wires, public_inputs=pis).to_bytes(),
           cd.prove(torch.from_numpy(wires.view(np.int64)).cuda(), public_inputs=pis).to_bytes(),
           cd.prove_routed(wires[:80], public_inputs=pis).to_bytes()]
    ok = all(g == expect for g in got)
    try:
        cd.verify(expect); comp = cd.compress(expect); ok &= cd.decompress(comp).to_bytes() == expect
    except Exception as e:
        ok = False; print("verify/compress failed", e)
    if not ok:
        bad += 1; print("MISMATCH", d, mix, seed, npi, nw, [g == expect for g in got])
    cd.close()
print("configs", N, "mismatches", bad)
