"""Randomised GPU-vs-oracle differential run: random (degree, gate mix, seed, public inputs, width), random
structure of the unused wire columns (zeroed, extra rows, values in the PoseidonGate rows), random knobs
(zero_columns / virtual_columns / half_gates toggled between proofs on one handle), every entry point (host matrix, resident,
routed when the witness allows it, sparse with a random split)."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").getcwd())
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); orc = entry.load_oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for it in range(N):
    d = int(rng.integers(5, 15)); mix = ["arith", "sha", "ecdsa", "ecdsa", "grammar"][int(rng.integers(0, 5))]
    seed = int(rng.integers(1, 1 << 30)); npi = int(rng.choice([0, 0, 1, 4, 9, 17, 30])); nw = int(rng.choice([234, 234, 135]))
    routed_only = bool(rng.integers(0, 2))
    if nw == 135 and mix == "ecdsa" and npi == 0 and d < 6: d = 6
    try:
        out = pkg.make_circuit(d, mix, seed, num_public_inputs=npi, num_wires=nw, pi_row_routed_only=routed_only)
    except Exception as e:
        print("skip", d, mix, npi, nw, e); continue
    blob, wires = out[0], out[1]; pis = out[2] if npi else ()
    oc = orc.OracleCircuit(blob); cd = pkg.CircuitData(blob)
    W, n = nw, 1 << d
    ok = True
    for rnd in range(3):
        w = wires.copy().reshape(W, n)
        mutated = False
        if rnd:  # perturb the structure of random columns (the witness may stop satisfying the circuit: bytes still comparable)
            for c_ in rng.integers(0, W, size=int(rng.integers(1, 6))):
                k = int(rng.integers(0, 4))
                if k == 0: w[c_, :] = 0
                elif k == 1: w[c_, int(rng.integers(0, n))] = int(rng.integers(1, 1 << 62))
                elif k == 2: w[c_, :] = 0; w[c_, int(rng.integers(0, n))] = 5
                else: w[c_, rng.integers(0, n, size=3)] = 9
            mutated = True
        w = np.ascontiguousarray(w)
        cd.set("self_check", 0 if mutated else 1)
        cd.set("zero_columns", int(rng.integers(0, 4) != 0))
        cd.set("virtual_columns", int(rng.integers(0, 3) != 0))
        cd.set("half_gates", int(rng.integers(0, 3)))   # round 6: gates of degree <= 4 on the even cosets only (0 never, 1 where it pays, 2 always)
        expect, _ = oc.prove(w, public_inputs=pis)
        got = {"host": cd.prove(w, public_inputs=pis).to_bytes(),
               "dev": cd.prove(torch.from_numpy(w.view(np.int64)).cuda(), public_inputs=pis).to_bytes()}
        if routed_only and not mutated:
            got["routed"] = cd.prove_routed(w[:80], public_inputs=pis).to_bytes()
        # sparse: the smallest split this matrix allows for a random row, or a random larger one
        row = int(rng.integers(0, n))
        others = np.delete(w, row, axis=1)
        dense = np.nonzero(others.any(axis=1))[0]
        lo = int(dense.max()) + 1 if dense.size else 0
        ncols = int(rng.integers(lo, W + 1))
        got["sparse"] = cd.prove_sparse(w, ncols, row, public_inputs=pis).to_bytes()
        for k_, g in got.items():
            if g != expect:
                ok = False; print("MISMATCH", k_, "round", rnd, d, mix, seed, npi, nw, routed_only)
    try:
        cd.verify(expect) if not mutated else None
        comp = cd.compress(expect); ok &= cd.decompress(comp).to_bytes() == expect
    except Exception as e:
        ok = False; print("verify/compress failed", e)
    bad += 0 if ok else 1
    cd.close(); oc.close()
print("configs", N, "mismatches", bad)
