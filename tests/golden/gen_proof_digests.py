"""Regenerates tests/golden/proof_digests.json: SHA-256 of the oracle's proof bytes (minimum
PoW witness) for small synthetic circuits, plus their transcript challenges.  These pin the
oracle + workload generator across rounds (regression vectors); they are NOT reference outputs
-- the reference prover cannot run here.  Run from the repo root:
    python tests/golden/gen_proof_digests.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

CASES = [(5, "arith", 1), (6, "sha", 2), (7, "ecdsa", 3), (9, "ecdsa", 4), (10, "arith", 5)]


def main():
    entry.build()
    pkg, orc = entry.load_package(), entry.load_oracle()
    out = []
    for d, mix, seed in CASES:
        blob, wires = pkg.make_circuit(d, mix, seed)
        oc = orc.OracleCircuit(blob)
        proof, tr = oc.prove(wires)
        assert oc.verify(proof)
        out.append({
            "degree_bits": d, "mix": mix, "seed": seed,
            "blob_sha256": hashlib.sha256(blob.tobytes()).hexdigest(),
            "wires_sha256": hashlib.sha256(wires.tobytes()).hexdigest(),
            "constants_sigmas_cap_sha256": hashlib.sha256(oc.cap()).hexdigest(),
            "circuit_digest": oc.digest().hex(),
            "proof_len": len(proof),
            "proof_sha256": hashlib.sha256(proof).hexdigest(),
            "betas": [int(x) for x in tr.betas[:2]], "gammas": [int(x) for x in tr.gammas[:2]],
            "alphas": [int(x) for x in tr.alphas[:2]], "zeta": [int(x) for x in tr.zeta],
            "pow_witness": int(tr.pow_witness),
        })
    path = os.path.join(ROOT, "tests", "golden", "proof_digests.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
