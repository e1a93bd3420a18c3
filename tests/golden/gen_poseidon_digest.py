"""tests/golden/proof_digests_poseidon.json: the ORACLE's proof of the bench workload under PoseidonGoldilocksConfig (blob hasher 1:
Poseidon Merkle trees, challenger, PoW, circuit digest) at BASELINE configs[2]'s size, 2^20 LDE rows, and of a smaller circuit with
public inputs -- SHA-256 of the proof bytes, the cap and the circuit digest, the PoW witness.  Lets the GPU suite demand
bit-exactness of the X1 mode at full size without running the oracle on the GPU box (the oracle's plain-form Poseidon takes
minutes there).  Not reference output (the reference runs KeccakGoldilocksConfig, lib.rs:13).  Run from the repo root:
    python tests/golden/gen_poseidon_digest.py          (~3 min on 8 cores)"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

CASES = [(13, "ecdsa", 5, 3), (17, "sha", 1, 0)]


def main():
    pkg, orc = entry.load_package(), entry.load_oracle()
    out = []
    for d, mix, seed, npi in CASES:
        res = pkg.make_circuit(d, mix, seed, num_public_inputs=npi, hasher=1)
        blob, wires = res[0], res[1]
        pis = res[2] if npi else ()
        oc = orc.OracleCircuit(blob)
        t0 = time.time()
        proof, tr = oc.prove(wires, public_inputs=pis)
        assert oc.verify(proof)
        out.append({"degree_bits": d, "mix": mix, "seed": seed, "public_inputs": npi, "hasher": 1,
                    "blob_sha256": hashlib.sha256(blob.tobytes()).hexdigest(), "wires_sha256": hashlib.sha256(wires.tobytes()).hexdigest(),
                    "constants_sigmas_cap_sha256": hashlib.sha256(oc.cap()).hexdigest(), "circuit_digest": oc.digest().hex(),
                    "proof_len": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(), "pow_witness": int(tr.pow_witness),
                    "oracle_seconds": round(time.time() - t0, 1)})
        print(out[-1], flush=True)
        oc.close()
    with open(os.path.join(ROOT, "tests", "golden", "proof_digests_poseidon.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
