"""Regenerates tests/golden/proof_digests.json (small circuits) and, with --large,
tests/golden/proof_digests_large.json (the BASELINE sizes): SHA-256 of the oracle's proof bytes
(minimum PoW witness) and of every prover stage inside them (tests/golden/proof_stages.py), plus the
transcript challenges.  These pin the oracle + workload generator across rounds and let the GPU
tests demand bit-exactness at 2^20 / 2^22 LDE rows without running the oracle on the GPU box; they
are NOT reference outputs -- the reference prover cannot run here.  Run from the
repo root (the --large pass takes ~10 min and ~25 GB on 8 cores):
    python tests/golden/gen_proof_digests.py [--large]
--xlarge adds BASELINE configs[4] (2^24 LDE rows) to proof_digests_large.json without redoing the entries it holds.  The oracle's
row-major LDE of the wires is 31 GB there (11 GB for constants/sigmas) on a 62 GB build box: run it with
    ORC_SPILL_DIR=/tmp/orc_spill python tests/golden/gen_proof_digests.py --xlarge
(oracle/poly.c spill_malloc: those two buffers become file mappings; ~25 GB of RAM, ~45 GB of disk, ~20 min on 8 cores).
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mini_builder  # noqa: E402
import proof_stages  # noqa: E402

# (degree_bits, mix, seed, num_public_inputs)
CASES = [(5, "arith", 1, 0), (6, "sha", 2, 0), (7, "ecdsa", 3, 0), (9, "ecdsa", 4, 0), (10, "arith", 5, 0)]
# BASELINE.json configs[2] (SHA256 ~2^20 LDE rows: the bench workload, seed 1), the same size with every
# gate kind and with public inputs (PoseidonGate), and configs[3] (EcdsaSecp256k1 ~2^22 LDE rows)
LARGE = [(17, "sha", 1, 0), (17, "ecdsa", 1, 0), (17, "sha", 3, 4), (19, "ecdsa", 2, 0)]
# BASELINE.json configs[4] (zk-grammar ~2^24 LDE rows: the `grammar` mix -- arithmetic + SHA bit logic + RandomAccess /
# memory rows), and the bench workload's mix at the same size
XLARGE = [(21, "grammar", 1, 0), (21, "sha", 1, 0)]
# hand-written ACIR-equivalents (BASELINE configs[0]: the fibonacci example program)
HAND = {"fibonacci": mini_builder.fibonacci, "quadratic_example": mini_builder.quadratic_example}
SHA256_IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def sha256_compression_circuit(pkg, block=None, state=None):
    """BASELINE configs[2]'s named circuit: ONE Sha256Compression opcode through the package's restatement of the
    reference's translator (acvm-backend-plonky2_amd/translate.py; sha256_translator.rs:61-273).  Default
    input = the reference's own test vector (circuit_translation/tests/test_sha256_internal.rs:481-549): the
    padded empty message under the SHA-256 IV.  Returns (blob, wires, the 8 output words the circuit forces)."""
    cb = pkg.translate.CircuitBuilderFromAcirToPlonky2()
    cb.translate_circuit([("sha256_compression", list(range(16)), list(range(16, 24)), list(range(24, 32)))])
    block = [1 << 31] + [0] * 15 if block is None else block
    state = SHA256_IV if state is None else state
    wit = {i: v for i, v in enumerate(block)}
    wit.update({16 + i: v for i, v in enumerate(state)})
    blob, wires = cb.build(wit)
    return blob, wires, [cb.witness_value(24 + i) for i in range(8)]


def record(oc, blob, wires, pis, meta):
    proof, tr = oc.prove(wires, public_inputs=pis)
    assert oc.verify(proof)
    meta.update({
        "blob_sha256": hashlib.sha256(blob.tobytes()).hexdigest(),
        "wires_sha256": hashlib.sha256(wires.tobytes()).hexdigest(),
        "constants_sigmas_cap_sha256": hashlib.sha256(oc.cap()).hexdigest(),
        "circuit_digest": oc.digest().hex(),
        "proof_len": len(proof),
        "proof_sha256": hashlib.sha256(proof).hexdigest(),
        "stages": proof_stages.stage_digests(blob, proof),
        "betas": [int(x) for x in tr.betas[:2]], "gammas": [int(x) for x in tr.gammas[:2]],
        "alphas": [int(x) for x in tr.alphas[:2]], "zeta": [int(x) for x in tr.zeta],
        "pow_witness": int(tr.pow_witness),
    })
    return meta


def synth_cases(pkg, orc, cases):
    out = []
    for d, mix, seed, npi in cases:
        res = pkg.make_circuit(d, mix, seed, num_public_inputs=npi)
        blob, wires = res[0], res[1]
        pis = res[2] if npi else ()
        oc = orc.OracleCircuit(blob)
        out.append(record(oc, blob, wires, pis, {"degree_bits": d, "mix": mix, "seed": seed, "public_inputs": npi}))
        oc.close()
        print("done", d, mix, seed, npi, flush=True)
    return out


def main():
    entry.build()
    pkg, orc = entry.load_package(), entry.load_oracle()
    gold = os.path.join(ROOT, "tests", "golden")
    if "--xlarge" in sys.argv:
        path = os.path.join(gold, "proof_digests_large.json")
        have = json.load(open(path))
        for case in XLARGE:
            if any((g["degree_bits"], g["mix"], g["seed"], g["public_inputs"]) == case for g in have):
                continue
            have += synth_cases(pkg, orc, [case])
            with open(path, "w") as f:
                json.dump(have, f, indent=1)
        return
    if "--large" in sys.argv:
        with open(os.path.join(gold, "proof_digests_large.json"), "w") as f:
            json.dump(synth_cases(pkg, orc, LARGE), f, indent=1)
        return
    with open(os.path.join(gold, "proof_digests.json"), "w") as f:
        json.dump(synth_cases(pkg, orc, CASES), f, indent=1)
    hand = []
    for name, fn in HAND.items():
        blob, wires = fn()
        hand.append(record(orc.OracleCircuit(blob), blob, wires, (), {"name": name}))
    blob, wires, out = sha256_compression_circuit(pkg)
    hand.append(record(orc.OracleCircuit(blob), blob, wires, (), {"name": "sha256_compression", "outputs": out}))
    with open(os.path.join(gold, "proof_digests_hand.json"), "w") as f:
        json.dump(hand, f, indent=1)


if __name__ == "__main__":
    main()
