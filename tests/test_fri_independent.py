"""A third, independent reader of the FRI part of the proofs: tests/golden/fri_check.py (pure Python,
own Keccak, Lagrange interpolation from the definition; shares nothing with oracle/ or the product).

The reference ships no proof with a FRI reduction step (its two .proof files are 2^3-row circuits), so
the arity-16 fold is otherwise checked only by verifiers written alongside the prover.  Here:
  * on the reference's own proofs, the Python transcript + batched-quotient + final_poly check passes on
    the reference's BYTES (so this reader is itself pinned to the reference where the reference reaches);
  * on oracle proofs with 1, 2 and 3 reduction steps every fold holds from the definition, and a
    corrupted fold value / step path / query row is caught;
  * (-m gpu) the same on GPU proofs at the BASELINE sizes, incl. 2^22 rows = 4 steps, final_poly 2^3.
"""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import fri_check  # noqa: E402
import proof_stages  # noqa: E402
import reference_proofs as rp  # noqa: E402


def _cap_list(cap_bytes):
    return [cap_bytes[i:i + 25] for i in range(0, len(cap_bytes), 25)]


@pytest.mark.parametrize("name", ["basic_if", "basic_div"])
def test_python_reader_accepts_the_reference_proofs(orc, name):
    """The reference's own prover output (decompressed in pure Python): transcript, PoW, query indices,
    Merkle paths, batched quotient and final_poly all check out -- zero reduction steps, but everything
    else in fri_check.py runs on reference bytes."""
    case = rp.ReferenceCase(name)
    blob = case.blob()
    proof = case.uncompressed()
    oc = orc.OracleCircuit(blob)   # only to obtain the verifier key (constants_sigmas cap + digest)
    c = proof_stages.header(blob)
    assert c["steps"] == 0
    pi_hash = (0, 0, 0, 0)
    if case.public_inputs:
        out = np.zeros(4, dtype=np.uint64)
        pis = np.array(case.public_inputs, dtype=np.uint64)
        orc.lib().orc_poseidon_hash_no_pad(pis.ctypes.data, len(pis), out.ctypes.data)
        pi_hash = tuple(int(v) for v in out)
    res = fri_check.check(c, proof, oc.digest(), _cap_list(oc.cap()), pi_hash=pi_hash)
    assert res["query_indices"] == case.pr["indices"]          # the indices the reference file itself records
    assert res["zeta"] == tuple(case.zeta)                     # zeta recovered from the identity-sigma opening


@pytest.mark.parametrize("d,mix,steps", [(6, "sha", 1), (10, "ecdsa", 2), (14, "arith", 3)])
def test_folds_hold_from_the_definition(pkg, orc, d, mix, steps):
    blob, wires = pkg.make_circuit(d, mix, 41)
    oc = orc.OracleCircuit(blob)
    proof, tr = oc.prove(wires)
    c = proof_stages.header(blob)
    assert c["arity"] == [4] * steps
    cap = _cap_list(oc.cap())
    res = fri_check.check(c, proof, oc.digest(), cap)
    # the Python transcript equals the oracle's, challenge for challenge
    assert res["betas"] == [int(x) for x in tr.betas[:2]] and res["zeta"] == tuple(tr.zeta)
    assert res["fri_betas"] == [tuple(tr.fri_betas[i]) for i in range(steps)]
    assert res["query_indices"] == list(tr.query_indices[:28])
    # corruptions that leave the transcript alone must be caught by the fold / path checks
    st = proof_stages.stages(blob, proof)
    q0 = sum(len(st[k]) for k in ("wires_cap", "zs_partial_products_cap", "quotient_polys_cap", "openings", "fri_commit_caps"))
    init_len = sum(8 * n + 1 + 25 * (d + 3 - 4) for n in (c["NC"] + c["R"], c["W"], 20, 16))
    for what, off in (("initial row", q0 + 8), ("fold value", q0 + init_len + 16 * 5), ("fold path", q0 + init_len + 256 + 1 + 3)):
        bad = bytearray(proof)
        bad[off] ^= 1
        with pytest.raises(AssertionError):
            fri_check.check(c, bytes(bad), oc.digest(), cap)
        assert not oc.verify(bytes(bad)), what
    # a wrong verifier key is caught too
    with pytest.raises(AssertionError):
        fri_check.check(c, proof, oc.digest(), cap[1:] + cap[:1])


@pytest.mark.gpu
@pytest.mark.parametrize("d,mix", [(17, "sha"), (19, "ecdsa")])
def test_gpu_proof_folds_hold_from_the_definition(pkg, d, mix):
    """2^20 LDE rows: 3 steps, final_poly 2^5; 2^22 rows: 4 steps, final_poly 2^3."""
    blob, wires = pkg.make_circuit(d, mix, 1 if d == 17 else 2)
    cd = pkg.CircuitData(blob)
    proof = cd.prove(wires).to_bytes()
    c = proof_stages.header(blob)
    assert len(c["arity"]) == (3 if d == 17 else 4)
    fri_check.check(c, proof, cd.circuit_digest(), _cap_list(cd.constants_sigmas_cap()))
    cd.close()
