"""N4 (partial): the reference's ACIR -> CircuitBuilder translation, restated for the opcodes BASELINE's named
circuits need (acvm-backend-plonky2_amd/translate.py): AssertZero (assert_zero_translator.rs:25-38) and
Sha256Compression (sha256_translator.rs:61-273 over binary_digits_target.rs), on top of the library's build().

The reference's own check for the SHA-256 circuit is a known-answer test
(circuit_translation/tests/test_sha256_internal.rs:481-549: the padded empty message under the IV must give
e3b0c442 98fc1c14 ... 7852b855, and the proof must verify); the same vector is required here, plus random
blocks against hashlib's compression function, plus negatives."""
import hashlib
import json
import os
import struct
import sys

import numpy as np
import pytest

from conftest import GOLDEN, P

sys.path.insert(0, GOLDEN)
import gen_proof_digests as gen  # noqa: E402
import mini_builder  # noqa: E402


def _sha256_compress(state, block_words):
    """SHA-256 compression function (FIPS 180-4), plain Python: the expected outputs for random inputs."""
    k = gen.entry.load_package().translate.SHA256_K
    w = list(block_words)
    rotr = lambda x, n: ((x >> n) | (x << (32 - n))) & 0xFFFFFFFF  # noqa: E731
    for t in range(16, 64):
        s0 = rotr(w[t - 15], 7) ^ rotr(w[t - 15], 18) ^ (w[t - 15] >> 3)
        s1 = rotr(w[t - 2], 17) ^ rotr(w[t - 2], 19) ^ (w[t - 2] >> 10)
        w.append((w[t - 16] + s0 + w[t - 7] + s1) & 0xFFFFFFFF)
    a, b, c, d, e, f, g, h = state
    for t in range(64):
        t1 = (h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g & 0xFFFFFFFF)) + k[t] + w[t]) & 0xFFFFFFFF
        t2 = ((rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))) & 0xFFFFFFFF
        a, b, c, d, e, f, g, h = (t1 + t2) & 0xFFFFFFFF, a, b, c, (d + t1) & 0xFFFFFFFF, e, f, g
    return [(x + y) & 0xFFFFFFFF for x, y in zip(state, (a, b, c, d, e, f, g, h))]


@pytest.fixture(scope="module")
def sha_circuit(pkg):
    return gen.sha256_compression_circuit(pkg)


def test_sha256_compression_circuit_reference_vector(pkg, sha_circuit):
    blob, wires, out = sha_circuit
    assert out == [0xe3b0c442, 0x98fc1c14, 0x9afbf4c8, 0x996fb924, 0x27ae41e4, 0x649b934c, 0xa495991b, 0x7852b855]
    assert bytes.fromhex("".join("%08x" % v for v in out)) == hashlib.sha256(b"").digest()
    h = blob[:256].view(np.uint32)
    # ~3.7e5 arithmetic ops in 20-op ArithmeticGates + 24 + 8 BaseSum<2> rows: 2^15 gates = 2^18 LDE rows
    assert int(h[2]) == 15 and int(h[24]) == 0 and int(h[23]) == 6
    with open(os.path.join(GOLDEN, "proof_digests_hand.json")) as f:
        g = {x["name"]: x for x in json.load(f)}["sha256_compression"]
    assert hashlib.sha256(blob.tobytes()).hexdigest() == g["blob_sha256"]
    assert hashlib.sha256(wires.tobytes()).hexdigest() == g["wires_sha256"]


def test_sha256_compression_circuit_random_blocks(pkg):
    rng = np.random.default_rng(99)
    block = [int(x) for x in rng.integers(0, 1 << 32, size=16)]
    state = [int(x) for x in rng.integers(0, 1 << 32, size=8)]
    blob, wires, out = gen.sha256_compression_circuit(pkg, block, state)
    assert out == _sha256_compress(state, block)
    # the solver's output witnesses are checked against what the circuit forces: a wrong one is refused
    cb = pkg.translate.CircuitBuilderFromAcirToPlonky2()
    cb.translate_circuit([("sha256_compression", list(range(16)), list(range(16, 24)), list(range(24, 32)))])
    wit = {i: v for i, v in enumerate(block + state)}
    wit[24] = (out[0] + 1) & 0xFFFFFFFF
    with pytest.raises(ValueError):
        cb.build(wit)
    wit[24] = out[0]
    wit[3] = 1 << 32            # not a 32-bit word: the 32-limb decomposition cannot hold it
    with pytest.raises(ValueError):
        cb.build(wit)


def test_assert_zero_translation_equals_the_mini_builder(pkg):
    """Two independent restatements of the same builder calls (tests/golden/mini_builder.py is pure Python with
    its own blob assembly; translate.py goes through the library's p2gpu_build_blob): identical circuit blobs."""
    tr = pkg.translate
    cb = tr.CircuitBuilderFromAcirToPlonky2()
    cb.translate_circuit([("assert_zero", [], [(1, 0)], -377)])
    blob, wires = cb.build({0: 377})
    b2, w2 = mini_builder.fibonacci()
    assert blob.tobytes() == b2.tobytes() and np.array_equal(wires, w2)
    cb = tr.CircuitBuilderFromAcirToPlonky2()
    cb.translate_circuit([("assert_zero", [(1, 0, 1)], [(P - 1, 2)], 0), ("assert_zero", [], [(3, 0), (2, 1)], -17)])
    blob, wires = cb.build({0: 3, 1: 4, 2: 12})
    assert blob.tobytes() == mini_builder.quadratic_example()[0].tobytes()
    with pytest.raises(ValueError):
        cb2 = tr.CircuitBuilderFromAcirToPlonky2()
        cb2.translate_circuit([("assert_zero", [], [(1, 0)], -377)])
        cb2.build({0: 376})     # not F(14)


def test_oracle_proves_the_sha256_circuit(orc, sha_circuit):
    """The named BASELINE circuit through the CPU restatement: prove, verify, digests as committed."""
    blob, wires, _ = sha_circuit
    with open(os.path.join(GOLDEN, "proof_digests_hand.json")) as f:
        g = {x["name"]: x for x in json.load(f)}["sha256_compression"]
    oc = orc.OracleCircuit(blob)
    proof, _ = oc.prove(wires)
    assert oc.verify(proof) and hashlib.sha256(proof).hexdigest() == g["proof_sha256"]


@pytest.mark.gpu
def test_gpu_proves_the_sha256_circuit(pkg, sha_circuit):
    """The same on the MI355X: bit-exact against the oracle's proof, stage by stage, via the committed digests."""
    import proof_stages

    blob, wires, _ = sha_circuit
    with open(os.path.join(GOLDEN, "proof_digests_hand.json")) as f:
        g = {x["name"]: x for x in json.load(f)}["sha256_compression"]
    cd = pkg.CircuitData(blob)
    assert cd.circuit_digest().hex() == g["circuit_digest"]
    proof = cd.prove(wires)
    assert proof_stages.first_difference(blob, proof.to_bytes(), g["stages"]) is None
    assert hashlib.sha256(proof.to_bytes()).hexdigest() == g["proof_sha256"]
    cd.verify(proof)
    cd.close()


def test_range_and_xor_opcodes(pkg, orc):
    """BlackBoxFuncCall::RANGE / AND / XOR (circuit_translation/mod.rs:131-155, 222-238), the way the reference's
    test_blackbox.rs exercises them: right values prove and verify, an out-of-range witness is refused."""
    tr = pkg.translate
    cb = tr.CircuitBuilderFromAcirToPlonky2()
    cb.translate_circuit([("range", 0, 8), ("range", 1, 33), ("and", 0, 2, 3, 8), ("xor", 0, 2, 4, 32),
                          ("assert_zero", [], [(1, 3), (1, 4), (P - 1, 5)], 0)])
    a, c = 0xB7, 0x5D
    blob, wires = cb.build({0: a, 1: (1 << 33) - 1, 2: c, 5: (a & c) + (a ^ c)})   # the ACVM solver hands over every witness
    assert cb.witness_value(3) == a & c and cb.witness_value(4) == a ^ c and cb.witness_value(5) == (a & c) + (a ^ c)
    oc = orc.OracleCircuit(blob)
    proof, _ = oc.prove(wires)
    assert oc.verify(proof)
    for bad in ({0: 256, 1: 5, 2: c}, {0: a, 1: 1 << 33, 2: c}):          # 256 is not 8 bits, 2^33 not 33 bits
        cb2 = tr.CircuitBuilderFromAcirToPlonky2()
        cb2.translate_circuit([("range", 0, 8), ("range", 1, 33), ("and", 0, 2, 3, 8)])
        with pytest.raises(ValueError):
            cb2.build(bad)


def test_identical_arithmetic_operations_share_their_output(pkg):
    """plonky2's CircuitBuilder::arithmetic keeps `base_arithmetic_results`: the same (const_0, const_1, multiplicand_0,
    multiplicand_1, addend) returns the earlier output target instead of a new ArithmeticGate slot -- so does the restated
    builder, otherwise gate counts, sigma and the circuit digest drift from the reference's build()."""
    cb = pkg.translate.CircuitBuilder()
    x, y, z = cb.add_virtual_target(), cb.add_virtual_target(), cb.add_virtual_target()
    a = cb.mul_add(x, y, z)
    ops_before = sum(len(r["ops"]) for r in cb.rows if r["kind"] == "arith")
    assert cb.mul_add(x, y, z) == a                      # cached
    assert cb.arithmetic(1, 1, x, y, z) == a
    assert sum(len(r["ops"]) for r in cb.rows if r["kind"] == "arith") == ops_before
    b = cb.mul_add(y, x, z)                              # operand order is part of the key, as upstream
    assert b != a and sum(len(r["ops"]) for r in cb.rows if r["kind"] == "arith") == ops_before + 1
    assert cb.mul(x, y) != a and cb.mul(x, y) == cb.mul(x, y)


# ---- the translator against circuits the REFERENCE built -------------------------------------------------------------------------
# The two proofs the reference ships (example_programs/basic_{if,div}/proofs/*.proof) leak their whole circuit and witness
# (tests/golden/reference_proofs.py).  Decoding the ArithmeticGate rows of those circuits with the reference's own translator in
# hand (assert_zero_translator.rs:25-38: constant first, then the linear terms, then the quadratic ones, every step a
# builder.add(term, acc)) gives back the ACIR the programs had been compiled to -- src/main.nr of each program says the same
# thing in Noir -- and the restated translator must then rebuild the reference's circuit BIT FOR BIT: gate table, selector
# columns, gate constants (ConstantGate rows in the order of the constants' canonical values), sigma polynomials, and every wire
# value except the ones build() randomises in the PublicInputGate row.  The proofs predate wide_ecc_config: 135 wires.
def _reference_programs():
    a, b, cond, ncond, t1, t2, r, ret = range(8)
    basic_if = dict(            # fn main(a, b, cond: bool) -> pub Field { if cond { a } else { b } }    a = 4, b = 2, cond = 1
        opcodes=[("range", cond, 1),
                 ("assert_zero", [], [(P - 1, cond), (P - 1, ncond)], 1),          # ncond = 1 - cond
                 ("assert_zero", [(1, a, cond)], [(P - 1, t1)], 0),                 # t1 = a * cond
                 ("assert_zero", [(1, b, ncond)], [(P - 1, t2)], 0),                # t2 = b * ncond
                 ("assert_zero", [], [(P - 1, r), (1, t1), (1, t2)], 0),           # r = t1 + t2
                 ("assert_zero", [], [(1, r), (P - 1, ret)], 0)],                  # return value
        public=(), private=(a, b, cond), witness={a: 4, b: 2, cond: 1, ncond: 0, t1: 4, t2: 0, r: 4, ret: 4})
    x, y, inv, q, ret = range(5)
    half = pow(2, P - 2, P)
    basic_div = dict(           # fn main(x, y: pub Field) -> pub Field { y / x }    x = 2, y = 1
        opcodes=[("assert_zero", [(1, x, inv)], [], P - 1),                        # x * inv = 1 (inv comes from a Brillig call)
                 ("assert_zero", [(1, y, inv)], [(P - 1, q)], 0),                   # q = y * inv
                 ("assert_zero", [], [(1, q), (P - 1, ret)], 0)],
        public=(y,), private=(x,), witness={x: 2, y: 1, inv: half, q: half, ret: half})
    return {"basic_if": basic_if, "basic_div": basic_div}


@pytest.mark.parametrize("name", ["basic_if", "basic_div"])
def test_translator_rebuilds_the_reference_circuits(pkg, orc, name):
    sys.path.insert(0, GOLDEN)
    import reference_proofs as rp

    ref, prog = rp.ReferenceCase(name), _reference_programs()[name]
    cb = pkg.translate.CircuitBuilderFromAcirToPlonky2(num_wires=135)
    cb.translate_circuit(prog["opcodes"], public_parameters=prog["public"], private_parameters=prog["private"])
    blob, wires = cb.build(prog["witness"])
    assert blob.tobytes() == ref.blob().tobytes()          # header, gate table, k_is, selectors + gate constants, sigmas
    assert cb.public_inputs() == list(ref.public_inputs)
    pi_row = cb.builder.pi_row
    mine, theirs = wires.reshape(135, -1).copy(), ref.wires.copy()
    assert (theirs[4:, pi_row] != 0).all()                 # build() randomised every unused wire of the PublicInputGate row
    mine[4:, pi_row] = theirs[4:, pi_row]                  # ... which is the one thing a rebuild cannot reproduce
    assert np.array_equal(mine, theirs)
    # and from there the reference's proof itself: same circuit, same witness, the reference's PoW witness
    oc = orc.OracleCircuit(blob)
    proof, _ = oc.prove(mine, public_inputs=ref.public_inputs, pow_hint=ref.pow_witness)
    assert proof == ref.uncompressed()
    oc.close()


def test_public_parameters_are_hashed_in_circuit(pkg, orc):
    """Public parameters (mod.rs:290-310) in the shape the reference's own opcode tests use them (test_assert_zero.rs:72-103:
    3 x + 4 y = 7 with both witnesses public, 9 more to cross the 8-input rate of the in-circuit sponge): one PoseidonGate row
    per 8 inputs, the proof echoes the public inputs, the verifier accepts, another value is refused."""
    tr = pkg.translate
    cb = tr.CircuitBuilderFromAcirToPlonky2()
    pub = list(range(11))
    ops = [("assert_zero", [], [(3, 0), (4, 1)], P - 7)] + [("assert_zero", [], [(P - 1, i), (1, i + 1)], P - 1) for i in range(2, 10)]
    cb.translate_circuit(ops, public_parameters=pub)
    wit = {0: 1, 1: 1}
    wit.update({i: 100 + i for i in range(2, 11)})
    blob, wires = cb.build(wit)
    assert sum(1 for r in cb.builder.rows if r["kind"] == "poseidon") == 2 and int(blob[:256].view(np.uint32)[24]) == 11
    pis = cb.public_inputs()
    assert pis == [1, 1] + list(range(102, 111))
    oc = orc.OracleCircuit(blob)
    proof, _ = oc.prove(wires, public_inputs=pis)
    assert oc.verify(proof) and proof[-8 * 11:] == b"".join(int(v).to_bytes(8, "little") for v in pis)
    bad = list(pis)
    bad[3] += 1
    assert not oc.verify(proof[:-8 * 11] + b"".join(int(v).to_bytes(8, "little") for v in bad))
    oc.close()
    # the reference keeps the parameters in BTreeSets (mod.rs:290-303): a caller's order (or a duplicate) does not matter
    cb2 = tr.CircuitBuilderFromAcirToPlonky2()
    cb2.translate_circuit(ops, public_parameters=[7, 3, 10, 0, 1, 9, 2, 8, 4, 6, 5, 3])
    blob2, wires2 = cb2.build(wit)
    assert cb2.public_inputs() == pis and np.array_equal(blob2, blob) and np.array_equal(wires2, wires)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["basic_if", "basic_div"])
def test_gpu_proves_the_translated_reference_programs(pkg, name):
    """Opcodes -> restated translator -> p2gpu_build_blob -> MI355X: the reference's own proof bytes (its PoW witness as hint)."""
    sys.path.insert(0, GOLDEN)
    import reference_proofs as rp

    ref, prog = rp.ReferenceCase(name), _reference_programs()[name]
    cb = pkg.translate.CircuitBuilderFromAcirToPlonky2(num_wires=135)
    cb.translate_circuit(prog["opcodes"], public_parameters=prog["public"], private_parameters=prog["private"])
    blob, wires = cb.build(prog["witness"])
    w = wires.reshape(135, -1).copy()
    w[4:, cb.builder.pi_row] = ref.wires[4:, cb.builder.pi_row]
    cd = pkg.CircuitData(blob)
    cd.set("pow_hint", ref.pow_witness)
    proof = cd.prove(np.ascontiguousarray(w), public_inputs=cb.public_inputs())
    assert proof.to_bytes() == ref.uncompressed()
    assert cd.compress(proof.to_bytes()) == ref.compressed      # ... and the file the reference's CLI wrote, byte for byte
    cd.close()
