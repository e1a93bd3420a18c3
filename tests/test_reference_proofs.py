"""Parity against the reference's OWN prover output.

The reference ships two proofs written by its `prove` action (fixtures:
tests/golden/reference/*.proof.hex, see tests/golden/reference_proofs.py for provenance and for
how the circuit and the full witness are recovered from them).  Given that circuit, that witness
and the reference's PoW witness, the oracle's prover -- and on the GPU box the product -- must
reproduce the reference's proof bytes exactly, and both verifiers must accept the reference's proof.

Pinned by these two files: Goldilocks generators, coset/LDE/bit-reversal conventions, Keccak-256/25
leaf + node hashing and cap order, the circuit digest, the whole Fiat-Shamir transcript (Keccak
duplex challenger, hash onion), Poseidon public-input hash (basic_div), permutation argument and
partial products, selector filters, Noop/Constant/PublicInput/BaseSum/Arithmetic/Poseidon gate
constraints, quotient chunking, opening set order, FRI batch reduction and final polynomial,
PoW check, query index derivation, proof byte layout.  Not reachable from them: FRI reduction
steps (2^3 rows need none), RandomAccessGate, the five custom gates.
"""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import reference_proofs as rp  # noqa: E402

NAMES = ["basic_if", "basic_div"]


@pytest.fixture(scope="module")
def cases():
    return {n: rp.ReferenceCase(n) for n in NAMES}


def test_keccak256_known_answers():
    # the fixture module's own Keccak (original padding): standard vectors
    assert rp.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert rp.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert rp.keccak256(bytes(200)).hex() != rp.keccak256(bytes(199)).hex()


@pytest.mark.parametrize("name", NAMES)
def test_artefact_is_internally_consistent(cases, name):
    """Layout, Merkle paths, low degree and openings -- no oracle, no product involved."""
    c = cases[name]
    pr = c.pr
    assert len(pr["init"]) == len(set(pr["indices"])) and max(pr["indices"]) < 64
    assert len(c.known_cap_entries()) >= 10
    assert c.openings_from_polynomials() == pr["openings"]
    # Z(1) = 1 and the public inputs sit where the PublicInputGate row's hash wires say
    g = rp.root_of_unity(rp.D)
    for k in range(rp.K):
        assert rp.poly_eval(c.polys[2][k], 1) == 1
    assert all(int(v) < rp.P for v in c.wires.ravel())
    assert c.public_inputs == ([] if name == "basic_if" else [1])
    # selector column: gate index of every row; rows 6,7 are NoopGate padding
    assert [int(v) for v in c.constants[0][-2:]] == [0, 0]
    assert g != 1


@pytest.mark.parametrize("name", NAMES)
def test_oracle_pieces_match_the_reference(cases, orc, name):
    """Each oracle primitive against data only the reference could have produced."""
    c = cases[name]
    # Keccak: the oracle's hash == the fixture module's on real leaves
    x0 = c.pr["indices"][0]
    leaf = c.pr["init"][x0][1][0]
    raw = np.array(leaf, dtype=np.uint64).tobytes()
    assert orc.keccak256(raw)[:25] == rp.keccak256(raw)[:25]
    # coset LDE: the recovered coefficients, extended by the oracle, give the opened rows
    coeffs = np.array(c.polys[1][0], dtype=np.uint64)
    lde = orc.coset_lde(coeffs, rp.RATE_BITS)
    for x in c.pr["indices"]:
        assert int(lde[rp.bitrev(x, 6)]) == c.pr["init"][x][1][0][0]
    # constants_sigmas commitment: every cap entry the proof reveals
    oc = orc.OracleCircuit(c.blob())
    cap = oc.cap()
    for i, v in c.known_cap_entries().items():
        assert cap[25 * i:25 * i + 25] == v


@pytest.mark.parametrize("name", NAMES)
def test_oracle_verifier_accepts_reference_proof(cases, orc, name):
    c = cases[name]
    oc = orc.OracleCircuit(c.blob())
    ref = c.uncompressed()
    assert oc.verify(ref)
    bad = bytearray(ref)
    bad[1300] ^= 1
    assert not oc.verify(bytes(bad))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_prover_reproduces_reference_proof(cases, orc, name):
    """Bit-exact: same circuit, same witness, the reference's PoW witness -> the reference's bytes."""
    c = cases[name]
    oc = orc.OracleCircuit(c.blob())
    proof, tr = oc.prove(c.wires, public_inputs=c.public_inputs, pow_hint=c.pow_witness)
    assert [int(v) for v in tr.zeta] == list(c.zeta)
    assert [int(v) for v in tr.query_indices[:rp.QUERIES]] == c.pr["indices"]
    assert proof == c.uncompressed()
    # grinding for the minimum witness instead changes only what depends on it
    mine, tr2 = oc.prove(c.wires, public_inputs=c.public_inputs)
    assert int(tr2.pow_witness) <= c.pow_witness and oc.verify(mine)
    assert mine[:3 * 16 * 25 + 16 * len(c.pr["openings"])] == proof[:3 * 16 * 25 + 16 * len(c.pr["openings"])]


@pytest.mark.parametrize("name", NAMES)
def test_product_verifier_accepts_reference_proof(cases, orc, pkg, name):
    """libp2gpu's host verifier (no GPU needed) on the reference's proof.  The verifier key needs
    the full constants_sigmas cap; the proof reveals 12 of its 16 entries (checked above against
    the oracle's), the rest come from the oracle's commitment of the recovered columns."""
    from test_verifier import vk_blob

    c = cases[name]
    blob = c.blob()
    oc = orc.OracleCircuit(blob)
    vd = pkg.VerifierCircuitData(vk_blob(blob, oc.cap(), oc.digest()))
    ref = c.uncompressed()
    vd.verify(ref)
    bad = bytearray(ref)
    bad[-9] ^= 1  # the PoW witness
    with pytest.raises(pkg.P2GpuError):
        vd.verify(bytes(bad))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_prover_reproduces_reference_proof(cases, pkg, name):
    """The product on MI355X, through the C ABI: the reference's bytes, exactly."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    c = cases[name]
    cd = pkg.CircuitData(c.blob())
    cap = cd.constants_sigmas_cap()
    for i, v in c.known_cap_entries().items():
        assert cap[25 * i:25 * i + 25] == v
    cd.set("pow_hint", c.pow_witness)
    proof = cd.prove(c.wires, public_inputs=c.public_inputs)
    assert proof.to_bytes() == c.uncompressed()
    cd.verify(c.uncompressed())
    cd.verifier_data().verify(c.uncompressed())
    cd.close()


@pytest.mark.parametrize("name", NAMES)
def test_compressed_format_round_trips_the_reference_file(cases, orc, pkg, name):
    """N3: the product's compress / decompress / verify_compressed on the reference's own file
    (`proof.compress(..).to_bytes()`, prove_action.rs:75-78): decompress gives the bytes the
    fixture module rebuilt independently, compress gives the file back, verify_compressed accepts."""
    from test_verifier import vk_blob

    c = cases[name]
    blob = c.blob()
    oc = orc.OracleCircuit(blob)
    vd = pkg.VerifierCircuitData(vk_blob(blob, oc.cap(), oc.digest()))
    assert vd.decompress(c.compressed).to_bytes() == c.uncompressed()
    assert vd.compress(c.uncompressed()) == c.compressed
    vd.verify_compressed(c.compressed)
    bad = bytearray(c.compressed)
    bad[len(bad) // 2] ^= 4
    with pytest.raises(pkg.P2GpuError):
        vd.verify_compressed(bytes(bad))
    with pytest.raises(pkg.P2GpuError):
        vd.verify_compressed(c.compressed[:-3])


@pytest.mark.parametrize("name", NAMES)
def test_verify_tool_accepts_the_reference_file_as_shipped(cases, orc, pkg, tmp_path, name):
    """`p2gpu-verify <vk> <file>` on the reference's .proof file exactly as it lies in its tree
    (hex text of the compressed proof): the plain-C counterpart of `plonky2-backend verify`."""
    import subprocess

    from conftest import ROOT
    from test_verifier import vk_blob

    c = cases[name]
    blob = c.blob()
    oc = orc.OracleCircuit(blob)
    (tmp_path / "vk.blob").write_bytes(vk_blob(blob, oc.cap(), oc.digest()))
    exe = os.path.join(ROOT, "acvm-backend-plonky2_amd", "p2gpu-verify")
    src = os.path.join(GOLDEN, "reference", name + ".proof.hex")
    r = subprocess.run([exe, str(tmp_path / "vk.blob"), src], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "accepted" in r.stderr, r.stderr
    text = open(src).read()
    (tmp_path / "bad.hex").write_text(text[:5000] + ("0" if text[5000] != "0" else "1") + text[5001:])
    r = subprocess.run([exe, str(tmp_path / "vk.blob"), str(tmp_path / "bad.hex")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3, r.stderr


def _pi_row(c):
    # the PublicInputGate row: its unused wires hold RandomValueGenerator output (not derivable)
    gi = [g[0] for g in rp.CASES[c.name]["gates"]].index(rp.G_PUBLIC_INPUT)
    rows = [r for r in range(1 << rp.D) if int(c.constants[0][r]) == gi]
    assert len(rows) == 1
    return rows[0]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_generators_rebuild_the_reference_witness(cases, orc, name):
    """N1 against the reference's real witness: drop every gate-internal (non-routed) column, let the
    row-local generators refill them -- the PoseidonGate row of basic_div (all 135 wires), the
    BaseSum/Arithmetic rows -- and get the reference's wire matrix back."""
    c = cases[name]
    oc = orc.OracleCircuit(c.blob())
    part = c.wires.copy()
    part[rp.R:, :] = 0
    got = oc.fill_witness(part)
    keep = [r for r in range(1 << rp.D) if r != _pi_row(c)]
    assert np.array_equal(got[:, keep], c.wires[:, keep])
    assert np.array_equal(got[:rp.R], c.wires[:rp.R])
    if name == "basic_div":
        assert np.count_nonzero(c.wires[rp.R:, keep]) >= 50  # the Poseidon row really has internal wires


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_generators_rebuild_the_reference_witness(cases, pkg, name):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    c = cases[name]
    cd = pkg.CircuitData(c.blob())
    part = c.wires.copy()
    part[rp.R:, :] = 0
    dev = torch.from_numpy(part.view(np.int64)).cuda()
    cd.fill_witness(dev)
    got = dev.cpu().numpy().view(np.uint64)
    keep = [r for r in range(1 << rp.D) if r != _pi_row(c)]
    assert np.array_equal(got[:, keep], c.wires[:, keep])
    cd.close()


def _batch_values(c, tree):
    """Values on the subgroup H of the columns of prover tree 1 (wires), 2 (Z + partial products) or
    3 (quotient chunks), from the recovered coefficients."""
    g = rp.root_of_unity(rp.D)
    H = [pow(g, i, rp.P) for i in range(1 << rp.D)]
    return np.array([[rp.poly_eval(col, h) for h in H] for col in c.polys[tree]], dtype=np.uint64)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_commitments_equal_the_reference_caps(cases, orc, name):
    """Stage level: PolynomialBatch::from_values (iNTT -> 8x coset LDE -> Keccak tree) of the three
    prover batches gives exactly the three Merkle caps in the reference's proof -- all 16 entries."""
    c = cases[name]
    for tree in (1, 2, 3):
        cap = orc.commit_values(_batch_values(c, tree), rp.RATE_BITS, rp.CAP_H)
        assert cap == b"".join(c.pr["caps"][tree - 1]), tree


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_commitments_equal_the_reference_caps(cases, pkg, name):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    c = cases[name]
    for tree in (1, 2, 3):
        vals = _batch_values(c, tree)
        assert pkg.commit_values(vals, rp.RATE_BITS, rp.CAP_H) == b"".join(c.pr["caps"][tree - 1]), tree
        # and the LDE itself: the opened rows of the reference proof
        lde = pkg.lde_batch(pkg.ifft_batch(vals), rp.RATE_BITS)
        for x in c.pr["indices"]:
            assert [int(v) for v in lde[:, rp.bitrev(x, rp.D + rp.RATE_BITS)]] == c.pr["init"][x][tree][0]
