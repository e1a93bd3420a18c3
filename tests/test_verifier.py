"""The product's verifier (libp2gpu.so p2gpu_verify, host code only) against the oracle's prover on a
CPU-only host: two independently written implementations of the protocol must agree -- the
product's verifier accepts every proof the oracle's prover writes and rejects every proof the
oracle's verifier rejects.

Mirrors the reference's `verify` action (plonky2-backend/src/actions/verify_action.rs:11-17) and
the `circuit_data.verify(proof)` assertion its tests end with (tests/factories/utils.rs:26-27); the
negatives follow its `should_panic` tests (circuit_translation/tests/test_blackbox.rs:17,36,55).
"""
import re

import numpy as np
import pytest

from conftest import P


def vk_blob(blob, cap, digest):
    """The verifier's share of a synth circuit blob (include/p2gpu.h: header | gate table | cap |
    k_is, flags 0b11), with the cap and digest the oracle computed on the CPU."""
    h = blob[:256].copy().view(np.uint32)
    assert h[25] == 0  # synth blobs carry neither cap nor digest
    ng, R, cap_h = int(h[23]), int(h[4]), int(h[10])
    h[25] = 3
    hb = bytearray(h.tobytes())
    nb = len(digest)                  # 25 (KeccakHash<25>) or 32 (PoseidonHash)
    hb[128:128 + nb] = digest
    gates = blob[256:256 + 48 * ng].tobytes()
    capb = b"".join(cap[nb * i:nb * (i + 1)] + bytes(32 - nb) for i in range(1 << cap_h))
    k_is = blob[256 + 48 * ng:256 + 48 * ng + 8 * R].tobytes()
    return bytes(hb) + gates + capb + k_is


def make(pkg, orc, d, mix, seed, npi=0, num_wires=234, hasher=0):
    out = pkg.make_circuit(d, mix, seed, num_public_inputs=npi, num_wires=num_wires, hasher=hasher)
    blob, wires = out[0], out[1]
    pis = out[2] if npi else ()
    oc = orc.OracleCircuit(blob)
    vd = pkg.VerifierCircuitData(vk_blob(blob, oc.cap(), oc.digest()))
    return oc, vd, wires, pis


@pytest.mark.parametrize("d,mix,seed,npi,nw", [
    (5, "arith", 1, 0, 234),    # no FRI reduction step: final polynomial of 2^5 coefficients
    (6, "sha", 2, 0, 234),
    (8, "ecdsa", 3, 0, 234),
    (9, "ecdsa", 4, 4, 234),    # public inputs: PoseidonGate + PublicInputGate
    (7, "arith", 5, 1, 135),    # standard_recursion_config width
    (10, "sha", 6, 0, 234),     # two reduction steps
])
def test_accepts_oracle_proofs(pkg, orc, d, mix, seed, npi, nw):
    oc, vd, wires, pis = make(pkg, orc, d, mix, seed, npi, nw)
    proof, _ = oc.prove(wires, public_inputs=pis)
    assert oc.verify(proof)
    vd.verify(proof)  # raises when rejected
    assert vd.circuit_digest() == oc.digest() and vd.constants_sigmas_cap() == oc.cap()
    assert vd.num_public_inputs == npi and vd.degree_bits == d


def test_rejects_what_the_oracle_rejects(pkg, orc):
    """Flip one bit anywhere: caps, openings, fold caps, query rows, Merkle paths, final
    polynomial, PoW witness, public inputs.  Both verifiers must say no, every time."""
    oc, vd, wires, pis = make(pkg, orc, 8, "ecdsa", 7, npi=2)
    proof, _ = oc.prove(wires, public_inputs=pis)
    vd.verify(proof)
    rng = np.random.default_rng(5)
    n = len(proof)
    fixed = [0, 24, 25 * 16, 3 * 25 * 16, 3 * 25 * 16 + 8, n - 1, n - 8, n - 16, n - 17, n - 24, n - 25]
    positions = fixed + [int(x) for x in rng.integers(0, n, size=150)]
    reasons = set()
    for pos in positions:
        bad = bytearray(proof)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        assert not oc.verify(bad), pos
        with pytest.raises(pkg.P2GpuError) as ei:
            vd.verify(bad)
        assert ei.value.code == -9, pos
        reasons.add(re.sub(r"\d+", "#", str(ei.value).split("proof rejected: ")[1]))
    assert len(reasons) >= 4, reasons  # the mutations reached several different checks
    # wrong length
    for bad in (proof[:-1], proof + b"\0", proof[:100], b""):
        assert not oc.verify(bad) if bad else True
        with pytest.raises(pkg.P2GpuError) as ei:
            vd.verify(bad)
        assert ei.value.code in (-9, -7)


def test_rejects_proof_of_unsatisfied_witness(pkg, orc):
    """A witness that breaks a gate constraint or a copy constraint still yields proof bytes from
    the oracle's prover (upstream's prover does not check either); no verifier may accept them."""
    oc, vd, wires, _ = make(pkg, orc, 7, "ecdsa", 9)
    for (col, row) in ((3, 5), (0, 2), (100, 40)):
        bad = wires.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % P
        proof, _ = oc.prove(bad)
        assert not oc.verify(proof)
        with pytest.raises(pkg.P2GpuError) as ei:
            vd.verify(proof)
        assert ei.value.code == -9 and "vanishing" in str(ei.value)


def test_rejects_proof_for_another_circuit(pkg, orc):
    oc1, vd1, w1, _ = make(pkg, orc, 6, "sha", 2)
    oc2, vd2, w2, _ = make(pkg, orc, 6, "sha", 3)
    p1, _ = oc1.prove(w1)
    vd1.verify(p1)
    with pytest.raises(pkg.P2GpuError):
        vd2.verify(p1)


def test_verifier_handle_has_no_prover(pkg, orc):
    """A verifier-only handle holds no device state: the prove entry points refuse it (and the
    refusal is not a fallback to some CPU prover)."""
    import ctypes

    oc, vd, wires, _ = make(pkg, orc, 5, "arith", 1)
    lib = pkg.load_library()
    out = np.zeros(1 << 20, dtype=np.uint8)
    plen = ctypes.c_size_t(out.nbytes)
    w = np.ascontiguousarray(wires)
    rc = lib.p2gpu_prove(vd._h, w.ctypes.data, None, 0, out.ctypes.data, ctypes.byref(plen), None)
    assert rc == -7 and b"verifier-only" in lib.p2gpu_last_error()
    rc = lib.p2gpu_prove_routed(vd._h, w.ctypes.data, None, 0, out.ctypes.data, ctypes.byref(plen), None)
    assert rc == -7
    assert lib.p2gpu_circuit_set(vd._h, b"profile", 1) == -7


def test_verifier_blob_validation(pkg, orc):
    blob, wires = pkg.make_circuit(5, "arith", 1)
    oc = orc.OracleCircuit(blob)
    vk = vk_blob(blob, oc.cap(), oc.digest())
    with pytest.raises(pkg.P2GpuError) as ei:  # a blob without cap/digest is not a verifier key
        pkg.VerifierCircuitData(blob.tobytes())
    assert ei.value.code == -1
    with pytest.raises(pkg.P2GpuError):
        pkg.VerifierCircuitData(vk[:-8])  # truncated k_is
    with pytest.raises(pkg.P2GpuError):
        pkg.VerifierCircuitData(b"\0" * 300)
    # a tampered cap makes every proof fail at the first constants_sigmas Merkle path
    proof, _ = oc.prove(wires)
    ng = int(blob[:256].view(np.uint32)[23])
    bad = bytearray(vk)
    for i in range(16):  # every cap entry: whichever subtree the first query falls into
        bad[256 + 48 * ng + 32 * i + 3] ^= 1
    vd = pkg.VerifierCircuitData(bytes(bad))
    with pytest.raises(pkg.P2GpuError) as ei:
        vd.verify(proof)
    assert "initial oracle 0" in str(ei.value)


@pytest.mark.parametrize("d,mix,seed,npi", [(5, "arith", 1, 0), (9, "ecdsa", 4, 4), (10, "sha", 6, 0), (11, "arith", 8, 1)])
def test_compress_round_trip(pkg, orc, d, mix, seed, npi):
    """The reference's on-disk format (compressed proof, prove_action.rs:75-78) with fold steps:
    compress then decompress is the identity, the compressed proof is smaller and is accepted by
    verify_compressed; queries that share a fold-step leaf (certain at these sizes: 28 queries into
    2^(d+3-4) leaves and fewer) exercise the shared-entry / inferred-element logic."""
    oc, vd, wires, pis = make(pkg, orc, d, mix, seed, npi)
    proof, tr = oc.prove(wires, public_inputs=pis)
    comp = vd.compress(proof)
    assert len(comp) < len(proof)
    assert vd.decompress(comp).to_bytes() == proof
    vd.verify_compressed(comp)
    # layout: indices sit right after caps | openings | fold caps
    h = np.frombuffer(vd.to_bytes()[:256], dtype=np.uint32)
    nsteps, nc, w = int(h[13]), int(h[5]), int(h[3])
    at = 3 * 16 * 25 + 16 * (nc + 80 + w + 2 + 2 + 18 + 16) + nsteps * 16 * 25
    assert list(np.frombuffer(comp[at:at + 4 * 28], dtype=np.uint32)) == [int(x) for x in tr.query_indices[:28]]
    # any flipped bit is caught (by the decompressor or by the verifier behind it)
    rng = np.random.default_rng(d)
    for pos in [int(x) for x in rng.integers(0, len(comp), size=60)] + [at, at + 111]:
        bad = bytearray(comp)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        with pytest.raises(pkg.P2GpuError):
            vd.verify_compressed(bytes(bad))


def test_garbage_never_crashes_the_host_parsers(pkg, orc):
    """The verifier and the (de)compressor read untrusted bytes: truncations, random tails and pure
    noise must come back as error codes (the process surviving this test is the assertion)."""
    oc, vd, wires, pis = make(pkg, orc, 9, "ecdsa", 4, 4)
    proof, _ = oc.prove(wires, public_inputs=pis)
    comp = vd.compress(proof)
    rng = np.random.default_rng(11)
    cases = []
    for src in (proof, comp):
        for cut in [int(x) for x in rng.integers(0, len(src), size=25)]:
            cases.append(src[:cut])
            cases.append(src[:cut] + rng.bytes(len(src) - cut))
        cases.append(src + b"\0" * 7)
    cases += [rng.bytes(n) for n in (0, 1, 24, 1200, len(comp), len(proof))]
    for blob in cases:
        for fn in (vd.verify, vd.verify_compressed, vd.decompress, vd.compress):
            with pytest.raises(pkg.P2GpuError):
                fn(blob)


@pytest.mark.parametrize("d,mix,npi", [(5, "arith", 0), (8, "ecdsa", 3), (10, "sha", 0)])
def test_poseidon_hasher_oracle_prover_vs_product_verifier(pkg, orc, d, mix, npi):
    """PoseidonGoldilocksConfig (blob hasher = 1): Poseidon Merkle trees / challenger / circuit digest.  The
    oracle's prover and the product's host verifier are independent implementations of that mode too: the
    verifier accepts the oracle's proofs, rejects tampered ones, and the compressed format round-trips with
    32-byte digests.  (The reference itself runs KeccakGoldilocksConfig, plonky2-backend/src/lib.rs:13.)"""
    oc, vd, wires, pis = make(pkg, orc, d, mix, 9, npi, hasher=1)
    assert vd.hash_bytes() == 32 and len(oc.digest()) == 32
    proof, _ = oc.prove(wires, public_inputs=pis)
    assert oc.verify(proof)
    vd.verify(proof)
    comp = vd.compress(proof)
    assert len(comp) < len(proof) and vd.decompress(comp).to_bytes() == proof
    vd.verify_compressed(comp)
    rng = np.random.default_rng(3)
    for off in [int(x) for x in rng.integers(0, len(proof), size=12)]:
        bad = bytearray(proof)
        bad[off] ^= 1 << int(rng.integers(0, 8))
        assert not oc.verify(bytes(bad))
        with pytest.raises(pkg.P2GpuError):
            vd.verify(bytes(bad))
    # a Keccak verifier key does not accept a Poseidon proof of the same circuit
    ock, vdk, _, _ = make(pkg, orc, d, mix, 9, npi, hasher=0)
    with pytest.raises(pkg.P2GpuError):
        vdk.verify(proof)
    # Poseidon digests are field elements: w and w + p hash alike, so a cap word re-encoded as w + p (when that still
    # fits 64 bits) would be a second byte string for the same proof -- only the canonical encoding is accepted, in the
    # proof (first cap entry = first bytes) and in the verifier key
    w = int.from_bytes(proof[:8], "little")
    small = [i for i in range(0, 4 * 16 * 3 * 8, 8) if int.from_bytes(proof[i:i + 8], "little") < (1 << 32) - 1]
    if small:
        i = small[0]
        enc = bytearray(proof)
        enc[i:i + 8] = (int.from_bytes(proof[i:i + 8], "little") + P).to_bytes(8, "little")
        with pytest.raises(pkg.P2GpuError):
            vd.verify(bytes(enc))
    vk = bytearray(vd.to_bytes())
    capoff = 256 + 48 * int(np.frombuffer(bytes(vk[:256]), dtype=np.uint32)[23])
    vk[capoff:capoff + 8] = ((1 << 64) - 1).to_bytes(8, "little")      # >= p: not a field element
    with pytest.raises(pkg.P2GpuError):
        pkg.VerifierCircuitData(bytes(vk))


@pytest.mark.parametrize("d,mix,seed,npi,nw,hasher", [
    (5, "arith", 1, 0, 234, 0),
    (8, "ecdsa", 3, 0, 234, 0),     # all twelve gate kinds minus the public-input pair
    (9, "ecdsa", 4, 4, 234, 0),     # + PoseidonGate / PublicInputGate, two selector groups
    (7, "arith", 5, 1, 135, 0),
    (10, "sha", 6, 0, 234, 0),
    (8, "sha", 9, 0, 234, 1),       # PoseidonGoldilocksConfig: 32-byte digests
])
def test_reference_vk_file_format_round_trip(pkg, orc, d, mix, seed, npi, nw, hasher):
    """`write_vk` / `verify` of the reference exchange `VerifierCircuitData::to_bytes(&BackendGateSerializer)`
    (write_vk_action.rs:77-80, noir_and_plonky2_serialization.rs:16-22).  The restatement is UNPINNED (no VK file
    in the reference tree); what is tested: export -> import is lossless (same verifier blob, same bytes again),
    the imported key accepts the oracle's proof and rejects a tampered one, the layout's fixed points (wide_ecc_config constants,
    cap height, cap, digest at the end, gate tags of write_vk_action.rs:39-61) sit where the layout says."""
    oc, vd, wires, pis = make(pkg, orc, d, mix, seed, npi, nw, hasher=hasher)
    proof, _ = oc.prove(wires, public_inputs=pis)
    vk = vd.to_plonky2_bytes()
    vd2 = pkg.VerifierCircuitData.from_plonky2_bytes(vk, hasher=hasher)
    assert vd2.to_bytes() == vd.to_bytes()
    assert vd2.to_plonky2_bytes() == vk
    vd2.verify(proof)
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 4
    with pytest.raises(pkg.P2GpuError):
        vd2.verify(bytes(bad))
    hb = 32 if hasher else 25
    u64 = lambda off: int.from_bytes(vk[off:off + 8], "little")
    # VerifierCircuitData::to_bytes = CommonCircuitData first, VerifierOnlyCircuitData (cap height, cap, digest) last
    vo = len(vk) - (8 + 17 * hb)
    assert u64(vo) == 4 and vk[vo + 8:vo + 8 + 16 * hb] == oc.cap() and vk[vo + 8 + 16 * hb:] == oc.digest()
    cfg = 0
    assert [u64(cfg + 8 * i) for i in range(6)] == [nw, 80, 2, 100, 2, 8] and vk[cfg + 48:cfg + 50] == b"\x01\x00"
    # FriConfig: rate_bits 3, cap_height 4, 28 queries, 16 PoW bits, ConstantArityBits(4, 5)
    fri = cfg + 50
    assert [u64(fri), u64(fri + 8), u64(fri + 16)] == [3, 4, 28] and vk[fri + 24:fri + 28] == (16).to_bytes(4, "little")
    assert vk[fri + 28] == 1 and [u64(fri + 29), u64(fri + 37)] == [4, 5]
    # truncations and bit flips never crash the reader: rejected, or parsed into some key (configuration words the
    # blob does not keep, e.g. security_bits, may change without changing the key)
    rng = np.random.default_rng(seed)
    for cut in [0, 7, 8, fri + 10, vo, vo + 8, len(vk) - 1]:
        with pytest.raises(pkg.P2GpuError):
            pkg.VerifierCircuitData.from_plonky2_bytes(vk[:cut], hasher=hasher)
    for _ in range(60):
        m = bytearray(vk)
        m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
        try:
            pkg.VerifierCircuitData.from_plonky2_bytes(bytes(m), hasher=hasher).close()
        except pkg.P2GpuError:
            continue
