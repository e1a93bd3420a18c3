"""The oracle's primitives against known answers (SURVEY.md 8(c) golden vectors).

Pins: Goldilocks constants (test_assert_zero.rs:275-285 modulus pin), Keccak-256 (original
padding) KATs + an independent pure-Python Keccak validated against hashlib's SHA-3, the
regenerated Poseidon constants and plonky2's permutation test vectors (tests/golden/poseidon.json),
the challenger's hash-onion permutation, NTT identities and Merkle cap layout.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GEN, GOLDEN, P, ROOT32


def test_field_modulus_and_generators(orc):
    # plonky2-backend/src/circuit_translation/tests/test_assert_zero.rs:275-285
    assert P == 18446744069414584321 == 2**64 - 2**32 + 1
    L = orc.lib()
    assert L.orc_gl_mul(P - 1, P - 1) == 1
    # plonky2's MULTIPLICATIVE_GROUP_GENERATOR / POWER_OF_TWO_GENERATOR (the pair the reference's own
    # proof artefacts are consistent with, tests/test_reference_proofs.py -- NOT 7 / 7^((p-1)/2^32))
    g = L.orc_gl_pow(GEN, (P - 1) >> 32)
    assert g == ROOT32 == 7277203076849721926
    assert L.orc_gl_pow(g, 1 << 31) == P - 1 and L.orc_gl_pow(g, 1 << 32) == 1
    assert L.orc_gl_pow(g, 1 << 26) == 8   # w_64 = 8: the radix-64 twiddles are shifts
    # GEN generates the multiplicative group: GEN^((p-1)/q) != 1 for every prime q | p-1
    for q in (2, 3, 5, 17, 257, 65537):
        assert L.orc_gl_pow(GEN, (P - 1) // q) != 1
    for a in (1, 2, 7, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1234567891011121314 % P):
        assert L.orc_gl_mul(a, L.orc_gl_inv(a)) == 1
    rng = np.random.default_rng(5)
    for _ in range(200):
        a, b = (int(x) for x in rng.integers(0, P, size=2, dtype=np.uint64))
        assert L.orc_gl_mul(a, b) == a * b % P


# ---- Keccak -------------------------------------------------------------------------
_RC, _ROT = [], {}


def _init_keccak():
    r = 1
    x, y = 1, 0
    for t in range(24):
        _ROT[(x, y)] = ((t + 1) * (t + 2) // 2) % 64
        x, y = y, (2 * x + 3 * y) % 5
    lfsr = 1
    for _ in range(24):
        rc = 0
        for j in range(7):
            if lfsr & 1:
                rc |= 1 << ((1 << j) - 1)
            lfsr = ((lfsr << 1) ^ (0x71 if lfsr & 0x80 else 0)) & 0xFF
        _RC.append(rc)


def _keccak_f(a):
    m = (1 << 64) - 1
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & m if n else v
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = rol(a[x][y], _ROT.get((x, y), 0))
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y] & m) for y in range(5)] for x in range(5)]
        a[0][0] ^= _RC[rnd]
    return a


def py_keccak(data, pad):
    """Spec-level sponge (rate 136, 256-bit output) with domain byte `pad` (0x01 Keccak, 0x06 SHA-3)."""
    if not _RC:
        _init_keccak()
    msg = bytearray(data) + bytes([pad]) + bytes((-len(data) - 1) % 136)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), 136):
        for i in range(17):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
        a = _keccak_f(a)
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


def test_keccak256_known_answers(orc):
    assert orc.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert orc.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"


def test_keccak256_against_independent_sponge(orc):
    rng = np.random.default_rng(11)
    for n in (0, 1, 7, 50, 96, 135, 136, 137, 271, 272, 273, 1872, 2000):
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        # the pure-Python sponge is validated on the same input against hashlib (SHA-3 padding) ...
        assert py_keccak(data, 0x06) == hashlib.sha3_256(data).digest()
        # ... and then, with the original Keccak padding, checks the oracle
        assert orc.keccak256(data) == py_keccak(data, 0x01)


def test_keccak_hash_onion_permutation(orc):
    st = np.arange(12, dtype=np.uint64)
    exp, h = [], py_keccak(st.tobytes(), 0x01)
    while len(exp) < 12:
        for i in range(4):
            w = int.from_bytes(h[8 * i: 8 * i + 8], "little")
            if w < P and len(exp) < 12:
                exp.append(w)
        h = py_keccak(h, 0x01)
    got = st.copy()
    orc.lib().orc_keccak_permutation(got.ctypes.data)
    assert [int(x) for x in got] == exp


def test_challenger_overwrite_duplex(orc):
    """observe 3 elements, squeeze 10: outputs pop from the END of the rate (C.3)."""
    obs = np.array([5, 6, 7], dtype=np.uint64)
    out = np.zeros(10, dtype=np.uint64)
    orc.lib().orc_challenger_squeeze(obs.ctypes.data, 3, out.ctypes.data, 10)
    st = np.zeros(12, dtype=np.uint64)
    st[:3] = obs
    orc.lib().orc_keccak_permutation(st.ctypes.data)
    first = [int(st[7 - i]) for i in range(8)]
    orc.lib().orc_keccak_permutation(st.ctypes.data)
    assert [int(x) for x in out] == first + [int(st[7]), int(st[6])]


# ---- Poseidon --------------------------------------------------------------------------
def test_poseidon_constants_and_vectors(orc):
    with open(os.path.join(GOLDEN, "poseidon.json")) as f:
        gold = json.load(f)
    rc = np.zeros(360, dtype=np.uint64)
    orc.lib().orc_poseidon_round_constants(rc.ctypes.data)
    assert [int(x) for x in rc] == gold["round_constants"]
    assert int(rc[0]) == 0xB585F766F2144405 and int(rc[11]) == 0xC54302F225DB2C76
    for v in gold["permutation_vectors"]:
        st = np.array(v["input"], dtype=np.uint64)
        orc.lib().orc_poseidon_permute(st.ctypes.data)
        assert [int(x) for x in st] == v["output"]


def test_poseidon_fused_layers_on_the_host(tmp_path):
    """Round 6: the device runs the partial rounds' linear layers three at a time on products of the small-integer MDS matrix
    (csrc/poseidon.hpp POSEIDON_FUSED: M, M P M, M P M P M; the PoseidonGate evaluator takes the same route).  The PRODUCT's header,
    compiled for the host with g++: the permutation in that form -- integer row sums over the 32-bit halves, one reduction per row,
    the poseidon_device_constants form of the round constants -- equals the plain form on plonky2's vectors, on 20 000 random and
    extreme states, and no half-row sum leaves 58 bits.  (The GPU suite then checks the device code itself against the oracle.)"""
    import subprocess

    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "acvm-backend-plonky2_amd", "csrc", "tests", "poseidon_fused_selftest.cpp")
    exe = str(tmp_path / "poseidon_fused_selftest")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, src], check=True, timeout=300)
    with open(os.path.join(GOLDEN, "poseidon.json")) as f:
        gold = json.load(f)
    vecs = gold["permutation_vectors"]
    r = subprocess.run([exe], input="".join(" ".join(str(x) for x in v["input"]) + "\n" for v in vecs), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-500:]
    lines = r.stdout.strip().splitlines()
    assert [[int(x) for x in ln.split()] for ln in lines[:len(vecs)]] == [v["output"] for v in vecs]
    assert lines[-1].startswith("mismatches 0 ") and int(lines[-1].split()[-1]) <= 58


def test_poseidon_hash_no_pad(orc):
    out = np.ones(4, dtype=np.uint64)
    orc.lib().orc_poseidon_hash_no_pad(None, 0, out.ctypes.data)
    assert list(out) == [0, 0, 0, 0]  # public_inputs_hash of a circuit without public inputs
    inp = np.arange(1, 10, dtype=np.uint64)  # 9 elements: two absorptions, overwrite mode
    orc.lib().orc_poseidon_hash_no_pad(inp.ctypes.data, 9, out.ctypes.data)
    st = np.zeros(12, dtype=np.uint64)
    st[:8] = inp[:8]
    orc.lib().orc_poseidon_permute(st.ctypes.data)
    st[0] = inp[8]
    orc.lib().orc_poseidon_permute(st.ctypes.data)
    assert list(out) == list(st[:4])


# ---- NTT / LDE / Merkle ---------------------------------------------------------------
def _py_dft(a, w):
    n = len(a)
    return [sum(a[j] * pow(w, j * k, P) for j in range(n)) % P for k in range(n)]


@pytest.mark.parametrize("lg", [0, 1, 3, 5])
def test_ntt_matches_naive_dft(orc, lg):
    rng = np.random.default_rng(lg)
    a = rng.integers(0, P, size=1 << lg, dtype=np.uint64)
    w = pow(ROOT32, 1 << (32 - lg), P)
    assert [int(x) for x in orc.ntt(a)] == _py_dft([int(x) for x in a], w)
    assert np.array_equal(orc.ntt(orc.ntt(a), inverse=True), a)


def test_coset_lde_evaluates_the_polynomial(orc):
    rng = np.random.default_rng(3)
    d = 4
    c = rng.integers(0, P, size=1 << d, dtype=np.uint64)
    lde = orc.coset_lde(c, 3)
    wN = pow(ROOT32, 1 << (32 - d - 3), P)
    for i in (0, 1, 7, 8, 100, 127):
        x = GEN * pow(wN, i, P) % P
        assert int(lde[i]) == sum(int(c[j]) * pow(x, j, P) for j in range(1 << d)) % P


def test_merkle_cap_layout(orc):
    """cap[k] = root of leaves [k*L/16, (k+1)*L/16), Keccak-256/25 two_to_one on 25-byte digests."""
    rng = np.random.default_rng(9)
    leaves = rng.integers(0, P, size=(32, 5), dtype=np.uint64)
    cap = np.zeros(25 * 16, dtype=np.uint8)
    orc.lib().orc_merkle_cap(leaves.ctypes.data, 32, 5, 4, cap.ctypes.data)
    for k in range(16):
        l = py_keccak(leaves[2 * k].tobytes(), 0x01)[:25]
        r = py_keccak(leaves[2 * k + 1].tobytes(), 0x01)[:25]
        assert cap[25 * k: 25 * k + 25].tobytes() == py_keccak(l + r, 0x01)[:25]
    # hash_or_noop: rows of <= 3 elements are copied, not hashed
    small = np.array([[1, 2, 3]], dtype=np.uint64)
    assert orc.hash_rows(small)[0].tobytes() == small.tobytes() + b"\0"
