"""Reference-generated golden vectors: the two proof files the reference ships.

`tests/golden/reference/basic_if.proof.hex` and `basic_div.proof.hex` are byte-for-byte copies of
the DATA files `plonky2-backend/example_programs/basic_{if,div}/proofs/basic_{if,div}.proof` of the
reference tree: hex of `proof.compress(..).to_bytes()` written by its own `prove` action
(plonky2-backend/src/actions/prove_action.rs:64-79) for the Noir programs
`example_programs/basic_if/src/main.nr` (a=4, b=2, cond=1; no public input) and
`example_programs/basic_div/src/main.nr` (x=2, y=1; one public input).  They are the only outputs
of the reference's prover that exist in this environment (it cannot be built here), and they are
enough to pin almost the whole protocol, because a plonk proof of a 2^3-row circuit opened at 28
queries leaks everything:

  * the file parses, to the last byte, as plonky2's CompressedProofWithPublicInputs for
    standard_recursion_config shapes (135 wires, 80 routed, 2 challenges, cap height 4, 28 queries,
    degree 2^3, no FRI reduction step);
  * the compressed Merkle paths of the three prover trees lead to the caps in the proof;
  * 25 distinct LDE rows are opened, every column is a polynomial of degree < 8, so ALL columns
    (constants, sigmas, wires, Z, partial products, quotient chunks) are recovered by
    interpolation -- i.e. the circuit's constants/sigmas and the complete witness;
  * an identity sigma column (sigma_79 = k_79 X) gives zeta from its opening; every one of the
    256/257 openings then equals the recovered polynomial at zeta (or g zeta);
  * from these: a circuit blob + wire matrix in this repository's input format, and the
    decompressed proof in plonky2's uncompressed `ProofWithPublicInputs::to_bytes` layout.

`tests/test_reference_proofs.py` then demands that the oracle's prover (and on the GPU box the
product) reproduce those bytes exactly when handed the same circuit, witness and PoW witness.

Nothing here uses oracle/ or the product: the arithmetic is Python integers and the Keccak-256
below is a from-the-spec implementation (original Keccak padding 0x01, as in tiny-keccak 2.0.2
`Keccak::v256`).  What the two files can NOT pin: FRI reduction steps (2^3 rows need none) and the
gates they do not contain (RandomAccess and the five custom u32/comparison gates).
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
P = 0xFFFFFFFF00000001
GEN = 14293326489335486720     # GoldilocksField::MULTIPLICATIVE_GROUP_GENERATOR (coset shift, k_is base)
ROOT32 = 7277203076849721926   # GoldilocksField::POWER_OF_TWO_GENERATOR = GEN^((p-1)/2^32)

# standard_recursion_config, as used by the version of the reference that wrote the files
W, R, K, QF, PP, RATE_BITS, CAP_H, QUERIES, POW_BITS = 135, 80, 2, 8, 9, 3, 4, 28, 16
D = 3  # degree bits of both circuits (query indices < 2^6)

G_NOOP, G_CONSTANT, G_PUBLIC_INPUT, G_ARITHMETIC, G_BASE_SUM, G_RANDOM_ACCESS, G_POSEIDON = range(7)
# Gate lists in plonky2's order (sorted by degree, then id), read off the selector columns and the
# witness rows:  (kind, p0, p1, selector index, group start, group end, #constraints, degree, #constants)
CASES = {
    "basic_if": dict(num_constants=3, num_pi=0, gates=[
        (G_NOOP, 0, 0, 0, 0, 5, 0, 0, 0), (G_CONSTANT, 2, 0, 0, 0, 5, 2, 1, 2), (G_PUBLIC_INPUT, 0, 0, 0, 0, 5, 4, 1, 0),
        (G_BASE_SUM, 2, 63, 0, 0, 5, 64, 2, 0), (G_ARITHMETIC, 20, 0, 0, 0, 5, 20, 3, 2)]),
    "basic_div": dict(num_constants=4, num_pi=1, gates=[
        (G_NOOP, 0, 0, 0, 0, 4, 0, 0, 0), (G_CONSTANT, 2, 0, 0, 0, 4, 2, 1, 2), (G_PUBLIC_INPUT, 0, 0, 0, 0, 4, 4, 1, 0),
        (G_ARITHMETIC, 20, 0, 0, 0, 4, 20, 3, 2), (G_POSEIDON, 0, 0, 1, 4, 5, 123, 7, 0)]),
}


# ---- Keccak-256 (FIPS 202 permutation, original pad10*1 with domain byte 0x01) ----------------
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
       0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
       0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
       0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
       0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _keccak_f(a):
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        dd = [c[(x - 1) % 5] ^ (((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63)) & _M64) for x in range(5)]
        a = [[a[x][y] ^ dd[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                r = _ROT[x][y]
                v = a[x][y]
                b[y][(2 * x + 3 * y) % 5] = ((v << r) | (v >> (64 - r))) & _M64 if r else v
        a = [[b[x][y] ^ (~b[(x + 1) % 5][y] & _M64 & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data):
    rate = 136
    msg = bytearray(data) + b"\x01"
    msg += bytes(-len(msg) % rate)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        a = _keccak_f(a)
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


def hash_or_noop(vals):
    """KeccakHash<25>::hash_or_noop of a leaf of field elements."""
    raw = b"".join(struct.pack("<Q", v) for v in vals)
    return (raw + bytes(25))[:25] if len(raw) <= 25 else keccak256(raw)[:25]


def two_to_one(left, right):
    return keccak256(left + right)[:25]


# ---- field helpers --------------------------------------------------------------------------
def inv(a):
    return pow(a, P - 2, P)


def root_of_unity(bits):
    return pow(ROOT32, 1 << (32 - bits), P)


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def ext_scale(a, s):
    return (a[0] * s % P, a[1] * s % P)


def ext_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def ext_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def poly_eval(c, x):
    r = 0
    for a in reversed(c):
        r = (r * x + a) % P
    return r


def poly_eval_ext(c, z):
    r = (0, 0)
    for a in reversed(c):
        r = ext_add(ext_mul(r, z), (a, 0))
    return r


def interpolate(xs, ys):
    n = len(xs)
    coeffs = [0] * n
    for i in range(n):
        num, den = [1], 1
        for j in range(n):
            if j != i:
                num = [(-xs[j] * num[0]) % P] + [((num[k - 1]) - xs[j] * (num[k] if k < len(num) else 0)) % P
                                                 for k in range(1, len(num) + 1)]
                den = den * (xs[i] - xs[j]) % P
        s = ys[i] * inv(den) % P
        for k in range(n):
            coeffs[k] = (coeffs[k] + s * num[k]) % P
    return coeffs


# ---- the compressed proof -------------------------------------------------------------------
class _Reader:
    def __init__(self, b):
        self.b, self.p = b, 0

    def take(self, n):
        r = self.b[self.p:self.p + n]
        assert len(r) == n, "proof too short"
        self.p += n
        return r

    def u64s(self, n):
        return list(struct.unpack("<%dQ" % n, self.take(8 * n)))


def load_bytes(name):
    with open(os.path.join(HERE, "reference", name + ".proof.hex")) as f:
        return bytes.fromhex(f.read().strip())


def parse_compressed(data, num_constants, num_pi):
    """plonky2 util/serialization: write_compressed_proof_with_public_inputs."""
    r = _Reader(data)
    ncap = 1 << CAP_H
    pr = {"caps": [[r.take(25) for _ in range(ncap)] for _ in range(3)]}
    nop = num_constants + R + W + K + K + K * PP + K * QF
    pr["openings"] = [tuple(r.u64s(2)) for _ in range(nop)]
    pr["indices"] = list(struct.unpack("<%dI" % QUERIES, r.take(4 * QUERIES)))
    cols = [num_constants + R, W, K * (1 + PP), K * QF]
    pr["init"] = {}
    for x in sorted(set(pr["indices"])):   # HashMap entries are written sorted by index
        ent = []
        for o in range(4):
            leaf = r.u64s(cols[o])
            ent.append((leaf, [r.take(25) for _ in range(r.take(1)[0])]))
        pr["init"][x] = ent
    pr["final_poly"] = [tuple(r.u64s(2)) for _ in range(1 << D)]   # no reduction step: 2^3 coefficients
    pr["pow_witness"] = r.u64s(1)[0]
    pr["public_inputs"] = r.u64s(num_pi)
    assert r.p == len(data), "trailing bytes"
    return pr


def decompress_tree(pr, tree, cap):
    """hash/merkle_proofs.rs decompress_merkle_proofs: rebuild every node on the query paths
    (heap numbering, root = 1) and, when the cap is known, check the paths end in it."""
    height = D + RATE_BITS
    nl = 1 << height
    seen = {x + nl: hash_or_noop(pr["init"][x][tree][0]) for x in pr["indices"]}
    its = [iter(pr["init"][x][tree][1]) for x in pr["indices"]]
    for layer in range(height - CAP_H):
        for x, it in zip(pr["indices"], its):
            node = (x + nl) >> layer
            if node ^ 1 not in seen:
                seen[node ^ 1] = next(it)
            par = two_to_one(seen[node], seen[node ^ 1]) if node % 2 == 0 else two_to_one(seen[node ^ 1], seen[node])
            assert seen.setdefault(node >> 1, par) == par
    if cap is not None:
        for x in pr["indices"]:
            ci = (x + nl) >> (height - CAP_H)
            assert seen[ci] == cap[ci - (1 << CAP_H)], "Merkle path does not lead to the cap"
    return seen


def recover_polynomials(pr):
    """Coefficients of every committed column, from the opened LDE rows: leaf x of a tree holds
    the natural LDE row bitrev(x), i.e. the point GEN * w_64^bitrev(x)."""
    lg, n = D + RATE_BITS, 1 << D
    w = root_of_unity(lg)
    idx = sorted(pr["init"])
    pts = [GEN * pow(w, bitrev(x, lg), P) % P for x in idx]
    polys = []
    for t in range(4):
        cols = []
        for j in range(len(pr["init"][idx[0]][t][0])):
            ys = [pr["init"][x][t][0][j] for x in idx]
            c = interpolate(pts[:n], ys[:n])
            assert all(poly_eval(c, xx) == yy for xx, yy in zip(pts[n:], ys[n:])), "column is not of degree < n"
            cols.append(c)
        polys.append(cols)
    return polys


class ReferenceCase:
    """Everything derived from one artefact."""

    def __init__(self, name):
        cfg = CASES[name]
        self.name, self.num_constants, self.num_pi = name, cfg["num_constants"], cfg["num_pi"]
        self.compressed = load_bytes(name)
        self.pr = parse_compressed(self.compressed, self.num_constants, self.num_pi)
        pr, NC, n = self.pr, self.num_constants, 1 << D
        self.trees = [decompress_tree(pr, t, pr["caps"][t - 1] if t else None) for t in range(4)]
        self.polys = recover_polynomials(pr)
        g = root_of_unity(D)
        H = [pow(g, i, P) for i in range(n)]
        on_h = lambda cols: np.array([[poly_eval(c, h) for h in H] for c in cols], dtype=np.uint64)  # noqa: E731
        self.constants, self.sigmas, self.wires = on_h(self.polys[0][:NC]), on_h(self.polys[0][NC:]), on_h(self.polys[1])
        # zeta from an identity sigma column
        k79 = pow(GEN, R - 1, P)
        assert self.polys[0][NC + R - 1] == [0, k79] + [0] * (n - 2)
        self.zeta = ext_scale(pr["openings"][NC + R - 1], inv(k79))
        self.pow_witness, self.public_inputs = pr["pow_witness"], pr["public_inputs"]

    def known_cap_entries(self):
        """constants_sigmas cap entries the query paths pass through: {cap index: 25 bytes}."""
        return {k - (1 << CAP_H): v for k, v in self.trees[0].items() if (1 << CAP_H) <= k < (2 << CAP_H)}

    def blob(self):
        """The circuit in include/p2gpu.h blob format (no cap / digest: the library computes them)."""
        cfg = CASES[self.name]
        gates = cfg["gates"]
        h = np.zeros(64, dtype=np.uint32)
        h[0], h[1], h[2], h[3], h[4], h[5] = 0x43473250, 1, D, W, R, self.num_constants
        h[6] = 1 + max(g[3] for g in gates)
        h[7], h[8], h[9], h[10], h[11], h[12], h[13] = K, QF, RATE_BITS, CAP_H, POW_BITS, QUERIES, 0
        h[22], h[23], h[24], h[25], h[26] = 0, len(gates), self.num_pi, 0, PP
        gt = np.zeros((len(gates), 12), dtype=np.uint32)
        for i, (kind, p0, p1, sel, gs, ge, nc, deg, nk) in enumerate(gates):
            gt[i] = [kind, p0, p1, 0, 0, sel, gs, ge, nc, deg, nk, 0]
        k_is = np.array([pow(GEN, j, P) for j in range(R)], dtype=np.uint64)
        raw = h.tobytes() + gt.tobytes() + k_is.tobytes() + self.constants.tobytes() + self.sigmas.tobytes()
        return np.frombuffer(raw, dtype=np.uint8).copy()

    def uncompressed(self):
        """The same proof in plonky2's uncompressed ProofWithPublicInputs::to_bytes layout."""
        pr = self.pr
        out = bytearray()
        for cap in pr["caps"]:
            out += b"".join(cap)
        for a, b in pr["openings"]:
            out += struct.pack("<QQ", a, b)
        height = D + RATE_BITS
        for x in pr["indices"]:
            for t in range(4):
                out += b"".join(struct.pack("<Q", v) for v in pr["init"][x][t][0])
                out += bytes([height - CAP_H])
                node = x + (1 << height)
                for layer in range(height - CAP_H):
                    out += self.trees[t][(node >> layer) ^ 1]
        for a, b in pr["final_poly"]:
            out += struct.pack("<QQ", a, b)
        out += struct.pack("<Q", pr["pow_witness"])
        for v in pr["public_inputs"]:
            out += struct.pack("<Q", v)
        return bytes(out)

    def openings_from_polynomials(self):
        """OpeningSet recomputed from the recovered coefficients, in serialisation order."""
        z = self.zeta
        gz = ext_scale(z, root_of_unity(D))
        zs = self.polys[2][:K]
        cols = self.polys[0] + self.polys[1] + zs
        return ([poly_eval_ext(c, z) for c in cols] + [poly_eval_ext(c, gz) for c in zs]
                + [poly_eval_ext(c, z) for c in self.polys[2][K:] + self.polys[3]])
