"""An independent, from-the-definition check of the FRI part of a proof (test infrastructure).

Why: the two proofs the reference ships (tests/golden/reference/) are 2^3-row circuits with NO FRI
reduction step, so the arity-16 fold, the step-tree leaf layout and the fold orientation are pinned
by no reference byte; the oracle's verifier (oracle/verifier.c) and the product's (csrc/verify.hip)
were written by the same hand as the prover.  This file is a third reader that shares nothing with
them: pure Python integers, the from-the-spec Keccak of reference_proofs.py, no oracle/, no product,
no index helper in common.  It checks, for every query of an uncompressed proof:

  1. the Fiat-Shamir transcript (Keccak duplex challenger) re-derived here gives
     the query indices, and the proof-of-work response has its leading zero bits;
  2. the four initial Merkle paths lead to the caps;
  3. the value the FRI oracle must take at the query point follows from the opened rows and the
     opening set by the DEFINITION of the batched quotient
        E(x) = alpha^|B1| * (sum_j alpha^j f_j(x) - sum_j alpha^j f_j(zeta)) / (x - zeta)
                          + (sum_j alpha^j z_j(x) - sum_j alpha^j z_j(g zeta)) / (x - g zeta);
  4. every reduction step, from the definition of the FRI fold: the 16 values of the step leaf are
     the values of the current polynomial on the 16 preimages {u : u^16 = y}; the unique polynomial of
     degree < 16 through them (plain Lagrange interpolation, no barycentric shortcut), evaluated at
     beta, must be the next polynomial's value at y -- which is what the next leaf (or final_poly)
     holds; the leaf's Merkle path leads to the step's cap;
  5. final_poly, evaluated at the last point, equals the last folded value, and has the degree the
     rate allows.

What stays recollection-only after this check passes (it cannot be otherwise without upstream code or
a reference proof with d >= 6): that upstream plonky2 0.2.2 uses the SAME point <-> position map as
this file -- position t of step leaf q is the point shift * w^bitrev(16 q + t), queries walk x -> x >> 4
-- and the same challenge order.  The position <-> point map of the INITIAL trees is the one the
reference's own proofs confirm (reference_proofs.recover_polynomials interpolates with it).
"""
import struct

from reference_proofs import GEN, P, bitrev, ext_add, ext_mul, hash_or_noop, inv, keccak256, root_of_unity, two_to_one


# ---- F_p^2 = F_p[X]/(X^2 - 7) ----------------------------------------------------------------
def ext_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def ext_inv(a):
    n = inv((a[0] * a[0] - 7 * a[1] * a[1]) % P)
    return (a[0] * n % P, (-a[1]) * n % P)


def ext_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = ext_mul(r, a)
        a = ext_mul(a, a)
        e >>= 1
    return r


# ---- Challenger<F, KeccakHash<25>> (iop/challenger.rs, hash/keccak.rs KeccakPermutation) ------
class Challenger:
    def __init__(self):
        self.state, self.inp, self.out = [0] * 12, [], []

    def _permute(self):
        h = keccak256(b"".join(struct.pack("<Q", v) for v in self.state))
        words = []
        while len(words) < 12:
            words += [w for w in struct.unpack("<4Q", h) if w < P]
            h = keccak256(h)
        self.state = words[:12]

    def _duplex(self):
        self.state[:len(self.inp)] = self.inp
        self.inp = []
        self._permute()
        self.out = self.state[:8]

    def observe(self, e):
        self.out = []
        self.inp.append(e)
        if len(self.inp) == 8:
            self._duplex()

    def observe_ext(self, x):
        self.observe(x[0])
        self.observe(x[1])

    def observe_hash(self, h25):
        for i in range(0, 25, 7):
            self.observe(int.from_bytes(h25[i:i + 7], "little"))

    def observe_cap(self, cap):
        for h in cap:
            self.observe_hash(h)

    def challenge(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()

    def ext_challenge(self):
        a = self.challenge()
        return (a, self.challenge())


# ---- proof bytes ----------------------------------------------------------------------------
class _R:
    def __init__(self, b):
        self.b, self.p = b, 0

    def take(self, n):
        r = self.b[self.p:self.p + n]
        assert len(r) == n, "proof too short"
        self.p += n
        return r

    def u64s(self, n):
        return list(struct.unpack("<%dQ" % n, self.take(8 * n)))

    def path(self):
        return [self.take(25) for _ in range(self.take(1)[0])]


def parse(c, proof):
    """c: proof_stages.header(blob).  Returns the proof as a dict of Python ints/bytes."""
    r = _R(proof)
    ncap = 1 << c["cap_h"]
    pr = {"caps": [[r.take(25) for _ in range(ncap)] for _ in range(3)]}
    sizes = [("constants", c["NC"]), ("sigmas", c["R"]), ("wires", c["W"]), ("zs", c["K"]), ("zs_next", c["K"]),
             ("pps", c["K"] * c["PP"]), ("quotient", c["K"] * c["QF"])]
    pr["openings"] = {k: [tuple(r.u64s(2)) for _ in range(n)] for k, n in sizes}
    pr["step_caps"] = [[r.take(25) for _ in range(ncap)] for _ in range(c["steps"])]
    cols = [c["NC"] + c["R"], c["W"], c["K"] * (1 + c["PP"]), c["K"] * c["QF"]]
    pr["queries"] = []
    for _ in range(c["queries"]):
        init = [(r.u64s(cols[t]), r.path()) for t in range(4)]
        steps = []
        for a in c["arity"]:
            ev = r.u64s(2 << a)
            steps.append(([(ev[2 * i], ev[2 * i + 1]) for i in range(1 << a)], r.path()))
        pr["queries"].append((init, steps))
    pr["final_poly"] = [tuple(r.u64s(2)) for _ in range(1 << (c["d"] - sum(c["arity"])))]
    pr["pow_witness"] = r.u64s(1)[0]
    pr["public_inputs"] = r.u64s(c["n_pi"])
    assert r.p == len(proof), "trailing bytes"
    return pr


def merkle_root_from_path(leaf_hash, index, path):
    cur = leaf_hash
    for s in path:
        cur = two_to_one(s, cur) if index & 1 else two_to_one(cur, s)
        index >>= 1
    return cur, index


def lagrange_eval(points, values, at):
    """Value at `at` of the polynomial of degree < len(points) through (points[i], values[i]);
    points in F_p, values and `at` in F_p^2.  Textbook Lagrange: sum_i v_i prod_{j != i} (at - x_j)/(x_i - x_j)."""
    acc = (0, 0)
    for i, (xi, vi) in enumerate(zip(points, values)):
        num, den = (1, 0), 1
        for j, xj in enumerate(points):
            if j != i:
                num = ext_mul(num, ext_sub(at, (xj, 0)))
                den = den * (xi - xj) % P
        acc = ext_add(acc, ext_mul(vi, (num[0] * inv(den) % P, num[1] * inv(den) % P)))
    return acc


def check(c, proof, circuit_digest, constants_sigmas_cap, pow_bits=16, pi_hash=(0, 0, 0, 0)):
    """Raises AssertionError with the failed check; returns a summary dict when everything holds.
    constants_sigmas_cap: list of 2^cap_h 25-byte digests (verifier key)."""
    pr = parse(c, proof)
    d, rate = c["d"], c["rate_bits"]
    lg = d + rate
    N = 1 << lg
    # 1. transcript
    ch = Challenger()
    ch.observe_hash(circuit_digest)
    for v in pi_hash:
        ch.observe(v)
    ch.observe_cap(pr["caps"][0])
    betas = [ch.challenge() for _ in range(c["K"])]
    gammas = [ch.challenge() for _ in range(c["K"])]
    ch.observe_cap(pr["caps"][1])
    alphas = [ch.challenge() for _ in range(c["K"])]
    ch.observe_cap(pr["caps"][2])
    zeta = ch.ext_challenge()
    op = pr["openings"]
    batch0 = op["constants"] + op["sigmas"] + op["wires"] + op["zs"] + op["pps"] + op["quotient"]
    batch1 = op["zs_next"]
    for v in batch0 + batch1:
        ch.observe_ext(v)
    alpha = ch.ext_challenge()
    fri_betas = []
    for cap in pr["step_caps"]:
        ch.observe_cap(cap)
        fri_betas.append(ch.ext_challenge())
    for v in pr["final_poly"]:
        ch.observe_ext(v)
    ch.observe(pr["pow_witness"])
    response = ch.challenge()
    assert response >> (64 - pow_bits) == 0, "proof-of-work response has too few leading zeros"
    indices = [ch.challenge() % N for _ in range(c["queries"])]
    # 5a. final_poly degree: 2^(d - sum arity) coefficients are all the proof can hold (rate 1/8 of the last domain)
    assert len(pr["final_poly"]) << rate == N >> sum(c["arity"])
    # reduced openings  sum_j alpha^j y_j
    def reduce_ext(vals):
        acc = (0, 0)
        for v in reversed(vals):
            acc = ext_add(ext_mul(acc, alpha), v)
        return acc
    red0, red1 = reduce_ext(batch0), reduce_ext(batch1)
    g_n = root_of_unity(d)
    g_zeta = (zeta[0] * g_n % P, zeta[1] * g_n % P)
    alpha_b1 = ext_pow(alpha, len(batch1))
    w_N = root_of_unity(lg)
    caps4 = [constants_sigmas_cap] + pr["caps"]
    for qi, (x_index, (init, steps)) in enumerate(zip(indices, pr["queries"])):
        # 2. initial trees: leaf x_index, cap entry = x_index >> (lg - cap_h)
        for t in range(4):
            leaf, path = init[t]
            root, top = merkle_root_from_path(hash_or_noop(leaf), x_index, path)
            assert len(path) == max(lg - c["cap_h"], 0) and root == caps4[t][top], f"query {qi}: initial tree {t} path"
        # 3. the point of leaf x_index is shift * w_N^bitrev(x_index) (pinned by the reference's proofs)
        x = GEN * pow(w_N, bitrev(x_index, lg), P) % P
        row0 = init[0][0] + init[1][0] + init[2][0] + init[3][0]  # all columns of the four oracles, in order
        assert len(row0) == len(batch0)
        f0 = reduce_ext([(v, 0) for v in row0])
        f1 = reduce_ext([(v, 0) for v in init[2][0][:c["K"]]])
        e = ext_add(ext_mul(alpha_b1, ext_mul(ext_sub(f0, red0), ext_inv(ext_sub((x, 0), zeta)))),
                    ext_mul(ext_sub(f1, red1), ext_inv(ext_sub((x, 0), g_zeta))))
        # 4. the folds
        size_bits, shift, idx = lg, GEN, x_index
        for s, (a, (evals, path)) in enumerate(zip(c["arity"], steps)):
            q, t0 = idx >> a, idx & ((1 << a) - 1)
            assert evals[t0] == e, f"query {qi} step {s}: leaf does not hold the previous value"
            w = root_of_unity(size_bits)
            pts = [shift * pow(w, bitrev((q << a) + t, size_bits), P) % P for t in range(1 << a)]
            y = pow(pts[0], 1 << a, P)
            assert all(pow(u, 1 << a, P) == y for u in pts) and len(set(pts)) == 1 << a
            e = lagrange_eval(pts, evals, fri_betas[s])
            flat = [v for pair in evals for v in pair]
            root, top = merkle_root_from_path(hash_or_noop(flat), q, path)
            assert len(path) == max(size_bits - a - c["cap_h"], 0) and root == pr["step_caps"][s][top], f"query {qi} step {s}: path"
            size_bits -= a
            shift = pow(shift, 1 << a, P)
            idx = q
            assert y == shift * pow(root_of_unity(size_bits), bitrev(idx, size_bits), P) % P
        # 5. final polynomial at the last point
        last = shift * pow(root_of_unity(size_bits), bitrev(idx, size_bits), P) % P
        acc = (0, 0)
        for cf in reversed(pr["final_poly"]):
            acc = ext_add((acc[0] * last % P, acc[1] * last % P), cf)
        assert acc == e, f"query {qi}: final_poly does not match the last folded value"
    return {"betas": betas, "gammas": gammas, "alphas": alphas, "zeta": zeta, "alpha_fri": alpha, "fri_betas": fri_betas,
            "pow_response": response, "query_indices": indices}
