"""The algebra behind the half-domain gate evaluation of round 6 (csrc/plonk.hip gate_sums_kernel / gate_sums_cross_kernel, handle.hip
half_cross), from the definition, in Python integers -- no GPU, no library: a polynomial S of degree < 4n, given by its values on the four
EVEN cosets g w_8n^(2m) <w_n> of the LDE domain, is carried to the four ODD cosets by
    per-coset interpolation (size n)  ->  P'_m[e] = P_m[e] s_2m^-e  ->  R'_m'[e] = sum_m F[m'][m] P'_m[e],  F[m'][m] = 1/4 sum_j w_8^((2m'+1-2m) j)
    ->  R_m'[e] = R'_m'[e] s_(2m'+1)^e  ->  per-coset evaluation (size n)
and the result is S itself on those points (what makes the GPU's proofs bit-exact with the knob on).  Also: degree 4n is one too many."""
import random

P = 0xFFFFFFFF00000001
GEN = 14293326489335486720          # plonky2's MULTIPLICATIVE_GROUP_GENERATOR: the LDE coset shift
ROOT32 = 7277203076849721926        # POWER_OF_TWO_GENERATOR (order 2^32)


def root(k):
    g = ROOT32
    for _ in range(32 - k):
        g = g * g % P
    return g


def ev(coef, x):
    acc = 0
    for c in reversed(coef):
        acc = (acc * x + c) % P
    return acc


def carry_even_to_odd(vals_even, d):
    n = 1 << d
    wn, wN, w8 = root(d), root(d + 3), root(3)
    ninv, quarter = pow(n, P - 2, P), pow(4, P - 2, P)
    s = [GEN * pow(wN, r, P) % P for r in range(8)]
    # per-coset interpolants of the even cosets, unscaled: P'_m[e] = (1/n sum_k v[m][k] w_n^(-k e)) s_2m^-e
    Pp = [[ninv * sum(vals_even[m][k] * pow(wn, (-k * e) % n, P) for k in range(n)) % P * pow(s[2 * m], P - 1 - e % (P - 1), P) % P for e in range(n)] for m in range(4)]
    F = [[quarter * sum(pow(w8, ((2 * mo + 1 - 2 * m) * j) % 8, P) for j in range(4)) % P for m in range(4)] for mo in range(4)]
    out = []
    for mo in range(4):
        R = [sum(F[mo][m] * Pp[m][e] for m in range(4)) % P * pow(s[2 * mo + 1], e, P) % P for e in range(n)]
        out.append([sum(R[e] * pow(wn, (k * e) % n, P) for e in range(n)) % P for k in range(n)])
    return out


def cosets(coef, d, parity):
    n = 1 << d
    wn, wN = root(d), root(d + 3)
    return [[ev(coef, GEN * pow(wN, 2 * m + parity, P) % P * pow(wn, k, P) % P) for k in range(n)] for m in range(4)]


def test_a_polynomial_below_degree_4n_is_carried_exactly():
    rng = random.Random(6)
    for d in (2, 3, 4):
        n = 1 << d
        for deg in (4 * n, 4 * n - 1, 3 * n + 1, 1):          # number of coefficients
            coef = [rng.randrange(P) for _ in range(deg)]
            assert carry_even_to_odd(cosets(coef, d, 0), d) == cosets(coef, d, 1)


def test_degree_4n_is_one_too_many():
    rng = random.Random(7)
    d = 3
    coef = [rng.randrange(P) for _ in range(4 * (1 << d) + 1)]   # degree exactly 4n: the even cosets no longer determine it
    assert carry_even_to_odd(cosets(coef, d, 0), d) != cosets(coef, d, 1)


def test_a_product_of_four_degree_n_columns_qualifies():
    """What a range check of a 2-bit limb is: v (v - 1) (v - 2) (v - 3) of a column of degree < n has degree <= 4 (n - 1) < 4n."""
    rng = random.Random(8)
    d = 3
    n = 1 << d
    v = [rng.randrange(P) for _ in range(n)]

    def mul(a, b):
        out = [0] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % P
        return out
    poly = [1]
    for c in range(4):
        poly = mul(poly, [(v[0] - c) % P] + v[1:])
    assert len(poly) == 4 * (n - 1) + 1 < 4 * n + 1
    assert carry_even_to_odd(cosets(poly, d, 0), d) == cosets(poly, d, 1)
