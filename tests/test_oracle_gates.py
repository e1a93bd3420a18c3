"""Gate-constraint evaluators of the oracle against the reference's own gate tests
(SURVEY.md section 4 "Custom-gate tests"): satisfying assignments give all-zero constraint
vectors, the canonicity negative gives a non-zero one, wire indices match, and the
base-field evaluator agrees with the extension-field evaluator (`test_eval_fns`)."""
import numpy as np
import pytest

from conftest import P
from gate_wires import *  # noqa: F401,F403


def _rng(seed):
    return np.random.default_rng(seed)


def _u32s(rng, n):
    return [int(x) for x in rng.integers(0, 1 << 32, size=n, dtype=np.uint64)]


def _pad(w, n=234):
    return np.array(list(w) + [0] * (n - len(w)), dtype=np.uint64)


def test_u32_arithmetic_gate_constraint(orc):
    # arithmetic_u32.rs:532-570: NUM_U32_ARITHMETIC_OPS = 3, random u32 operands
    rng = _rng(1)
    w = u32_arithmetic_wires(_u32s(rng, 3), _u32s(rng, 3), _u32s(rng, 3))
    out = orc.gate_eval(G_U32_ARITHMETIC, [3], _pad(w))
    assert len(out) == 3 * 36 and not out.any()
    # maximal output: hi = 0xFFFFFFFE.., lo arbitrary
    w = u32_arithmetic_wires([0xFFFFFFFF] * 3, [0xFFFFFFFF] * 3, [0xFFFFFFFF] * 3)
    assert not orc.gate_eval(G_U32_ARITHMETIC, [3], _pad(w)).any()


def test_u32_arithmetic_canonicity(orc):
    # arithmetic_u32.rs:572-605: a non-canonical addend 0xFFFFFFFF00000001 must NOT pass
    w = u32_arithmetic_wires([0] * 3, [0] * 3, [0xFFFFFFFF00000001] * 3)
    assert orc.gate_eval(G_U32_ARITHMETIC, [3], _pad(w)).any()


def test_u32_arithmetic_wide_config_shape(orc):
    # arithmetic_u32.rs:39-42 with 234 wires / 80 routed: 6 ops, 228 wires, 216 constraints
    rng = _rng(2)
    w = u32_arithmetic_wires(_u32s(rng, 6), _u32s(rng, 6), _u32s(rng, 6))
    assert len(w) == 228
    out = orc.gate_eval(G_U32_ARITHMETIC, [6], _pad(w))
    assert len(out) == 216 and not out.any()


@pytest.mark.parametrize("num_addends,num_ops", [(10, 3), (3, 9), (16, 4), (2, 5)])
def test_u32_add_many_gate_constraint(orc, num_addends, num_ops):
    # add_many_u32.rs:416-491
    rng = _rng(num_addends)
    addends = [_u32s(rng, num_addends) for _ in range(num_ops)]
    carries = _u32s(rng, num_ops) if num_addends < 16 else [0] * num_ops
    w, _ = u32_add_many_wires(addends, carries)
    out = orc.gate_eval(G_U32_ADD_MANY, [num_addends, num_ops], _pad(w, 400))
    assert len(out) == num_ops * 21 and not out.any()
    bad = list(w)
    bad[num_addends + 1] = (bad[num_addends + 1] + 1) % P  # wrong result limb
    assert orc.gate_eval(G_U32_ADD_MANY, [num_addends, num_ops], _pad(bad, 400)).any()


def test_u32_subtraction_gate_constraint(orc):
    # subtraction_u32.rs:397-474, incl. borrow-out cases x < y
    rng = _rng(3)
    xs, ys = _u32s(rng, 11), _u32s(rng, 11)
    xs[0], ys[0] = 5, 7          # borrow
    xs[1], ys[1] = 9, 9          # zero
    bs = [int(b) for b in rng.integers(0, 2, size=11)]
    w = u32_subtraction_wires(xs, ys, bs)
    assert len(w) == 231
    out = orc.gate_eval(G_U32_SUBTRACTION, [11], _pad(w))
    assert len(out) == 11 * 19 and not out.any()
    bad = list(w)
    bad[4] = 2  # output borrow must be a bit
    assert orc.gate_eval(G_U32_SUBTRACTION, [11], _pad(bad)).any()


def test_u32_range_check_gate_constraint(orc):
    # range_check_u32.rs:262-334: 8 input limbs (nonnative.rs:331-332 instantiation)
    rng = _rng(4)
    w = u32_range_check_wires(_u32s(rng, 8))
    assert len(w) == 136
    out = orc.gate_eval(G_U32_RANGE_CHECK, [8], _pad(w))
    assert len(out) == 136 and not out.any()
    # a limb >= 2^32 cannot be decomposed into 16 base-4 digits
    bad = u32_range_check_wires([1 << 32] + [0] * 7)
    assert orc.gate_eval(G_U32_RANGE_CHECK, [8], _pad(bad)).any()


def test_comparison_wire_indices():
    # comparison.rs:564-592 (num_bits 40, num_chunks 5): fixed wire positions
    nc, cb = 5, 8
    assert (4, 4 + nc - 1) == (4, 8) and (4 + nc, 4 + 2 * nc - 1) == (9, 13)
    assert (4 + 2 * nc, 4 + 3 * nc - 1) == (14, 18) and (4 + 3 * nc, 4 + 4 * nc - 1) == (19, 23)
    assert (4 + 4 * nc, 4 + 5 * nc - 1) == (24, 28) and (4 + 5 * nc, 4 + 5 * nc + cb) == (29, 37)
    assert len(comparison_wires(1, 2, 40, 5)) == 4 + 5 * nc + cb + 1 == 38


@pytest.mark.parametrize("nb,nc", [(40, 5), (32, 16)])
def test_comparison_gate_constraint(orc, nb, nc):
    # comparison.rs:613-743: less-than and equal inputs both satisfy the gate
    rng = _rng(nb)
    a = int(rng.integers(0, 1 << (nb - 1)))
    b = int(rng.integers(a, 1 << (nb - 1)))
    for x, y in ((a, b), (a, a), (b, a), (0, 0), ((1 << nb) - 1, 0)):
        out = orc.gate_eval(G_COMPARISON, [nb, nc], _pad(comparison_wires(x, y, nb, nc)))
        assert len(out) == 6 + 5 * nc + nb // nc and not out.any()
    bad = comparison_wires(a, b, nb, nc)
    bad[2] ^= 1  # wrong result bool
    assert orc.gate_eval(G_COMPARISON, [nb, nc], _pad(bad)).any()


def test_stock_gates(orc):
    rng = _rng(6)
    f = lambda k: [int(x) for x in rng.integers(0, P, size=k, dtype=np.uint64)]
    # ArithmeticGate{20}: out = c0*m0*m1 + c1*addend
    c0, c1 = f(2)
    w = []
    for _ in range(20):
        m0, m1, ad = f(3)
        w += [m0, m1, ad, (c0 * m0 * m1 + c1 * ad) % P]
    out = orc.gate_eval(G_ARITHMETIC, [20], _pad(w), consts=[c0, c1])
    assert len(out) == 20 and not out.any()
    w[3] = (w[3] + 1) % P
    assert orc.gate_eval(G_ARITHMETIC, [20], _pad(w), consts=[c0, c1]).any()
    # ConstantGate{2}, PublicInputGate
    assert not orc.gate_eval(G_CONSTANT, [2], _pad([11, 12]), consts=[11, 12]).any()
    assert orc.gate_eval(G_CONSTANT, [2], _pad([11, 13]), consts=[11, 12]).any()
    assert not orc.gate_eval(G_PUBLIC_INPUT, [], _pad([1, 2, 3, 4]), pi_hash=(1, 2, 3, 4)).any()
    # BaseSumGate<2>{32}, BaseSumGate<4>{16}
    v = 0xDEADBEEF
    assert not orc.gate_eval(G_BASE_SUM, [2, 32], _pad([v] + [(v >> i) & 1 for i in range(32)])).any()
    assert not orc.gate_eval(G_BASE_SUM, [4, 16], _pad([v] + [(v >> (2 * i)) & 3 for i in range(16)])).any()
    assert orc.gate_eval(G_BASE_SUM, [2, 32], _pad([v] + [2] + [(v >> i) & 1 for i in range(1, 32)])).any()
    # RandomAccessGate{bits 4, 4 copies, 2 extra constants}
    w, bits = [], []
    for c in range(4):
        items, idx = f(16), int(rng.integers(0, 16))
        w += [idx, items[idx]] + items
        bits += [(idx >> k) & 1 for k in range(4)]
    e0, e1 = f(2)
    w += [e0, e1] + bits
    out = orc.gate_eval(G_RANDOM_ACCESS, [4, 4, 2], _pad(w), consts=[e0, e1])
    assert len(out) == 4 * 6 + 2 and not out.any()
    w[1] = (w[1] + 1) % P
    assert orc.gate_eval(G_RANDOM_ACCESS, [4, 4, 2], _pad(w), consts=[e0, e1]).any()


def test_poseidon_gate_constraint(pkg, orc):
    """PoseidonGate (gates/poseidon.rs, 135 wires, 123 constraints, degree 7): the rows the
    generator emits for public-input hashing satisfy it; breaking an S-box wire does not."""
    blob, wires, pis = pkg.make_circuit(6, "arith", 5, num_public_inputs=9)
    for row in (2, 3):  # two permutations for 9 inputs (rate 8)
        w = wires[:, row].copy()
        out = orc.gate_eval(6, [], w)
        assert len(out) == 123 and not out.any()
        # the outputs are the Poseidon permutation of the inputs
        st = w[:12].copy()
        orc.lib().orc_poseidon_permute(st.ctypes.data)
        assert np.array_equal(st, w[12:24])
        bad = w.copy()
        bad[65 + 7] = (int(bad[65 + 7]) + 1) % P  # a partial-round S-box input
        assert orc.gate_eval(6, [], bad).any()
    # second permutation: capacity lanes carried over from the first one's outputs
    assert np.array_equal(wires[8:12, 3], wires[20:24, 2]) and wires[0, 3] == pis[8]
    # swap = 1 with matching deltas is also accepted by the gate (Merkle-path mode)
    w = wires[:, 2].copy()
    w[24] = 1
    for i in range(4):
        w[25 + i] = (int(w[i + 4]) - int(w[i])) % P
    st = np.concatenate([w[4:8], w[0:4], w[8:12]]).copy()
    out = orc.gate_eval(6, [], w)
    assert out[:5].tolist() == [0] * 5 and out[5:].any()  # swap/delta constraints hold; the trace was for swap = 0


@pytest.mark.parametrize("kind,params", [
    (6, []),
    (G_ARITHMETIC, [20]), (G_BASE_SUM, [2, 32]), (G_BASE_SUM, [4, 16]), (G_RANDOM_ACCESS, [4, 4, 2]),
    (G_U32_ARITHMETIC, [6]), (G_U32_ADD_MANY, [3, 9]), (G_U32_SUBTRACTION, [11]), (G_U32_RANGE_CHECK, [8]),
    (G_COMPARISON, [32, 16]), (G_CONSTANT, [2]), (G_PUBLIC_INPUT, []),
])
def test_eval_fns_base_vs_extension(orc, kind, params):
    """gate_testing.rs:85-159 `test_eval_fns`: on random wires the base evaluator equals the
    extension evaluator restricted to the base field; and the extension evaluator is F_p-linear
    enough to agree coordinate-wise on (w, 0) inputs."""
    rng = _rng(kind * 7 + len(params))
    w = rng.integers(0, P, size=234, dtype=np.uint64)
    cs = [int(x) for x in rng.integers(0, P, size=2, dtype=np.uint64)]
    pih = tuple(int(x) for x in rng.integers(0, P, size=4, dtype=np.uint64))
    base = orc.gate_eval(kind, params, w, consts=cs, pi_hash=pih)
    wext = np.zeros(2 * 234, dtype=np.uint64)
    wext[0::2] = w
    cext = []
    for c in cs:
        cext += [c, 0]
    ext = orc.gate_eval(kind, params, wext, consts=cext, pi_hash=pih, ext=True)
    assert np.array_equal(ext[0::2], base) and not ext[1::2].any()


@pytest.mark.parametrize("kind,params,degree,nconst", [
    (G_CONSTANT, [2], 1, 2), (G_PUBLIC_INPUT, [], 1, 0), (G_ARITHMETIC, [20], 3, 2), (G_BASE_SUM, [2, 32], 2, 0), (G_BASE_SUM, [4, 16], 4, 0),
    (G_RANDOM_ACCESS, [4, 4, 2], 5, 2), (6, [], 7, 0),
    (G_U32_ARITHMETIC, [6], 4, 0), (G_U32_ADD_MANY, [3, 9], 4, 0), (G_U32_SUBTRACTION, [11], 4, 0), (G_U32_RANGE_CHECK, [8], 4, 0),
    (G_COMPARISON, [32, 16], 4, 0),
])
def test_low_degree(orc, kind, params, degree, nconst):
    """gate_testing.rs:20-63 `test_low_degree`, the property test every gate file of the reference runs
    (arithmetic_u32.rs:520-525, add_many_u32.rs:404-409, subtraction_u32.rs:385-390, range_check_u32.rs:250-255,
    comparison.rs:594-601): wires and constants that are random polynomials of degree < 32, evaluated on a domain
    2^ceil(log2(degree + 1)) times larger; every constraint's values interpolate to a polynomial of degree <= 31 * degree --
    the DECLARED degree of the gate, which is what the selector grouping and the quotient degree factor rely on."""
    rng = _rng(1000 + 31 * kind + len(params))
    wsize = 32
    rate_bits = int(np.ceil(np.log2(degree + 1)))
    m = wsize << rate_bits

    def low_degree_column():
        c = np.zeros(m, dtype=np.uint64)
        c[:wsize] = rng.integers(0, P, size=wsize, dtype=np.uint64)
        return orc.ntt(c)          # values of a degree < 32 polynomial on the size-m subgroup

    wires = np.stack([low_degree_column() for _ in range(234)])
    consts = np.stack([low_degree_column() for _ in range(nconst)]) if nconst else np.zeros((0, m), dtype=np.uint64)
    pih = tuple(int(x) for x in rng.integers(0, P, size=4, dtype=np.uint64))
    rows = [orc.gate_eval(kind, params, np.ascontiguousarray(wires[:, i]), consts=[int(x) for x in consts[:, i]], pi_hash=pih) for i in range(m)]
    vals = np.stack(rows, axis=1)   # [num_constraints][m]
    assert vals.shape[0] > 0
    worst = 0
    for t in range(vals.shape[0]):
        coeffs = orc.ntt(np.ascontiguousarray(vals[t]), inverse=True)
        nz = np.nonzero(coeffs)[0]
        deg = int(nz.max()) if nz.size else 0
        worst = max(worst, deg)
        assert deg <= (wsize - 1) * degree, (t, deg)
    # the bound is tight for at least one constraint of every gate with non-linear constraints (a declared degree that is
    # too generous would waste selector groups; plonky2's test only checks the upper bound, this is an extra sanity check)
    if degree > 1:
        assert worst > (wsize - 1) * (degree - 1), worst
