"""The reference's in-tree gate vectors on the HIP path ("Missing 2").

tests/test_oracle_gates.py replays the satisfying assignments and the negatives of the reference's own gate tests
(arithmetic_u32.rs:533-605 incl. the 0xFFFFFFFF00000001 canonicity negative at :584, add_many_u32.rs:417-491,
subtraction_u32.rs:398-474, range_check_u32.rs:262-334 incl. the bad limb at :329, comparison.rs:564-743) on the ORACLE's
evaluator only.  Here the same wire vectors (tests/gate_wires.py, the restated `get_wires` helpers) become the rows of
one-gate circuits that are proved on the MI355X through the C ABI, so they reach the device's base-field evaluator
(csrc/gates.hpp inside quotient_kernel) and the host's extension-field one (the plonk-identity self-check):
  * a satisfying row set proves, the bytes equal the oracle's proof, both verifiers accept;
  * every negative of the reference's tests (and a few more single-wire corruptions) comes back as
    P2GPU_E_UNSATISFIED (-5) -- with the self-check off the product emits a proof and both verifiers reject it,
    i.e. the device evaluator really produced a non-vanishing quotient for that row.
"""
import numpy as np
import pytest

from conftest import P
from gate_wires import (G_COMPARISON, G_U32_ADD_MANY, G_U32_ARITHMETIC, G_U32_RANGE_CHECK, G_U32_SUBTRACTION, comparison_wires,
                        u32_add_many_wires, u32_arithmetic_wires, u32_range_check_wires, u32_subtraction_wires)

pytestmark = pytest.mark.gpu
G_NOOP = 0
E_UNSATISFIED = -5
E_VERIFY = -9
W = 234


@pytest.fixture(scope="module")
def gpu(pkg):
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    assert "gfx950" in pkg.device_info()["name"]
    return True


def _u32s(rng, n):
    return [int(x) for x in rng.integers(0, 1 << 32, size=n, dtype=np.uint64)]


def one_gate_circuit(pkg, kind, params, degree, rows, d=3, consts=None):
    """2^d rows: `rows` (wire vectors) under the gate, the rest NoopGate; no copy constraints, no public inputs.
    gates sorted by (degree, id) like CommonCircuitData.gates: Noop (degree 0) first.  consts: [num_constants][n] gate
    constants per row (ConstantGate, ArithmeticGate, RandomAccessGate's extra constants)."""
    n = 1 << d
    assert len(rows) <= n
    rc = np.zeros((0, n), dtype=np.uint64) if consts is None else np.ascontiguousarray(consts, dtype=np.uint64)
    gates = [(G_NOOP, (), 0, 0), (kind, tuple(params), degree, rc.shape[0])]
    row_gate = [1] * len(rows) + [0] * (n - len(rows))
    blob = pkg.build_blob(d, gates, row_gate, rc, np.zeros((0, 4), dtype=np.uint32))
    wires = np.zeros((W, n), dtype=np.uint64)
    for r, w in enumerate(rows):
        assert len(w) <= W and all(0 <= int(v) < P for v in w)
        wires[:len(w), r] = np.array([int(v) for v in w], dtype=np.uint64)
    return blob, wires


def prove_both(pkg, orc, blob, wires):
    cd = pkg.CircuitData(blob)
    oc = orc.OracleCircuit(blob)
    try:
        got = cd.prove(wires).to_bytes()
        want, _ = oc.prove(wires)
        assert got == want, "GPU proof differs from the oracle's"
        assert oc.verify(got)
        cd.verify(got)
    finally:
        cd.close()
        oc.close()


def expect_unsatisfied(pkg, orc, blob, wires):
    cd = pkg.CircuitData(blob)
    oc = orc.OracleCircuit(blob)
    try:
        with pytest.raises(pkg.P2GpuError) as e:
            cd.prove(wires)
        assert e.value.code == E_UNSATISFIED, e.value
        # the same witness with the host-side self-check off: the device evaluator's quotient does not vanish on H, so
        # the proof that comes out is rejected by both verifiers (upstream's prover would have panicked in trim_to_len)
        cd.set("self_check", 0)
        # (no try / except around this prove: with the self-check off the only acceptable outcomes are proof bytes that the
        # verifiers then reject -- a device fault or any other error code here is a FAILURE, not a rejection)
        bad = cd.prove(wires).to_bytes()
        assert not oc.verify(bad)
        with pytest.raises(pkg.P2GpuError) as ev:
            cd.verify(bad)
        assert ev.value.code == E_VERIFY, ev.value
    finally:
        cd.close()
        oc.close()


# ---- G1 U32ArithmeticGate (arithmetic_u32.rs) ---------------------------------------------------------------------
@pytest.mark.parametrize("num_ops", [3, 6])   # 3 = the reference test's NUM_U32_ARITHMETIC_OPS, 6 = new_from_config at 234 / 80
def test_u32_arithmetic_vectors_prove_on_gpu(pkg, orc, gpu, num_ops):
    rng = np.random.default_rng(1)
    rows = [u32_arithmetic_wires(_u32s(rng, num_ops), _u32s(rng, num_ops), _u32s(rng, num_ops)) for _ in range(5)]
    rows.append(u32_arithmetic_wires([0xFFFFFFFF] * num_ops, [0xFFFFFFFF] * num_ops, [0xFFFFFFFF] * num_ops))  # maximal output
    rows.append(u32_arithmetic_wires([0] * num_ops, [0] * num_ops, [0] * num_ops))
    blob, wires = one_gate_circuit(pkg, G_U32_ARITHMETIC, [num_ops], 4, rows)
    prove_both(pkg, orc, blob, wires)


def test_u32_arithmetic_canonicity_negative_on_gpu(pkg, orc, gpu):
    # arithmetic_u32.rs:572-605 (the addend 0xFFFFFFFF00000001 at :584): output_high = 2^32 - 1 with a non-zero
    # output_low is the non-canonical encoding the gate's first constraint exists to reject
    rng = np.random.default_rng(2)
    good = u32_arithmetic_wires(_u32s(rng, 3), _u32s(rng, 3), _u32s(rng, 3))
    bad = u32_arithmetic_wires([0] * 3, [0] * 3, [0xFFFFFFFF00000001] * 3)
    blob, wires = one_gate_circuit(pkg, G_U32_ARITHMETIC, [3], 4, [good, bad, good])
    expect_unsatisfied(pkg, orc, blob, wires)
    # a wrong limb and a wrong product, one wire each
    for idx, delta in ((18, 1), (3, 1)):
        w = list(good)
        w[idx] = (w[idx] + delta) % P
        blob, wires = one_gate_circuit(pkg, G_U32_ARITHMETIC, [3], 4, [good, w])
        expect_unsatisfied(pkg, orc, blob, wires)


# ---- G2 U32AddManyGate (add_many_u32.rs) ---------------------------------------------------------------------------
@pytest.mark.parametrize("num_addends,num_ops", [(10, 3), (3, 9), (16, 4), (2, 5)])   # (10, 3): add_many_u32.rs:424-426
def test_u32_add_many_vectors_on_gpu(pkg, orc, gpu, num_addends, num_ops):
    rng = np.random.default_rng(num_addends)
    rows = []
    for _ in range(6):
        addends = [_u32s(rng, num_addends) for _ in range(num_ops)]
        carries = _u32s(rng, num_ops) if num_addends < 16 else [0] * num_ops
        rows.append(u32_add_many_wires(addends, carries)[0])
    blob, wires = one_gate_circuit(pkg, G_U32_ADD_MANY, [num_addends, num_ops], 4, rows)
    prove_both(pkg, orc, blob, wires)
    bad = list(rows[0])
    bad[num_addends + 1] = (bad[num_addends + 1] + 1) % P   # wrong result
    blob, wires = one_gate_circuit(pkg, G_U32_ADD_MANY, [num_addends, num_ops], 4, [rows[1], bad])
    expect_unsatisfied(pkg, orc, blob, wires)


# ---- G3 U32SubtractionGate (subtraction_u32.rs) ---------------------------------------------------------------------
@pytest.mark.parametrize("num_ops", [3, 11])   # 3 = the reference test, 11 = new_from_config at 234 / 80
def test_u32_subtraction_vectors_on_gpu(pkg, orc, gpu, num_ops):
    rng = np.random.default_rng(3)
    rows = []
    for _ in range(6):
        xs, ys = _u32s(rng, num_ops), _u32s(rng, num_ops)
        xs[0], ys[0] = 5, 7                  # borrow out
        xs[1], ys[1] = 9, 9                  # zero
        rows.append(u32_subtraction_wires(xs, ys, [int(b) for b in rng.integers(0, 2, size=num_ops)]))
    blob, wires = one_gate_circuit(pkg, G_U32_SUBTRACTION, [num_ops], 4, rows)
    prove_both(pkg, orc, blob, wires)
    bad = list(rows[0])
    bad[4] = 2                               # output borrow must be a bit
    blob, wires = one_gate_circuit(pkg, G_U32_SUBTRACTION, [num_ops], 4, [rows[1], bad])
    expect_unsatisfied(pkg, orc, blob, wires)


# ---- G4 U32RangeCheckGate (range_check_u32.rs) ----------------------------------------------------------------------
def test_u32_range_check_vectors_on_gpu(pkg, orc, gpu):
    rng = np.random.default_rng(4)
    rows = [u32_range_check_wires(_u32s(rng, 8)) for _ in range(7)] + [u32_range_check_wires([0xFFFFFFFF] * 8)]
    blob, wires = one_gate_circuit(pkg, G_U32_RANGE_CHECK, [8], 4, rows)
    prove_both(pkg, orc, blob, wires)
    # range_check_u32.rs:318-334 (test_gate_constraint_bad): a limb >= 2^32 has no 16-digit base-4 decomposition
    bad = u32_range_check_wires([1 << 32] + [0] * 7)
    blob, wires = one_gate_circuit(pkg, G_U32_RANGE_CHECK, [8], 4, [rows[0], bad])
    expect_unsatisfied(pkg, orc, blob, wires)
    # an auxiliary limb outside {0, 1, 2, 3} whose recomposition still matches
    w = list(rows[1])
    w[8], w[9] = (w[8] + 4) % P, (w[9] - 1) % P
    blob, wires = one_gate_circuit(pkg, G_U32_RANGE_CHECK, [8], 4, [w])
    expect_unsatisfied(pkg, orc, blob, wires)


# ---- G5 ComparisonGate (comparison.rs), as instantiated by multiple_comparison.rs:28-34: new(32, 16) -------------------
def test_comparison_vectors_on_gpu(pkg, orc, gpu):
    nb, nc = 32, 16
    rng = np.random.default_rng(5)
    a = int(rng.integers(0, 1 << (nb - 1)))
    b = int(rng.integers(a, 1 << (nb - 1)))
    rows = [comparison_wires(x, y, nb, nc) for x, y in ((a, b), (a, a), (b, a), (0, 0), ((1 << nb) - 1, 0), (0, (1 << nb) - 1), (a, a + 1))]
    blob, wires = one_gate_circuit(pkg, G_COMPARISON, [nb, nc], 4, rows)
    prove_both(pkg, orc, blob, wires)
    bad = comparison_wires(a, b, nb, nc)
    bad[2] ^= 1                              # wrong result bool (comparison.rs:700-743 negative)
    blob, wires = one_gate_circuit(pkg, G_COMPARISON, [nb, nc], 4, [rows[0], bad])
    expect_unsatisfied(pkg, orc, blob, wires)


# ---- every gate kind on edge words: 0, 1, 2^32 - 1, 2^32, p - 1 ... ---------------------------------------------------------
# The device evaluator keeps some intermediates as congruent, not canonical, words (gl.hpp _nc forms, round 3); gate rows
# whose wires sit on the boundaries of the field and of the 32-bit limbs push its carry chains where random witnesses never
# go.  The rows do not satisfy their gates: with the self-check off both provers still emit bytes, which must be equal
# (the quotient's coefficients are what they are), and the verifiers must reject them.  Model: arithmetic_u32.rs:572-605.
EDGE = [0, 1, 2, 3, (1 << 32) - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, 1 << 63, P - (1 << 32), P - (1 << 32) + 1, P - 2, P - 1]
KINDS = [  # (kind, params, degree, num_constants): include/p2gpu.h "Gate kinds"
    (1, (2,), 1, 2), (3, (20,), 3, 2), (4, (2, 32), 2, 0), (4, (4, 16), 4, 0), (5, (2, 4, 2), 4, 2),
    (G_U32_ARITHMETIC, (6,), 4, 0), (G_U32_ADD_MANY, (5, 4), 4, 0), (G_U32_SUBTRACTION, (11,), 4, 0), (G_U32_RANGE_CHECK, (8,), 4, 0),
    (G_COMPARISON, (32, 16), 4, 0),
]


@pytest.mark.parametrize("kind,params,degree,nconst", KINDS)
def test_gate_rows_on_edge_words_match_the_oracle(pkg, orc, gpu, kind, params, degree, nconst):
    rng = np.random.default_rng(100 + kind + len(params))
    n = 8
    rows = []
    for r in range(n - 1):
        if r == 0:
            w = [P - 1] * W
        elif r == 1:
            w = [(1 << 32) - 1] * W
        elif r == 2:
            w = [0] * W
        else:
            w = [EDGE[int(i)] for i in rng.integers(0, len(EDGE), size=W)]
        rows.append(w)
    consts = np.array([[EDGE[int(i)] for i in rng.integers(0, len(EDGE), size=n)] for _ in range(nconst)], dtype=np.uint64).reshape(nconst, n)
    blob, wires = one_gate_circuit(pkg, kind, params, degree, rows, consts=consts if nconst else None)
    cd, oc = pkg.CircuitData(blob), orc.OracleCircuit(blob)
    try:
        cd.set("self_check", 0)
        got = cd.prove(wires).to_bytes()
        want, _ = oc.prove(wires)
        assert got == want, "GPU bytes differ from the oracle's on edge-word gate rows"
        assert not oc.verify(got)
        with pytest.raises(pkg.P2GpuError):
            cd.verify(got)
        cd.set("self_check", 1)
        with pytest.raises(pkg.P2GpuError) as e:
            cd.prove(wires)
        assert e.value.code == E_UNSATISFIED
    finally:
        cd.close()
        oc.close()
