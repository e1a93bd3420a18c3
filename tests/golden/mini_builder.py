"""A miniature `CircuitBuilder` for hand-written ACIR-equivalent circuits (test infrastructure).

BASELINE.json configs[0] names the reference's `fibonacci` example program; SURVEY.md 8(d) maps it to
"a hand-written ACIR-equivalent of example_programs/fibonacci/src/main.nr:1-10 through the CPU
restatement + verifier".  Neither nargo nor Rust exist here, so the ACIR is written down by hand and
translated by the few builder calls the reference's AssertZero translator makes
(plonky2-backend/src/circuit_translation/assert_zero_translator.rs:25-38, 60-115):

    constant(q_c); for each linear term: mul_const(factor, w) then add(term, acc);
    for each mul term: mul(w1, w2), mul_const(factor, .), add(., acc); assert_zero(acc)

on top of a restatement of the small part of plonky2's builder those calls reach (gadgets/arithmetic.rs
`arithmetic` + its constant-folding special cases, ConstantGate allocation, the PublicInputGate row
`build()` always adds, Noop padding, selector column, WirePartition -> sigma).  The row order follows
what the reference's own proofs show for tiny circuits (tests/golden/reference_proofs.py: user gates,
PublicInputGate, ConstantGates, Noops).  Shapes: wide_ecc_config (circuit_translation/mod.rs:69):
234 wires, 80 routed, ArithmeticGate with 20 ops.  Output: the circuit blob of include/p2gpu.h and the
full wire matrix.  Pure Python/numpy; uses neither oracle/ nor the product.

The exact constant/slot allocation order of upstream `build()` is not pinned by anything the reference
ships for this program (there is no fibonacci proof in the tree), so this is a *counterpart* circuit --
same gates, same constraints -- not a byte-level claim about what `plonky2-backend prove` would emit.
"""
import numpy as np

P = 0xFFFFFFFF00000001
GEN = 14293326489335486720
ROOT32 = 7277203076849721926
W, R, K, QF, RATE_BITS, CAP_H, POW_BITS, QUERIES = 234, 80, 2, 8, 3, 4, 16, 28
NUM_OPS = 20
G_NOOP, G_CONSTANT, G_PUBLIC_INPUT, G_ARITHMETIC = 0, 1, 2, 3


def root_of_unity(bits):
    g = ROOT32
    for _ in range(bits, 32):
        g = g * g % P
    return g


class MiniBuilder:
    def __init__(self, seed=2024):
        self.ops = []            # (c0, c1, m0, m1, addend, out) targets
        self.consts = {}         # value -> target id
        self.parent = []         # union-find over targets
        self.values = {}         # witness values of ACIR witnesses: target -> value
        self.rng = np.random.default_rng(seed)

    # -- targets ---------------------------------------------------------------------------
    def add_virtual_target(self):
        self.parent.append(len(self.parent))
        return len(self.parent) - 1

    def _find(self, x):
        while self.parent[x] != x:
            self.parent[x] = self.parent[self.parent[x]]
            x = self.parent[x]
        return x

    def connect(self, a, b):
        ra, rb = self._find(a), self._find(b)
        if ra != rb:
            self.parent[max(ra, rb)] = min(ra, rb)

    def constant(self, c):
        c %= P
        if c not in self.consts:
            self.consts[c] = self.add_virtual_target()
        return self.consts[c]

    def zero(self):
        return self.constant(0)

    def one(self):
        return self.constant(1)

    def _as_const(self, t):
        for c, tt in self.consts.items():
            if tt == t:
                return c
        return None

    # -- gadgets/arithmetic.rs ---------------------------------------------------------------
    def arithmetic(self, c0, c1, m0, m1, addend):
        """c0 * m0 * m1 + c1 * addend, with upstream's special cases (no gate when foldable)."""
        c0 %= P
        c1 %= P
        zero = self.zero()
        k0, k1, ka = self._as_const(m0), self._as_const(m1), self._as_const(addend)
        first_zero = c0 == 0 or m0 == zero or m1 == zero
        second_zero = c1 == 0 or addend == zero
        first_const = 0 if first_zero else (k0 * k1 * c0 % P if k0 is not None and k1 is not None else None)
        second_const = 0 if second_zero else (ka * c1 % P if ka is not None else None)
        if first_const is not None and second_const is not None:
            return self.constant((first_const + second_const) % P)
        if first_zero and c1 == 1:
            return addend
        if second_zero:
            if k0 is not None and k0 * c0 % P == 1:
                return m1
            if k1 is not None and k1 * c0 % P == 1:
                return m0
        out = self.add_virtual_target()
        self.ops.append((c0, c1, m0, m1, addend, out))
        return out

    def mul(self, x, y):
        return self.arithmetic(1, 0, x, y, x)

    def add(self, x, y):
        return self.arithmetic(1, 1, x, self.one(), y)

    def mul_const(self, c, x):
        return self.mul(self.constant(c), x)

    def assert_zero(self, x):
        self.connect(x, self.zero())

    # -- AssertZero translation (assert_zero_translator.rs:25-38) -------------------------------
    def translate_assert_zero(self, witness_targets, mul_terms, linear, q_c):
        """witness_targets: {acir witness index: target}; mul_terms [(f, w1, w2)], linear [(f, w)]."""
        for _, a, b in mul_terms:
            for w in (a, b):
                witness_targets.setdefault(w, self.add_virtual_target())
        for _, w in linear:
            witness_targets.setdefault(w, self.add_virtual_target())
        acc = self.constant(q_c)
        for f, w in linear:
            acc = self.add(self.mul_const(f, witness_targets[w]), acc)
        for f, a, b in mul_terms:
            acc = self.add(self.mul_const(f, self.mul(witness_targets[a], witness_targets[b])), acc)
        self.assert_zero(acc)

    # -- build() + witness ------------------------------------------------------------------
    def build(self, witness_values):
        """witness_values: {target: value} for the ACIR witnesses.  Returns (blob, wires)."""
        zero = self.zero()
        # ArithmeticGate rows: ops with equal (c0, c1) share a row (upstream's slot reuse)
        rows, row_consts, open_rows = [], [], {}
        cell = {}                       # target -> list of (row, col)
        for (c0, c1, m0, m1, ad, out) in self.ops:
            key = (c0, c1)
            if key not in open_rows or len(rows[open_rows[key]]) == NUM_OPS:
                open_rows[key] = len(rows)
                rows.append([])
                row_consts.append(key)
            r = open_rows[key]
            k = len(rows[r])
            rows[r].append((m0, m1, ad, out))
            for j, t in enumerate((m0, m1, ad, out)):
                cell.setdefault(t, []).append((r, 4 * k + j))
        n_arith = len(rows)
        pi_row = n_arith
        for i in range(4):              # hash of zero public inputs = four copies of `zero`
            cell.setdefault(zero, []).append((pi_row, i))
        const_list = sorted(self.consts.items())   # plonky2 build(): constants_to_targets sorted by canonical value (pinned by the reference circuits)
        const_rows = (len(const_list) + 1) // 2
        for i, (c, t) in enumerate(const_list):
            cell.setdefault(t, []).append((pi_row + 1 + i // 2, i % 2))
        used = pi_row + 1 + const_rows
        d = max(2, (used - 1).bit_length())
        n = 1 << d
        # build() pads with NoopGates only when the row count is not a power of two already
        gates = ([(G_NOOP, 0, 0, 0, 0)] if used < n else []) + [(G_CONSTANT, 2, 2, 1, 2), (G_PUBLIC_INPUT, 0, 4, 1, 0),
                                                               (G_ARITHMETIC, NUM_OPS, NUM_OPS, 3, 2)]
        o = len(gates) - 3
        NC = 3
        constants = np.zeros((NC, n), dtype=np.uint64)
        row_gate = [o + 2] * n_arith + [o + 1] + [o] * const_rows + [0] * (n - used)
        constants[0, :] = row_gate
        for r, (c0, c1) in enumerate(row_consts):
            constants[1, r], constants[2, r] = c0, c1
        for i, (c, _) in enumerate(const_list):
            constants[1 + i % 2, pi_row + 1 + i // 2] = c
        # witness: constants, ACIR witnesses, then ops in creation order
        val = {t: c for c, t in const_list}
        val.update({t: v % P for t, v in witness_values.items()})
        def get(t):
            rt = self._find(t)
            for u, v in val.items():
                if self._find(u) == rt:
                    return v
            raise KeyError(t)
        for (c0, c1, m0, m1, ad, out) in self.ops:
            val[out] = (c0 * get(m0) % P * get(m1) + c1 * get(ad)) % P
        wires = np.zeros((W, n), dtype=np.uint64)
        for t, cells in cell.items():
            for (r, c) in cells:
                wires[c, r] = get(t)
        wires[4:R, pi_row] = self.rng.integers(0, P, size=R - 4, dtype=np.uint64)  # randomize_unused_pi_wires:
        wires[R:, pi_row] = self.rng.integers(0, P, size=W - R, dtype=np.uint64)    # routed or not
        # sigma: classes listed row by row, each cell maps to the next of its class
        classes = {}
        for t, cells in cell.items():
            classes.setdefault(self._find(t), []).extend(cells)
        g = root_of_unity(d)
        sub = [pow(g, i, P) for i in range(n)]
        k_is = [pow(GEN, j, P) for j in range(R)]
        sig = np.zeros((R, n), dtype=np.uint64)
        for c in range(R):
            for r in range(n):
                sig[c, r] = k_is[c] * sub[r] % P
        for cells in classes.values():
            cells = sorted(set(cells))
            assert all(c < R for _, c in cells)
            for (r, c), (r2, c2) in zip(cells, cells[1:] + cells[:1]):
                sig[c, r] = k_is[c2] * sub[r2] % P
        h = np.zeros(64, dtype=np.uint32)
        h[0], h[1], h[2], h[3], h[4], h[5], h[6] = 0x43473250, 1, d, W, R, NC, 1
        h[7], h[8], h[9], h[10], h[11], h[12] = K, QF, RATE_BITS, CAP_H, POW_BITS, QUERIES
        arity = []
        db = d
        while db > 5 and db + RATE_BITS - 4 >= CAP_H:
            arity.append(4)
            db -= 4
        h[13] = len(arity)
        for i, a in enumerate(arity):
            h[14 + i] = a
        h[22], h[23], h[24], h[25], h[26] = 0, len(gates), 0, 0, (R + QF - 1) // QF - 1
        gt = np.zeros((len(gates), 12), dtype=np.uint32)
        for i, (kind, p0, ncons, deg, nk) in enumerate(gates):
            gt[i] = [kind, p0, 0, 0, 0, 0, 0, len(gates), ncons, deg, nk, 0]
        raw = h.tobytes() + gt.tobytes() + np.array(k_is, dtype=np.uint64).tobytes() + constants.tobytes() + sig.tobytes()
        return np.frombuffer(raw, dtype=np.uint8).copy(), wires


def fibonacci():
    """example_programs/fibonacci/src/main.nr:1-10 -- no inputs, 13 additions folded by the Noir
    compiler to the constant 377; the ACIR that is left is one opcode binding the return witness:
        EXPR [ (1, _0) -377 ]        (return_values = [_0]; not a Plonky2 public input,
                                      circuit_translation/mod.rs:290-296)"""
    a, b = 0, 1
    for _ in range(13):
        a, b = b, a + b
    mb = MiniBuilder()
    wt = {}
    mb.translate_assert_zero(wt, [], [(1, 0)], -b)
    return mb.build({wt[0]: b})


def quadratic_example():
    """A second tiny ACIR program with mul terms (x * y - z = 0 and 3 x + 2 y - 17 = 0) so the
    ArithmeticGate's multiplicative path and two different constant pairs are exercised."""
    mb = MiniBuilder(seed=7)
    wt = {}
    x, y = 3, 4
    mb.translate_assert_zero(wt, [(1, 0, 1)], [(P - 1, 2)], 0)
    mb.translate_assert_zero(wt, [], [(3, 0), (2, 1)], -17)
    return mb.build({wt[0]: x, wt[1]: y, wt[2]: x * y})
