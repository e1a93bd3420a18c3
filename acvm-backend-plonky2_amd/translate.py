"""ACIR opcodes -> Plonky2 circuit: a restatement of the slice of the reference's translation layer that
BASELINE's named circuits need, on top of the library's own `build()` (p2gpu_build_blob).

Mirrors, with the reference's names:
  * ``CircuitBuilderFromAcirToPlonky2`` (plonky2-backend/src/circuit_translation/mod.rs:37-330):
    `translate_circuit`, `binary_number_target_for_witness/_constant`, `convert_binary_number_to_number`;
  * ``AssertZeroTranslator`` (assert_zero_translator.rs:25-38, 60-115);
  * ``Sha256CompressionTranslator`` (sha256_translator.rs:61-273) and ``BinaryDigitsTarget``
    (binary_digits_target.rs: rotate_right 19-41, shift_right 43-64, choose 66-83, majority 85-106, xor/and
    121-149, bit_xor 181-187, add_module_32_bits 189-224);
  * the part of plonky2 0.2.2's ``CircuitBuilder`` those reach (gadgets/arithmetic.rs `arithmetic` with its
    constant-folding special cases, `mul/add/sub/mul_sub/mul_add/mul_const_add`, `and/or/not/select`,
    `assert_bool`, gadgets/split_join.rs `split_le` / `le_sum` with BaseSumGate<2>, constants in ConstantGates,
    the PublicInputGate row of `build()`, Noop padding) -- recollection of an un-vendored crate, checked by what
    can be checked here: every circuit it emits is satisfiable only by the right values (the reference's own
    SHA-256 compression vector, tests/test_sha256_internal.rs:481-549, comes out of it), and the oracle and the
    GPU prove it to identical bytes.

What it is NOT: an ACIR *reader*.  The reference deserialises gzip + bincode `Program` bytes through the acvm
crate (noir_and_plonky2_serialization.rs:42-64); no compiled program exists in the tree to test a reader on, so
programs are given as Python data: [("assert_zero", mul_terms, linear, q_c), ("sha256_compression", inputs16,
hash8, outputs8), ("range", w, num_bits), ("and" | "xor", lhs, rhs, output, num_bits)] (mod.rs:131-155).  Public parameters (which add PoseidonGate rows in `build()`) are not supported here.

Host-side Python by design: translation runs once per circuit, off the hot path (north_star keeps this layer in
Rust; this file exists so that the named SHA256 circuit can be built and proved without it).
"""
import numpy as np

from .prover import build_blob
from .synth import poseidon_gate_row

P = 0xFFFFFFFF00000001
NUM_WIRES, NUM_ROUTED, NUM_OPS, BASE_SUM_LIMBS = 234, 80, 20, 63
G_NOOP, G_CONSTANT, G_PUBLIC_INPUT, G_ARITHMETIC, G_BASE_SUM, G_POSEIDON = 0, 1, 2, 3, 4, 6


class CircuitBuilder:
    """The slice of plonky2's CircuitBuilder<GoldilocksField, 2> (wide_ecc_config, mod.rs:69) the translators use."""

    def __init__(self, seed=2024, num_wires=NUM_WIRES):
        # num_wires: 234 = wide_ecc_config (what the reference builds with today, mod.rs:69); 135 = standard_recursion_config
        # (what the two proofs the reference ships were built with: tests/test_translate.py reproduces their circuits).
        # Both have 80 routed wires, so ArithmeticGate holds 20 operations and BaseSumGate<2> 63 limbs in either.
        self.num_wires = num_wires
        self.public_inputs = []   # targets, in registration order (circuit_builder.rs register_public_input)
        self.parent = []
        self.const_of, self.consts = {}, {}
        self.rows = []            # gate instances in creation order
        self.free_arith = {}      # (c0, c1) -> row with a free slot
        self.arith_results = {}   # (c0, c1, m0, m1, addend) -> output target of the identical earlier operation
        self.events = []          # witness generators in creation order
        self.rng = np.random.default_rng(seed)

    # -- targets, copy constraints --------------------------------------------------------------
    def add_virtual_target(self):
        self.parent.append(len(self.parent))
        return len(self.parent) - 1

    def find(self, x):
        p = self.parent
        r = x
        while p[r] != r:
            r = p[r]
        while p[x] != r:
            p[x], x = r, p[x]
        return r

    def connect(self, a, b):
        ra, rb = self.find(a), self.find(b)
        if ra != rb:
            if ra < rb:
                self.parent[rb] = ra
            else:
                self.parent[ra] = rb

    def constant(self, c):
        c %= P
        t = self.consts.get(c)
        if t is None:
            t = self.add_virtual_target()
            self.consts[c] = t
            self.const_of[t] = c
        return t

    def zero(self):
        return self.constant(0)

    def one(self):
        return self.constant(1)

    def two(self):
        return self.constant(2)

    def constant_bool(self, b):
        return self.constant(1 if b else 0)

    def _false(self):
        return self.constant_bool(False)

    def assert_zero(self, x):
        self.connect(x, self.zero())

    def register_public_input(self, t):
        self.public_inputs.append(t)

    # -- gadgets/arithmetic.rs --------------------------------------------------------------------
    def arithmetic(self, c0, c1, m0, m1, addend):
        c0 %= P
        c1 %= P
        zero = self.zero()
        k0, k1, ka = self.const_of.get(m0), self.const_of.get(m1), self.const_of.get(addend)
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
        # plonky2's `base_arithmetic_results`: an identical operation (same constants, same targets) returns the earlier
        # output instead of a fresh ArithmeticGate slot (gadgets/arithmetic.rs CircuitBuilder::arithmetic)
        op = (c0, c1, m0, m1, addend)
        cached = self.arith_results.get(op)
        if cached is not None:
            return cached
        key = (c0, c1)
        r = self.free_arith.get(key)
        if r is None or len(self.rows[r]["ops"]) == NUM_OPS:
            r = len(self.rows)
            self.rows.append({"kind": "arith", "c": key, "ops": []})
            self.free_arith[key] = r
        out = self.add_virtual_target()
        self.rows[r]["ops"].append((m0, m1, addend, out))
        self.events.append(("arith", c0, c1, m0, m1, addend, out))
        self.arith_results[op] = out
        return out

    def mul(self, x, y):
        return self.arithmetic(1, 0, x, y, x)

    def add(self, x, y):
        return self.arithmetic(1, 1, x, self.one(), y)

    def sub(self, x, y):
        return self.arithmetic(1, P - 1, x, self.one(), y)

    def mul_add(self, x, y, z):
        return self.arithmetic(1, 1, x, y, z)

    def mul_sub(self, x, y, z):
        return self.arithmetic(1, P - 1, x, y, z)

    def mul_const(self, c, x):
        return self.mul(self.constant(c), x)

    def mul_const_add(self, c, x, y):
        return self.mul_add(self.constant(c), x, y)

    # -- booleans (gadgets/arithmetic.rs, gadgets/select.rs) ---------------------------------------
    def assert_bool(self, b):
        self.connect(self.mul_sub(b, b, b), self.zero())

    def add_virtual_bool_target_safe(self):
        b = self.add_virtual_target()
        self.assert_bool(b)
        return b

    def and_(self, a, b):
        return self.mul(a, b)

    def or_(self, a, b):
        return self.add(self.arithmetic(P - 1, 1, a, b, a), b)

    def not_(self, b):
        return self.sub(self.one(), b)

    def select(self, b, x, y):
        return self.mul_sub(b, x, self.mul_sub(b, y, y))

    # -- gadgets/split_join.rs ----------------------------------------------------------------------
    def _base_sum_row(self, num_limbs):
        row = {"kind": "basesum", "L": num_limbs, "sum": self.add_virtual_target(),
               "limbs": [self.add_virtual_target() for _ in range(num_limbs)]}
        self.rows.append(row)
        return row

    def split_le(self, integer, num_bits):
        if num_bits == 0:
            return []
        k = -(-num_bits // BASE_SUM_LIMBS)
        gates = [self._base_sum_row(BASE_SUM_LIMBS) for _ in range(k)]
        bits = [t for g in gates for t in g["limbs"]]
        for b in bits[num_bits:]:
            self.assert_zero(b)
        bits = bits[:num_bits]
        acc = self.zero()
        for g in reversed(gates):
            acc = self.mul_const_add(pow(2, BASE_SUM_LIMBS, P), acc, g["sum"])
        self.connect(acc, integer)
        self.events.append(("split", integer, gates))
        return bits

    def le_sum(self, bits):
        bits = list(bits)
        n = len(bits)
        if n == 0:
            return self.zero()
        if n - 1 <= NUM_OPS:
            s = bits[-1]
            for b in reversed(bits[:-1]):
                s = self.mul_add(self.two(), s, b)
            return s
        row = self._base_sum_row(n)
        for b, l in zip(bits, row["limbs"]):
            self.connect(b, l)
        self.events.append(("lesum", row))
        return row["sum"]

    # -- build(): rows -> blob (library) + witness ----------------------------------------------------
    def build(self, witness_values):
        """witness_values: {target: value}.  Returns (blob, wires) for CircuitData(blob).prove(wires)."""
        zero = self.zero()
        # circuit_builder.rs build(): the public inputs are hashed IN CIRCUIT -- hash_n_to_hash_no_pad::<PoseidonHash>, an
        # overwrite-mode sponge: one PoseidonGate row per 8 inputs, swap wire tied to zero, the state starts as twelve copies of
        # `zero` -- and the first four outputs are routed to a PublicInputGate added right behind; no public inputs: the hash is
        # four copies of `zero` and no PoseidonGate exists (the reference's basic_div / basic_if circuits show both cases)
        state = [zero] * 12
        for off in range(0, len(self.public_inputs), 8):
            chunk = self.public_inputs[off:off + 8]
            state = list(chunk) + state[len(chunk):]
            row = {"kind": "poseidon", "in": state, "out": [self.add_virtual_target() for _ in range(12)], "swap": zero}
            self.rows.append(row)
            self.events.append(("poseidon", row))
            state = row["out"]
        pi_hash = state[:4]
        rows = list(self.rows)
        pi_row = len(rows)
        # ConstantGate rows: plonky2 walks constants_to_targets sorted by the constant's canonical value, two per gate (both
        # reference circuits: 0, 1 | 2^63, p - 1 and 0, 1 | p - 1, -)
        const_list = sorted(self.consts.items())
        const_rows = (len(const_list) + 1) // 2
        used = pi_row + 1 + const_rows
        d = max(2, (used - 1).bit_length())
        n = 1 << d
        # gate set, sorted by (degree, id) like CommonCircuitData.gates
        kinds = {("noop",)} if used < n else set()
        kinds |= {("const",), ("pi",)}
        for r in rows:
            kinds.add(("arith",) if r["kind"] == "arith" else ("poseidon",) if r["kind"] == "poseidon" else ("basesum", r["L"]))
        spec = {("noop",): (0, "NoopGate", (G_NOOP, (0, 0, 0, 0), 0, 0)),
                ("const",): (1, "ConstantGate { num_consts: 2 }", (G_CONSTANT, (2, 0, 0, 0), 1, 2)),
                ("pi",): (1, "PublicInputGate", (G_PUBLIC_INPUT, (0, 0, 0, 0), 1, 0)),
                ("arith",): (3, "ArithmeticGate { num_ops: 20 }", (G_ARITHMETIC, (NUM_OPS, 0, 0, 0), 3, 2)),
                ("poseidon",): (7, "PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>",
                                (G_POSEIDON, (0, 0, 0, 0), 7, 0))}
        for k in kinds:
            if k[0] == "basesum":
                spec[k] = (2, "BaseSumGate { num_limbs: %d } + Base: 2" % k[1], (G_BASE_SUM, (2, k[1], 0, 0), 2, 0))
        order = sorted(kinds, key=lambda k: (spec[k][0], spec[k][1]))
        index = {k: i for i, k in enumerate(order)}
        gates = [spec[k][2] for k in order]
        row_gate = np.zeros(n, dtype=np.uint32)
        row_consts = np.zeros((2, n), dtype=np.uint64)
        cells = {}                                  # class root -> [(row, col)]

        def put(t, r, c):
            cells.setdefault(self.find(t), []).append((r, c))

        for r, g in enumerate(rows):
            if g["kind"] == "arith":
                row_gate[r] = index[("arith",)]
                row_consts[0, r], row_consts[1, r] = g["c"]
                for k, op in enumerate(g["ops"]):
                    for j, t in enumerate(op):
                        put(t, r, 4 * k + j)
            elif g["kind"] == "poseidon":
                row_gate[r] = index[("poseidon",)]
                g["row"] = r
                for j, t in enumerate(g["in"]):
                    put(t, r, j)
                for j, t in enumerate(g["out"]):
                    put(t, r, 12 + j)
                put(g["swap"], r, 24)
            else:
                row_gate[r] = index[("basesum", g["L"])]
                put(g["sum"], r, 0)
                for j, t in enumerate(g["limbs"]):
                    put(t, r, 1 + j)
        row_gate[pi_row] = index[("pi",)]
        for i in range(4):                          # (no public inputs: four copies of `zero`)
            put(pi_hash[i], pi_row, i)
        for i, (c, t) in enumerate(const_list):
            r = pi_row + 1 + i // 2
            row_gate[r] = index[("const",)]
            row_consts[i % 2, r] = c
            put(t, r, i % 2)
        if const_rows:
            row_gate[pi_row + 1:pi_row + 1 + const_rows] = index[("const",)]
        if used < n:
            row_gate[used:] = index[("noop",)]
        copies = []
        for cl in cells.values():
            for (r0, c0), (r1, c1) in zip(cl, cl[1:]):
                copies.append((r0, c0, r1, c1))
        blob = build_blob(d, gates, row_gate, row_consts, np.array(copies, dtype=np.uint32).reshape(-1, 4),
                          num_public_inputs=len(self.public_inputs), num_wires=self.num_wires)
        # witness: the generators, in creation order, until nothing changes
        val = {}
        for c, t in const_list:
            val[self.find(t)] = c
        for t, v in witness_values.items():
            rt = self.find(t)
            v %= P
            if rt in val and val[rt] != v:
                raise ValueError("witness value contradicts the circuit")
            val[rt] = v
        pending = list(self.events)
        while pending:
            rest = []
            for ev in pending:
                if ev[0] == "arith":
                    _, c0, c1, m0, m1, ad, out = ev
                    a = val.get(self.find(m0)) if c0 else 0
                    b = val.get(self.find(m1)) if c0 else 0
                    c = val.get(self.find(ad)) if c1 else 0
                    if a is None or b is None or c is None:
                        rest.append(ev)
                        continue
                    self._set(val, out, (c0 * a % P * b + c1 * c) % P)
                elif ev[0] == "poseidon":
                    ins = [val.get(self.find(t)) for t in ev[1]["in"]]
                    if any(v is None for v in ins):
                        rest.append(ev)
                        continue
                    ev[1]["wires"] = poseidon_gate_row(ins)
                    for j, t in enumerate(ev[1]["out"]):
                        self._set(val, t, int(ev[1]["wires"][12 + j]))
                elif ev[0] == "split":
                    v = val.get(self.find(ev[1]))
                    if v is None:
                        rest.append(ev)
                        continue
                    for g in ev[2]:
                        part = v & ((1 << g["L"]) - 1)
                        v >>= g["L"]
                        self._set(val, g["sum"], part)
                        for j, l in enumerate(g["limbs"]):
                            self._set(val, l, (part >> j) & 1)
                else:
                    bits = [val.get(self.find(l)) for l in ev[1]["limbs"]]
                    if any(b is None for b in bits):
                        rest.append(ev)
                        continue
                    self._set(val, ev[1]["sum"], sum(b << j for j, b in enumerate(bits)) % P)
            if len(rest) == len(pending):
                raise ValueError("witness generation is stuck: some inputs were not assigned")
            pending = rest
        W = self.num_wires
        wires = np.zeros((W, n), dtype=np.uint64)
        for g in rows:                              # the PoseidonGate's internal wires (deltas, S-box inputs): its generator's
            if g["kind"] == "poseidon":
                wires[:135, g["row"]] = g["wires"]
        for rt, cl in cells.items():
            v = val.get(rt)
            if v is None:
                # the dummy addend of a product-only op whose operand is otherwise unconstrained
                raise ValueError("a wire has no value")
            for (r, c) in cl:
                wires[c, r] = v
        # circuit_builder.rs randomize_unused_pi_wires: every wire of the PublicInputGate row after the hash, routed
        # or not, gets a random value (so no wire column of a real witness is zero in every row)
        wires[4:NUM_ROUTED, pi_row] = self.rng.integers(0, P, size=NUM_ROUTED - 4, dtype=np.uint64)
        wires[NUM_ROUTED:, pi_row] = self.rng.integers(0, P, size=W - NUM_ROUTED, dtype=np.uint64)
        self.public_input_values = [val[self.find(t)] for t in self.public_inputs]
        self.pi_row = pi_row
        self.values = val
        return blob, wires

    def _set(self, val, t, v):
        rt = self.find(t)
        if rt in val and val[rt] != v:
            raise ValueError("unsatisfiable: a generator contradicts an assigned value")
        val[rt] = v

    def value_of(self, t):
        return self.values[self.find(t)]


class BinaryDigitsTarget:
    """binary_digits_target.rs: a number as its bits, most significant first."""

    def __init__(self, bits):
        self.bits = list(bits)

    @staticmethod
    def rotate_right(t, times, b):
        n = len(t.bits)
        new = []
        for i in list(range(n - times, n)) + list(range(0, n - times)):
            nb = b.add_virtual_bool_target_safe()
            b.connect(t.bits[i], nb)
            new.append(nb)
        return BinaryDigitsTarget(new)

    @staticmethod
    def shift_right(t, times, b):
        new = [b.constant(0) for _ in range(times)]
        for i in range(len(t.bits) - times):
            nb = b.add_virtual_bool_target_safe()
            b.connect(t.bits[i], nb)
            new.append(nb)
        return BinaryDigitsTarget(new)

    @staticmethod
    def choose(chooser, on_true, on_false, b):
        return BinaryDigitsTarget([b.select(c, t, f) for c, t, f in zip(chooser.bits, on_true.bits, on_false.bits)])

    @staticmethod
    def majority(a, bb, c, b):
        out = []
        for b0, b1, b2 in zip(c.bits, a.bits, bb.bits):
            on_true = b.or_(b1, b2)
            on_false = b.and_(b1, b2)
            out.append(b.select(b0, on_true, on_false))
        return BinaryDigitsTarget(out)

    @staticmethod
    def bit_xor(x, y, b):
        x_or_y = b.or_(x, y)
        x_and_y = b.and_(x, y)
        return b.and_(x_or_y, b.not_(x_and_y))

    @staticmethod
    def xor(x, y, b):
        return BinaryDigitsTarget([BinaryDigitsTarget.bit_xor(p, q, b) for p, q in zip(x.bits, y.bits)])

    @staticmethod
    def add_module_32_bits(x, y, b):
        assert len(x.bits) == len(y.bits)
        partial_sum = [BinaryDigitsTarget.bit_xor(p, q, b) for p, q in zip(x.bits, y.bits)]
        partial_carries = [b.and_(p, q) for p, q in zip(x.bits, y.bits)]
        carry_in = b._false()
        out = []
        for i in reversed(range(len(x.bits))):
            s = BinaryDigitsTarget.bit_xor(partial_sum[i], carry_in, b)
            pair = b.and_(carry_in, partial_sum[i])
            carry_in = b.or_(partial_carries[i], pair)
            out.append(s)
        out.reverse()
        return BinaryDigitsTarget(out)


SHA256_K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]


class CircuitBuilderFromAcirToPlonky2:
    """mod.rs:37-330, for programs given as Python data (see the module docstring)."""

    def __init__(self, num_wires=NUM_WIRES):
        self.builder = CircuitBuilder(num_wires=num_wires)
        self.witness_target_map = {}

    def _target(self, w):
        if w not in self.witness_target_map:
            self.witness_target_map[w] = self.builder.add_virtual_target()
        return self.witness_target_map[w]

    def binary_number_target_for_witness(self, w, digits):
        return BinaryDigitsTarget(reversed(self.builder.split_le(self._target(w), digits)))

    def binary_number_target_for_constant(self, constant, digits):
        return BinaryDigitsTarget(reversed([self.builder.constant_bool(bool(constant & (1 << i))) for i in range(digits)]))

    def convert_binary_number_to_number(self, a):
        return self.builder.le_sum(reversed(a.bits))

    # assert_zero_translator.rs:25-38
    def translate_assert_zero(self, mul_terms, linear, q_c):
        b = self.builder
        for _, w1, w2 in mul_terms:
            self._target(w1)
            self._target(w2)
        for _, w in linear:
            self._target(w)
        acc = b.constant(q_c)
        for f, w in linear:
            acc = b.add(b.mul_const(f, self._target(w)), acc)
        for f, w1, w2 in mul_terms:
            acc = b.add(b.mul_const(f, b.mul(self._target(w1), self._target(w2))), acc)
        b.assert_zero(acc)

    # sha256_translator.rs:61-117
    def translate_sha256_compression(self, inputs, hash_values, outputs):
        b, B = self.builder, BinaryDigitsTarget
        w = [self.binary_number_target_for_witness(i, 32) for i in inputs]

        def sig(t, r1, r2, r3, last_is_shift):
            x1, x2 = B.rotate_right(t, r1, b), B.rotate_right(t, r2, b)
            x3 = B.shift_right(t, r3, b) if last_is_shift else B.rotate_right(t, r3, b)
            return B.xor(B.xor(x1, x2, b), x3, b)

        for t in range(16, 64):
            s1 = B.add_module_32_bits(sig(w[t - 2], 17, 19, 10, True), w[t - 7], b)
            s2 = B.add_module_32_bits(sig(w[t - 15], 7, 18, 3, True), w[t - 16], b)
            w.append(B.add_module_32_bits(s1, s2, b))
        k = [self.binary_number_target_for_constant(c, 32) for c in SHA256_K]
        h0 = [self.binary_number_target_for_witness(i, 32) for i in hash_values]
        a, bb, c, d, e, f, g, h = h0
        for t in range(64):
            big1 = sig(e, 6, 11, 25, False)
            ch = B.choose(e, f, g, b)
            s0 = B.add_module_32_bits(k[t], w[t], b)
            s1 = B.add_module_32_bits(h, big1, b)
            s2 = B.add_module_32_bits(ch, s0, b)
            t1 = B.add_module_32_bits(s1, s2, b)
            big0 = sig(a, 2, 13, 22, False)
            maj = B.majority(a, bb, c, b)
            t2 = B.add_module_32_bits(big0, maj, b)
            a, bb, c, d, e, f, g, h = B.add_module_32_bits(t1, t2, b), a, bb, c, B.add_module_32_bits(d, t1, b), e, f, g
        for ow, x0, x1 in zip(outputs, h0, (a, bb, c, d, e, f, g, h)):
            self.witness_target_map[ow] = self.convert_binary_number_to_number(B.add_module_32_bits(x0, x1, b))

    # mod.rs:131-139: BlackBoxFuncCall::RANGE -> builder.range_check = split_le
    def translate_range(self, w, num_bits):
        assert num_bits <= 33, "Range checks with more than 33 bits are not allowed yet while using Plonky2 prover"
        self.builder.split_le(self._target(w), num_bits)

    # mod.rs:140-155, 222-238: AND / XOR through the bit decompositions
    def translate_bitwise(self, lhs, rhs, output, num_bits, xor):
        x = self.binary_number_target_for_witness(lhs, num_bits)
        y = self.binary_number_target_for_witness(rhs, num_bits)
        b = self.builder
        bits = ([BinaryDigitsTarget.bit_xor(p, q, b) for p, q in zip(x.bits, y.bits)] if xor
                else [b.and_(p, q) for p, q in zip(x.bits, y.bits)])
        self.witness_target_map[output] = self.convert_binary_number_to_number(BinaryDigitsTarget(bits))

    def translate_circuit(self, opcodes, public_parameters=(), private_parameters=()):
        # mod.rs:290-310 _register_witnesses_from_acir_circuit: public parameters first -- a fresh target each, registered as a
        # Plonky2 public input -- then the private ones (return values are NOT public inputs).  The reference holds
        # both in BTreeSets (acir Circuit::public_parameters / private_parameters): iteration is by ascending witness index
        # whatever order the caller lists them in, and that order fixes the public-input order (hence the in-circuit
        # Poseidon hash, the circuit digest and the proof bytes)
        for w in sorted(set(public_parameters)):
            t = self.builder.add_virtual_target()
            self.builder.register_public_input(t)
            self.witness_target_map[w] = t
        for w in sorted(set(private_parameters)):
            self._target(w)
        for op in opcodes:
            if op[0] == "assert_zero":
                self.translate_assert_zero(*op[1:])
            elif op[0] == "sha256_compression":
                self.translate_sha256_compression(*op[1:])
            elif op[0] == "range":
                self.translate_range(*op[1:])
            elif op[0] in ("and", "xor"):
                self.translate_bitwise(*op[1:], xor=op[0] == "xor")
            else:
                raise NotImplementedError(op[0])

    def build(self, acir_witness):
        """acir_witness: {witness index: value} for every witness the solver would hand over (at least the
        inputs; outputs given are checked against what the circuit forces).  Returns (blob, wires)."""
        vals = {self.witness_target_map[w]: v for w, v in acir_witness.items() if w in self.witness_target_map}
        return self.builder.build(vals)

    def public_inputs(self):
        """Values of the registered public inputs, in order (after build): what p2gpu_prove takes beside the wires."""
        return list(self.builder.public_input_values)

    def witness_value(self, w):
        return self.builder.value_of(self.witness_target_map[w])
