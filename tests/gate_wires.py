"""Satisfying wire assignments for the reference's five custom gates, restated from the
`get_wires` helpers / generators of the reference's own gate tests:
  arithmetic_u32.rs:486-530 (test get_wires) and :376-426 (generator)
  add_many_u32.rs:416-491, subtraction_u32.rs:397-474, range_check_u32.rs:262-334,
  comparison.rs:613-743
Wire layouts follow the gates' `wire_*` accessors (file:line in oracle/circuit.h)."""
P = 0xFFFFFFFF00000001
G_CONSTANT, G_PUBLIC_INPUT, G_ARITHMETIC, G_BASE_SUM, G_RANDOM_ACCESS = 1, 2, 3, 4, 5
G_U32_ARITHMETIC, G_U32_ADD_MANY, G_U32_SUBTRACTION, G_U32_RANGE_CHECK, G_COMPARISON = 7, 8, 9, 10, 11


def inv(a):
    return pow(a, P - 2, P)


def u32_arithmetic_wires(m0s, m1s, addends):
    """arithmetic_u32.rs test get_wires: routed wires of all ops first, then all limbs."""
    v0, v1 = [], []
    for m0, m1, a in zip(m0s, m1s, addends):
        out = (m0 * m1 + a) % (1 << 64)  # the reference test computes in u64 (non-canonical addends allowed)
        lo, hi = out & 0xFFFFFFFF, out >> 32
        diff = 0xFFFFFFFF - hi
        v0 += [m0, m1, a % P, lo, hi, inv(diff) if diff else 0]
        v1 += [(out >> (2 * j)) & 3 for j in range(32)]
    return v0 + v1


def u32_add_many_wires(addends, carries):
    na = len(addends[0])
    v0, v1 = [], []
    for ads, c in zip(addends, carries):
        s = sum(ads) + c
        res, oc = s & 0xFFFFFFFF, s >> 32
        v0 += list(ads) + [c, res, oc]
        v1 += [(res >> (2 * j)) & 3 for j in range(16)] + [(oc >> (2 * j)) & 3 for j in range(2)]
    return v0 + v1, na


def u32_subtraction_wires(xs, ys, borrows):
    v0, v1 = [], []
    for x, y, b in zip(xs, ys, borrows):
        init = (x - y - b) % P
        bout = 1 if init > (1 << 32) else 0
        res = (init + (bout << 32)) % P
        v0 += [x, y, b, res, bout]
        v1 += [(res >> (2 * j)) & 3 for j in range(16)]
    return v0 + v1


def u32_range_check_wires(limbs):
    aux = []
    for v in limbs:
        aux += [(v >> (2 * j)) & 3 for j in range(16)]
    return list(limbs) + aux


def comparison_wires(a, b, num_bits, num_chunks):
    cb = -(-num_bits // num_chunks)
    cs = 1 << cb
    fc = [(a >> (cb * i)) % cs for i in range(num_chunks)]
    sc = [(b >> (cb * i)) % cs for i in range(num_chunks)]
    eq = [1 if f == s else 0 for f, s in zip(fc, sc)]
    dummy = [1 if f == s else inv((s - f) % P) for f, s in zip(fc, sc)]
    msd, inter = 0, []
    for f, s in zip(fc, sc):
        if f != s:
            msd = (s - f) % P
            inter.append(0)
        else:
            inter.append(msd)
    t = (cs + msd) % P
    bits = [(t >> i) & 1 for i in range(cb + 1)]
    return [a, b, 1 if a <= b else 0, msd] + fc + sc + dummy + eq + inter + bits
