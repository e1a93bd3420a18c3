#!/usr/bin/env python3
"""LIBRARY of gen_keccak_fixed.py (which imports the operation list and the list scheduler from here and emits keccak_fixed.inc,
the file keccak.hpp includes).  Its own output, keccak_sched.inc -- the device Keccak-f[1600] round pair as an explicitly ORDERED
instruction stream on compiler-allocated registers -- was the round-3 form; the file left the tree in round 5.

Why (profiles/r03_ubench.txt, scratch/ubench/gen_issue.py + gen_bank.py, MI355X): a gfx950 SIMD retires a wave64
v_bitop3_b32 in 2.2 cycles but a v_alignbit_b32 (every rotate of Keccak) in 4.15, and a stream that alternates
"bitop3, bitop3, alignbit" runs at 3.13 cycles per instruction where the same instructions in long single-kind runs
(what hipcc emits for theta / rho+pi / chi written phase by phase) take 3.46-3.63.  Keccak-f has 120 three-input bit
ops and 58 rotate halves per round -- exactly that 2 : 1 mix -- so the round is emitted as one `asm volatile` per
instruction (the compiler allocates the registers, the statement order is the issue order), list-scheduled here into
the B,B,A pattern: row by row, theta of the five lanes that feed a row of B, their rho rotations, chi of the row, and
the column parities of the NEXT round accumulated as the rows come out.

The operation list is checked in this script against a plain Python Keccak-f (python gen_keccak_sched.py --check).
Stand-alone use today: `python gen_keccak_sched.py --check` (the self-check); the printed stream is no longer included anywhere.
"""
import sys

# (b index, a index, rotation): B[y][2x+3y] = rot(A[x][y]); indices are x + 5y
RHO_PI = [(0, 0, 0), (10, 1, 1), (20, 2, 62), (5, 3, 28), (15, 4, 27), (16, 5, 36), (1, 6, 44), (11, 7, 6), (21, 8, 55),
          (6, 9, 20), (7, 10, 3), (17, 11, 10), (2, 12, 43), (12, 13, 25), (22, 14, 39), (23, 15, 41), (8, 16, 45),
          (18, 17, 15), (3, 18, 21), (13, 19, 8), (14, 20, 18), (24, 21, 2), (9, 22, 61), (19, 23, 56), (4, 24, 14)]
SRC_OF_B = {b: (a, r) for b, a, r in RHO_PI}


class Op:
    __slots__ = ("kind", "dst", "srcs", "imm", "order")

    def __init__(self, kind, dst, srcs, imm, order):
        self.kind, self.dst, self.srcs, self.imm, self.order = kind, dst, srcs, imm, order


def round_ops(rnd, a, c):
    """One round on symbolic 32-bit values.  a[i] = (lo, hi) names of lane i, c[x] = (lo, hi) of the column parities
    of THIS state.  Returns (ops in natural order, new a, new c).  Kinds: 'X3' xor3, 'CHI' a^(~b&c), 'ROT'
    alignbit(s0, s1, sh), 'IOTA' xor with half h of round constant rnd."""
    ops = []
    cnt = [0]

    def new(tag):
        cnt[0] += 1
        return f"r{rnd}_{tag}{cnt[0]}"

    def emit(kind, srcs, imm, tag):
        d = new(tag)
        ops.append(Op(kind, d, tuple(srcs), imm, len(ops)))
        return d

    def rot(v, n, tag):
        lo, hi = v
        if n == 0:
            return v
        if n == 32:
            return (hi, lo)
        if n < 32:
            nh = emit("ROT", (hi, lo), 32 - n, tag)
            nl = emit("ROT", (lo, hi), 32 - n, tag)
        else:
            nh = emit("ROT", (lo, hi), 64 - n, tag)
            nl = emit("ROT", (hi, lo), 64 - n, tag)
        return (nl, nh)

    r = [rot(c[x], 1, "rc") for x in range(5)]
    na = [None] * 25
    part = [None] * 5
    nc = [None] * 5
    for Y in range(5):
        b = [None] * 5
        for X in range(5):
            ai, n = SRC_OF_B[X + 5 * Y]
            x = ai % 5
            t = tuple(emit("X3", (a[ai][h], c[(x + 4) % 5][h], r[(x + 1) % 5][h]), 0x96, "t") for h in range(2))
            b[X] = rot(t, n, "b")
        for X in range(5):
            na[X + 5 * Y] = tuple(emit("CHI", (b[X][h], b[(X + 1) % 5][h], b[(X + 2) % 5][h]), 0xD2, "n") for h in range(2))
        if Y == 0:
            na[0] = tuple(emit("IOTA", (na[0][h],), h, "i") for h in range(2))
        if Y == 2:
            for x in range(5):
                part[x] = tuple(emit("X3", (na[x][h], na[x + 5][h], na[x + 10][h]), 0x96, "p") for h in range(2))
        if Y == 4:
            for x in range(5):
                nc[x] = tuple(emit("X3", (part[x][h], na[x + 15][h], na[x + 20][h]), 0x96, "c") for h in range(2))
    return ops, na, nc


def schedule(ops, live_in):
    """List scheduling into the pattern B, B, A (B = bit op, A = rotate); falls back to the other kind when the wanted
    one has nothing ready.  Priority: natural (row by row) order."""
    # hipcc treats every asm statement that writes a VGPR as a possible dst_sel (SDWA / op_sel) producer and puts an
    # `s_nop 0` in front of a consumer that follows it within two statements (gfx940 "dst_sel forwarding" hazard, which these
    # instructions do not have): a value is therefore never consumed by the very next statement (GAP = 3: two other statements in between)
    GAP = 3
    pos = {v: -GAP for v in live_in}
    remaining = list(ops)
    out = []
    slot = 0
    pattern = "BBA"
    while remaining:
        t = len(out)
        avail = [o for o in remaining if all(s in pos and pos[s] + GAP <= t for s in o.srcs)]
        if not avail:  # nothing can go without the nop: take the oldest op whose sources exist at all
            avail = [o for o in remaining if all(s in pos for s in o.srcs)][:1]
        want_a = pattern[slot % 3] == "A"
        pick = None
        for o in avail:
            if (o.kind == "ROT") == want_a:
                pick = o
                break
        if pick is None:
            pick = avail[0]
        else:
            slot += 1
        out.append(pick)
        pos[pick.dst] = t
        remaining.remove(pick)
    return out


RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b, 0x0000000080000001,
      0x8000000080008081, 0x8000000000008009, 0x000000000000008a, 0x0000000000000088, 0x0000000080008009, 0x000000008000000a,
      0x000000008000808b, 0x800000000000008b, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
      0x000000000000800a, 0x800000008000000a, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]


def build_pair():
    a = [(f"a{i}l", f"a{i}h") for i in range(25)]
    c = [(f"c{x}l", f"c{x}h") for x in range(5)]
    live = [v for p in a + c for v in p]
    ops0, a1, c1 = round_ops(0, a, c)
    ops1, a2, c2 = round_ops(1, a1, c1)
    return live, schedule(ops0 + ops1, live), a2, c2


def simulate(ops, env, rc_pair):
    M = 0xFFFFFFFF
    for o in ops:
        s = [env[x] for x in o.srcs]
        if o.kind == "X3":
            v = s[0] ^ s[1] ^ s[2]
        elif o.kind == "CHI":
            v = s[0] ^ (~s[1] & s[2] & M)
        elif o.kind == "ROT":
            v = (((s[0] << 32) | s[1]) >> o.imm) & M
        else:
            rnd = int(o.dst[1])
            v = s[0] ^ ((rc_pair[rnd] >> (32 * o.imm)) & M)
        env[o.dst] = v & M


def ref_keccak_f(st):
    M = (1 << 64) - 1
    rotl = lambda x, n: ((x << n) | (x >> (64 - n))) & M if n else x
    a = list(st)
    for rnd in range(24):
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        b = [0] * 25
        for bi, ai, n in RHO_PI:
            b[bi] = rotl(a[ai], n)
        a = [b[i] ^ (~b[5 * (i // 5) + (i + 1) % 5] & M & b[5 * (i // 5) + (i + 2) % 5]) for i in range(25)]
        a[0] ^= RC[rnd]
    return a


def check():
    import random
    live, ops, a2, c2 = build_pair()
    rng = random.Random(1)
    for _ in range(3):
        st = [rng.getrandbits(64) for _ in range(25)]
        cur = list(st)
        col = [cur[x] ^ cur[x + 5] ^ cur[x + 10] ^ cur[x + 15] ^ cur[x + 20] for x in range(5)]
        for rp in range(12):
            env = {}
            for i in range(25):
                env[f"a{i}l"], env[f"a{i}h"] = cur[i] & 0xFFFFFFFF, cur[i] >> 32
            for x in range(5):
                env[f"c{x}l"], env[f"c{x}h"] = col[x] & 0xFFFFFFFF, col[x] >> 32
            simulate(ops, env, RC[2 * rp:2 * rp + 2])
            cur = [env[a2[i][0]] | (env[a2[i][1]] << 32) for i in range(25)]
            col = [env[c2[x][0]] | (env[c2[x][1]] << 32) for x in range(5)]
        assert cur == ref_keccak_f(st), "scheduled op list != Keccak-f"
    import hashlib
    # and the plain reference against hashlib (SHA3-256 of the empty string = one permutation of the padded block)
    st = [0] * 25
    st[0] ^= 0x06
    st[16] ^= 1 << 63
    out = ref_keccak_f(st)
    assert b"".join(w.to_bytes(8, "little") for w in out[:4]) == hashlib.sha3_256(b"").digest()
    kinds = [("A" if o.kind == "ROT" else "B") for o in ops]
    print("ok:", len(ops), "instructions per round pair;", "".join(kinds[:90]), "...", file=sys.stderr)


def main():
    if "--check" in sys.argv:
        check()
        return
    live, ops, a2, c2 = build_pair()
    w = sys.stdout.write
    w("// GENERATED by gen_keccak_sched.py -- do not edit.  Two Keccak-f rounds as an ordered instruction stream\n")
    w("// (v_bitop3_b32 / v_alignbit_b32 in the B,B,A issue pattern; see the generator's docstring).\n")
    w("// In: a<i>l/a<i>h (state halves), c<x>l/c<x>h (column parities of the state), rc0l.. rc1h (round constants).\n")
    w("// Out: the same names hold the state and its column parities two rounds later.\n")
    w("#define P2_KS_B(d, a, b, c, imm) asm volatile(\"v_bitop3_b32 %0, %1, %2, %3 bitop3:\" #imm : \"=v\"(d) : \"v\"(a), \"v\"(b), \"v\"(c))\n")
    w("#define P2_KS_A(d, a, b, sh) asm volatile(\"v_alignbit_b32 %0, %1, %2, \" #sh : \"=v\"(d) : \"v\"(a), \"v\"(b))\n")
    w("#define P2_KS_I(d, a, k) asm volatile(\"v_xor_b32 %0, %1, %2\" : \"=v\"(d) : \"s\"(k), \"v\"(a))\n")
    w("#define P2_KECCAK_ROUND_PAIR_SCHED \\\n  {\\\n")
    names = sorted({o.dst for o in ops})
    for i in range(0, len(names), 12):
        w("    uint32_t " + ", ".join(names[i:i + 12]) + ";\\\n")
    for o in ops:
        if o.kind in ("X3", "CHI"):
            w(f"    P2_KS_B({o.dst}, {o.srcs[0]}, {o.srcs[1]}, {o.srcs[2]}, {hex(o.imm)});\\\n")
        elif o.kind == "ROT":
            w(f"    P2_KS_A({o.dst}, {o.srcs[0]}, {o.srcs[1]}, {o.imm});\\\n")
        else:
            w(f"    P2_KS_I({o.dst}, {o.srcs[0]}, rc{o.dst[1]}{'lh'[o.imm]});\\\n")
    for i in range(25):
        w(f"    a{i}l = {a2[i][0]}; a{i}h = {a2[i][1]};\\\n")
    for x in range(5):
        w(f"    c{x}l = {c2[x][0]}; c{x}h = {c2[x][1]};\\\n")
    w("  }\n")


if __name__ == "__main__":
    main()
