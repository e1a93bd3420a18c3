#!/usr/bin/env python3
"""Generator of keccak_fixed.inc -- Keccak-f[1600] as ONE straight-line instruction stream on explicit physical VGPRs.

Second step after gen_keccak_sched.py (same operation list, same B,B,A issue order).  What the compiler-allocated version
cannot control (profiles/r03_ubench.txt): a v_bitop3_b32 whose three sources sit in ONE VGPR bank (index mod 4) retires at
half rate, and hipcc's allocation leaves 23 of the 240 per round pair that way; it also guards every asm-written register
with s_nop for a hazard these instructions do not have.  Here the 24 rounds are unrolled (the lane -> register map changes
every round, so nothing has to be moved back), registers are assigned by a linear scan that picks, for every result, a free
register in a bank that cannot complete an all-same-bank triple in any of its consumers, and the round constants are
literals.  The kernel that includes this keeps the sponge state in the fixed registers for its whole life
(`__attribute__((amdgpu_num_vgpr(KF_BASE)))` keeps the compiler below them).

Registers: v[KF_BASE .. KF_BASE + KF_COUNT).  State lane i (x + 5y) lives in v[KF_BASE + 2i] (low half) and
v[KF_BASE + 2i + 1] (high half) on entry AND on exit of the permutation block.
  python gen_keccak_fixed.py --check          semantic check of the scheduled + allocated stream against a plain Keccak-f
  python gen_keccak_fixed.py BASE > keccak_fixed.inc
"""
import sys

from gen_keccak_sched import RC, Op, ref_keccak_f, round_ops, schedule

NTEMP = 36  # registers beyond the 50 of the state


def build(base):
    a = [(f"a{i}l", f"a{i}h") for i in range(25)]
    live = [v for p in a for v in p]
    # column parities of the input state: computed inside the block (20 xor3)
    ops = []
    c = []
    for x in range(5):
        halves = []
        for h in range(2):
            p = f"cin{x}{'lh'[h]}p"
            q = f"cin{x}{'lh'[h]}"
            ops.append(Op("X3", p, (a[x][h], a[x + 5][h], a[x + 10][h]), 0x96, len(ops)))
            ops.append(Op("X3", q, (p, a[x + 15][h], a[x + 20][h]), 0x96, len(ops)))
            halves.append(q)
        c.append(tuple(halves))
    cur_a, cur_c = a, c
    for r in range(24):
        rops, cur_a, cur_c = round_ops(r, cur_a, cur_c)
        ops.extend(rops)
    # dead code: the column parities of the final state (and whatever only feeds them)
    needed = {v for p in cur_a for v in p}
    keep = []
    for o in reversed(ops):
        if o.dst in needed:
            keep.append(o)
            needed.update(o.srcs)
    ops = list(reversed(keep))
    # B,B,A list scheduling over the whole permutation (priority: program order), then registers
    return live, g_schedule(ops, live), cur_a


def g_schedule(ops, live_in, gap=1):
    pos = {v: -gap for v in live_in}
    remaining = list(ops)
    out = []
    slot = 0
    pattern = "BBA"
    while remaining:
        t = len(out)
        avail = [o for o in remaining if all(s in pos and pos[s] + gap <= t for s in o.srcs)]
        if not avail:
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


def allocate(live, ops, final_a, base):
    """Linear scan.  reg[v] = physical index.  Returns (list of (op, dst reg, src regs)), final fix-up moves)."""
    # zero-constant iota halves are aliases: rename their uses to the source value up front
    alias = {}
    for o in ops:
        if o.kind == "IOTA" and ((RC[int(o.dst[1:].split("_")[0])] >> (32 * o.imm)) & 0xFFFFFFFF) == 0:
            alias[o.dst] = alias.get(o.srcs[0], o.srcs[0])
    if alias:
        ops = [Op(o.kind, o.dst, tuple(alias.get(s_, s_) for s_ in o.srcs), o.imm, o.order) for o in ops if o.dst not in alias]
        final_a = [tuple(alias.get(v, v) for v in p) for p in final_a]
    last_use = {}
    for t, o in enumerate(ops):
        for s in o.srcs:
            last_use[s] = t
    finals = [v for p in final_a for v in p]
    for v in finals:
        last_use[v] = len(ops) + 1
    consumers = {}
    for t, o in enumerate(ops):
        for s in o.srcs:
            consumers.setdefault(s, []).append(o)
    reg = {v: base + i for i, v in enumerate(live)}  # a<i>l -> base + 2i, a<i>h -> base + 2i + 1
    free = list(range(base + 50, base + 50 + NTEMP))
    out = []
    conflicts = 0
    for t, o in enumerate(ops):
        srcs = [reg[s] for s in o.srcs]
        if o.kind in ("X3", "CHI") and len({r % 4 for r in srcs}) == 1:
            conflicts += 1
        # sources whose last use this is give their registers back BEFORE the destination is chosen (in-place is fine)
        for s in set(o.srcs):
            if last_use[s] == t:
                free.append(reg[s])
        # banks to avoid: for every three-source consumer of this value, the bank shared by its other two sources
        avoid_hard, avoid_soft = set(), set()
        for cns in consumers.get(o.dst, []):
            if cns.kind not in ("X3", "CHI"):
                continue
            others = [s for s in cns.srcs if s != o.dst]
            banks = [reg[s] % 4 for s in others if s in reg]
            if len(others) == 2 and len(banks) == 2 and banks[0] == banks[1]:
                avoid_hard.add(banks[0])
            else:
                avoid_soft.update(banks)
        best = None
        for r_ in free:
            b = r_ % 4
            score = (b in avoid_hard) * 100 + (b in avoid_soft) * 1
            if best is None or score < best[0]:
                best = (score, r_)
        if best is None:
            raise SystemExit(f"out of registers at op {t} (NTEMP = {NTEMP})")
        free.remove(best[1])
        reg[o.dst] = best[1]
        out.append((o, best[1], srcs))
    # final placement: value of lane i must sit in its canonical pair
    moves = []
    want = {base + i: reg[v] for i, v in enumerate(finals)}  # dst <- src
    pending = {d: s for d, s in want.items() if d != s}
    tmp = None
    spare = [r_ for r_ in range(base + 50, base + 50 + NTEMP) if r_ not in set(want.values())]
    while pending:
        progressed = False
        for d, s in list(pending.items()):
            if d not in pending.values():  # nobody still needs d's old content
                moves.append((d, s))
                del pending[d]
                progressed = True
        if not progressed:  # a cycle: park one source in a spare register
            d, s = next(iter(pending.items()))
            tmp = spare[0]
            moves.append((tmp, d))
            for k in pending:
                if pending[k] == d:
                    pending[k] = tmp
    return out, moves, conflicts


def simulate(alloc, moves, st, base):
    M = 0xFFFFFFFF
    regs = {}
    for i in range(25):
        regs[base + 2 * i] = st[i] & M
        regs[base + 2 * i + 1] = st[i] >> 32
    for o, d, srcs in alloc:
        s = [regs[r] for r in srcs]
        if o.kind == "X3":
            v = s[0] ^ s[1] ^ s[2]
        elif o.kind == "CHI":
            v = s[0] ^ (~s[1] & s[2] & M)
        elif o.kind == "ROT":
            v = (((s[0] << 32) | s[1]) >> o.imm) & M
        else:
            rnd = int(o.dst[1:].split("_")[0])
            v = s[0] ^ ((RC[rnd] >> (32 * o.imm)) & M)
        regs[d] = v & M
    for d, s in moves:
        regs[d] = regs[s]
    return [regs[base + 2 * i] | (regs[base + 2 * i + 1] << 32) for i in range(25)]


def main():
    base = 36
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    if args:
        base = int(args[0])
    live, ops, final_a = build(base)
    alloc, moves, conflicts = allocate(live, ops, final_a, base)
    if "--check" in sys.argv:
        import random
        rng = random.Random(2)
        for _ in range(3):
            st = [rng.getrandbits(64) for _ in range(25)]
            assert simulate(alloc, moves, st, base) == ref_keccak_f(st)
        kinds = "".join("A" if o.kind == "ROT" else "B" for o, _, _ in alloc)
        print(f"ok: {len(alloc)} instructions + {len(moves)} moves, {conflicts} three-source instructions with all sources in one bank;",
              kinds[:60], "...", file=sys.stderr)
        return
    w = sys.stdout.write
    top = base + 50 + NTEMP
    w("// GENERATED by gen_keccak_fixed.py -- do not edit.  Keccak-f[1600], 24 rounds unrolled, explicit VGPRs (see the generator).\n")
    w(f"#define P2_KF_BASE {base}\n#define P2_KF_TOP {top}  /* first register above the block's */\n")
    # every instruction of the block is 8 bytes (VOP3, or VOP2 + literal; iota halves that are inline constants are forced to
    # the 64-bit encoding) and the block starts 8-byte aligned: a stream of 8-byte instructions that sits at 4 mod 8 fetches
    # measurably slower (MI355X_MICROARCH.md, "code-placement sensitivity"; P2_KF_PHASE_TEST shifts it on purpose)
    # every instruction of the block is 8 bytes (VOP3, or VOP2 + literal; iota halves that are inline constants are forced to
    # the 64-bit encoding), so the block's start decides where ALL of them sit relative to the 8-byte fetch granule -- and a
    # MIXED stream of full-rate and half-rate 8-byte instructions is sensitive to that (profiles/r03_ubench.txt, sweep 4):
    # starting at 4 mod 8 it runs at the blended rate (Keccak-f 11.3 Gperm/s with 2-4 waves per SIMD), starting 8-byte aligned
    # everything runs at the half rate (8.9) -- except for a LONE wave, which is faster aligned (8.0 vs 6.8).  PH = number of
    # s_nop behind a 64-byte alignment: 1 for throughput kernels, 0 for the latency-bound tree tails.
    w("#define P2_KECCAK_FIXED_PERMUTE() P2_KECCAK_FIXED_PERMUTE_PH(1)\n")
    w("#define P2_KECCAK_FIXED_PERMUTE_PH(PH) asm volatile(\".p2align 6\\n .rept \" #PH \"\\n s_nop 0\\n .endr\\n\" \\\n")
    for o, d, srcs in alloc:
        if o.kind in ("X3", "CHI"):
            w(f'  "v_bitop3_b32 v{d}, v{srcs[0]}, v{srcs[1]}, v{srcs[2]} bitop3:{hex(o.imm)}\\n" \\\n')
        elif o.kind == "ROT":
            w(f'  "v_alignbit_b32 v{d}, v{srcs[0]}, v{srcs[1]}, {o.imm}\\n" \\\n')
        else:
            rnd = int(o.dst[1:].split("_")[0])
            k = (RC[rnd] >> (32 * o.imm)) & 0xFFFFFFFF
            if k <= 64:  # an inline constant: VOP2 would be 4 bytes
                w(f'  "v_xor_b32_e64 v{d}, {k}, v{srcs[0]}\\n" \\\n')
            else:
                w(f'  "v_xor_b32 v{d}, {hex(k)}, v{srcs[0]}\\n" \\\n')
    for d, s in moves:
        w(f'  "v_mov_b32 v{d}, v{s}\\n" \\\n')
    w(f'  ::: "v{top - 1}")\n')


if __name__ == "__main__":
    main()
