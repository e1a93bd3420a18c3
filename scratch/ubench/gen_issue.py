#!/usr/bin/env python3
"""Generator of issue.hip -- the VALU issue-rate micro-benchmark behind DESIGN.md section 3b (round 3 rewrite).

What round 2's version got wrong (Weak 2): rows of 0.1-0.5 ms (launch ramp inside the number), the clock
taken from s_memtime.  This one:
  * every row is ONE dispatch of >= 20 ms (hundreds of thousands of loop trips of a 64-instruction block);
  * the block is inline asm on EXPLICIT physical registers, so the number of independent dependency chains
    (1, 2, 4, 8), the operand form (2 / 3 register sources) and the VGPR banks of the sources (bank = index mod 4:
    "friendly" = three different banks, "hostile" = all sources in one bank) are what the row says they are;
  * K = 1, 2, 4, 8 waves per SIMD (256 CUs x K blocks of 256 threads, every SIMD gets exactly K waves);
  * wall time from HIP events (the un-profiled arm), and -- from a separate rocprofv3 --pmc pass of the same binary
    (run_issue.sh, joined by join_issue.py) -- GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_INSTS_VALU per
    dispatch: clock = GRBM_GUI_ACTIVE / wall (guide, "DVFS give-back"), cycles per wave-instruction per SIMD =
    GRBM_GUI_ACTIVE / (SQ_INSTS_VALU / 1024 SIMDs).
Usage: python gen_issue.py > issue.hip; hipcc --offload-arch=gfx950 -O2 issue.hip -o issue
"""
import sys

# name, asm template, number of VALU instructions per template, registers per chain value (1 = 32 bit, 2 = 64 bit),
# number of register sources (for the bank variants).  {D} chain register (dst and first source), {E} second chain
# register (carry pairs), {A} {B} loop-invariant sources (pairs for the 64-bit forms)
MODES = [
    ("v_fma_f32", "v_fma_f32 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_mul_f32", "v_mul_f32 {D}, {D}, {A}", 1, 1, 2),
    ("v_pk_fma_f32", "v_pk_fma_f32 {D2}, {D2}, {A2}, {B2}", 1, 2, 3),
    ("v_add_u32", "v_add_u32 {D}, {D}, {A}", 1, 1, 2),
    ("v_xor_b32", "v_xor_b32 {D}, {D}, {A}", 1, 1, 2),
    ("v_bitop3_b32", "v_bitop3_b32 {D}, {D}, {A}, {B} bitop3:0x96", 1, 1, 3),
    ("v_alignbit_b32", "v_alignbit_b32 {D}, {D}, {A}, 7", 1, 1, 2),
    ("v_alignbit_b32(v)", "v_alignbit_b32 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_add3_u32", "v_add3_u32 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_lshl_add_u64", "v_lshl_add_u64 {D2}, {D2}, 0, {A2}", 1, 2, 2),
    ("v_lshlrev_b64", "v_lshlrev_b64 {D2}, 13, {D2}", 1, 2, 1),
    ("v_add_co+v_addc_co", "v_add_co_u32 {D}, vcc, {D}, {A}\n v_addc_co_u32 {E}, vcc, {E}, {A}, vcc", 2, 2, 2),
    ("v_sub_co+v_subb_co", "v_sub_co_u32 {D}, vcc, {D}, {A}\n v_subb_co_u32 {E}, vcc, {E}, {A}, vcc", 2, 2, 2),
    ("v_cmp_lt_u64+v_cndmask", "v_cmp_lt_u64 vcc, {A2}, {D2}\n v_cndmask_b32 {D}, {D}, {A}, vcc", 2, 2, 2),
    ("v_mad_u64_u32", "v_mad_u64_u32 {D2}, s[20:21], {A}, {B}, {D2}", 1, 2, 3),
    ("v_mul_lo_u32", "v_mul_lo_u32 {D}, {D}, {A}", 1, 1, 2),
    ("v_mul_hi_u32", "v_mul_hi_u32 {D}, {D}, {A}", 1, 1, 2),
    ("v_mul_u32_u24", "v_mul_u32_u24 {D}, {D}, {A}", 1, 1, 2),
]
# round 3, second sweep (SURVEY=1): which opcodes run at the full (2-cycle) rate, and which pairs of source banks cost the
# half rate on a three-source instruction.  {A}/{B} banks are chosen per row through the bank pattern.
SURVEY_MODES = [
    ("v_sub_u32", "v_sub_u32 {D}, {D}, {A}", 1, 1, 2),
    ("v_and_b32", "v_and_b32 {D}, {D}, {A}", 1, 1, 2),
    ("v_or_b32", "v_or_b32 {D}, {D}, {A}", 1, 1, 2),
    ("v_lshlrev_b32", "v_lshlrev_b32 {D}, 3, {D}", 1, 1, 1),
    ("v_lshrrev_b32", "v_lshrrev_b32 {D}, 3, {D}", 1, 1, 1),
    ("v_lshlrev_b32(v)", "v_lshlrev_b32 {D}, {A}, {D}", 1, 1, 2),
    ("v_mov_b32", "v_mov_b32 {D}, {A}", 1, 1, 1),
    ("v_cndmask_b32(s)", "v_cndmask_b32 {D}, {D}, {A}, s[22:23]", 1, 1, 2),
    ("v_min_u32", "v_min_u32 {D}, {D}, {A}", 1, 1, 2),
    ("v_lshl_or_b32", "v_lshl_or_b32 {D}, {D}, 3, {A}", 1, 1, 2),
    ("v_and_or_b32", "v_and_or_b32 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_or3_b32", "v_or3_b32 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_lshl_add_u32", "v_lshl_add_u32 {D}, {D}, 3, {A}", 1, 1, 2),
    ("v_add_lshl_u32", "v_add_lshl_u32 {D}, {D}, {A}, 3", 1, 1, 2),
    ("v_xad_u32", "v_xad_u32 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_bfe_u32", "v_bfe_u32 {D}, {D}, 3, 20", 1, 1, 1),
    ("v_bfi_b32", "v_bfi_b32 {D}, {A}, {D}, {B}", 1, 1, 3),
    ("v_perm_b32", "v_perm_b32 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_alignbyte_b32", "v_alignbyte_b32 {D}, {D}, {A}, 1", 1, 1, 2),
    ("v_mad_u32_u24", "v_mad_u32_u24 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_mad_u32_u16", "v_mad_u32_u16 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_mul_lo_u16", "v_mul_lo_u16 {D}, {D}, {A}", 1, 1, 2),
    ("v_pk_add_u16", "v_pk_add_u16 {D}, {D}, {A}", 1, 1, 2),
    ("v_pk_mul_lo_u16", "v_pk_mul_lo_u16 {D}, {D}, {A}", 1, 1, 2),
    ("v_pk_mad_u16", "v_pk_mad_u16 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_dot4_u32_u8", "v_dot4_u32_u8 {D}, {A}, {B}, {D}", 1, 1, 3),
    ("v_dot2_u32_u16", "v_dot2_u32_u16 {D}, {A}, {B}, {D}", 1, 1, 3),
    ("v_sad_u32", "v_sad_u32 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 {D}, {D}", 1, 1, 1),
    ("v_cvt_u32_f32", "v_cvt_u32_f32 {D}, {D}", 1, 1, 1),
    ("v_fma_f64", "v_fma_f64 {D2}, {D2}, {A2}, {B2}", 1, 2, 3),
    ("v_mul_f64", "v_mul_f64 {D2}, {D2}, {A2}", 1, 2, 2),
    ("v_add_f64", "v_add_f64 {D2}, {D2}, {A2}", 1, 2, 2),
    ("v_add_co_u32(sgpr)", "v_add_co_u32 {D}, s[22:23], {D}, {A}", 1, 1, 2),
    ("v_cmp_lt_u32(vcc)", "v_cmp_lt_u32 vcc, {D}, {A}", 1, 1, 2),
    ("v_cmp_lt_u32(sgpr)", "v_cmp_lt_u32 s[22:23], {D}, {A}", 1, 1, 2),
    ("v_add_u32_dpp", "v_add_u32_dpp {D}, {D}, {A} row_shr:1 row_mask:0xf bank_mask:0xf", 1, 1, 2),
    ("v_mov_b32_dpp", "v_mov_b32_dpp {D}, {A} row_shr:1 row_mask:0xf bank_mask:0xf", 1, 1, 1),
    ("v_xor_b32(sgpr)", "v_xor_b32 {D}, s24, {D}", 1, 1, 1),
    ("v_xor_b32(lit)", "v_xor_b32 {D}, 0x12345678, {D}", 1, 1, 1),
    ("v_bitop3_b32(s)", "v_bitop3_b32 {D}, {D}, {A}, s24 bitop3:0x96", 1, 1, 2),
    ("v_pk_mov_b32", "v_pk_mov_b32 {D2}, {A2}, {B2} op_sel:[0,1]", 1, 2, 2),
    ("v_mov_b64", "v_mov_b64 {D2}, {A2}", 1, 2, 1),
]
# (name suffix, bank of A, bank of B) with D in bank 0: which pairs collide
BANK_PATTERNS = [("D0.A1.B2", 1, 2), ("D0.A0.B1", 0, 1), ("D0.A1.B0", 1, 0), ("D0.A1.B1", 1, 1), ("D0.A0.B0", 0, 0)]
BANK_MODES = [
    ("v_bitop3_b32", "v_bitop3_b32 {D}, {D}, {A}, {B} bitop3:0x96", 1, 1, 3),
    ("v_fma_f32", "v_fma_f32 {D}, {D}, {A}, {B}", 1, 1, 3),
    ("v_bitop3(dst!=src)", "v_bitop3_b32 {D}, {C}, {A}, {B} bitop3:0x96", 1, 1, 3),
    ("v_alignbit(dst!=src)", "v_alignbit_b32 {D}, {C}, {A}, 7", 1, 1, 3),
]
BLOCK = 64  # templates per asm block


def regs(c, bank):
    """chain c (0..7): D = v[32+4c] (bank 0), E = v[33+4c].  friendly: A in bank 2, B in bank 3 (pairs 2-3 / 6-7 style);
    hostile: A and B in bank 0 like D."""
    d = 32 + 4 * c
    if isinstance(bank, tuple):  # (bank of A, bank of B); C = a third register in bank 0 that is not the destination
        a, b = 64 + bank[0], 68 + bank[1]
        return dict(D=f"v{d}", C="v72", A=f"v{a}", B=f"v{b}")
    if bank == "hostile":
        a, b = 64, 68
    else:
        a, b = 66, 71  # banks 2 and 3
    return dict(D=f"v{d}", E=f"v{d+1}", D2=f"v[{d}:{d+1}]", A=f"v{a}", B=f"v{b}", A2=f"v[{a}:{a+1}]", B2=f"v[{b & ~1}:{(b & ~1)+1}]")


def body(tpl, chains, bank):
    lines = [tpl.format(**regs(i % chains, bank)) for i in range(BLOCK)]
    return "\\n\"\n      \"".join(l.replace("\n", "\\n") for l in lines)


def main():
    out = []
    out.append("// GENERATED by gen_issue.py -- do not edit; see the generator's docstring.")
    out.append("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <cstdlib>\n#include <cstring>")
    clob = ", ".join(f'"v{i}"' for i in range(32, 74)) + ', "vcc", "s20", "s21", "s22", "s23", "s24"'
    rows = []
    kid = 0
    survey = len(sys.argv) > 1 and sys.argv[1] == "survey"
    table = []
    if survey:
        for m in SURVEY_MODES:
            table.append((m, [(8, "friendly")]))
        for m in BANK_MODES:
            table.append((m, [(8, (pn, ba, bb)) for pn, ba, bb in BANK_PATTERNS]))
    else:
        for m in MODES:
            v = [(c, "friendly") for c in (1, 2, 4, 8)]
            if m[4] >= 2:
                v.append((8, "hostile"))
            table.append((m, v))
    for (name, tpl, ninstr, width, nsrc), variants in table:
        for chains, bank in variants:
            bank_name = bank
            if isinstance(bank, tuple):
                bank_name, bank = bank[0], (bank[1], bank[2])
            fn = f"k{kid}"
            out.append(f"__global__ __launch_bounds__(256) void {fn}(uint32_t *out, int iters) {{")
            # initialise every register the block reads (values are irrelevant for timing; keep floats finite)
            out.append("  asm volatile(\"" + "\\n\"\n      \"".join(f"v_mov_b32 v{i}, 0x3f800000" for i in range(32, 74)) + "\\n s_mov_b64 s[22:23], 0x55\\n s_mov_b32 s24, 77" + f"\" ::: {clob});")
            out.append("  for (int it = 0; it < iters; it++)")
            out.append(f"    asm volatile(\"{body(tpl, chains, bank)}\" ::: {clob});")
            out.append("  uint32_t r;\n  asm volatile(\"v_mov_b32 %0, v32\" : \"=v\"(r) :: " + clob + ");")
            out.append("  if (r == 0x12345u) out[threadIdx.x] = r;\n}")
            rows.append((fn, name, chains, bank_name, ninstr))
            kid += 1
    out.append("struct Row { void (*fn)(uint32_t *, int); const char *name; int chains; const char *bank; int ninstr; };")
    out.append("static const Row ROWS[] = {")
    for fn, name, chains, bank, ninstr in rows:
        out.append(f'  {{{fn}, "{name}", {chains}, "{bank}", {ninstr}}},')
    out.append("};")
    out.append(r"""
int main(int argc, char **argv) {
  // issue [iters] [only-mode-substring]
  const int iters = argc > 1 ? atoi(argv[1]) : 100000;
  const char *only = argc > 2 ? argv[2] : nullptr;
  uint32_t *out;
  hipMalloc(&out, 4096);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("# device %s, %d CUs, clockRate %d kHz; 64 templates per loop trip, %d trips per row\n", p.gcnArchName, p.multiProcessorCount,
         p.clockRate, iters);
  printf("# if a SIMD retires one wave64 VALU per 2 cycles: %.1f T lane-instr/s at 2.4 GHz; per 4 cycles: %.1f T\n",
         256 * 4 * 32 * 2.4e9 / 1e12, 256 * 4 * 16 * 2.4e9 / 1e12);
  printf("%-24s %6s %-9s %2s %10s %10s %14s\n", "instruction", "chains", "banks", "K", "ms", "T lane-i/s", "wave-instr");
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (const Row &r : ROWS) {
    if (only && !strstr(r.name, only)) continue;
    for (int K : {1, 2, 4, 8}) {
      if (r.chains != 8 && r.chains != 1 && K != 4 && K != 8) continue;  // chain sweep at the occupancies that matter
      const int blocks = 256 * K;
      hipLaunchKernelGGL(r.fn, dim3(blocks), dim3(256), 0, 0, out, 64);  // warm: code object, clocks
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipLaunchKernelGGL(r.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      const double wave_instr = (double)iters * 64 * r.ninstr * blocks * 4;
      printf("%-24s %6d %-9s %2d %10.3f %10.2f %14.0f\n", r.name, r.chains, r.bank, K, ms, wave_instr * 64 / (ms * 1e-3) / 1e12, wave_instr);
      fflush(stdout);
    }
  }
  return 0;
}
""")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
