// witness.hip -- row-local witness filling on the GPU (SURVEY.md 8(f) "N1").
//
// After `generate_partial_witness` has propagated the copy constraints (CPU, Rust), what is
// left of witness generation is row-local: every gate's own SimpleGenerator derives the rest
// of its row -- limb decompositions, carries, inverses, S-box inputs -- from the row's input
// wires.  For the reference's circuits that is every NON-ROUTED column (154 of 234): the
// host only has to provide the 80 routed columns (84 MB instead of 245 MB over PCIe at 2^17
// rows) and this kernel fills the rest in HBM, one lane per row, column-major (unit-stride
// across the wave for every column).
//
// Restated generators (plonky2-backend/src/plonky2_ecdsa/biguint/gates/):
//   arithmetic_u32.rs:376-426, add_many_u32.rs:329-378, subtraction_u32.rs:298-343,
//   range_check_u32.rs:198-220, comparison.rs:439-537
// and the stock plonky2 ones (ArithmeticBaseGenerator, BaseSplitGenerator,
// RandomAccessGenerator, ConstantGate wires, PoseidonGenerator).
#include "internal.hpp"
#include "poseidon.hpp"

namespace p2 {

struct FillArgs {
  gl_t *wires;             // [W][n], in place
  const uint8_t *row_gate; // [n] index into gates
  const GateDesc *gates;
  const gl_t *gconsts;     // [ngc][n] gate-constant columns (after the selector columns)
  const gl_t *prc;         // 360 Poseidon round constants (wave-uniform index -> scalar loads)
  uint32_t d, ngc;
};

__global__ __launch_bounds__(256) void fill_witness_kernel(FillArgs a) {
  const size_t n = (size_t)1 << a.d;
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const GateDesc g = a.gates[a.row_gate[row]];
  gl_t *w = a.wires + row;
#define Wv(col) w[(size_t)(col) * n]
  auto LC = [&](uint32_t i) { return i < a.ngc ? a.gconsts[(size_t)i * n + row] : (gl_t)0; };
  switch (g.kind) {
  case G_CONSTANT:
    for (uint32_t i = 0; i < g.p[0]; i++) Wv(i) = LC(i);
    break;
  case G_ARITHMETIC: {
    const gl_t c0 = LC(0), c1 = LC(1);
    for (uint32_t i = 0; i < g.p[0]; i++)
      Wv(4 * i + 3) = gl_add(gl_mul(gl_mul(Wv(4 * i), Wv(4 * i + 1)), c0), gl_mul(Wv(4 * i + 2), c1));
    break;
  }
  case G_BASE_SUM: {
    uint64_t v = Wv(0);
    const uint32_t B = g.p[0];
    for (uint32_t i = 0; i < g.p[1]; i++) {
      Wv(1 + i) = v % B;
      v /= B;
    }
    break;
  }
  case G_RANDOM_ACCESS: {
    const uint32_t bits = g.p[0], copies = g.p[1], extra = g.p[2], vec = 1u << bits;
    const uint32_t routed = (2 + vec) * copies + extra;
    for (uint32_t cp = 0; cp < copies; cp++) {
      const uint32_t base = (2 + vec) * cp;
      const uint64_t idx = Wv(base);
      Wv(base + 1) = Wv(base + 2 + (uint32_t)(idx & (vec - 1)));
      for (uint32_t k = 0; k < bits; k++) Wv(routed + cp * bits + k) = (idx >> k) & 1;
    }
    for (uint32_t i = 0; i < extra; i++) Wv((2 + vec) * copies + i) = LC(i);
    break;
  }
  case G_POSEIDON: {
    gl_t st[12];
    const gl_t swap = Wv(24);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const gl_t l = Wv(i), r = Wv(i + 4);
      const gl_t dl = gl_mul(swap, gl_sub(r, l));
      Wv(25 + i) = dl;
      st[i] = gl_add(l, dl);
      st[i + 4] = gl_sub(r, dl);
    }
#pragma unroll
    for (int i = 8; i < 12; i++) st[i] = Wv(i);
#pragma unroll 1
    for (int r = 0; r < 30; r++) {
#pragma unroll
      for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], a.prc[12 * r + i]);
      const bool full = r < 4 || r >= 26;
      if (full) {
        if (r != 0) {
          const uint32_t base = r < 4 ? 29 + 12 * (r - 1) : 87 + 12 * (r - 26);
#pragma unroll
          for (int i = 0; i < 12; i++) Wv(base + i) = st[i];
        }
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = poseidon_sbox(st[i]);
      } else {
        Wv(65 + (r - 4)) = st[0];
        st[0] = poseidon_sbox(st[0]);
      }
      poseidon_mds(st);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) Wv(12 + i) = st[i];
    break;
  }
  case G_U32_ARITHMETIC: {
    const uint32_t ops = g.p[0];
    for (uint32_t i = 0; i < ops; i++) {
      uint64_t o = gl_add(gl_mul(Wv(6 * i), Wv(6 * i + 1)), Wv(6 * i + 2));
      const uint64_t hi = o >> 32, lo = o & 0xFFFFFFFFULL;
      Wv(6 * i + 3) = lo;
      Wv(6 * i + 4) = hi;
      const uint64_t diff = 0xFFFFFFFFULL - hi;
      Wv(6 * i + 5) = diff ? gl_inv(diff) : 0;
      for (uint32_t j = 0; j < 32; j++) {
        Wv(6 * ops + 32 * i + j) = o & 3;
        o >>= 2;
      }
    }
    break;
  }
  case G_U32_ADD_MANY: {
    const uint32_t na = g.p[0], ops = g.p[1];
    for (uint32_t i = 0; i < ops; i++) {
      const uint32_t b = (na + 3) * i;
      gl_t sum = 0;
      for (uint32_t j = 0; j <= na; j++) sum = gl_add(sum, Wv(b + j));
      const uint64_t res = sum & 0xFFFFFFFFULL, carry = sum >> 32;
      Wv(b + na + 1) = res;
      Wv(b + na + 2) = carry;
      for (uint32_t j = 0; j < 16; j++) Wv((na + 3) * ops + 18 * i + j) = (res >> (2 * j)) & 3;
      for (uint32_t j = 0; j < 2; j++) Wv((na + 3) * ops + 18 * i + 16 + j) = (carry >> (2 * j)) & 3;
    }
    break;
  }
  case G_U32_SUBTRACTION: {
    const uint32_t ops = g.p[0];
    for (uint32_t i = 0; i < ops; i++) {
      const gl_t init = gl_sub(gl_sub(Wv(5 * i), Wv(5 * i + 1)), Wv(5 * i + 2));
      const gl_t bout = init > (1ULL << 32) ? 1 : 0;
      const gl_t res = gl_add(init, gl_mul(bout, 1ULL << 32));
      Wv(5 * i + 3) = res;
      Wv(5 * i + 4) = bout;
      for (uint32_t j = 0; j < 16; j++) Wv(5 * ops + 16 * i + j) = (res >> (2 * j)) & 3;
    }
    break;
  }
  case G_U32_RANGE_CHECK: {
    const uint32_t nl = g.p[0];
    for (uint32_t i = 0; i < nl; i++) {
      const uint32_t v = (uint32_t)Wv(i);
      for (uint32_t j = 0; j < 16; j++) Wv(nl + 16 * i + j) = (v >> (2 * j)) & 3;
    }
    break;
  }
  case G_COMPARISON: {
    const uint32_t nb = g.p[0], nc = g.p[1], cb = (nb + nc - 1) / nc;
    const uint64_t a0 = Wv(0), b0 = Wv(1), cs = 1ULL << cb;
    uint64_t ta = a0, tb = b0;
    Wv(2) = a0 <= b0 ? 1 : 0;
    gl_t msd = 0;
    for (uint32_t i = 0; i < nc; i++) {
      const gl_t f = ta % cs, s = tb % cs;
      ta /= cs;
      tb /= cs;
      Wv(4 + i) = f;
      Wv(4 + nc + i) = s;
      Wv(4 + 2 * nc + i) = (f == s) ? 1 : gl_inv(gl_sub(s, f));
      Wv(4 + 3 * nc + i) = (f == s) ? 1 : 0;
      if (f != s) {
        msd = gl_sub(s, f);
        Wv(4 + 4 * nc + i) = 0;
      } else {
        Wv(4 + 4 * nc + i) = msd;
      }
    }
    Wv(3) = msd;
    uint64_t v = gl_add(cs, msd);
    for (uint32_t i = 0; i < cb + 1; i++) {
      Wv(4 + 5 * nc + i) = v & 1;
      v >>= 1;
    }
    break;
  }
  default:
    break;
  }
#undef Wv
}

void fill_witness(hipStream_t st, gl_t *wires, const uint8_t *row_gate, const GateDesc *gates, const gl_t *gconsts,
                  const gl_t *prc, uint32_t d, uint32_t ngc, uint32_t num_wires) {
  const size_t n = (size_t)1 << d;
  FillArgs a;
  a.wires = wires; a.row_gate = row_gate; a.gates = gates; a.gconsts = gconsts; a.prc = prc; a.d = d; a.ngc = ngc;
  ProfScope ps("fill_witness_kernel", 8.0 * (double)num_wires * (double)n);
  hipLaunchKernelGGL(fill_witness_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}

}  // namespace p2
