// plonk.hip -- permutation argument (Z / partial products) and the
// vanishing-polynomial / quotient evaluation over the LDE domain.
//
// Replaces, inside `circuit_data.prove` (plonky2-backend/src/actions/prove_action.rs:96):
//   plonky2 0.2.2 plonk/prover.rs  wires_permutation_partial_products_and_zs
//                                  compute_quotient_polys                    (P9)
//                 plonk/vanishing_poly.rs eval_vanishing_poly_base_batch,
//                                  evaluate_gate_constraints_base_batch
//                 gates/selectors.rs compute_filter
// and restates the constraint formulas of the reference's in-tree custom gates:
//   plonky2-backend/src/plonky2_ecdsa/biguint/gates/arithmetic_u32.rs:289-348
//   .../add_many_u32.rs:151-192, subtraction_u32.rs:234-270,
//   .../range_check_u32.rs:95-117, comparison.rs:337-414
// plus the stock gates the translator emits.
//
// Layout: one lane per LDE row (coset r, index k); every column access
// lde[r][col][k] is unit-stride across the wave.  Each gate's constraints are
// folded straight into the two alpha-combinations (sum_t alpha_c^t term_t) with
// a precomputed alpha-power table read through scalar loads, so no per-row
// constraint vector is ever materialised.  Z(g x) is row k+1 of the same coset
// (i + 8 = 8(k+1) + r): no halo.  Integer-ALU bound (modmuls), not MFMA.
#include "internal.hpp"
#include "gates.hpp"

namespace p2 {

// Poseidon round constants for PoseidonGate rows (scalar loads: the index is wave-uniform)
__constant__ gl_t c_poseidon_rc[360];
int poseidon_upload_constants() {
  gl_t rc[360];
  poseidon_round_constants_host(rc);
  gl_t dev[360];  // the gate evaluator takes the form in which a partial round has a constant for word 0 only (gates.hpp)
  poseidon_device_constants(rc, dev);
  return hipMemcpyToSymbol(HIP_SYMBOL(c_poseidon_rc), dev, sizeof dev) == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------
// x = w_n^i from the half table tw[j << shift] = w_n^j, j < n/2
__device__ __forceinline__ gl_t root_pow(const gl_t *tw, uint32_t shift, uint32_t d, uint32_t i) {
  uint32_t half = 1u << (d - 1);
  if (d == 0) return 1;
  return i < half ? tw[(size_t)i << shift] : gl_neg(tw[(size_t)(i - half) << shift]);
}

// ---- Z / partial products -------------------------------------------------------
// position of (column, row i) in the chunk-quotient scratch (ZsArgs::cp): [i >> sb][column][i & (2^sb - 1)]
__device__ __forceinline__ size_t zs_idx(const ZsArgs &a, uint32_t col, uint32_t i) {
  const uint32_t cols = a.K * (a.nchunks + 1);
  return ((((size_t)(i >> a.sb) * cols + col) << a.sb) | (i & ((1u << a.sb) - 1u)));
}
__global__ __launch_bounds__(256) void zs_chunk_kernel(ZsArgs a) {
  const uint32_t n = 1u << a.d;
  const uint32_t i = a.row0 + blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = blockIdx.y;
  if (i >= a.row0 + a.rows) return;
  const gl_t x = root_pow(a.sub_tw, a.tw_shift, a.d, i);
  const gl_t beta = a.betas[c], gamma = a.gammas[c];
  const gl_t bx = gl_mul(beta, x);
  gl_t np[16], dp[16];
  for (uint32_t m = 0; m < a.nchunks; m++) {
    // the factors and the running products are only ever multiplied: congruent u64s (gl.hpp _nc), canonical once per chunk
    uint64_t pn = 1, pd = 1;
    for (uint32_t j = m * a.QF; j < (m + 1) * a.QF && j < a.R; j++) {
      gl_t wv = a.wires[(size_t)j * n + i];
      const gl_t wg = gl_add(wv, gamma);
      const uint64_t num = gl_mul_add_nc(bx, a.k_is[j], wg);
      const uint64_t den = gl_mul_add_nc(beta, a.sigmas[(size_t)j * n + i], wg);
      pn = gl_mul_nc(pn, num);
      pd = gl_mul_nc(pd, den);
    }
    np[m] = gl_canon(pn);
    dp[m] = gl_canon(pd);
  }
  // Montgomery batch inversion of the chunk denominators
  gl_t pre[16];
  gl_t acc = 1;
  for (uint32_t m = 0; m < a.nchunks; m++) {
    pre[m] = acc;
    acc = gl_mul(acc, dp[m]);
  }
  gl_t inv = gl_inv(acc);
  gl_t rowp = 1;
  for (uint32_t m = a.nchunks; m-- > 0;) {
    gl_t di = gl_mul(inv, pre[m]);
    inv = gl_mul(inv, dp[m]);
    gl_t q = gl_mul(np[m], di);
    a.cp[zs_idx(a, c * a.nchunks + m, i)] = q;
    rowp = gl_mul(rowp, q);
  }
  a.cp[zs_idx(a, a.K * a.nchunks + c, i)] = rowp;
}

// exclusive multiplicative scan across a block (blockDim <= 1024)
__device__ __forceinline__ gl_t block_scan_mul(gl_t v, gl_t *lds, gl_t *total) {
  const uint32_t t = threadIdx.x, nt = blockDim.x;
  lds[t] = v;
  __syncthreads();
  for (uint32_t off = 1; off < nt; off <<= 1) {
    gl_t x = lds[t];
    gl_t y = t >= off ? lds[t - off] : 1;
    __syncthreads();
    lds[t] = gl_mul(x, y);
    __syncthreads();
  }
  gl_t incl = lds[t];
  gl_t excl = t ? lds[t - 1] : 1;
  *total = lds[nt - 1];
  (void)incl;
  return excl;
}

__global__ __launch_bounds__(256) void scan_local_kernel(const ZsArgs a, uint32_t n, gl_t *local, gl_t *bsum, uint32_t nblocks) {
  __shared__ gl_t lds[256];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = blockIdx.y;
  gl_t v = i < n ? a.cp[zs_idx(a, a.K * a.nchunks + c, i)] : 1;
  gl_t total;
  gl_t excl = block_scan_mul(v, lds, &total);
  if (i < n) local[(size_t)c * n + i] = excl;
  if (threadIdx.x == 0) bsum[(size_t)c * nblocks + blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void scan_blocks_kernel(gl_t *bsum, uint32_t nblocks) {
  __shared__ gl_t lds[1024];
  const uint32_t c = blockIdx.x;
  gl_t *b = bsum + (size_t)c * nblocks;
  const uint32_t per = (nblocks + blockDim.x - 1) / blockDim.x;
  const uint32_t lo = threadIdx.x * per;
  gl_t prod = 1;
  for (uint32_t j = lo; j < lo + per && j < nblocks; j++) prod = gl_mul(prod, b[j]);
  gl_t total;
  gl_t excl = block_scan_mul(prod, lds, &total);
  gl_t acc = excl;
  for (uint32_t j = lo; j < lo + per && j < nblocks; j++) {
    gl_t v = b[j];
    b[j] = acc;  // exclusive prefix of block j
    acc = gl_mul(acc, v);
  }
}
__global__ __launch_bounds__(256) void zs_finish_kernel(ZsArgs a, const gl_t *local, const gl_t *bpre,
                                                        uint32_t nblocks) {
  const uint32_t n = 1u << a.d;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = blockIdx.y;
  if (i >= n) return;
  const uint32_t PP = a.nchunks - 1;
  gl_t z = gl_mul(bpre[(size_t)c * nblocks + blockIdx.x], local[(size_t)c * n + i]);
  a.zp[(size_t)c * n + i] = z;
  gl_t acc = z;
  for (uint32_t m = 0; m < PP; m++) {
    acc = gl_mul(acc, a.cp[zs_idx(a, c * a.nchunks + m, i)]);
    a.zp[((size_t)a.K + (size_t)c * PP + m) * n + i] = acc;
  }
}

// chunk quotients and row products of the rows [a.row0, a.row0 + a.rows)
void zs_chunks(hipStream_t st, const ZsArgs &a) {
  if (!a.rows) return;
  ProfScope ps("zs_chunk_kernel", 8.0 * a.K * (double)a.rows * (2.0 * a.R + a.nchunks + 1));
  hipLaunchKernelGGL(zs_chunk_kernel, dim3((a.rows + 255) / 256, a.K), dim3(256), 0, st, a);
}
// multiplicative scan of the row products + Z and the partial products of every row; scan_tmp: at least K * (n + ceil(n/256)) elements
void zs_scan_finish(hipStream_t st, const ZsArgs &a, gl_t *scan_tmp) {
  const uint32_t n = 1u << a.d;
  const uint32_t threads = 256;
  const uint32_t nblocks = (n + threads - 1) / threads;
  gl_t *local = scan_tmp;
  gl_t *bsum = scan_tmp + (size_t)a.K * n;
  ProfScope ps2("zs_scan_finish", 8.0 * a.K * (double)n * (2.0 * a.nchunks + 4));
  hipLaunchKernelGGL(scan_local_kernel, dim3(nblocks, a.K), dim3(threads), 0, st, a, n, local, bsum, nblocks);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(a.K), dim3(1024), 0, st, bsum, nblocks);
  hipLaunchKernelGGL(zs_finish_kernel, dim3(nblocks, a.K), dim3(threads), 0, st, a, local, bsum, nblocks);
}
void zs_partial_products(hipStream_t st, const ZsArgs &a, gl_t *scan_tmp) {
  zs_chunks(st, a);
  zs_scan_finish(st, a, scan_tmp);
}

// ---- gate constraints -------------------------------------------------------------
// sum_t alpha^t c_t for both challenges, accumulated unreduced (gl.hpp Acc160) and reduced once per
// gate: 13 instead of 29 VALU per term.
struct Consumer {
  Acc160 acc0, acc1;
  const gl_t *ap0, *ap1;
  uint32_t t;
  __device__ __forceinline__ void emit(gl_t c) {
    acc0.mac(ap0[t], c);
    acc1.mac(ap1[t], c);
    t++;
  }
};

// grid: x = k blocks, y = coset.  G = 1: one lane per LDE row does everything (block 256 x 1).
// G = 4 (heavy gate mixes): block 64 x 4 -- the four waves of a block share one 64-row tile and
// each evaluates its share of the gates (GateDesc.pad = group; group 0 also does the permutation
// argument), so the tile's wires are pulled from HBM once per block instead of once per gate
// (reuse distance ~120 KB per wave defeats L2 otherwise); partial sums meet in LDS.
#ifndef P2_QUOT_WAVES
#define P2_QUOT_WAVES 6  // waves per SIMD the register allocator must leave room for (80 VGPRs; 2-4 spilled): measured
                         // 3.85 -> 3.27 ms (all gate kinds) and 0.83 -> 0.80 ms (sha mix) at 2^20 rows against no bound
#endif
template <bool POSEIDON, int G>
__global__ __launch_bounds__(256, P2_QUOT_WAVES) void quotient_kernel(const QuotArgs a) {
  __shared__ gl_t red[2][G > 1 ? G : 1][G > 1 ? 64 : 1];
  const uint32_t grp = G > 1 ? threadIdx.y : 0;
  const uint32_t n = 1u << a.d;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t z = blockIdx.y;                             // local coset
  const uint32_t r = a.coset_first + z * a.coset_stride;     // global coset
  if (k >= n) return;
  const uint32_t ncs = a.NC + a.R, nzp = a.K * (1 + a.PP);
  const gl_t *cs = a.cs_lde + (size_t)r * ncs * n + k;
  const gl_t *wl = a.wires_lde + (size_t)z * a.W * n + k;
  const gl_t *zl = a.zp_lde + (size_t)z * nzp * n;  // indexed with explicit k (next row)
  const uint32_t kn = (k + 1) & (n - 1);
  const gl_t x = gl_mul(a.qconst[r], root_pow(a.tw, a.tw_shift, a.d, k));
  Consumer out;
  out.acc0.clear();
  out.acc1.clear();
  out.ap0 = a.apow;
  out.ap1 = a.apow + a.nterms;
  out.t = 0;
  if (grp == 0) {
  // L_0(x) (Z_c(x) - 1),  L_0(x) = Z_H(x) / (n (x - 1))
  const gl_t l0 = a.l0[(size_t)r * n + k];
  for (uint32_t c = 0; c < a.K; c++) out.emit(gl_mul(l0, gl_sub(zl[(size_t)c * n + k], 1)));
  // partial-product checks: prev * prod(num) - next * prod(den), chunk by chunk, both challenges
  {
    const gl_t bx0 = gl_mul(a.betas[0], x), bx1 = gl_mul(a.betas[1], x);
    const uint32_t t_base = out.t;
    for (uint32_t m = 0; m < a.nchunks; m++) {
      // running products: congruent u64s, only ever multiplied (gl.hpp _nc).  They START at the chunk's first factor (every chunk
      // has one: nchunks = ceil(R / QF)) instead of at 1: four products less per chunk, 40 of ~900 per row
      uint64_t n0, d0, n1 = 1, d1 = 1;
      {
        const uint32_t j = m * a.QF;
        const gl_t wv = wl[(size_t)j * n];
        const gl_t sg = cs[(size_t)(a.NC + j) * n];
        const gl_t kj = a.k_is[j];
        const gl_t wg0 = gl_add(wv, a.gammas[0]);
        n0 = gl_mul_add_nc(bx0, kj, wg0);
        d0 = gl_mul_add_nc(a.betas[0], sg, wg0);
        if (a.K > 1) {
          const gl_t wg1 = gl_add(wv, a.gammas[1]);
          n1 = gl_mul_add_nc(bx1, kj, wg1);
          d1 = gl_mul_add_nc(a.betas[1], sg, wg1);
        }
      }
#pragma unroll 7
      for (uint32_t j = m * a.QF + 1; j < (m + 1) * a.QF && j < a.R; j++) {
        const gl_t wv = wl[(size_t)j * n];
        const gl_t sg = cs[(size_t)(a.NC + j) * n];
        const gl_t kj = a.k_is[j];
        const gl_t wg0 = gl_add(wv, a.gammas[0]);
        n0 = gl_mul_nc(n0, gl_mul_add_nc(bx0, kj, wg0));
        d0 = gl_mul_nc(d0, gl_mul_add_nc(a.betas[0], sg, wg0));
        if (a.K > 1) {
          const gl_t wg1 = gl_add(wv, a.gammas[1]);
          n1 = gl_mul_nc(n1, gl_mul_add_nc(bx1, kj, wg1));
          d1 = gl_mul_nc(d1, gl_mul_add_nc(a.betas[1], sg, wg1));
        }
      }
      for (uint32_t c = 0; c < a.K; c++) {
        const gl_t prev = m == 0 ? zl[(size_t)c * n + k] : zl[((size_t)a.K + c * a.PP + m - 1) * n + k];
        const gl_t next = m == a.nchunks - 1 ? zl[(size_t)c * n + kn] : zl[((size_t)a.K + c * a.PP + m) * n + k];
        const gl_t term = c == 0 ? gl_sub(gl_mul(prev, n0), gl_mul(next, d0)) : gl_sub(gl_mul(prev, n1), gl_mul(next, d1));
        out.t = t_base + c * a.nchunks + m;
        out.emit(term);
      }
    }
    out.t = t_base + a.K * a.nchunks;
  }
  }
  // gate constraints: every gate on every row, masked by its selector filter
  const uint32_t t_gates = a.K + a.K * a.nchunks;
  gl_t tot0 = out.acc0.value(), tot1 = out.acc1.value();
  auto W = [&](uint32_t c) { return wl[(size_t)c * n]; };
  auto LC = [&](uint32_t i) { return cs[(size_t)(a.num_selectors + i) * n]; };
  for (uint32_t gi = 0; gi < a.num_gates; gi++) {
    const GateDesc g = a.gates[gi];
    if (g.num_constraints == 0) continue;
    if (!POSEIDON && g.kind == G_POSEIDON) continue;  // evaluated by poseidon_gate_kernel
    if (G > 1 && gate_group(g, a.use_half) != grp) continue;
    const gl_t s = cs[(size_t)g.sel_index * n];
    const gl_t f = gate_filter<BaseOps>(g, gi, a.num_selectors, s);
    if (const uint32_t slot1 = a.use_half ? gate_half_slot(g) : 0u) {
      // a gate of degree <= 4: its alpha-folded sum was evaluated on the even cosets and extended to the odd ones (gate_sums)
      const gl_t *h = a.hsum + ((size_t)((r & 1u) * 4u + (r >> 1)) * a.nsk + (size_t)(slot1 - 1) * a.K) * n + k;
      tot0 = gl_mul_add(f, h[0], tot0);
      if (a.K > 1) tot1 = gl_mul_add(f, h[n], tot1);
      continue;
    }
    out.acc0.clear();
    out.acc1.clear();
    out.t = t_gates;
    eval_gate<BaseOps, POSEIDON>(g, W, LC, a.pi_hash, c_poseidon_rc, out);
    tot0 = gl_add(tot0, gl_mul(f, out.acc0.value()));
    tot1 = gl_add(tot1, gl_mul(f, out.acc1.value()));
  }
  if constexpr (G > 1) {
    red[0][grp][threadIdx.x] = tot0;
    red[1][grp][threadIdx.x] = tot1;
    __syncthreads();
    if (grp != 0) return;
    for (int q = 1; q < G; q++) {
      tot0 = gl_add(tot0, red[0][q][threadIdx.x]);
      tot1 = gl_add(tot1, red[1][q][threadIdx.x]);
    }
  }
  const gl_t zi = a.qconst[16 + r];
  a.out[((size_t)0 * a.ncosets + z) * n + k] = gl_mul(tot0, zi);
  if (a.K > 1) a.out[((size_t)1 * a.ncosets + z) * n + k] = gl_mul(tot1, zi);
}

// ---- gates of degree <= 4 on HALF the LDE domain -----------------------------------------------------------------------------
// The alpha-folded constraint sum of a gate, S_g(x) = sum_t alpha^t c_t(x), is a polynomial of degree < deg(g) * n before the
// selector filter multiplies it.  The reference's five custom gates, BaseSum<4> and everything else built on 2-bit limb range
// checks have degree 4 (arithmetic_u32.rs:44 `1 << Self::limb_bits()`, comparison.rs `1 << self.chunk_bits()`): S_g is fixed by
// its values on ANY 4n points -- the four even cosets r = 0, 2, 4, 6 are the coset g <w_4n> -- and the values on the odd cosets
// follow by interpolation: four size-n inverse transforms, a 4 x 4 cross-coset matrix, four size-n transforms, per
// (gate, challenge) instead of ~200 constraints x 77 VALU on every row of four more cosets.  Exact field arithmetic: the same
// words the direct evaluation gives (tests: proofs bit-exact, knob "half_gates" 0 / 1).  synth(19, ecdsa): 855 of 940
// constraints per row are such gates.
// out: hsum [parity][m][slot * K + c][n] (coset r = 2 m + parity); this kernel writes parity 0.
#ifndef P2_SUMS_WAVES
#define P2_SUMS_WAVES 8  // 64 VGPRs: measured 1.02 ms against 1.065 at 6 and 1.14 at 5 waves per SIMD (synth(17, ecdsa), staged sums)
#endif
// STAGE: the sums wait in LDS until every gate is done and go to memory at the end.  Not for bandwidth: a global store inside
// the gate loop makes every later uniform load (the alpha powers: two per constraint) a VECTOR load -- hipcc may only use the
// scalar unit for memory no store of the kernel can have touched -- and each of those waits out a full L2 round trip in front of
// its multiply-accumulate (round 6, first version: 73 % of the wave cycles in s_waitcnt, 1.46 ms; profiles/r06_gate_sums.md).
template <int G, bool STAGE>
__global__ __launch_bounds__(256, P2_SUMS_WAVES) void gate_sums_kernel(const QuotArgs a) {
  extern __shared__ gl_t sums_stage[];  // [nsk][rows of the block]
  const uint32_t grp = G > 1 ? threadIdx.y : 0;
  const uint32_t n = 1u << a.d;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t m = blockIdx.y, r = 2u * m;
  if (k >= n) return;
  const uint32_t ncs = a.NC + a.R;
  // (column base uniform, the lane's row a 32-bit byte offset: hipcc folds the lane's part into ONE 64-bit base and forms each
  // column's address with a scalar shift + one v_lshl_add_u64 -- it does not take the SGPR-base form of global_load here)
  const gl_t *csb = a.cs_lde + (size_t)r * ncs * n;
  const gl_t *wlb = a.wires_lde + (size_t)r * a.W * n;   // (only with every coset local: z = r)
  Consumer out;
  out.ap0 = a.apow;
  out.ap1 = a.apow + a.nterms;
  const uint32_t kb = k * 8u;  // (n <= 2^24 rows: a 32-bit byte offset)
  auto W = [&](uint32_t c) { return *(const gl_t *)((const char *)(wlb + (size_t)c * n) + kb); };
  auto LC = [&](uint32_t i) { return *(const gl_t *)((const char *)(csb + (size_t)(a.num_selectors + i) * n) + kb); };
  for (uint32_t gi = 0; gi < a.num_gates; gi++) {
    const GateDesc g = a.gates[gi];
    const uint32_t slot1 = gate_half_slot(g);
    if (!slot1) continue;
    if (G > 1 && gate_sums_group(g) != grp) continue;
    out.acc0.clear();
    out.acc1.clear();
    out.t = a.K + a.K * a.nchunks;
    eval_gate<BaseOps, false>(g, W, LC, a.pi_hash, c_poseidon_rc, out);
    if constexpr (STAGE) {
      const uint32_t rb = blockDim.x, sk = (slot1 - 1) * a.K;
      sums_stage[sk * rb + threadIdx.x] = out.acc0.value();
      if (a.K > 1) sums_stage[(sk + 1) * rb + threadIdx.x] = out.acc1.value();
    } else {
      gl_t *h = a.hsum + ((size_t)m * a.nsk + (size_t)(slot1 - 1) * a.K) * n + k;
      h[0] = out.acc0.value();
      if (a.K > 1) h[n] = out.acc1.value();
    }
  }
  if constexpr (STAGE) {
    // every lane reads back what it wrote itself (same wave, LDS operations in order): no barrier
    for (uint32_t gi = 0; gi < a.num_gates; gi++) {
      const GateDesc g = a.gates[gi];
      const uint32_t slot1 = gate_half_slot(g);
      if (!slot1 || (G > 1 && gate_sums_group(g) != grp)) continue;
      const uint32_t rb = blockDim.x, sk = (slot1 - 1) * a.K;
      gl_t *h = a.hsum + ((size_t)m * a.nsk + sk) * n + k;
      h[0] = sums_stage[sk * rb + threadIdx.x];
      if (a.K > 1) h[n] = sums_stage[(sk + 1) * rb + threadIdx.x];
    }
  }
}
// even cosets -> odd cosets, between the per-coset inverse and forward transforms: in [m][col][n] = coefficients (bit-reversed
// positions) of P_m(y) = S(s_2m y); with inv_scale[r][p] = s_r^-e (e = bitrev p) P'_m = sum_j w_8^(2 m j) A_j, A_j = (g^n)^j S_j[e]
// (S = sum_j x^(j n) S_j), and the odd coset 2 m' + 1 wants sum_j w_8^((2 m' + 1) j) A_j: out[m'] = sum_m F[m'][m] P'_m,
// F[m'][m] = 1/4 sum_j w_8^((2 m' + 1 - 2 m) j) (host: CircuitState::half_cross, handle.hip).  The forward transform applies s_(2m'+1)^e itself.
struct CrossMat {
  gl_t f[4][4];
};
__global__ __launch_bounds__(256) void gate_sums_cross_kernel(const gl_t *__restrict__ in, const gl_t *__restrict__ inv_scale, gl_t *__restrict__ out,
                                                             uint32_t d, uint32_t cols, CrossMat F) {
  const uint32_t n = 1u << d;
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
  if (p >= n) return;
  gl_t P[4];
#pragma unroll
  for (uint32_t m = 0; m < 4; m++) P[m] = gl_mul(in[((size_t)m * cols + col) * n + p], inv_scale[(size_t)(2 * m) * n + p]);
#pragma unroll
  for (uint32_t mo = 0; mo < 4; mo++) {
    Acc160 acc;
    acc.clear();
#pragma unroll
    for (uint32_t m = 0; m < 4; m++) acc.mac(F.f[mo][m], P[m]);
    out[((size_t)mo * cols + col) * n + p] = acc.value();
  }
}
void gate_sums_eval(hipStream_t st, const QuotArgs &a, uint32_t groups) {
  const uint32_t n = 1u << a.d;
  const size_t lds = (size_t)a.nsk * (groups == 4 ? 64 : 256) * sizeof(gl_t);
  static const bool stage_ok = [] {
    const char *e = getenv("P2GPU_SUMS_STAGE");  // 0: store from inside the gate loop (A/B measurements)
    return !(e && *e == '0');
  }();
  const bool stage = stage_ok && lds <= 24 * 1024;
  // (same spelling as rocprofv3's demangled names)
  const char *name = groups == 4 ? (stage ? "gate_sums_kernel<4, true>" : "gate_sums_kernel<4, false>") : (stage ? "gate_sums_kernel<1, true>" : "gate_sums_kernel<1, false>");
  ProfScope ps(name, 8.0 * (double)n * 4 * (a.W + a.num_selectors + 1.0 * a.nsk));
  if (groups == 4) {
    if (stage) hipLaunchKernelGGL((gate_sums_kernel<4, true>), dim3(n / 64, 4), dim3(64, 4), lds, st, a);
    else hipLaunchKernelGGL((gate_sums_kernel<4, false>), dim3(n / 64, 4), dim3(64, 4), 0, st, a);
  } else {
    if (stage) hipLaunchKernelGGL((gate_sums_kernel<1, true>), dim3((n + 255) / 256, 4), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((gate_sums_kernel<1, false>), dim3((n + 255) / 256, 4), dim3(256), 0, st, a);
  }
}
void gate_sums_cross(hipStream_t st, const gl_t *in, const gl_t *inv_scale, gl_t *out, uint32_t d, uint32_t cols, const gl_t F[16]) {
  const uint32_t n = 1u << d;
  CrossMat M;
  for (int i = 0; i < 16; i++) M.f[i / 4][i % 4] = F[i];
  ProfScope ps("gate_sums_cross_kernel", 8.0 * (double)n * cols * 8);
  hipLaunchKernelGGL(gate_sums_cross_kernel, dim3((n + 255) / 256, cols), dim3(256), 0, st, in, inv_scale, out, d, cols, M);
}

// The PoseidonGate term on its own grid (circuits with public inputs): 118 S-boxes and 30 MDS layers
// per LDE row are as much work as the rest of the row together, and inside the generic kernel they
// set its register budget (155 VGPRs -> 3 waves per SIMD).  Adds filter(x) * sum_t alpha^t c_t / Z_H(x)
// to what quotient_kernel wrote.
#ifndef P2_PGATE_WAVES
#define P2_PGATE_WAVES 6  // 80 VGPRs: 0.646 ms against 0.685 at 4 waves per SIMD (107 VGPRs), synth(17, sha) + 4 public inputs
#endif
__global__ __launch_bounds__(256, P2_PGATE_WAVES) void poseidon_gate_kernel(const QuotArgs a, uint32_t gi) {
  const uint32_t n = 1u << a.d;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t z = blockIdx.y;
  const uint32_t r = a.coset_first + z * a.coset_stride;
  if (k >= n) return;
  const GateDesc g = a.gates[gi];
  const uint32_t ncs = a.NC + a.R;
  const gl_t *cs = a.cs_lde + (size_t)r * ncs * n + k;
  const gl_t *wl = a.wires_lde + (size_t)z * a.W * n + k;
  Consumer out;
  out.acc0.clear();
  out.acc1.clear();
  out.ap0 = a.apow;
  out.ap1 = a.apow + a.nterms;
  out.t = a.K + a.K * a.nchunks;
  auto W = [&](uint32_t c) { return wl[(size_t)c * n]; };
  eval_poseidon_gate<BaseOps>(W, c_poseidon_rc, out);
  const gl_t f = gate_filter<BaseOps>(g, gi, a.num_selectors, cs[(size_t)g.sel_index * n]);
  const gl_t fz = gl_mul(f, a.qconst[16 + r]);
  gl_t *o0 = a.out + ((size_t)0 * a.ncosets + z) * n + k;
  *o0 = gl_mul_add(fz, out.acc0.value(), *o0);
  if (a.K > 1) {
    gl_t *o1 = a.out + ((size_t)1 * a.ncosets + z) * n + k;
    *o1 = gl_mul_add(fz, out.acc1.value(), *o1);
  }
}

__global__ __launch_bounds__(256) void l0_table_kernel(const gl_t *qconst, const gl_t *tw, uint32_t tw_shift, uint32_t d, gl_t n_inv,
                                                       gl_t *out) {
  const uint32_t n = 1u << d;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (k >= n) return;
  const gl_t x = gl_mul(qconst[r], root_pow(tw, tw_shift, d, k));
  out[(size_t)r * n + k] = gl_mul(gl_mul(qconst[8 + r], n_inv), gl_inv(gl_sub(x, 1)));
}
void fill_l0_table(hipStream_t st, const gl_t *qconst, const gl_t *tw, uint32_t tw_shift, uint32_t d, uint32_t cosets, gl_t n_inv, gl_t *out) {
  const uint32_t n = 1u << d, threads = n >= 256 ? 256 : 64;
  hipLaunchKernelGGL(l0_table_kernel, dim3((n + threads - 1) / threads, cosets), dim3(threads), 0, st, qconst, tw, tw_shift, d, n_inv, out);
}

void quotient_eval(hipStream_t st, const QuotArgs &a) {
  const uint32_t n = 1u << a.d;
  const uint32_t threads = n >= 256 ? 256 : 64;
  const bool split = (a.use_half ? a.gate_groups_half : a.gate_groups) == 4 && n >= 64;
  // same spelling as rocprofv3's demangled names
  {
    const char *name = split ? "quotient_kernel<false, 4>" : "quotient_kernel<false, 1>";
    ProfScope ps(name, 8.0 * (double)n * a.ncosets * (a.NC + a.R + a.W + a.K * (1 + a.PP) + 2.0 * a.K));
    if (split) {
      hipLaunchKernelGGL((quotient_kernel<false, 4>), dim3(n / 64, a.ncosets), dim3(64, 4), 0, st, a);
    } else {
      hipLaunchKernelGGL((quotient_kernel<false, 1>), dim3((n + threads - 1) / threads, a.ncosets), dim3(threads), 0, st, a);
    }
  }
  if (a.has_poseidon) {
    uint32_t gi = 0;
    while (gi < a.num_gates && a.host_gates[gi].kind != G_POSEIDON) gi++;
    ProfScope ps("poseidon_gate_kernel", 8.0 * (double)n * a.ncosets * (135 + 1 + 4.0 * a.K));
    hipLaunchKernelGGL(poseidon_gate_kernel, dim3((n + threads - 1) / threads, a.ncosets), dim3(threads), 0, st, a, gi);
  }
}

// in [K][C][n]: coefficients (bit-reversed storage) of the per-coset interpolants
// in the variable y = x / s_r; out [K*C][n]: chunk polynomials Q_m, m < C.
// `in` is [world][K][C/world][n]: the slice of rank q holds its cosets r = q + z * world (world = 1: [K][C][n])
__global__ __launch_bounds__(256) void quotient_chunks_kernel(const gl_t *in, const gl_t *inv_scale, gl_t *out,
                                                             uint32_t d, uint32_t rate_bits, gl_t w_inv, gl_t gn_inv,
                                                             gl_t rate_inv, uint32_t world, uint32_t K) {
  const uint32_t n = 1u << d, C = 1u << rate_bits;
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = blockIdx.y;
  if (p >= n) return;
  gl_t P[8];
  const uint32_t own = C / world;
  for (uint32_t r = 0; r < C; r++) {
    const uint32_t q = r % world, z = r / world;
    P[r] = gl_mul(in[(((size_t)q * K + c) * own + z) * n + p], inv_scale[(size_t)r * n + p]);
  }
  gl_t wm = 1;       // w^-m
  gl_t scale = rate_inv;  // (7^-n)^m / C
  for (uint32_t m = 0; m < C; m++) {
    gl_t acc = 0, wr = 1;  // w^(-r m)
    for (uint32_t r = 0; r < C; r++) {
      acc = gl_add(acc, gl_mul(P[r], wr));
      wr = gl_mul(wr, wm);
    }
    out[((size_t)c * C + m) * n + p] = gl_mul(acc, scale);
    wm = gl_mul(wm, w_inv);
    scale = gl_mul(scale, gn_inv);
  }
}
void quotient_chunks(hipStream_t st, const gl_t *in, const gl_t *inv_scale, gl_t *out, uint32_t d, uint32_t K,
                     uint32_t rate_bits, gl_t w_inv, gl_t gn_inv, gl_t rate_inv, uint32_t world) {
  const uint32_t n = 1u << d;
  const uint32_t threads = n >= 256 ? 256 : 64;
  ProfScope ps("quotient_chunks_kernel", 8.0 * (double)n * (1u << rate_bits) * (2.0 * K + 1));
  hipLaunchKernelGGL(quotient_chunks_kernel, dim3((n + threads - 1) / threads, K), dim3(threads), 0, st, in, inv_scale,
                     out, d, rate_bits, w_inv, gn_inv, rate_inv, world, K);
}

}  // namespace p2
