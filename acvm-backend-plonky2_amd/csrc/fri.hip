// fri.hip -- opening evaluation, FRI batch reduction, folding, proof-of-work
// grinding and query gathers.
//
// Replaces, inside `circuit_data.prove` (plonky2-backend/src/actions/prove_action.rs:96):
//   plonky2 0.2.2 plonk/proof.rs OpeningSet::new / PolynomialCoeffs::eval   
//                 fri/oracle.rs  PolynomialBatch::prove_openings             (P11)
//                 fri/prover.rs  fri_committed_trees, fri_proof_of_work,
//                                fri_prover_query_rounds                     (P12)
//
// Coefficients live in bit-reversed positions (ntt.hip), so every coefficient
// pass here is position-agnostic: openings are dot products with the table
// zeta^(bitrev p); the batch combination sum_j alpha^j f_j is a column sweep;
// (F(X) - F(z)) / (X - z) is taken pointwise on the size-n subgroup (z is never
// in it) instead of by sequential synthetic division; the arity-16 fold reads
// 16 unit-stride runs.  All HBM-bound streaming kernels.
#include "internal.hpp"

namespace p2 {

// sq[b] = base^(2^b) of the two bases (host: d squarings each): a lane multiplies the ones its exponent's bits select --
// half the products of squaring in every lane, and both tables of a proof (zeta, g*zeta) in ONE launch (grid.y)
struct ExtSquares {
  ext_t sq[2][32];
};
__global__ __launch_bounds__(256) void ext_powers_bitrev_kernel(ExtSquares S, uint32_t d, gl_t *out0, gl_t *out1) {
  const uint32_t n = 1u << d;
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t which = blockIdx.y;
  const uint32_t e = bitrev32(p, d);
  ext_t acc = ext_from(1);
  for (uint32_t b = 0; b < d; b++)
    if ((e >> b) & 1) acc = ext_mul(acc, S.sq[which][b]);
  gl_t *out = which ? out1 : out0;
  out[p] = acc.c0;
  out[(size_t)n + p] = acc.c1;
}
void ext_powers_bitrev2(hipStream_t st, ext_t base0, ext_t base1, uint32_t d, gl_t *out0, gl_t *out1) {
  ExtSquares S;
  ext_t a = base0, b = base1;
  for (uint32_t i = 0; i < 32; i++) {
    S.sq[0][i] = a;
    S.sq[1][i] = b;
    if (i + 1 < d) {
      a = ext_mul(a, a);
      b = ext_mul(b, b);
    }
  }
  uint32_t n = 1u << d;
  hipLaunchKernelGGL(ext_powers_bitrev_kernel, dim3((n + 255) / 256, 2), dim3(256), 0, st, S, d, out0, out1);
}

// one (part, column) block of an opening: the dot product of 1/parts of the column's coefficients with the powers of the
// point (both coordinates), left as a partial sum for the host to add up
__device__ __forceinline__ void eval_column_block(const gl_t *__restrict__ coeffs, uint32_t d, const gl_t *__restrict__ pw,
                                                  uint32_t parts, gl_t *__restrict__ partial, const uint32_t *__restrict__ colnz,
                                                  const gl_t *__restrict__ colval, const gl_t *__restrict__ basis_partial,
                                                  uint32_t part, uint32_t col) {
  __shared__ gl_t s0[256], s1[256];
  const uint32_t n = 1u << d;
  if (colnz != nullptr && colnz[col] < 2) {  // (class 3 columns have their coefficients in memory: the plain dot product)
    // class 0: the zero polynomial opens to zero; class 1: v times the unit column's polynomial, whose partial
    // sums an earlier launch left in basis_partial [parts][2]
    if (threadIdx.x == 0) {
      gl_t r0 = 0, r1 = 0;
      if (colnz[col] == 1) {
        const gl_t v = colval[col];
        r0 = gl_mul(v, basis_partial[2 * part]);
        r1 = gl_mul(v, basis_partial[2 * part + 1]);
      }
      partial[((size_t)col * parts + part) * 2] = r0;
      partial[((size_t)col * parts + part) * 2 + 1] = r1;
    }
    return;
  }
  const uint32_t per = n / parts;
  const gl_t *c = coeffs + (size_t)col * n + (size_t)part * per;
  const gl_t *p0 = pw + (size_t)part * per, *p1 = pw + n + (size_t)part * per;
  Acc160 l0, l1;  // unreduced dot products (gl.hpp)
  l0.clear();
  l1.clear();
  // four elements per trip, their twelve loads issued before the first product (one element per trip left a single load in flight
  // per lane: 75 us for 200 MB)
  uint32_t i = threadIdx.x;
#pragma unroll 1
  for (; i + 3 * blockDim.x < per; i += 4 * blockDim.x) {
    gl_t v[4], q0[4], q1[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      v[u] = c[i + u * blockDim.x];
      q0[u] = p0[i + u * blockDim.x];
      q1[u] = p1[i + u * blockDim.x];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      l0.mac(v[u], q0[u]);
      l1.mac(v[u], q1[u]);
    }
  }
  for (; i < per; i += blockDim.x) {
    gl_t v = c[i];
    l0.mac(v, p0[i]);
    l1.mac(v, p1[i]);
  }
  s0[threadIdx.x] = l0.value();
  s1[threadIdx.x] = l1.value();
  __syncthreads();
  for (uint32_t off = blockDim.x >> 1; off; off >>= 1) {
    if (threadIdx.x < off) {
      s0[threadIdx.x] = gl_add(s0[threadIdx.x], s0[threadIdx.x + off]);
      s1[threadIdx.x] = gl_add(s1[threadIdx.x], s1[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[((size_t)col * parts + part) * 2] = s0[0];
    partial[((size_t)col * parts + part) * 2 + 1] = s1[0];
  }
}
// grid (parts, cols)
__global__ __launch_bounds__(256) void eval_columns_kernel(const gl_t *__restrict__ coeffs, uint32_t d,
                                                           const gl_t *__restrict__ pw, uint32_t parts,
                                                           gl_t *__restrict__ partial, const uint32_t *__restrict__ colnz,
                                                           const gl_t *__restrict__ colval,
                                                           const gl_t *__restrict__ basis_partial) {
  eval_column_block(coeffs, d, pw, parts, partial, colnz, colval, basis_partial, blockIdx.x, blockIdx.y);
}
// every batch of a proof's openings in ONE launch (grid.y runs over the columns of all segments): six launches of which four
// were too small to fill the chip were 0.14 ms on the critical path of a proof
__global__ __launch_bounds__(256) void eval_columns_multi_kernel(EvalSegs S, uint32_t d, uint32_t parts, uint32_t col0) {
  uint32_t col = col0 + blockIdx.y, k = 0;
  while (k + 1 < S.count && col >= S.seg[k].cols) {
    col -= S.seg[k].cols;
    k++;
  }
  const bool h = k == S.hinted;
  eval_column_block(S.seg[k].coeffs, d, S.seg[k].pw, parts, S.seg[k].partial, h ? S.cls : nullptr, h ? S.val : nullptr,
                    h ? S.basis_partial : nullptr, blockIdx.x, col);
}
void eval_columns(hipStream_t st, const gl_t *coeffs, uint32_t cols, uint32_t d, const gl_t *pw, uint32_t parts,
                  gl_t *partial, const ColHints *hints, const gl_t *basis_partial) {
  if (!cols) return;
  ProfScope ps("eval_columns_kernel", 8.0 * cols * (double)((size_t)1 << d));
  hipLaunchKernelGGL(eval_columns_kernel, dim3(parts, cols), dim3(256), 0, st, coeffs, d, pw, parts, partial,
                     hints ? hints->cls : nullptr, hints ? hints->val : nullptr, basis_partial);
}
void eval_columns_multi(hipStream_t st, const EvalSegs &S, uint32_t d, uint32_t parts, uint32_t col0, uint32_t ncols) {
  uint32_t cols = 0;
  for (uint32_t k = 0; k < S.count; k++) cols += S.seg[k].cols;
  if (col0 >= cols) return;
  cols = std::min(cols - col0, ncols);
  if (!cols) return;
  ProfScope ps("eval_columns_multi_kernel", 8.0 * cols * (double)((size_t)1 << d));
  hipLaunchKernelGGL(eval_columns_multi_kernel, dim3(parts, cols), dim3(256), 0, st, S, d, parts, col0);
}

// block = 64 positions x 4 column groups: group g sums the columns j = g mod 4 (8 loads in flight
// per lane), the four partial sums meet in LDS.  (One lane per position alone would be 2 waves per
// SIMD at n = 2^17, each walking 354 columns with a single load in flight.)
// LIST = false: every column through direct indices.  LIST = true: only the dense columns of nzlist (nzlist[0] =
// their number, nzlist[1..] their indices, ascending; scalar loads) -- the zero polynomials add no term, and the
// class 1 columns (v_j times the unit column's polynomial `basis`) are one term together, sum_j apow_j v_j
// basis[p], the sum precomputed in fold[2].  Both instantiations are launched when a list exists and the one the
// batch does not call for returns at once: whether the list pays (fewer than 3/4 of the columns dense) is only
// known on the device, and one kernel holding both loops needs 66 instead of 46 VGPRs.  The structured columns'
// coefficients are in memory like anybody's (structured_fill_kernel), so the direct loop is always correct.
template <bool LIST>
__global__ __launch_bounds__(256) void reduce_columns_kernel(const gl_t *__restrict__ coeffs, uint32_t cols, uint32_t d,
                                                             const gl_t *__restrict__ apow, uint32_t j0,
                                                             gl_t *__restrict__ acc, int accumulate,
                                                             const uint32_t *__restrict__ nzlist,
                                                             const gl_t *__restrict__ basis, const gl_t *__restrict__ fold) {
  __shared__ gl_t red[2][4][64];
  const uint32_t n = 1u << d;
  const uint32_t p = blockIdx.x * 64 + threadIdx.x;
  const uint32_t g = threadIdx.y;
  if (nzlist != nullptr && (nzlist[0] * 4u < cols * 3u) != LIST) return;
  const bool live = p < n;
  gl_t a0 = 0, a1 = 0;
  if (live) {
    Acc160 l0, l1;  // unreduced sums of products (gl.hpp)
    l0.clear();
    l1.clear();
    uint32_t j = g;
    if constexpr (!LIST) {
#pragma unroll 1
      for (; j + 28 < cols; j += 32) {
        gl_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = coeffs[(size_t)(j + 4 * u) * n + p];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          l0.mac(v[u], apow[2 * (j0 + j + 4 * u)]);
          l1.mac(v[u], apow[2 * (j0 + j + 4 * u) + 1]);
        }
      }
      for (; j < cols; j += 4) {
        const gl_t v = coeffs[(size_t)j * n + p];
        l0.mac(v, apow[2 * (j0 + j)]);
        l1.mac(v, apow[2 * (j0 + j) + 1]);
      }
    } else {
      const uint32_t cnt = nzlist[0];
#pragma unroll 1
      for (; j + 28 < cnt; j += 32) {
        gl_t v[8];
        uint32_t cj[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          cj[u] = nzlist[1 + j + 4 * u];
          v[u] = coeffs[(size_t)cj[u] * n + p];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          l0.mac(v[u], apow[2 * (j0 + cj[u])]);
          l1.mac(v[u], apow[2 * (j0 + cj[u]) + 1]);
        }
      }
      for (; j < cnt; j += 4) {
        const uint32_t cj = nzlist[1 + j];
        const gl_t v = coeffs[(size_t)cj * n + p];
        l0.mac(v, apow[2 * (j0 + cj)]);
        l1.mac(v, apow[2 * (j0 + cj) + 1]);
      }
      if (basis != nullptr && g == 3) {
        const gl_t b = basis[p];
        l0.mac(b, fold[0]);
        l1.mac(b, fold[1]);
      }
    }
    a0 = l0.value();
    a1 = l1.value();
  }
  red[0][g][threadIdx.x] = a0;
  red[1][g][threadIdx.x] = a1;
  __syncthreads();
  if (g != 0 || !live) return;
  for (int q = 1; q < 4; q++) {
    a0 = gl_add(a0, red[0][q][threadIdx.x]);
    a1 = gl_add(a1, red[1][q][threadIdx.x]);
  }
  if (accumulate) {
    a0 = gl_add(a0, acc[p]);
    a1 = gl_add(a1, acc[(size_t)n + p]);
  }
  acc[p] = a0;
  acc[(size_t)n + p] = a1;
}
// list[0] = count, list[1..] = indices of the DENSE columns (class 2; or class 3: coefficients in memory, no shortcut),
// ascending.  One block: ballot + popcount per wave, wave totals through LDS (a serial loop over a few hundred dependent
// loads was 28 us on the critical path of every proof)
__global__ __launch_bounds__(256) void compact_nonzero_kernel(const uint32_t *flags, uint32_t cols, uint32_t *list) {
  __shared__ uint32_t wcnt[4];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  uint32_t base = 0;
  for (uint32_t j0 = 0; j0 < cols; j0 += 256) {
    const uint32_t j = j0 + threadIdx.x;
    const bool f = j < cols && flags[j] >= 2;
    const unsigned long long m = __ballot(f);
    if (lane == 0) wcnt[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = base, tot = 0;
    for (uint32_t q = 0; q < 4; q++) {
      if (q < w) off += wcnt[q];
      tot += wcnt[q];
    }
    if (f) list[1 + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = j;
    __syncthreads();
    base += tot;
  }
  if (threadIdx.x == 0) list[0] = base;
}
void compact_nonzero(hipStream_t st, const uint32_t *flags, uint32_t cols, uint32_t *list) {
  hipLaunchKernelGGL(compact_nonzero_kernel, dim3(1), dim3(256), 0, st, flags, cols, list);
}
// fold[e] = sum over the class 1 columns j of apow[2 (j0 + j) + e] * val[j]   (e = 0, 1)
__global__ __launch_bounds__(256) void class1_fold_kernel(const uint32_t *cls, const gl_t *val, uint32_t cols, const gl_t *apow,
                                                          uint32_t j0, gl_t *fold) {
  __shared__ gl_t s0[256], s1[256];
  gl_t a0 = 0, a1 = 0;
  for (uint32_t j = threadIdx.x; j < cols; j += blockDim.x)
    if (cls[j] == 1) {
      a0 = gl_add(a0, gl_mul(apow[2 * (j0 + j)], val[j]));
      a1 = gl_add(a1, gl_mul(apow[2 * (j0 + j) + 1], val[j]));
    }
  s0[threadIdx.x] = a0;
  s1[threadIdx.x] = a1;
  __syncthreads();
  for (uint32_t off = blockDim.x >> 1; off; off >>= 1) {
    if (threadIdx.x < off) {
      s0[threadIdx.x] = gl_add(s0[threadIdx.x], s0[threadIdx.x + off]);
      s1[threadIdx.x] = gl_add(s1[threadIdx.x], s1[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    fold[0] = s0[0];
    fold[1] = s1[0];
  }
}
void class1_fold(hipStream_t st, const ColHints &h, uint32_t cols, const gl_t *apow, uint32_t j0, gl_t *fold) {
  hipLaunchKernelGGL(class1_fold_kernel, dim3(1), dim3(256), 0, st, h.cls, h.val, cols, apow, j0, fold);
}
void reduce_columns(hipStream_t st, const gl_t *coeffs, uint32_t cols, uint32_t d, const gl_t *apow, uint32_t j0,
                    gl_t *acc, bool accumulate, const uint32_t *nzlist, const gl_t *basis, const gl_t *fold) {
  uint32_t n = 1u << d;
  ProfScope ps("reduce_columns_kernel", 8.0 * (cols + 4.0) * (double)n);
  hipLaunchKernelGGL(reduce_columns_kernel<false>, dim3((n + 63) / 64), dim3(64, 4), 0, st, coeffs, cols, d, apow, j0, acc,
                     accumulate ? 1 : 0, nzlist, basis, fold);
  if (nzlist)
    hipLaunchKernelGGL(reduce_columns_kernel<true>, dim3((n + 63) / 64), dim3(64, 4), 0, st, coeffs, cols, d, apow, j0, acc,
                       accumulate ? 1 : 0, nzlist, basis, fold);
}

// out[p] = sum_q parts[q][p] (p < len): the partial sums of a column-sharded batch reduction, one part per rank
__global__ __launch_bounds__(256) void sum_parts_kernel(const gl_t *__restrict__ parts, uint32_t nparts, size_t len, gl_t *__restrict__ out) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= len) return;
  gl_t a = parts[p];
  for (uint32_t q = 1; q < nparts; q++) a = gl_add(a, parts[(size_t)q * len + p]);
  out[p] = a;
}
void sum_parts(hipStream_t st, const gl_t *parts, uint32_t nparts, size_t len, gl_t *out) {
  ProfScope ps("sum_parts_kernel", 8.0 * (nparts + 1.0) * (double)len);
  hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, parts, nparts, len, out);
}

__global__ __launch_bounds__(256) void fri_quotient_values_kernel(const gl_t *F0, const gl_t *F1, uint32_t d,
                                                                  const gl_t *tw, uint32_t tw_shift, ext_t zeta,
                                                                  ext_t gzeta, ext_t f0z, ext_t f1z, ext_t aK,
                                                                  gl_t *out) {
  const uint32_t n = 1u << d;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t half = n >> 1;
  gl_t x = d == 0 ? 1 : (k < half ? tw[(size_t)k << tw_shift] : gl_neg(tw[(size_t)(k - half) << tw_shift]));
  ext_t v0 = ext_make(F0[k], F0[(size_t)n + k]), v1 = ext_make(F1[k], F1[(size_t)n + k]);
  ext_t q0 = ext_mul(ext_sub(v0, f0z), ext_inv(ext_sub(ext_from(x), zeta)));
  ext_t q1 = ext_mul(ext_sub(v1, f1z), ext_inv(ext_sub(ext_from(x), gzeta)));
  ext_t r = ext_add(ext_mul(aK, q0), q1);
  out[k] = r.c0;
  out[(size_t)n + k] = r.c1;
}
void fri_quotient_values(hipStream_t st, const gl_t *F0, const gl_t *F1, uint32_t d, const gl_t *tw, uint32_t tw_shift,
                         ext_t zeta, ext_t gzeta, ext_t f0z, ext_t f1z, ext_t aK, gl_t *out) {
  uint32_t n = 1u << d;
  uint32_t threads = n >= 256 ? 256 : 64;
  hipLaunchKernelGGL(fri_quotient_values_kernel, dim3((n + threads - 1) / threads), dim3(threads), 0, st, F0, F1, d, tw,
                     tw_shift, zeta, gzeta, f0z, f1z, aK, out);
}

// coefficients c[j] live at bitrev_d(j).  out[q] (= c'[bitrev_{d-ab}(q)... ]) =
// sum_t beta^t c[arity * m + t] with q = bitrev(m): source position bitrev_ab(t) * n' + q.
__global__ __launch_bounds__(256) void fri_fold_kernel(const gl_t *in, uint32_t d, uint32_t ab, ext_t beta, gl_t *out) {
  const uint32_t n = 1u << d, n2 = n >> ab, arity = 1u << ab;
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n2) return;
  ext_t acc = ext_from(0);
  for (uint32_t t = arity; t-- > 0;) {
    size_t pos = (size_t)bitrev32(t, ab) * n2 + q;
    acc = ext_add(ext_mul(acc, beta), ext_make(in[pos], in[(size_t)n + pos]));
  }
  out[q] = acc.c0;
  out[(size_t)n2 + q] = acc.c1;
}
void fri_fold(hipStream_t st, const gl_t *in, uint32_t d, uint32_t ab, ext_t beta, gl_t *out) {
  uint32_t n2 = (1u << d) >> ab;
  uint32_t threads = n2 >= 256 ? 256 : 64;
  hipLaunchKernelGGL(fri_fold_kernel, dim3((n2 + threads - 1) / threads), dim3(threads), 0, st, in, d, ab, beta, out);
}

struct PowState {
  gl_t s[12];
};
template <int H>
__global__ __launch_bounds__(256) void pow_kernel(PowState st0, uint32_t pos, uint32_t pow_bits, uint64_t base,
                                                  uint64_t count, unsigned long long *result, const gl_t *__restrict__ prc) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint64_t w = base + i;
  gl_t st[12];
#pragma unroll
  for (int j = 0; j < 12; j++) st[j] = st0.s[j];
#pragma unroll
  for (int j = 0; j < 12; j++)
    if ((uint32_t)j == pos) st[j] = w;
  if constexpr (H == 1) {  // Challenger<F, PoseidonHash>: the sponge permutation is Poseidon
    poseidon_permute_dev(st, prc);  // (prc = poseidon_device_constants)
  } else {
    keccak_permutation12<8>(st);  // the response is word 7: the third layer of the onion (words 8..11) is never looked at
  }
  const uint64_t resp = st[7];
  if (pow_bits == 0 || (resp >> (64 - pow_bits)) == 0) atomicMin(result, (unsigned long long)w);
}
void pow_search(hipStream_t st, const gl_t state[12], uint32_t pos, uint32_t pow_bits, uint64_t base, uint64_t count,
                unsigned long long *result, const gl_t *prc) {
  PowState s;
  for (int i = 0; i < 12; i++) s.s[i] = state[i];
  ProfScope ps("pow_kernel", 0.0);
  if (prc) hipLaunchKernelGGL(pow_kernel<1>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, s, pos, pow_bits, base, count, result, prc);
  else hipLaunchKernelGGL(pow_kernel<0>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, s, pos, pow_bits, base, count, result, prc);
}

__global__ void gather_kernel(const uint64_t *addr, uint32_t count, gl_t *out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = addr[i] ? *(const gl_t *)(uintptr_t)addr[i] : 0;
}
void gather_u64(hipStream_t st, const uint64_t *addr, uint32_t count, gl_t *out) {
  if (!count) return;
  ProfScope ps("gather_kernel", 24.0 * count);
  hipLaunchKernelGGL(gather_kernel, dim3((count + 255) / 256), dim3(256), 0, st, addr, count, out);
}

}  // namespace p2
