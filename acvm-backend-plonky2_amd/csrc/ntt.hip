// ntt.hip -- batched radix-2 Goldilocks NTT for gfx950, LDS-tiled.
//
// Replaces (inside `circuit_data.prove`, plonky2-backend/src/actions/prove_action.rs:96)
// plonky2 0.2.2 field/src/fft.rs `fft_classic` / `ifft` and
// PolynomialCoeffs::lde + coset_fft.
//
// MI355X-first layout instead of plonky2's natural->natural transforms:
//   * values -> coefficients is a DIF transform (natural in, BIT-REVERSED out);
//     coefficients stay in bit-reversed positions for their whole life;
//   * the 8x LDE on the coset 7<w_N> is 8 independent size-n DIT transforms
//     (bit-reversed in, natural out) with coset shifts 7*w_N^r, r < 8; natural
//     LDE row i = 8k + r is output k of coset r.  No transpose, no bit-reversal
//     pass and no zero-padded 8n-point FFT ever touches HBM.
// A transform of 2^d points is split into passes of <= 12 layers; each pass
// stages a 2^12-element tile (32 KB + padding) in LDS, keeps 16 elements per
// lane in registers for 4 butterfly layers at a time, and reads/writes HBM in
// runs of >= 128 contiguous bytes.  HBM-bound target; 64-bit modular products
// come from 32-bit multiplies (no MFMA: integer prime-field work).
#include "internal.hpp"

namespace p2 {

// LDS index padding: one extra slot per 16 so that 16-element strides do not
// collide on LDS banks
__device__ __forceinline__ uint32_t pidx(uint32_t e) { return e + (e >> 4); }

struct PassArgs {
  const gl_t *src;   // [cols][n] (coset passes read the same src for every coset)
  gl_t *dst;         // [cosets][cols][n]
  const gl_t *tw;    // root powers w_m^i, i < m/2, for some m >= n (stride tw_stride)
  const gl_t *scale; // optional per-position scale [cosets][n] applied on load (DIT first pass)
  gl_t post;         // scale applied on store (1/n for the inverse), 1 = none
  uint32_t d;        // log2 n
  uint32_t s;        // log2 global stride of the tile's lowest butterfly layer
  uint32_t a;        // number of layers in this pass
  uint32_t tb;       // log2 contiguous run (tile = 2^(a+tb) elements)
  uint32_t tw_shift; // log2(m / n)
  uint32_t cols;
  uint32_t src_coset_stride_zero;  // 1: src is not indexed by coset
};

// global index of tile element e
__device__ __forceinline__ uint32_t gidx(uint32_t e, uint32_t hi_base, uint32_t lo0, uint32_t s, uint32_t tb) {
  return hi_base + ((e >> tb) << s) + lo0 + (e & ((1u << tb) - 1));
}

template <int DIT, int LOGR>
__device__ __forceinline__ void round_regs(gl_t *lds, const PassArgs &A, uint32_t TB, uint32_t beta0, uint32_t lo0) {
  constexpr int R = 1 << LOGR;
  const uint32_t ngroups = 1u << (TB - LOGR);
  for (uint32_t g = threadIdx.x; g < ngroups; g += blockDim.x) {
    uint32_t low = g & ((1u << beta0) - 1), high = g >> beta0;
    uint32_t base = (high << (beta0 + LOGR)) | low;
    gl_t v[R];
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = lds[pidx(base | ((uint32_t)j << beta0))];
    // position of `base` in the global index space, modulo the layer stride
    if (DIT) {
#pragma unroll
      for (int lam = 0; lam < LOGR; lam++) {
        uint32_t beta = beta0 + lam;
        uint32_t lgS = beta - A.tb + A.s;  // log2 global stride of this layer
#pragma unroll
        for (int j = 0; j < R; j++) {
          if (j & (1 << lam)) continue;
          uint32_t e = base | ((uint32_t)j << beta0);
          uint32_t gm = (((e & ((1u << beta) - 1)) >> A.tb) << A.s) + lo0 + (e & ((1u << A.tb) - 1));
          gl_t w = A.tw[(size_t)gm << (A.d - 1 - lgS + A.tw_shift)];
          gl_t u = v[j], t = gl_mul(v[j | (1 << lam)], w);
          v[j] = gl_add(u, t);
          v[j | (1 << lam)] = gl_sub(u, t);
        }
      }
    } else {
#pragma unroll
      for (int lam = LOGR - 1; lam >= 0; lam--) {
        uint32_t beta = beta0 + lam;
        uint32_t lgS = beta - A.tb + A.s;
#pragma unroll
        for (int j = 0; j < R; j++) {
          if (j & (1 << lam)) continue;
          uint32_t e = base | ((uint32_t)j << beta0);
          uint32_t gm = (((e & ((1u << beta) - 1)) >> A.tb) << A.s) + lo0 + (e & ((1u << A.tb) - 1));
          gl_t w = A.tw[(size_t)gm << (A.d - 1 - lgS + A.tw_shift)];
          gl_t u = v[j], x = v[j | (1 << lam)];
          v[j] = gl_add(u, x);
          v[j | (1 << lam)] = gl_mul(gl_sub(u, x), w);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < R; j++) lds[pidx(base | ((uint32_t)j << beta0))] = v[j];
  }
}

template <int DIT>
__device__ __forceinline__ void do_round(gl_t *lds, const PassArgs &A, uint32_t TB, uint32_t beta0, uint32_t logr,
                                         uint32_t lo0) {
  switch (logr) {
  case 4: round_regs<DIT, 4>(lds, A, TB, beta0, lo0); break;
  case 3: round_regs<DIT, 3>(lds, A, TB, beta0, lo0); break;
  case 2: round_regs<DIT, 2>(lds, A, TB, beta0, lo0); break;
  default: round_regs<DIT, 1>(lds, A, TB, beta0, lo0); break;
  }
}

// grid: x = tile index within a column, y = column, z = coset
template <int DIT>
__global__ __launch_bounds__(256) void ntt_pass_kernel(PassArgs A) {
  extern __shared__ gl_t lds[];
  const uint32_t TB = A.a + A.tb;
  const uint32_t tile = blockIdx.x;
  const uint32_t col = blockIdx.y, coset = blockIdx.z;
  const size_t n = (size_t)1 << A.d;
  // tile -> (hi, lo0): tiles enumerate lo-runs fastest
  const uint32_t runs = 1u << (A.s - A.tb);  // number of lo-runs per hi block (s >= tb)
  const uint32_t hi = tile / runs, lo0 = (tile % runs) << A.tb;
  const uint32_t hi_base = hi << (A.s + A.a);
  const gl_t *src = A.src + ((size_t)(A.src_coset_stride_zero ? 0 : coset) * A.cols + col) * n;
  gl_t *dst = A.dst + ((size_t)coset * A.cols + col) * n;
  const gl_t *scale = A.scale ? A.scale + (size_t)coset * n : nullptr;
  const uint32_t tsize = 1u << TB;
  for (uint32_t e = threadIdx.x; e < tsize; e += blockDim.x) {
    uint32_t g = gidx(e, hi_base, lo0, A.s, A.tb);
    gl_t x = src[g];
    if (scale) x = gl_mul(x, scale[g]);
    lds[pidx(e)] = x;
  }
  __syncthreads();
  // layers live on tile bits [tb, tb + a)
  if (DIT) {
    uint32_t beta = A.tb, left = A.a;
    while (left) {
      uint32_t r = left >= 4 ? 4 : left;
      do_round<1>(lds, A, TB, beta, r, lo0);
      __syncthreads();
      beta += r;
      left -= r;
    }
  } else {
    uint32_t left = A.a;
    while (left) {
      uint32_t r = left >= 4 ? 4 : left;
      do_round<0>(lds, A, TB, A.tb + left - r, r, lo0);
      __syncthreads();
      left -= r;
    }
  }
  const bool post = A.post != 1;
  for (uint32_t e = threadIdx.x; e < tsize; e += blockDim.x) {
    gl_t x = lds[pidx(e)];
    if (post) x = gl_mul(x, A.post);
    dst[gidx(e, hi_base, lo0, A.s, A.tb)] = x;
  }
}

static inline size_t lds_bytes(uint32_t TB) { return (((size_t)1 << TB) + ((size_t)1 << TB) / 16 + 1) * sizeof(gl_t); }

// Plan: contiguous pass over the low `b` bits, strided passes over the rest.
// DIF (natural -> bitrev) runs the strided passes first; DIT runs them last.
void ntt_batch(hipStream_t st, int dit, const gl_t *src, gl_t *dst, uint32_t d, uint32_t cols, uint32_t cosets,
               const gl_t *tw, uint32_t tw_shift, const gl_t *scale, gl_t post, bool src_per_coset) {
  if (cols == 0) return;
  const uint32_t TBMAX = 12;
  struct P { uint32_t s, a, tb; };
  P passes[8];
  int np = 0;
  uint32_t b = d < TBMAX ? d : TBMAX;
  if (d == 0) {  // size-1 transform: copy (+scale)
    b = 0;
  }
  // contiguous pass: s = 0, a = b, tb = 0
  // strided passes: split the remaining d - b high bits into chunks of <= 8
  P strided[8];
  int ns = 0;
  uint32_t rem = d - b, s = b;
  while (rem) {
    uint32_t a = rem > 8 ? 8 : rem;
    uint32_t tb = TBMAX - a;
    if (tb > s) tb = s;
    strided[ns++] = P{s, a, tb};
    s += a;
    rem -= a;
  }
  if (dit) {
    passes[np++] = P{0, b, 0};
    for (int i = 0; i < ns; i++) passes[np++] = strided[i];
  } else {
    for (int i = ns - 1; i >= 0; i--) passes[np++] = strided[i];
    passes[np++] = P{0, b, 0};
  }
  for (int i = 0; i < np; i++) {
    PassArgs A;
    A.src = (i == 0) ? src : dst;
    A.dst = dst;
    A.tw = tw;
    A.scale = (i == 0 && dit) ? scale : nullptr;
    A.post = (i == np - 1) ? post : 1;
    A.d = d;
    A.s = passes[i].s;
    A.a = passes[i].a;
    A.tb = passes[i].tb;
    A.tw_shift = tw_shift;
    A.cols = cols;
    A.src_coset_stride_zero = (i == 0 && !src_per_coset) ? 1 : 0;
    uint32_t TB = A.a + A.tb;
    uint32_t tiles = 1u << (d - TB);
    dim3 grid(tiles, cols, cosets);
    uint32_t threads = TB >= 8 ? 256 : 64;
    ProfScope ps(dit ? "ntt_pass_kernel<1>" : "ntt_pass_kernel<0>", 16.0 * ((double)cols * cosets * ((size_t)1 << d)));
    if (dit)
      hipLaunchKernelGGL(ntt_pass_kernel<1>, grid, dim3(threads), lds_bytes(TB), st, A);
    else
      hipLaunchKernelGGL(ntt_pass_kernel<0>, grid, dim3(threads), lds_bytes(TB), st, A);
  }
}

// ---- tables -------------------------------------------------------------------
// tw[i] = root^i for i < count
__global__ void powers_kernel(gl_t *out, gl_t root, uint32_t count) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = gl_pow(root, i);
}
// scale[c][p] = (shift * wN^c)^(bitrev_d(p)) * mult
__global__ void coset_scale_kernel(gl_t *out, gl_t shift, gl_t wN, uint32_t d, uint32_t cosets, gl_t mult) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = blockIdx.y;
  if (p >= (1u << d)) return;
  gl_t base = gl_mul(shift, gl_pow(wN, c));
  out[((size_t)c << d) + p] = gl_mul(gl_pow(base, bitrev32(p, d)), mult);
}

void fill_powers(hipStream_t st, gl_t *out, gl_t root, uint32_t count) {
  if (!count) return;
  hipLaunchKernelGGL(powers_kernel, dim3((count + 255) / 256), dim3(256), 0, st, out, root, count);
}
void fill_coset_scale(hipStream_t st, gl_t *out, gl_t shift, gl_t wN, uint32_t d, uint32_t cosets, gl_t mult) {
  uint32_t n = 1u << d;
  hipLaunchKernelGGL(coset_scale_kernel, dim3((n + 255) / 256, cosets), dim3(256), 0, st, out, shift, wN, d, cosets,
                     mult);
}

// bit-reversal permutation of columns (only for the stage-level test operators
// that speak plonky2's natural-order coefficient convention)
__global__ void bitrev_cols_kernel(const gl_t *in, gl_t *out, uint32_t d, uint32_t cols) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = blockIdx.y;
  if (p >= (1u << d)) return;
  out[((size_t)c << d) + bitrev32(p, d)] = in[((size_t)c << d) + p];
}
void bitrev_cols(hipStream_t st, const gl_t *in, gl_t *out, uint32_t d, uint32_t cols) {
  uint32_t n = 1u << d;
  hipLaunchKernelGGL(bitrev_cols_kernel, dim3((n + 255) / 256, cols), dim3(256), 0, st, in, out, d, cols);
}

}  // namespace p2
