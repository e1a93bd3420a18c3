// ntt.hip -- batched Goldilocks NTT for gfx950: LDS-tiled passes, radix-8/4/2
// register rounds with shift twiddles.
//
// Replaces (inside `circuit_data.prove`, plonky2-backend/src/actions/prove_action.rs:96)
// plonky2 0.2.2 field/src/fft.rs `fft_classic` / `ifft` and
// PolynomialCoeffs::lde + coset_fft.
//
// MI355X-first layout instead of plonky2's natural->natural transforms:
//   * values -> coefficients is a DIF transform (natural in, BIT-REVERSED out);
//     coefficients stay in bit-reversed positions for their whole life;
//   * the 8x LDE on the coset g<w_N> (g = GL_GEN) is 8 independent size-n DIT transforms
//     (bit-reversed in, natural out) with coset shifts g*w_N^r, r < 8; natural
//     LDE row i = 8k + r is output k of coset r.  No transpose, no bit-reversal
//     pass and no zero-padded 8n-point FFT ever touches HBM.
// A transform of 2^d points is split into passes of <= 12 layers; a pass stages a
// 2^12-element tile (32 KB, XOR-swizzled: pidx() below) in LDS and reads/writes HBM in runs of
// >= 128 contiguous bytes.  Inside a pass the layers are done in rounds of up to
// three: a lane holds 8 elements in registers, multiplies them by 7 per-group
// twiddles (one coalesced load each from a table packed per round -- no index
// arithmetic, no scattered root-table gathers) and then runs an 8-point DFT whose
// internal twiddles are powers of w_8 = 2^24, i.e. shifts (2 is a 192nd
// root of unity in Goldilocks): 7 general modmuls per 8 elements per three
// layers instead of 12.  This kernel is VALU-issue bound on gfx950, in the HALF-rate
// instruction class (carry chains, v_mad_u64_u32: 4.2 cycles per wave64 instruction,
// profiles/r03_ubench.txt): 0.73-0.82 of the 39.5 T lane-instr/s its mix allows.
// HBM traffic of a 2^17-point LDE: 25n * 8 B per dense column against 9n * 8 B
// algorithmic (the intermediate between the 12-layer and the 5-layer pass goes out
// and back: a 1 MB column-coset does not fit a CU's LDS) -- 2.5 TB/s while the
// kernel runs, not its limiter.  No MFMA.
#include "internal.hpp"
#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

namespace p2 {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// x * 2^E mod p for a compile-time 0 <= E < 96 (canonical in, canonical out)
#if P2_GL_DEV_ASM
// x = x0 + x1 * 2^32, E = 32 q + r: the shifted words go straight into the carry-chain reductions of gl.hpp
// (2^64 = eps, 2^96 = -1): 11-14 VALU for every E, where the portable form below costs 15 (E < 32), 20
// (E < 64) or 35 (two steps).
template <int E>
__device__ __forceinline__ gl_t mul_pow2(gl_t x) {
  if constexpr (E == 0) {
    return x;
  } else if constexpr (E < 32) {
    const uint64_t lo = x << E;
    return gl_reduce_add_eps((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)(x >> (64 - E)));
  } else if constexpr (E == 32) {
    return gl_reduce_words(0u, (uint32_t)x, (uint32_t)(x >> 32), 0u);
  } else if constexpr (E < 64) {
    const uint64_t y = x << (E - 32);
    return gl_reduce_words(0u, (uint32_t)y, (uint32_t)(y >> 32), (uint32_t)(x >> (96 - E)));
  } else if constexpr (E == 64) {
    // x * 2^64 = -x * 2^-32 = x0 * eps - x1
    return gl_reduce_eps_sub((uint32_t)x, (uint32_t)(x >> 32), 0u);
  } else {
    // x * 2^E = -x * 2^-s, s = 96 - E in (0, 32): x 2^-s = (x >> s) - z eps, z = low s bits of x at the top of a word
    constexpr int S = 96 - E;
    const uint64_t h = x >> S;
    return gl_reduce_eps_sub((uint32_t)x << (32 - S), (uint32_t)h, (uint32_t)(h >> 32));
  }
}
#else
template <int E>
__device__ __forceinline__ gl_t mul_pow2(gl_t x) {
  if constexpr (E == 0) {
    return x;
  } else if constexpr (E < 32) {
    const uint64_t lo = x << E;
    const uint32_t hi = (uint32_t)(x >> (64 - E));  // < 2^E
    const uint64_t t1 = ((uint64_t)hi << 32) - hi;  // hi * (2^32 - 1)
    uint64_t t2 = lo + t1;
    if (t2 < t1) t2 += GL_EPS;
    return gl_canon(t2);
  } else if constexpr (E < 64) {
    return gl_reduce128(x << E, x >> (64 - E));
  } else {
    return mul_pow2<32>(mul_pow2<E - 32>(x));
  }
}
#endif

// exponent of 2 for the constant twiddle w_{2^(lam+1)}^q (w_64 = 2^3), mod 192
__host__ __device__ constexpr int tw_exp(int lam, int q, bool inv) {
  int e = (3 * (32 >> lam) * q) % 192;
  return inv ? (192 - e) % 192 : e;
}
__host__ __device__ constexpr int brev_c(int x, int bits) {
  int r = 0;
  for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}

// in-register 2^LOGR-point DFT; DIT: bit-reversed in -> natural out; DIF: natural in -> bit-reversed out
template <int LOGR, int DIT, bool INV>
__device__ __forceinline__ void dft_regs(gl_t (&v)[1 << LOGR]) {
  constexpr int R = 1 << LOGR;
  static_for<0, LOGR>([&](auto lc) {
    constexpr int lam = DIT ? decltype(lc)::value : LOGR - 1 - decltype(lc)::value;
    static_for<0, R>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr ((j & (1 << lam)) == 0) {
        constexpr int k = j | (1 << lam);
        constexpr int e = tw_exp(lam, j & ((1 << lam) - 1), INV);
        if constexpr (DIT) {
          const gl_t u = v[j];
          if constexpr (e < 96) {
            const gl_t t = mul_pow2<e>(v[k]);
            v[j] = gl_add(u, t);
            v[k] = gl_sub(u, t);
          } else {
            const gl_t t = mul_pow2<e - 96>(v[k]);  // twiddle = -2^(e-96)
            v[j] = gl_sub(u, t);
            v[k] = gl_add(u, t);
          }
        } else {
          const gl_t u = v[j], x = v[k];
          v[j] = gl_add(u, x);
          if constexpr (e < 96) v[k] = mul_pow2<e>(gl_sub(u, x));
          else v[k] = mul_pow2<e - 96>(gl_sub(x, u));
        }
      }
    });
  });
}

// LDS index swizzle (round 3; replaces the "one pad slot per 16" of rounds 1-2, which left 61 % of the kernel's LDS cycles
// as bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS of profiles/r02i_sq_summary.json).  ds_read_b64 serves a
// wave in two groups of 32 lanes over 32 eight-byte slots, ds_write_b64 in four groups of 16 lanes over 16 slots
// (MI355X_MICROARCH.md, LDS).  In a round on tile bits [beta0, beta0 + 3) the lanes of a group differ in element bits
// {3..7} (beta0 = 0), {0,1,2,6,7} (beta0 = 3) or {0..4} (beta0 >= 5, and the load / store phases): the slot bits
//   s0 = e0^e4, s1 = e1^e5, s2 = e2^e6, s3 = e3^e6, s4 = e4^e7
// are a GF(2)-linear map of full rank on each of those sets (and s0..s3 on the first four bits of each for the
// stores), so every access of the d >= 12 plans is conflict-free with no padding at all.  Linear means
// swz(a ^ b) = swz(a) ^ swz(b): a lane's eight addresses are swz(base) ^ swz(j << beta0), the second term wave-uniform.
__device__ __forceinline__ uint32_t pidx(uint32_t e) { return e ^ ((e >> 4) & 7u) ^ ((e >> 3) & 0x18u); }

constexpr int MAX_ROUNDS = NTT_MAX_ROUNDS;
#ifndef NTT_TILE_BITS
#define NTT_TILE_BITS 12
#endif
// elements a lane holds in the full-tile kernel = largest round radix.  16: radix-16 rounds, 256 lanes per tile,
// ~100 VGPRs, 4 tiles (16 waves) per CU because of the 34 KB of LDS a tile needs.  8: radix-8 rounds, 512 lanes
// per tile, the same LDS but twice the waves per CU to cover barriers and HBM latency
#ifndef NTT_PER
#define NTT_PER 8
#endif
#define NTT_THREADS ((1 << NTT_TILE_BITS) / NTT_PER)
struct PassArgs {
  const gl_t *src;    // [cols][n] (or [cosets][cols][n])
  gl_t *dst;          // [cosets][cols][n]
  const gl_t *ptw;    // packed per-round twiddles of this plan
  const gl_t *scale;  // optional [cosets][n], multiplied into the input of the first DIT pass
  gl_t post;          // multiplied into the output of the last pass (1 = none)
  uint32_t d;         // log2 n
  uint32_t s;         // log2 global stride of the pass's lowest layer
  uint32_t a;         // layers in this pass
  uint32_t tb;        // log2 contiguous run; tile = 2^(a+tb)
  uint32_t cols;        // column stride between cosets in src / dst
  uint32_t tiles, cols_grid, cosets;  // launch shape (1-D grid, decoded XCD-aware in the kernel)
  uint32_t src_single;  // 1: src has no coset dimension
  uint32_t coset_first, coset_stride;  // global coset of grid.z = first + z * stride (indexes `scale`)
  uint32_t nrounds;
  uint32_t r[MAX_ROUNDS];       // layers per round, ascending tile bit
  uint32_t tw_off[MAX_ROUNDS];  // offset of the round's table in ptw
  const uint32_t *colnz;        // optional [cols]: class of every column (ColHints, internal.hpp); only class 2 (dense)
                                // columns are transformed here, the others are written by structured_fill_kernel
  const gl_t *ftw;              // direct DIT passes: folded twiddles of the round before the last (fold_table_kernel)
  const gl_t *ftw2;             // ntt_dit_head2_kernel: the same values in its lane order (head_fold_table_kernel)
};

__device__ __forceinline__ uint32_t gidx(uint32_t e, uint32_t hi_base, uint32_t lo0, uint32_t s, uint32_t tb) {
  return hi_base + ((e >> tb) << s) + lo0 + (e & ((1u << tb) - 1));
}

// one round: layers on tile bits [beta0, beta0 + LOGR); the group twiddle of position j is
// theta^(bitrev(j)), theta = w_{2^(s0+LOGR)}^(lo), read from the packed table T[(e-1)*M + lo].
// MUL = (s0 > 0): the round has general twiddles (the first round of a transform has none).
template <int DIT, bool INV, int LOGR, bool MUL>
__device__ __forceinline__ void round_regs(gl_t *lds, const PassArgs &A, uint32_t TB, uint32_t beta0, uint32_t lo0,
                                           const gl_t *tw) {
  constexpr int R = 1 << LOGR;
  const uint32_t ngroups = 1u << (TB - LOGR);
  const uint32_t s0 = beta0 - A.tb + A.s;  // log2 M
  // Index arithmetic is a tenth of this kernel's instructions if done per access: the swizzled LDS index of element
  // base | (j << beta0) is pidx(base) ^ pidx(j << beta0) (the swizzle is GF(2)-linear), the second term wave-uniform;
  // the packed twiddle table is walked with one 64-bit add per entry.
  uint32_t lj[R];
#pragma unroll
  for (int j = 0; j < R; j++) lj[j] = pidx((uint32_t)j << beta0);
  const size_t tstride = (size_t)1 << s0;
  for (uint32_t g = threadIdx.x; g < ngroups; g += blockDim.x) {
    const uint32_t low = g & ((1u << beta0) - 1), high = g >> beta0;
    const uint32_t base = (high << (beta0 + LOGR)) | low;
    const uint32_t lo = (((base & ((1u << beta0) - 1)) >> A.tb) << A.s) + lo0 + (base & ((1u << A.tb) - 1));
    gl_t v[R];
    uint32_t li[R];
    {
      const uint32_t l0 = pidx(base);
#pragma unroll
      for (int j = 0; j < R; j++) li[j] = l0 ^ lj[j];
    }
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = lds[li[j]];
    gl_t t[R];  // t[e]: twiddle of table row e - 1
    if constexpr (MUL) {
      const gl_t *tp = tw + lo;
#pragma unroll
      for (int e = 1; e < R; e++) {
#ifdef NTT_EXP_NOTW   // timing experiment only (wrong results): what waiting for the twiddle loads costs
        t[e] = (gl_t)(size_t)tp;
#else
        t[e] = *tp;
#endif
        tp += tstride;
      }
    }
    if constexpr (DIT && MUL) {
      static_for<1, R>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        v[j] = gl_mul(v[j], t[brev_c(j, LOGR)]);
      });
    }
    dft_regs<LOGR, DIT, INV>(v);
    if constexpr (!DIT && MUL) {
      static_for<1, R>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        v[j] = gl_mul(v[j], t[brev_c(j, LOGR)]);
      });
    }
#pragma unroll
    for (int j = 0; j < R; j++) lds[li[j]] = v[j];
  }
}

template <int DIT, bool INV>
__device__ __forceinline__ void do_round(gl_t *lds, const PassArgs &A, uint32_t TB, uint32_t beta0, uint32_t logr,
                                         uint32_t lo0, const gl_t *tw) {
  const bool mul = beta0 - A.tb + A.s > 0;
#define P2_ROUND(L)                                                        \
  do {                                                                     \
    if (mul) round_regs<DIT, INV, L, true>(lds, A, TB, beta0, lo0, tw);    \
    else round_regs<DIT, INV, L, false>(lds, A, TB, beta0, lo0, tw);       \
  } while (0)
  switch (logr) {
#if NTT_PER == 16
  case 4: P2_ROUND(4); break;
#endif
  case 3: P2_ROUND(3); break;
  case 2: P2_ROUND(2); break;
  default: P2_ROUND(1); break;
  }
#undef P2_ROUND
}

// One tile of one pass: global -> LDS (XOR-swizzled), the pass's rounds, LDS -> global.
// TBC = 12: the full 2^12-element tile with NTT_THREADS lanes -- the global loads/stores of a lane are NTT_PER
// independent accesses issued back to back (compile-time trip count) so their latencies overlap;
// TBC = 0: any smaller tile (small transforms), runtime loops.
// waves per SIMD the full-tile kernel is compiled for: 8 = four 512-lane workgroups per CU, i.e. <= 64 VGPRs.  hipcc reaches
// 59-60 on its own; saying so makes it schedule for that bound (LDE 1.22-1.24 -> 1.205-1.21 ms at 2^20 rows, round 4) and
// keeps an edit that would silently cost a quarter of the occupancy (every variant above 64 VGPRs measured 20-30 % slower)
// from compiling into one
#ifndef NTT_MIN_WAVES
#if NTT_PER != 8
#define NTT_MIN_WAVES 1
#else
#define NTT_MIN_WAVES 8
#endif
#endif
template <int DIT, bool INV, int TBC>
__device__ __forceinline__ void tile_body(gl_t *lds, const PassArgs &A, uint32_t tile, uint32_t col, uint32_t coset) {
  const uint32_t TB = TBC ? TBC : A.a + A.tb;
  const size_t n = (size_t)1 << A.d;
  const uint32_t runs = 1u << (A.s - A.tb);  // lo-runs per hi block
  const uint32_t hi = tile / runs, lo0 = (tile % runs) << A.tb;
  const uint32_t hi_base = hi << (A.s + A.a);
  const gl_t *src = A.src + ((size_t)(A.src_single ? 0 : coset) * A.cols + col) * n;
  gl_t *dst = A.dst + ((size_t)coset * A.cols + col) * n;
  const gl_t *scale = A.scale ? A.scale + (size_t)(A.coset_first + coset * A.coset_stride) * n : nullptr;
  const uint32_t tsize = 1u << TB;
  // lane e + i * NT of the tile: when a contiguous run (2^tb elements) divides NT, the global index is linear in i
  // (g0 + i * gstep) and so is the swizzled LDS index (l0 + i * NT: the swizzle only touches bits below log2 NT)
  constexpr int NT12 = NTT_THREADS;
  const bool glin = (1u << A.tb) <= (uint32_t)NT12;
  const uint32_t g0 = gidx(threadIdx.x, hi_base, lo0, A.s, A.tb), gstep = (NT12 >> (glin ? A.tb : 0)) << A.s;
  const uint32_t l0 = pidx(threadIdx.x);
  if constexpr (TBC != 0) {
    constexpr int PER = NTT_PER, NT = (1 << TBC) / PER;
    static_assert(NT == NT12, "full tile: NTT_THREADS lanes");
    gl_t x[PER];
    uint32_t g[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
      g[i] = glin ? g0 + (uint32_t)i * gstep : gidx(threadIdx.x + i * NT, hi_base, lo0, A.s, A.tb);
#ifdef NTT_EXP_NOLOAD   // timing experiment only (wrong results): what the load phase costs
      x[i] = (gl_t)g[i] * 0x9E3779B97F4A7C15ULL;
#else
      x[i] = src[g[i]];
#endif
    }
    if (scale) {
      gl_t sc[PER];
#pragma unroll
      for (int i = 0; i < PER; i++) sc[i] = scale[g[i]];
#pragma unroll
      for (int i = 0; i < PER; i++) x[i] = gl_mul(x[i], sc[i]);
    }
#pragma unroll
    for (int i = 0; i < PER; i++) lds[l0 + (uint32_t)i * NT] = x[i];
  } else {
    for (uint32_t e = threadIdx.x; e < tsize; e += blockDim.x) {
      const uint32_t g = gidx(e, hi_base, lo0, A.s, A.tb);
      gl_t x = src[g];
      if (scale) x = gl_mul(x, scale[g]);
      lds[pidx(e)] = x;
    }
  }
  __syncthreads();
  // A radix-NTT_PER round on tile bits [beta, beta + 3) below bit 9 of a full tile touches, in wave w, exactly the elements
  // whose bits [9, 12) are w (lane g = 64 w + l holds base = (g >> beta << (beta + 3)) | (g & (2^beta - 1)) and its 8
  // strides): consecutive such rounds exchange data inside a wave only, and a wave's LDS operations execute in order --
  // no workgroup barrier between them.
  // That argument needs: 512 lanes in wave64 (wave w = lanes [64 w, 64 w + 64)), ONE group per lane (g == threadIdx.x: the
  // trip count of round_regs' loop is 1) and round_regs' base = (g >> beta << (beta + 3)) | low split -- tied down here so that
  // an edit to any of them fails to compile instead of racing on LDS.
  static_assert(TBC != 12 || NTT_PER != 8 || NTT_THREADS == 512, "wave-private rounds: one radix-8 group per lane of a 2^12 tile");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "wave-private rounds assume wave64: gfx950 (CDNA has no wave32 mode) is the only target of this file"
#endif
  const bool one_group_per_lane = (1u << (TB - 3)) == blockDim.x;
  auto wave_private = [&](uint32_t beta, uint32_t r) { return TBC == 12 && NTT_PER == 8 && one_group_per_lane && r == 3 && beta + 3 <= 9; };
  if (DIT) {
    uint32_t beta = A.tb;
    for (uint32_t i = 0; i < A.nrounds; i++) {
      do_round<1, INV>(lds, A, TB, beta, A.r[i], lo0, A.ptw + A.tw_off[i]);
      if (i + 1 < A.nrounds && wave_private(beta, A.r[i]) && wave_private(beta + A.r[i], A.r[i + 1])) asm volatile("" ::: "memory");
      else __syncthreads();
      beta += A.r[i];
    }
  } else {
    uint32_t beta = A.tb + A.a;
    for (uint32_t i = A.nrounds; i-- > 0;) {
      beta -= A.r[i];
      do_round<0, INV>(lds, A, TB, beta, A.r[i], lo0, A.ptw + A.tw_off[i]);
      if (i > 0 && wave_private(beta, A.r[i]) && wave_private(beta - A.r[i - 1], A.r[i - 1])) asm volatile("" ::: "memory");
      else __syncthreads();
    }
  }
  const bool post = A.post != 1;
  if constexpr (TBC != 0) {
    constexpr int PER = NTT_PER, NT = (1 << TBC) / PER;
    gl_t x[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) x[i] = lds[l0 + (uint32_t)i * NT];
    if (post) {
#pragma unroll
      for (int i = 0; i < PER; i++) x[i] = gl_mul(x[i], A.post);
    }
#ifdef NTT_EXP_NOSTORE  // timing experiment only (wrong results): what the store phase costs
    if ((x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[4] ^ x[5] ^ x[6] ^ x[7]) == 0x1234567ULL)
#endif
#pragma unroll
    for (int i = 0; i < PER; i++) dst[glin ? g0 + (uint32_t)i * gstep : gidx(threadIdx.x + i * NT, hi_base, lo0, A.s, A.tb)] = x[i];
  } else {
    for (uint32_t e = threadIdx.x; e < tsize; e += blockDim.x) {
      gl_t x = lds[pidx(e)];
      if (post) x = gl_mul(x, A.post);
      dst[gidx(e, hi_base, lo0, A.s, A.tb)] = x;
    }
  }
}

// One pass as its own launch.  1-D grid decoded XCD-aware: the hardware deals consecutive block ids round-robin over
// the 8 XCDs, each with its own L2.  The 2^rate_bits coset transforms of one (tile, column) read the SAME coefficients
// (and neighbouring rows of the scale table): they get consecutive slots of ONE XCD, so the tile comes from HBM once
// and from that XCD's L2 afterwards (a speed choice only: any placement gives the same result).
template <int DIT, bool INV, int TBC>
__global__ __launch_bounds__(TBC ? NTT_THREADS : 256, TBC ? NTT_MIN_WAVES : 1) void ntt_pass_kernel(PassArgs A) {
  extern __shared__ gl_t lds[];
  const uint32_t units = A.tiles * A.cols_grid;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t coset0 = slot % A.cosets, unit = (slot / A.cosets) * 8u + xcd;
  if (unit >= units) return;
  const uint32_t tile = unit % A.tiles, col = unit / A.tiles;
  // Structured columns (internal.hpp ColHints): a transform is linear, so a zero column maps to zeros and
  // v * (unit column of the fixed row) to v * (that unit column's transform): structured_fill_kernel wrote them,
  // nothing to do here (block-uniform scalar branch).
  if (A.colnz != nullptr && A.colnz[col] != 2u) return;
  tile_body<DIT, INV, TBC>(lds, A, tile, col, coset0);
}

// ---- direct DIT passes (round 5) -----------------------------------------------------------------------------------------
// The forward (coefficients -> coset values) transforms are DIT passes on full 2^12 tiles.  ntt_pass_kernel above stages a tile in
// LDS, runs every round LDS -> registers -> LDS with a table twiddle on 7 of 8 inputs, and copies the tile out: five LDS round
// trips and three barriers for the 12-layer pass, three and three for the 5-layer one.  Two observations remove a third of that:
//   * The LAST round of a pass holds, in lane t, exactly the elements t | (j << beta) that lane t would copy to global memory
//     afterwards, and (strided passes) the FIRST round's elements base | (j << tb) are what a coalesced load hands a lane: the
//     first round runs straight from global memory, the last one stores straight to it.
//   * The twiddle of the last round is w_{2^(s0 + r)}^(lo * brev(j)) with lo = lo' + 2^s0' * k, k the OUTPUT index of the round
//     before: w^(lo' * brev(j)) is constant over that earlier round's butterfly group (fixed lo', fixed j = its `high` bits), so
//     it is multiplied into that round's input twiddles (one folded table: fold_table_kernel), and what is left,
//     w_{2^(r' + r)}^(k * brev(j)), is a power of w_64 = 2^3 -- a SHIFT -- with k uniform over the wave (k sits in bits >= 6 of the
//     lane's group index): a scalar branch on k, then compile-time shift amounts.
// 12-layer first pass: 5.25 -> 4.5 general products per element over the 17 layers of a 2^17-point transform together with the
// 5-layer pass, 8 -> 5 LDS round trips, 6 -> 3 barriers.  Same values (the field result of a butterfly network does not depend on
// how its twiddles are factored): bit-exact against the oracle and against ntt_pass_kernel (P2GPU_NTT_DIRECT=0).
template <int E>
__device__ __forceinline__ gl_t mul_pow2_any(gl_t x) {  // x * 2^E, 0 <= E < 192 (2^96 = -1)
  if constexpr (E == 0) return x;
  else if constexpr (E < 96) return mul_pow2<E>(x);
  else if constexpr (E == 96) return gl_sub((gl_t)0, x);
  else return gl_sub((gl_t)0, mul_pow2<E - 96>(x));
}
// the shift twiddles of a last round of RL layers after a round of RP layers: register j *= w_{2^(RP+RL)}^(k * brev(j))
template <bool INV, int RL, int RP, int K>
__device__ __forceinline__ void shift_twiddles(gl_t (&v)[1 << RL]) {
  constexpr int UNIT = 192 >> (RP + RL);
  static_for<1, (1 << RL)>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    constexpr int e0 = (UNIT * K * brev_c(j, RL)) % 192;
    constexpr int e = INV ? (192 - e0) % 192 : e0;
    v[j] = mul_pow2_any<e>(v[j]);
  });
}
template <bool INV, int RL, int RP>
__device__ __forceinline__ void shift_twiddles_k(gl_t (&v)[1 << RL], uint32_t k) {
  static_assert(RP == 2 || RP == 3, "round before the last: two or three layers");
  switch (__builtin_amdgcn_readfirstlane(k)) {  // k is wave-uniform: a scalar branch
  case 0: break;
  case 1: shift_twiddles<INV, RL, RP, 1>(v); break;
  case 2: shift_twiddles<INV, RL, RP, 2>(v); break;
  case 3: shift_twiddles<INV, RL, RP, 3>(v); break;
  case 4: if constexpr (RP == 3) shift_twiddles<INV, RL, RP, 4>(v); break;
  case 5: if constexpr (RP == 3) shift_twiddles<INV, RL, RP, 5>(v); break;
  case 6: if constexpr (RP == 3) shift_twiddles<INV, RL, RP, 6>(v); break;
  default: if constexpr (RP == 3) shift_twiddles<INV, RL, RP, 7>(v); break;
  }
}
// A round on tile bits [beta0, beta0 + LOGR) of a 2^12 tile in LDS whose input twiddles also carry the general part of the LAST
// round's (RL layers, the tile's top bits = this round's `high`): table row e = brev(j), then high, then lo.
template <bool INV, int LOGR>
__device__ __forceinline__ void round_folded(gl_t *lds, const PassArgs &A, uint32_t beta0, uint32_t rl, uint32_t lo0, const gl_t *ftw) {
  constexpr int R = 1 << LOGR;
  const uint32_t ngroups = 1u << (12 - LOGR);
  const uint32_t s0 = beta0 - A.tb + A.s;
  uint32_t lj[R];
#pragma unroll
  for (int j = 0; j < R; j++) lj[j] = pidx((uint32_t)j << beta0);
  const size_t estride = (size_t)1 << (s0 + rl);
  for (uint32_t g = threadIdx.x; g < ngroups; g += NTT_THREADS) {
    const uint32_t low = g & ((1u << beta0) - 1), high = g >> beta0;
    const uint32_t base = (high << (beta0 + LOGR)) | low;
    const uint32_t lo = ((low >> A.tb) << A.s) + lo0 + (low & ((1u << A.tb) - 1));
    gl_t v[R], t[R];
    uint32_t li[R];
    const uint32_t l0 = pidx(base);
#pragma unroll
    for (int j = 0; j < R; j++) li[j] = l0 ^ lj[j];
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = lds[li[j]];
    const gl_t *tp = ftw + ((size_t)high << s0) + lo;
#pragma unroll
    for (int e = 0; e < R; e++) {
      t[e] = *tp;
      tp += estride;
    }
    static_for<0, R>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      v[j] = gl_mul(v[j], t[brev_c(j, LOGR)]);
    });
    dft_regs<LOGR, 1, INV>(v);
#pragma unroll
    for (int j = 0; j < R; j++) lds[li[j]] = v[j];
  }
}
// The last round of a pass, LDS -> registers -> global: RL layers on the tile's top bits, shift twiddles (the round before had RP
// layers; its folded table carried the rest), then the pass's output scale.
// second swizzle, for the head pass whose first round runs in registers (ntt_dit_head2_kernel): its LDS accesses differ, within a
// lane group, in element bits {3,4,5,6} (16-lane store groups after round 0), {6..10} / {6..9} (round on bits 3-5, wave = bits 0-2),
// {3,4,5,9,10} / {3,4,5,9} (round on bits 6-8) and {0..4} (last round): slot bits s0 = e0^e3^e6, s1 = e1^e4^e7, s2 = e2^e5^e8,
// s3 = e3^e9, s4 = e4^e10 have full rank on each of them (found by exhaustive search over shift-and-mask forms: the only two-term
// one).  GF(2)-linear like pidx: pidx2(a ^ b) = pidx2(a) ^ pidx2(b).
__device__ __forceinline__ constexpr uint32_t pidx2(uint32_t e) { return e ^ ((e >> 3) & 7u) ^ ((e >> 6) & 31u); }
template <bool INV, int RL, int RP, int SW = 0>
__device__ __forceinline__ void round_to_global(const gl_t *lds, gl_t *dst, const PassArgs &A, uint32_t hi_base, uint32_t lo0) {
  constexpr int R = 1 << RL;
  constexpr uint32_t beta0 = 12 - RL;
  auto sw = [](uint32_t e) { return SW ? pidx2(e) : pidx(e); };
  uint32_t lj[R];
#pragma unroll
  for (int j = 0; j < R; j++) lj[j] = sw((uint32_t)j << beta0);
  const bool post = A.post != 1;
#pragma unroll
  for (uint32_t it = 0; it < ((1u << beta0) / NTT_THREADS); it++) {
    const uint32_t g = threadIdx.x + it * NTT_THREADS;
    gl_t v[R];
    const uint32_t l0 = sw(g);
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = lds[l0 ^ lj[j]];
    shift_twiddles_k<INV, RL, RP>(v, (g >> (beta0 - RP)) & ((1u << RP) - 1));
    dft_regs<RL, 1, INV>(v);
    if (post) {
#pragma unroll
      for (int j = 0; j < R; j++) v[j] = gl_mul(v[j], A.post);
    }
    const uint32_t a0 = hi_base + ((g >> A.tb) << A.s) + lo0 + (g & ((1u << A.tb) - 1));
#pragma unroll
    for (int j = 0; j < R; j++) dst[a0 + ((uint32_t)j << (beta0 - A.tb + A.s))] = v[j];
  }
}
// block -> (tile, column, coset): the XCD-aware decode of ntt_pass_kernel
__device__ __forceinline__ bool pass_unit(const PassArgs &A, uint32_t &tile, uint32_t &col, uint32_t &coset) {
  const uint32_t units = A.tiles * A.cols_grid;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  coset = slot % A.cosets;
  const uint32_t unit = (slot / A.cosets) * 8u + xcd;
  if (unit >= units) return false;
  tile = unit % A.tiles;
  col = unit / A.tiles;
  return !(A.colnz != nullptr && A.colnz[col] != 2u);
}
// The 12-layer first pass of a DIT transform of >= 2^12 points (s = 0, tb = 0: contiguous tiles): coalesced load (+ coset scale)
// -> LDS | rounds on bits 0-2, 3-5, 6-8 inside each wave (no barrier; the third with the folded table) | last round -> global.
template <bool INV>
__global__ __launch_bounds__(NTT_THREADS, NTT_MIN_WAVES) void ntt_dit_head_kernel(PassArgs A) {
  static_assert(NTT_THREADS == 512 && NTT_TILE_BITS == 12 && NTT_PER == 8, "direct passes: 512 lanes x 8 elements");
  extern __shared__ gl_t lds[];
  uint32_t tile, col, coset;
  if (!pass_unit(A, tile, col, coset)) return;
  const size_t n = (size_t)1 << A.d;
  const gl_t *src = A.src + ((size_t)(A.src_single ? 0 : coset) * A.cols + col) * n;
  gl_t *dst = A.dst + ((size_t)coset * A.cols + col) * n;
  const gl_t *scale = A.scale ? A.scale + (size_t)(A.coset_first + coset * A.coset_stride) * n : nullptr;
  const uint32_t tbase = tile << 12;
  {
    gl_t x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = src[tbase + threadIdx.x + (uint32_t)i * NTT_THREADS];
    if (scale) {
      gl_t sc[8];
#pragma unroll
      for (int i = 0; i < 8; i++) sc[i] = scale[tbase + threadIdx.x + (uint32_t)i * NTT_THREADS];
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = gl_mul(x[i], sc[i]);
    }
    const uint32_t l0 = pidx(threadIdx.x);  // (the swizzle moves bits below 5 only: + i * 512 commutes with it)
#pragma unroll
    for (int i = 0; i < 8; i++) lds[l0 + (uint32_t)i * NTT_THREADS] = x[i];
  }
  __syncthreads();
  // rounds on bits [0, 9): wave w works on the elements whose bits [9, 12) are w in all three, and a wave's LDS operations
  // execute in order (see wave_private in tile_body)
  round_regs<1, INV, 3, false>(lds, A, 12, 0, 0, nullptr);
  asm volatile("" ::: "memory");
  round_regs<1, INV, 3, true>(lds, A, 12, 3, 0, A.ptw + A.tw_off[1]);
  asm volatile("" ::: "memory");
  round_folded<INV, 3>(lds, A, 6, 3, 0, A.ftw);
  __syncthreads();
  round_to_global<INV, 3, 3>(lds, dst, A, tbase, 0);
}
// The same pass with its FIRST round in registers too, and the second one on shifts (w_64 = 2^3, so the 64-point
// transform on tile bits 0-5 needs no table): a lane loads 8 CONSECUTIVE words (64 B: the round on bits 0-2 is then its own eight
// registers; a wave still covers 4 KB contiguous), scales them, runs the twiddle-free 8-point transform and only then stores to LDS.
// The round on bits 3-5 multiplies input j by w_64^(k1 brev(j)), k1 = element bits 0-2: made WAVE-uniform by giving wave w the
// groups with k1 = w (lane = bits 6-11), so the eight shift amounts are compile-time behind one scalar branch; the round on bits
// 6-8 (folded table, laid out in this kernel's lane order: head_fold_table_kernel) keeps wave <-> k1 and therefore needs no barrier
// in between; then the last round -> global as in ntt_dit_head_kernel.  Against that kernel: 3 instead of 4 LDS round trips, 7 of 8
// general products per element gone, the same two barriers.  LDS addresses through pidx2 (every access conflict-free).
template <bool INV>
__global__ __launch_bounds__(NTT_THREADS, NTT_MIN_WAVES) void ntt_dit_head2_kernel(PassArgs A) {
  extern __shared__ gl_t lds[];
  uint32_t tile, col, coset;
  if (!pass_unit(A, tile, col, coset)) return;
  const size_t n = (size_t)1 << A.d;
  const gl_t *src = A.src + ((size_t)(A.src_single ? 0 : coset) * A.cols + col) * n;
  gl_t *dst = A.dst + ((size_t)coset * A.cols + col) * n;
  const gl_t *scale = A.scale ? A.scale + (size_t)(A.coset_first + coset * A.coset_stride) * n : nullptr;
  const uint32_t tbase = tile << 12, t = threadIdx.x, w = t >> 6, l = t & 63u;
  gl_t v[8];
  {
    const ulonglong2 *s2 = reinterpret_cast<const ulonglong2 *>(src + tbase + 8 * t);  // (columns and tiles are 64 B aligned)
    ulonglong2 x2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) x2[i] = s2[i];
    if (scale) {
      const ulonglong2 *c2 = reinterpret_cast<const ulonglong2 *>(scale + tbase + 8 * t);
      ulonglong2 sc[4];
#pragma unroll
      for (int i = 0; i < 4; i++) sc[i] = c2[i];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        v[2 * i] = gl_mul(x2[i].x, sc[i].x);
        v[2 * i + 1] = gl_mul(x2[i].y, sc[i].y);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        v[2 * i] = x2[i].x;
        v[2 * i + 1] = x2[i].y;
      }
    }
  }
  dft_regs<3, 1, INV>(v);  // bits 0-2: no twiddles
  {
    const uint32_t p0 = pidx2(8 * t);  // pidx2(k) = k for k < 8
#pragma unroll
    for (int k = 0; k < 8; k++) lds[p0 ^ (uint32_t)k] = v[k];
  }
  __syncthreads();
  {  // bits 3-5: element = w + 8 j + 64 l
    const uint32_t p0 = pidx2(w | (l << 6));
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = lds[p0 ^ pidx2((uint32_t)j << 3)];
    shift_twiddles_k<INV, 3, 3>(v, w);
    dft_regs<3, 1, INV>(v);
#pragma unroll
    for (int j = 0; j < 8; j++) lds[p0 ^ pidx2((uint32_t)j << 3)] = v[j];
  }
  asm volatile("" ::: "memory");  // the same wave, the same 512 elements (bits 0-2 = w): in-order LDS operations, no barrier
  {  // bits 6-8: element = w + 8 a + 64 i + 512 J, lane = a + 8 J; twiddles F2[e * 512 + t]
    const uint32_t p0 = pidx2(w | ((l & 7u) << 3) | ((l >> 3) << 9));
    gl_t tw[8];
    const gl_t *tp = A.ftw2 + t;
#pragma unroll
    for (int e = 0; e < 8; e++) tw[e] = tp[e * NTT_THREADS];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = lds[p0 ^ pidx2((uint32_t)i << 6)];
    static_for<0, 8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      v[i] = gl_mul(v[i], tw[brev_c(i, 3)]);
    });
    dft_regs<3, 1, INV>(v);
#pragma unroll
    for (int i = 0; i < 8; i++) lds[p0 ^ pidx2((uint32_t)i << 6)] = v[i];
  }
  __syncthreads();
  round_to_global<INV, 3, 3, 1>(lds, dst, A, tbase, 0);
}
// A strided DIT pass of a = 3 + RM + RL layers (RM = 0: no middle round; RL = 0: a = 3, no LDS at all) on 2^12-element tiles of
// 2^a rows x 2^tb contiguous words: first round global -> registers (its 8 elements are the rows base + j: one coalesced
// load each), [middle round in LDS,] last round -> global.  The round before the last uses the folded table.
template <bool INV, int RM, int RL>
__global__ __launch_bounds__(NTT_THREADS, NTT_MIN_WAVES) void ntt_dit_strided_kernel(PassArgs A) {
  extern __shared__ gl_t lds[];
  uint32_t tile, col, coset;
  if (!pass_unit(A, tile, col, coset)) return;
  const size_t n = (size_t)1 << A.d;
  const gl_t *src = A.src + ((size_t)(A.src_single ? 0 : coset) * A.cols + col) * n;
  gl_t *dst = A.dst + ((size_t)coset * A.cols + col) * n;
  const uint32_t runs = 1u << (A.s - A.tb);
  const uint32_t hi = tile / runs, lo0 = (tile % runs) << A.tb;
  const uint32_t hi_base = hi << (A.s + A.a);
  {
    // first round: group g = lane; low = its tb contiguous bits, high = the tile bits above the round
    const uint32_t g = threadIdx.x, low = g & ((1u << A.tb) - 1), high = g >> A.tb;
    const uint32_t base = (high << (A.tb + 3)) | low;
    const uint32_t a0 = hi_base + ((high << 3) << A.s) + lo0 + low, lo = lo0 + low;
    gl_t v[8], t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = src[a0 + ((uint32_t)j << A.s)];
    constexpr bool FOLD1 = RM == 0 && RL != 0;  // the first round is the one before the last
    if constexpr (FOLD1) {
      const gl_t *tp = A.ftw + ((size_t)high << A.s) + lo;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        t[e] = *tp;
        tp += (size_t)1 << (A.s + RL);
      }
      static_for<0, 8>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        v[j] = gl_mul(v[j], t[brev_c(j, 3)]);
      });
    } else {
      const gl_t *tp = A.ptw + A.tw_off[0] + lo;
#pragma unroll
      for (int e = 1; e < 8; e++) {
        t[e] = *tp;
        tp += (size_t)1 << A.s;
      }
      static_for<1, 8>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        v[j] = gl_mul(v[j], t[brev_c(j, 3)]);
      });
    }
    dft_regs<3, 1, INV>(v);
    if constexpr (RL == 0) {
      if (A.post != 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = gl_mul(v[j], A.post);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) dst[a0 + ((uint32_t)j << A.s)] = v[j];
      return;
    } else {
      const uint32_t l0 = pidx(base);
#pragma unroll
      for (int j = 0; j < 8; j++) lds[l0 ^ pidx((uint32_t)j << A.tb)] = v[j];
    }
  }
  if constexpr (RL != 0) {
    __syncthreads();
    if constexpr (RM != 0) {
      round_folded<INV, RM>(lds, A, A.tb + 3, RL, lo0, A.ftw);
      __syncthreads();
    }
    round_to_global<INV, RL, (RM != 0 ? RM : 3)>(lds, dst, A, hi_base, lo0);
  }
}

// ---- the mirror images for the inverse (DIF) transforms: values -> coefficients -------------------------------------------------
// A DIF round multiplies its OUTPUTS: after the round on [beta, beta + r) output j takes w^-(lo brev(j)), lo = the element's bits
// below beta.  For the pass's FIRST (top) round lo = lo' + 2^s0' k with k the INPUT index of the round below: w^-(2^s0' k brev(j)) is a
// power of w_64 -- a shift, k wave-uniform -- applied at once, and w^-(lo' brev(j)) is constant over the group of the round below
// (fixed lo', j = its `high`), so that round's output twiddles carry it (the same folded table, inverse root).  First round from
// global memory, last round to it.
template <bool INV, int RL, int RP, int SW = 0>
__device__ __forceinline__ void round_from_global(const gl_t *src, gl_t *lds, const PassArgs &A, uint32_t hi_base, uint32_t lo0) {
  constexpr int R = 1 << RL;
  constexpr uint32_t beta0 = 12 - RL;
  auto sw = [](uint32_t e) { return SW ? pidx2(e) : pidx(e); };
#pragma unroll
  for (uint32_t it = 0; it < ((1u << beta0) / NTT_THREADS); it++) {
    const uint32_t g = threadIdx.x + it * NTT_THREADS;
    const uint32_t a0 = hi_base + ((g >> A.tb) << A.s) + lo0 + (g & ((1u << A.tb) - 1));
    gl_t v[R];
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = src[a0 + ((uint32_t)j << (beta0 - A.tb + A.s))];
    dft_regs<RL, 0, INV>(v);
    shift_twiddles_k<INV, RL, RP>(v, (g >> (beta0 - RP)) & ((1u << RP) - 1));
    const uint32_t l0 = sw(g);
#pragma unroll
    for (int j = 0; j < R; j++) lds[l0 ^ sw((uint32_t)j << beta0)] = v[j];
  }
}
// DIF round in LDS whose OUTPUT twiddles are the folded table (its own general twiddle x the top round's deferred part)
template <bool INV, int LOGR>
__device__ __forceinline__ void round_folded_dif(gl_t *lds, const PassArgs &A, uint32_t beta0, uint32_t rl, uint32_t lo0, const gl_t *ftw) {
  constexpr int R = 1 << LOGR;
  const uint32_t ngroups = 1u << (12 - LOGR);
  const uint32_t s0 = beta0 - A.tb + A.s;
  uint32_t lj[R];
#pragma unroll
  for (int j = 0; j < R; j++) lj[j] = pidx((uint32_t)j << beta0);
  const size_t estride = (size_t)1 << (s0 + rl);
  for (uint32_t g = threadIdx.x; g < ngroups; g += NTT_THREADS) {
    const uint32_t low = g & ((1u << beta0) - 1), high = g >> beta0;
    const uint32_t base = (high << (beta0 + LOGR)) | low;
    const uint32_t lo = ((low >> A.tb) << A.s) + lo0 + (low & ((1u << A.tb) - 1));
    gl_t v[R], t[R];
    uint32_t li[R];
    const uint32_t l0 = pidx(base);
#pragma unroll
    for (int j = 0; j < R; j++) li[j] = l0 ^ lj[j];
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = lds[li[j]];
    const gl_t *tp = ftw + ((size_t)high << s0) + lo;
#pragma unroll
    for (int e = 0; e < R; e++) {
      t[e] = *tp;
      tp += estride;
    }
    dft_regs<LOGR, 0, INV>(v);
    static_for<0, R>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      v[j] = gl_mul(v[j], t[brev_c(j, LOGR)]);
    });
#pragma unroll
    for (int j = 0; j < R; j++) lds[li[j]] = v[j];
  }
}
// A strided DIF pass of a = 3 + RM + RL layers: top round (RL layers) global -> registers -> LDS with shift twiddles, [middle round
// in LDS with the folded table,] bottom round (3 layers) LDS -> registers -> global.  RL = 0: a = 3, one round, no LDS.
template <bool INV, int RM, int RL>
__global__ __launch_bounds__(NTT_THREADS, NTT_MIN_WAVES) void ntt_dif_strided_kernel(PassArgs A) {
  extern __shared__ gl_t lds[];
  uint32_t tile, col, coset;
  if (!pass_unit(A, tile, col, coset)) return;
  const size_t n = (size_t)1 << A.d;
  const gl_t *src = A.src + ((size_t)(A.src_single ? 0 : coset) * A.cols + col) * n;
  gl_t *dst = A.dst + ((size_t)coset * A.cols + col) * n;
  const uint32_t runs = 1u << (A.s - A.tb);
  const uint32_t hi = tile / runs, lo0 = (tile % runs) << A.tb;
  const uint32_t hi_base = hi << (A.s + A.a);
  if constexpr (RL != 0) {
    round_from_global<INV, RL, (RM != 0 ? RM : 3)>(src, lds, A, hi_base, lo0);
    __syncthreads();
    if constexpr (RM != 0) {
      round_folded_dif<INV, RM>(lds, A, A.tb + 3, RL, lo0, A.ftw);
      __syncthreads();
    }
  }
  {
    // bottom round: group g = lane; low = its tb contiguous bits, high = the tile bits above the round
    const uint32_t g = threadIdx.x, low = g & ((1u << A.tb) - 1), high = g >> A.tb;
    const uint32_t base = (high << (A.tb + 3)) | low;
    const uint32_t a0 = hi_base + ((high << 3) << A.s) + lo0 + low, lo = lo0 + low;
    gl_t v[8], t[8];
    if constexpr (RL == 0) {
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = src[a0 + ((uint32_t)j << A.s)];
    } else {
      const uint32_t l0 = pidx(base);
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = lds[l0 ^ pidx((uint32_t)j << A.tb)];
    }
    constexpr bool FOLD1 = RM == 0 && RL != 0;  // the bottom round is the one below the top
    if constexpr (FOLD1) {
      const gl_t *tp = A.ftw + ((size_t)high << A.s) + lo;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        t[e] = *tp;
        tp += (size_t)1 << (A.s + RL);
      }
    } else {
      const gl_t *tp = A.ptw + A.tw_off[0] + lo;
#pragma unroll
      for (int e = 1; e < 8; e++) {
        t[e] = *tp;
        tp += (size_t)1 << A.s;
      }
    }
    dft_regs<3, 0, INV>(v);
    static_for<(FOLD1 ? 0 : 1), 8>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      v[j] = gl_mul(v[j], t[brev_c(j, 3)]);
    });
    if (A.post != 1) {
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = gl_mul(v[j], A.post);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) dst[a0 + ((uint32_t)j << A.s)] = v[j];
  }
}
// The 12-layer LAST pass of a DIF transform (contiguous tiles), the mirror of ntt_dit_head2_kernel: round on bits 9-11 from a
// coalesced load with shift twiddles -> LDS | rounds on bits 6-8 (folded table) and 3-5 (shifts) with wave = element bits 0-2, no
// barrier between them | round on bits 0-2 in registers, output scale, 64 B per lane to global.
template <bool INV>
__global__ __launch_bounds__(NTT_THREADS, NTT_MIN_WAVES) void ntt_dif_tail2_kernel(PassArgs A) {
  extern __shared__ gl_t lds[];
  uint32_t tile, col, coset;
  if (!pass_unit(A, tile, col, coset)) return;
  const size_t n = (size_t)1 << A.d;
  const gl_t *src = A.src + ((size_t)(A.src_single ? 0 : coset) * A.cols + col) * n;
  gl_t *dst = A.dst + ((size_t)coset * A.cols + col) * n;
  const uint32_t tbase = tile << 12, t = threadIdx.x, w = t >> 6, l = t & 63u;
  round_from_global<INV, 3, 3, 1>(src, lds, A, tbase, 0);
  __syncthreads();
  gl_t v[8];
  {  // bits 6-8: element = w + 8 a + 64 i + 512 J, lane = a + 8 J; output twiddles F2[e * 512 + t]
    const uint32_t p0 = pidx2(w | ((l & 7u) << 3) | ((l >> 3) << 9));
    gl_t tw[8];
    const gl_t *tp = A.ftw2 + t;
#pragma unroll
    for (int e = 0; e < 8; e++) tw[e] = tp[e * NTT_THREADS];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = lds[p0 ^ pidx2((uint32_t)i << 6)];
    dft_regs<3, 0, INV>(v);
    static_for<0, 8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      v[i] = gl_mul(v[i], tw[brev_c(i, 3)]);
    });
#pragma unroll
    for (int i = 0; i < 8; i++) lds[p0 ^ pidx2((uint32_t)i << 6)] = v[i];
  }
  asm volatile("" ::: "memory");  // the same wave, the same 512 elements (bits 0-2 = w)
  {  // bits 3-5: element = w + 8 j + 64 l; output j takes w_64^-(w brev(j))
    const uint32_t p0 = pidx2(w | (l << 6));
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = lds[p0 ^ pidx2((uint32_t)j << 3)];
    dft_regs<3, 0, INV>(v);
    shift_twiddles_k<INV, 3, 3>(v, w);
#pragma unroll
    for (int j = 0; j < 8; j++) lds[p0 ^ pidx2((uint32_t)j << 3)] = v[j];
  }
  __syncthreads();
  {
    const uint32_t p0 = pidx2(8 * t);
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = lds[p0 ^ (uint32_t)k];
  }
  dft_regs<3, 0, INV>(v);  // bits 0-2: no twiddles
  if (A.post != 1) {
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = gl_mul(v[k], A.post);
  }
  ulonglong2 *d2 = reinterpret_cast<ulonglong2 *>(dst + tbase + 8 * t);
#pragma unroll
  for (int i = 0; i < 4; i++) d2[i] = make_ulonglong2(v[2 * i], v[2 * i + 1]);
}

static inline size_t lds_bytes(uint32_t TB) { return std::max<size_t>((size_t)1 << TB, 256) * sizeof(gl_t); }  // no padding: pidx() is a permutation of every 256-element block

// ---- plan -----------------------------------------------------------------------
// packed twiddles of one round: T[(e-1) * M + lo] = root_n^((lo * e) << (d - s0 - r)), e in [1, 2^r)
__global__ void round_table_kernel(gl_t *out, gl_t root_n, uint32_t d, uint32_t s0, uint32_t r) {
  const uint32_t M = 1u << s0, cnt = ((1u << r) - 1) * M;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const uint32_t e = i / M + 1, lo = i % M;
  out[i] = gl_pow(root_n, ((uint64_t)lo * e) << (d - s0 - r));
}

// folded twiddles of a round of r layers at s0 whose inputs also take the general part of the next (last, rl layers) round's:
// F[(e * 2^rl + J) * M + lo] = w_{2^(s0+r)}^(lo e) * w_{2^(s0+r+rl)}^(lo brev_rl(J)), M = 2^s0, e < 2^r, J < 2^rl
__global__ void fold_table_kernel(gl_t *out, gl_t root_n, uint32_t d, uint32_t s0, uint32_t r, uint32_t rl) {
  const uint32_t M = 1u << s0, cnt = (1u << (r + rl)) * M;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const uint32_t lo = i % M, J = (i / M) & ((1u << rl) - 1), e = i >> (s0 + rl);
  const uint64_t ex = (((uint64_t)lo * e) << (d - s0 - r)) + (((uint64_t)lo * bitrev32(J, rl)) << (d - s0 - r - rl));
  out[i] = gl_pow(root_n, ex);
}
// the head pass's folded table (s0 = 6, r = 3, rl = 3) in the lane order of ntt_dit_head2_kernel: F2[e * 512 + t] for lane t whose
// group is lo = (t >> 6) + 8 (t & 7), J = (t & 63) >> 3
__global__ void head_fold_table_kernel(gl_t *out, gl_t root_n, uint32_t d) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 8u * 512u) return;
  const uint32_t e = i >> 9, t = i & 511u, lo = (t >> 6) + 8u * (t & 7u), J = (t & 63u) >> 3;
  const uint64_t ex = (((uint64_t)lo * e) << (d - 9)) + (((uint64_t)lo * bitrev32(J, 3)) << (d - 12));
  out[i] = gl_pow(root_n, ex);
}
// P2GPU_NTT_HEAD=1: the head pass through ntt_dit_head_kernel (first two rounds in LDS with table twiddles) instead of ntt_dit_head2_kernel
static bool head2_on() {
  static const bool on = [] { const char *e = getenv("P2GPU_NTT_HEAD"); return !(e && *e == '1'); }();
  return on;
}
// P2GPU_NTT_DIRECT=0: every pass through ntt_pass_kernel (A/B measurements, and the reference the direct kernels are tested against)
static bool direct_on() {
  static const bool on = [] { const char *e = getenv("P2GPU_NTT_DIRECT"); return !(e && *e == '0'); }();
  return on;
}

static void split_rounds(uint32_t a, std::vector<uint32_t> &r) {
  // rounds of LMAX layers (LMAX = log2 NTT_PER), remainder fixed up without ever using a lone 1 when avoidable
  const uint32_t LMAX = NTT_PER == 16 ? 4 : 3;
  r.clear();
  uint32_t q = a / LMAX, rem = a % LMAX;
  if (rem == 1 && q >= 1 && LMAX >= 3) {  // L + 1 -> (L - 1) + 2
    for (uint32_t i = 0; i + 1 < q; i++) r.push_back(LMAX);
    r.push_back(LMAX - 1);
    r.push_back(2);
  } else {
    for (uint32_t i = 0; i < q; i++) r.push_back(LMAX);
    if (rem) r.push_back(rem);
  }
}

NttPlan *ntt_plan_create(hipStream_t st, uint32_t d, int dit, bool inverse) {
  NttPlan *p = new NttPlan();
  p->d = d;
  p->dit = dit;
  p->inverse = inverse;
  const uint32_t TBMAX = NTT_TILE_BITS;
  const uint32_t b = d < TBMAX ? d : TBMAX;
  struct P { uint32_t s, a, tb; };
  std::vector<P> strided;
  {
    uint32_t rem = d - b, s = b;
    uint32_t npass = rem == 0 ? 0 : (rem <= 9 ? 1 : 2);
    for (uint32_t i = 0; i < npass; i++) {
      uint32_t a = (npass == 1) ? rem : (i == 0 ? (rem + 1) / 2 : rem / 2);
      uint32_t tb = TBMAX - a;
      if (tb > s) tb = s;
      strided.push_back(P{s, a, tb});
      s += a;
    }
  }
  std::vector<P> passes;
  if (dit) {
    passes.push_back(P{0, b, 0});
    for (auto &x : strided) passes.push_back(x);
  } else {
    for (size_t i = strided.size(); i-- > 0;) passes.push_back(strided[i]);
    passes.push_back(P{0, b, 0});
  }
  // table layout
  size_t total = 0;
  std::vector<uint32_t> rr;
  for (auto &ps : passes) {
    NttPass np;
    np.s = ps.s; np.a = ps.a; np.tb = ps.tb;
    split_rounds(ps.a, rr);
    np.nrounds = (uint32_t)rr.size();
    if (np.nrounds > (uint32_t)MAX_ROUNDS) {  // (cannot happen for tiles of <= 2^16 elements)
      delete p;
      return nullptr;
    }
    uint32_t beta = ps.tb;
    for (uint32_t i = 0; i < np.nrounds; i++) {
      np.r[i] = rr[i];
      np.tw_off[i] = (uint32_t)total;
      uint32_t s0 = beta - ps.tb + ps.s;
      if (s0 > 0) total += (size_t)((1u << rr[i]) - 1) << s0;
      beta += rr[i];
    }
    // the direct form (DIT, full tiles): the 12-layer head [3,3,3,3], strided passes [3], [3,2], [3,3], [3,2,2], [3,3,2], [3,3,3]
    if (((dit && !inverse) || (!dit && inverse)) && NTT_TILE_BITS == 12 && NTT_PER == 8 && ps.a + ps.tb == 12 && np.r[0] == 3 &&
        (ps.s == 0 ? (ps.a == 12) : (ps.a == 3 || (ps.a >= 5 && ps.a <= 9)))) {
      np.direct = true;
      if (np.nrounds >= 2) {
        const uint32_t q = np.nrounds - 2, rl = np.r[np.nrounds - 1];  // the round before the last
        uint32_t bq = ps.tb;
        for (uint32_t i = 0; i < q; i++) bq += np.r[i];
        np.ftw_off = (uint32_t)total;
        total += (size_t)1 << (np.r[q] + rl + (bq - ps.tb + ps.s));
        if (ps.s == 0) {  // the head pass: the same 4096 values once more, in ntt_dit_head2_kernel's lane order
          np.ftw2_off = (uint32_t)total;
          total += 4096;
        }
      }
    }
    p->passes.push_back(np);
  }
  p->table_len = total;
  if (total >> 32) {  // tw_off / ftw_off / ftw2_off are 32-bit word offsets (d = 24 with its folded tables: ~2^25 words)
    delete p;
    return nullptr;
  }
  if (total) {
    if (hipMalloc((void **)&p->ptw, total * sizeof(gl_t)) != hipSuccess) {
      delete p;
      return nullptr;
    }
    gl_t root = gl_root(d);
    if (inverse) root = gl_inv(root);
    for (auto &np : p->passes) {
      uint32_t beta = np.tb;
      for (uint32_t i = 0; i < np.nrounds; i++) {
        uint32_t s0 = beta - np.tb + np.s;
        if (s0 > 0) {
          uint32_t cnt = ((1u << np.r[i]) - 1) << s0;
          hipLaunchKernelGGL(round_table_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, p->ptw + np.tw_off[i], root, d,
                             s0, np.r[i]);
        }
        beta += np.r[i];
      }
      if (np.direct && np.nrounds >= 2) {
        const uint32_t q = np.nrounds - 2, rl = np.r[np.nrounds - 1];
        uint32_t bq = np.tb;
        for (uint32_t i = 0; i < q; i++) bq += np.r[i];
        const uint32_t s0 = bq - np.tb + np.s, cnt = 1u << (np.r[q] + rl + s0);
        hipLaunchKernelGGL(fold_table_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, p->ptw + np.ftw_off, root, d, s0, np.r[q], rl);
        if (np.s == 0) hipLaunchKernelGGL(head_fold_table_kernel, dim3(16), dim3(256), 0, st, p->ptw + np.ftw2_off, root, d);
      }
    }
  }
  return p;
}
void ntt_plan_destroy(NttPlan *p) {
  if (!p) return;
  if (p->ptw) (void)hipFree(p->ptw);
  delete p;
}

// The whole transform of the structured columns of a batch (ColHints): class 0 -> zeros (not even stored when the
// column's clean mark says dst holds them already), class 1 -> val[c] * basis.  One lane per 2 elements, coalesced;
// blocks of dense columns return at once.  By linearity these are the canonical values the butterfly network
// would produce.
__global__ __launch_bounds__(256) void structured_fill_kernel(gl_t *dst, uint32_t d, uint32_t col_stride, uint32_t cosets,
                                                              ColHints h, uint32_t coset_first, uint32_t coset_stride) {
  const uint32_t col = blockIdx.y;
  const uint32_t cls = h.cls[col];
  if (cls == 2u) return;
  if (cls < 2u && col >= h.virt_first) return;  // never read from memory: the leaf hash and the query gather recompute it
  if (cls == 0u && h.clean != nullptr && h.clean[col] != 0u) return;
  const size_t n = (size_t)1 << d;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  if (cls == 3u) {  // zero outside the special rows: the same linear combination of their unit columns' transforms
    gl_t v[MAX_SPARSE_ROWS];
#pragma unroll
    for (uint32_t s = 0; s < MAX_SPARSE_ROWS; s++) v[s] = s < h.nrows ? h.val[(size_t)s * h.val_stride + col] : (gl_t)0;
    for (uint32_t coset = 0; coset < cosets; coset++) {
      gl_t *out = dst + ((size_t)coset * col_stride + col) * n;
      const gl_t *bs = h.basis + (h.basis_per_coset ? (size_t)(coset_first + coset * coset_stride) * n : 0);
      for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        Acc160 acc;
        acc.clear();
#pragma unroll
        for (uint32_t s = 0; s < MAX_SPARSE_ROWS; s++)
          if (s < h.nrows) acc.mac(v[s], bs[(size_t)s * h.basis_stride + i]);
        out[i] = acc.value();
      }
    }
    return;
  }
  const gl_t v = cls == 1u ? h.val[col] : 0;
  for (uint32_t coset = 0; coset < cosets; coset++) {  // one block covers its slice of every coset: 8x fewer blocks to retire
    gl_t *out = dst + ((size_t)coset * col_stride + col) * n;
    const gl_t *bs = h.basis + (h.basis_per_coset ? (size_t)(coset_first + coset * coset_stride) * n : 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) out[i] = cls == 1u ? gl_mul(v, bs[i]) : (gl_t)0;
  }
}

static void ntt_passes(hipStream_t st, const NttPlan *plan, const gl_t *src, gl_t *dst, uint32_t cols, uint32_t cosets,
                       const gl_t *scale, gl_t post, bool src_per_coset, CosetMap cm, uint32_t stride_all, const uint32_t *colnz,
                       uint32_t dense = 0);
void ntt_batch(hipStream_t st, const NttPlan *plan, const gl_t *src, gl_t *dst, uint32_t cols, uint32_t cosets,
               const gl_t *scale, gl_t post, bool src_per_coset, CosetMap cm, uint32_t stride_cols, const ColHints *hints) {
  if (cols == 0) return;
  if (hints) {
    const size_t n = (size_t)1 << plan->d;
    const uint32_t bx = (uint32_t)std::max<size_t>(1, n / (256 * 4));
    // (profile bytes: the structured columns that are actually stored -- those below virt_first; with no dense count from an earlier
    // proof every column is counted)
    const uint32_t stored = hints->virt_first < cols ? hints->virt_first : cols;
    const uint32_t filled = hints->dense_hint && hints->dense_hint <= cols ? (stored > hints->dense_hint ? stored - hints->dense_hint : 0) : cols;
    ProfScope ps("structured_fill_kernel", 8.0 * filled * cosets * (double)n);
    // Columns at or above virt_first are stored only when they are dense (the transform kernels) or of class 3, and class 3 exists
    // only with more than one special row (public-input circuits): otherwise the grid stops at virt_first -- 80 instead of 234
    // column rows of blocks for a circuit without ECC gates, most of which would only look at their class and leave
    const uint32_t fill_cols = hints->nrows > 1 ? cols : std::min(cols, hints->virt_first);
    if (fill_cols)
      hipLaunchKernelGGL(structured_fill_kernel, dim3(bx, fill_cols), dim3(256), 0, st, dst, plan->d,
                         stride_cols ? stride_cols : cols, cosets, *hints, cm.first, cm.stride);
    if (hints->fill_only) return;
  }
  const uint32_t d = plan->d;
  const size_t np = plan->passes.size();
  // A transform of several passes leaves its intermediate in dst between them.  With every column in one launch per pass
  // that intermediate (8n words per column for an LDE) is long gone from the 256 MB memory-side cache when the next pass
  // reads it; in groups of columns whose dst fits, the second pass finds it there and overwrites it in place: the
  // intermediate's round trip never reaches HBM (P2GPU_NTT_GROUP_MB: group size, 0 = one launch per pass as before)
  static const size_t group_bytes = [] {
    const char *e = getenv("P2GPU_NTT_GROUP_MB");
    return (size_t)(e ? atoi(e) : 0) << 20;
  }();
  const uint32_t stride_all = stride_cols ? stride_cols : cols;
  const uint32_t *colnz = hints ? hints->cls : nullptr;
  if (np >= 2 && group_bytes) {
    const size_t per_col = (size_t)cosets * 8 << d;
    const uint32_t gcols = (uint32_t)std::max<size_t>(1, group_bytes / per_col);
    if (gcols < cols) {
      for (uint32_t c0 = 0; c0 < cols; c0 += gcols)
        ntt_passes(st, plan, src + ((size_t)c0 << d), dst + ((size_t)c0 << d), std::min(gcols, cols - c0), cosets, scale, post,
                   src_per_coset, cm, stride_all, colnz ? colnz + c0 : nullptr);
      return;
    }
  }
  ntt_passes(st, plan, src, dst, cols, cosets, scale, post, src_per_coset, cm, stride_all, colnz,
             hints && hints->dense_hint && hints->dense_hint <= cols ? hints->dense_hint : 0);
}
static void fill_pass_args(PassArgs &A, const NttPlan *plan, size_t i, const gl_t *src, gl_t *dst, uint32_t cols, uint32_t cosets,
                           const gl_t *scale, gl_t post, bool src_per_coset, CosetMap cm, uint32_t stride_all, const uint32_t *colnz) {
  const size_t np = plan->passes.size();
  const NttPass &ps = plan->passes[i];
  A.src = (i == 0) ? src : dst;
  A.dst = dst;
  A.ptw = plan->ptw;
  A.scale = (i == 0 && plan->dit) ? scale : nullptr;
  A.post = (i == np - 1) ? post : 1;
  A.d = plan->d;
  A.s = ps.s; A.a = ps.a; A.tb = ps.tb;
  A.cols = stride_all;  // column stride between cosets (a launch may cover a column chunk)
  A.src_single = (i == 0 && !src_per_coset) ? 1 : 0;
  A.coset_first = cm.first;
  A.coset_stride = cm.stride;
  A.nrounds = ps.nrounds;
  A.colnz = colnz;
  A.ftw = ps.direct ? plan->ptw + ps.ftw_off : nullptr;
  A.ftw2 = ps.direct && ps.s == 0 ? plan->ptw + ps.ftw2_off : nullptr;
  for (int k = 0; k < MAX_ROUNDS; k++) { A.r[k] = ps.r[k]; A.tw_off[k] = ps.tw_off[k]; }
  A.tiles = 1u << (plan->d - (A.a + A.tb));
  A.cols_grid = cols; A.cosets = cosets;
}
// same spelling as rocprofv3's demangled kernel names, so the bench line and profiles/ agree
static const char *pass_kernel_name(const NttPlan *plan, bool full) {
  // string literals: the profiler keeps the pointer
  static const char *const names[2][2][2] = {
      {{"ntt_pass_kernel<0, false, 0>", "ntt_pass_kernel<0, true, 0>"}, {"ntt_pass_kernel<1, false, 0>", "ntt_pass_kernel<1, true, 0>"}},
      {{"ntt_pass_kernel<0, false, 12>", "ntt_pass_kernel<0, true, 12>"}, {"ntt_pass_kernel<1, false, 12>", "ntt_pass_kernel<1, true, 12>"}}};
  // (the "12" in the names is NTT_TILE_BITS of the default build)
  return names[full ? 1 : 0][plan->dit ? 1 : 0][plan->inverse ? 1 : 0];
}
// (Both passes of a two-pass transform in ONE launch -- per-XCD work queues, the intermediate handed over through L2 -- was built in
// round 4, bit-exact, and measured slower and heavier on HBM than two launches: profiles/r04_lde_fused.md; removed in round 5,
// `git show 33c653e:acvm-backend-plonky2_amd/csrc/ntt.hip` has it.)
static void ntt_passes(hipStream_t st, const NttPlan *plan, const gl_t *src, gl_t *dst, uint32_t cols, uint32_t cosets,
                       const gl_t *scale, gl_t post, bool src_per_coset, CosetMap cm, uint32_t stride_all, const uint32_t *colnz,
                       uint32_t dense) {
  const uint32_t d = plan->d;
  const size_t np = plan->passes.size();
  for (size_t i = 0; i < np; i++) {
    PassArgs A;
    fill_pass_args(A, plan, i, src, dst, cols, cosets, scale, post, src_per_coset, cm, stride_all, colnz);
    const uint32_t TB = A.a + A.tb;
    dim3 grid((((A.tiles * cols + 7u) / 8u) * 8u) * cosets);
    const uint32_t threads = TB >= 8 ? 256 : 64;
    // expected HBM bytes: every output element is written once; the input is read once per element, except
    // that the cosets of one (tile, column) share their source through one XCD's L2 (first LDE pass)
    // (structured columns are skipped by the kernels: `dense`, when the caller knows it, is what the launch really transforms)
    const double bytes = 8.0 * (double)(dense ? dense : cols) * ((size_t)1 << d) * (cosets + (A.src_single ? 1.0 : (double)cosets));
    const size_t lb = lds_bytes(TB);
    if (plan->passes[i].direct && direct_on()) {
      const NttPass &ps = plan->passes[i];
      const uint32_t rm = ps.nrounds == 3 ? ps.r[1] : 0, rl = ps.nrounds >= 2 ? ps.r[ps.nrounds - 1] : 0;
      // (profile names = rocprofv3's demangled symbols)
      if (!plan->dit) {
        if (ps.s == 0) {
          ProfScope psd("ntt_dif_tail2_kernel<true>", bytes);
          hipLaunchKernelGGL((ntt_dif_tail2_kernel<true>), grid, dim3(NTT_THREADS), lb, st, A);
        } else {
#define P2_STRIDED_DIF(RM, RL)                                                                                    \
  do {                                                                                                            \
    ProfScope psd("ntt_dif_strided_kernel<true, " #RM ", " #RL ">", bytes);                                       \
    hipLaunchKernelGGL((ntt_dif_strided_kernel<true, RM, RL>), grid, dim3(NTT_THREADS), (RL) ? lb : 0, st, A);     \
  } while (0)
          if (rm == 0 && rl == 0) P2_STRIDED_DIF(0, 0);
          else if (rm == 0 && rl == 2) P2_STRIDED_DIF(0, 2);
          else if (rm == 0 && rl == 3) P2_STRIDED_DIF(0, 3);
          else if (rm == 2 && rl == 2) P2_STRIDED_DIF(2, 2);
          else if (rm == 3 && rl == 2) P2_STRIDED_DIF(3, 2);
          else P2_STRIDED_DIF(3, 3);
#undef P2_STRIDED_DIF
        }
      } else if (ps.s == 0 && head2_on()) {
        ProfScope psd("ntt_dit_head2_kernel<false>", bytes);
        hipLaunchKernelGGL((ntt_dit_head2_kernel<false>), grid, dim3(NTT_THREADS), lb, st, A);
      } else if (ps.s == 0) {
        ProfScope psd("ntt_dit_head_kernel<false>", bytes);
        hipLaunchKernelGGL((ntt_dit_head_kernel<false>), grid, dim3(NTT_THREADS), lb, st, A);
      } else {
#define P2_STRIDED(RM, RL)                                                                                        \
  do {                                                                                                            \
    ProfScope psd("ntt_dit_strided_kernel<false, " #RM ", " #RL ">", bytes);                                      \
    hipLaunchKernelGGL((ntt_dit_strided_kernel<false, RM, RL>), grid, dim3(NTT_THREADS), (RL) ? lb : 0, st, A);    \
  } while (0)
        if (rm == 0 && rl == 0) P2_STRIDED(0, 0);
        else if (rm == 0 && rl == 2) P2_STRIDED(0, 2);
        else if (rm == 0 && rl == 3) P2_STRIDED(0, 3);
        else if (rm == 2 && rl == 2) P2_STRIDED(2, 2);
        else if (rm == 3 && rl == 2) P2_STRIDED(3, 2);
        else P2_STRIDED(3, 3);
#undef P2_STRIDED
      }
      continue;
    }
    ProfScope psx(pass_kernel_name(plan, TB == NTT_TILE_BITS), bytes);
#define P2_LAUNCH(DITV, INVV)                                                                                \
  do {                                                                                                       \
    if (TB == NTT_TILE_BITS) hipLaunchKernelGGL((ntt_pass_kernel<DITV, INVV, NTT_TILE_BITS>), grid, dim3(NTT_THREADS), lb, st, A);           \
    else hipLaunchKernelGGL((ntt_pass_kernel<DITV, INVV, 0>), grid, dim3(threads), lb, st, A);               \
  } while (0)
    if (plan->dit) {
      if (plan->inverse) P2_LAUNCH(1, true);
      else P2_LAUNCH(1, false);
    } else {
      if (plan->inverse) P2_LAUNCH(0, true);
      else P2_LAUNCH(0, false);
    }
#undef P2_LAUNCH
  }
}

// ---- zero-column flags --------------------------------------------------------------
// Class of every column of vals [cols][n] (flags zeroed by the launcher): 0 = zero in every row; 1 = zero in every
// row but `sparse_row`, whose value goes to scalar[c]; 2 = anything else.  plonky2's build() hangs a random value
// on every unused wire of the PublicInputGate row (circuit_builder.rs randomize_unused_pi_wires; visible in the
// reference's own proofs, tests/golden/reference_proofs.py), so in a real witness the wires no gate uses are
// class 1 with that row, not class 0.  sparse_row = UINT32_MAX: no such row.
__global__ __launch_bounds__(256) void column_nonzero_kernel(const gl_t *__restrict__ vals, uint32_t d, SparseRows rows,
                                                             uint32_t *flags, gl_t *scalar, uint32_t sstride) {
  const size_t n = (size_t)1 << d;
  const gl_t *p = vals + (size_t)blockIdx.y * n;
  uint64_t acc = 0;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  const uint32_t r0 = rows.row[0], r1 = rows.row[1], r2 = rows.row[2], r3 = rows.row[3];  // UINT32_MAX: never matches
#pragma unroll 8
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const gl_t v = p[i];
    acc |= (i == r0 || i == r1 || i == r2 || i == r3) ? (gl_t)0 : v;  // a select, not a branch: the loads stay batched
  }
  // dense: a plain store (every wave of a dense column would otherwise hammer one address with atomics); the other
  // classes are told apart from the special rows' values by column_class_kernel, launched behind this one
  if (__any(acc != 0) && (threadIdx.x & 63) == 0) flags[blockIdx.y] = 2u;
  if (blockIdx.x == 0 && threadIdx.x < rows.count) scalar[(size_t)threadIdx.x * sstride + blockIdx.y] = p[rows.row[threadIdx.x]];
}
__global__ void column_class_kernel(uint32_t *flags, const gl_t *scalar, uint32_t sstride, uint32_t nrows, uint32_t cols) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols || flags[c] == 2u) return;
  bool more = false;
  for (uint32_t s = 1; s < nrows; s++) more |= scalar[(size_t)s * sstride + c] != 0;
  flags[c] = more ? 3u : ((nrows && scalar[c] != 0) ? 1u : 0u);
}
// "clean" bookkeeping of the buffers a column's transforms are written to (coefficients + LDE), so that the zeros of
// an unused wire are stored once per handle instead of once per proof.  Both steps are stream-ordered around the
// transforms: BEFORE them a non-zero column loses its clean mark (its buffers are about to be overwritten), AFTER them
// a zero column gains it (its buffers now hold zeros).  Anything that aborts in between leaves marks only cleared.
__global__ void column_clean_kernel(const uint32_t *nz, uint32_t cols, uint32_t *clean, int after) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  if (after) {
    if (nz[c] == 0) clean[c] = 1;
  } else {
    if (nz[c] != 0) clean[c] = 0;
  }
}
void column_clean_update(hipStream_t st, const uint32_t *nz, uint32_t cols, uint32_t *clean, bool after) {
  if (!cols) return;
  hipLaunchKernelGGL(column_clean_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, nz, cols, clean, after ? 1 : 0);
}
void column_flags(hipStream_t st, const gl_t *vals, uint32_t cols, uint32_t d, const SparseRows &rows, uint32_t *flags,
                  gl_t *scalar, uint32_t sstride) {
  if (!cols) return;
  (void)hipMemsetAsync(flags, 0, sizeof(uint32_t) * cols, st);
  const size_t n = (size_t)1 << d;
  const uint32_t bx = (uint32_t)std::max<size_t>(1, n / (256 * 8));
  ProfScope ps("column_nonzero_kernel", 8.0 * cols * (double)n);
  hipLaunchKernelGGL(column_nonzero_kernel, dim3(bx, cols), dim3(256), 0, st, vals, d, rows, flags, scalar, sstride);
  hipLaunchKernelGGL(column_class_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, flags, scalar, sstride, rows.count, cols);
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

// ---- self-test of the field primitives (stage-level test operator p2gpu_field_selftest) -------------------
// a[i], b[i]: arbitrary u64.  Every carry-chain form of gl.hpp / mul_pow2 against the portable code, which is
// what the host and the oracle run: bad[0] canon, [1] add, [2] sub, [3] reduce128, [4] mul, [5] mul_add,
// [6] mul_pow2<1..95>, [7] Acc160, [8..13] the congruent-word (non-canonical) forms, [14..15] unused.
__global__ void field_selftest_kernel(const uint64_t *a, const uint64_t *b, uint32_t n, unsigned long long *bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t x = a[i], y = b[i];
  const gl_t xc = gl_canon_c(x), yc = gl_canon_c(y);
  if (gl_canon(x) != xc) atomicAdd(&bad[0], 1ULL);
  if (gl_add(xc, yc) != gl_add_c(xc, yc)) atomicAdd(&bad[1], 1ULL);
  if (gl_sub(xc, yc) != gl_sub_c(xc, yc)) atomicAdd(&bad[2], 1ULL);
  if (gl_reduce128(x, y) != gl_reduce128_c(x, y)) atomicAdd(&bad[3], 1ULL);
  const uint64_t plo = xc * yc, phi = __umul64hi(xc, yc);
  const gl_t prod = gl_reduce128_c(plo, phi);
  if (gl_mul(xc, yc) != prod) atomicAdd(&bad[4], 1ULL);
  {
    uint64_t lo = plo + xc, hi = phi + (lo < xc);
    if (gl_mul_add(xc, yc, xc) != gl_reduce128_c(lo, hi)) atomicAdd(&bad[5], 1ULL);
  }
  gl_t pw = 1;
  bool ok = true;
  static_for<1, 96>([&](auto ec) {
    constexpr int e = decltype(ec)::value;
    pw = gl_add_c(pw, pw);
    const uint64_t l = xc * pw, h = __umul64hi(xc, pw);
    if (mul_pow2<e>(xc) != gl_reduce128_c(l, h)) ok = false;
  });
  if (!ok) atomicAdd(&bad[6], 1ULL);
  {
    Acc160 acc;
    acc.clear();
    acc.mac(xc, yc);
    acc.mac(yc, yc);
    acc.mac(xc, xc);
    const uint64_t l2 = yc * yc, h2 = __umul64hi(yc, yc), l3 = xc * xc, h3 = __umul64hi(xc, xc);
    const gl_t want = gl_add_c(gl_add_c(prod, gl_reduce128_c(l2, h2)), gl_reduce128_c(l3, h3));
    if (acc.value() != want) atomicAdd(&bad[7], 1ULL);
  }
  // The congruent-word forms (gl.hpp, round 3): operands ANY u64 -- x, y are used raw, so the edge set's words in [p, 2^64)
  // reach every branch -- result some u64 congruent to the canonical portable value.  [8] gl_mul_nc, [9] gl_mul_add_nc (one
  // factor canonical: the product plus the addend stays below 2^128), [10] gl_reduce128_nc, [11] gl_add / [12] gl_sub with a
  // non-canonical FIRST operand (a congruent word comes out; canonical when both operands are), [13] a chain: congruent words fed back into the congruent forms.
  if (gl_canon(gl_mul_nc(x, y)) != prod) atomicAdd(&bad[8], 1ULL);
  {
    uint64_t lo = plo + xc, hi = phi + (lo < xc);
    if (gl_canon(gl_mul_add_nc(x, yc, xc)) != gl_reduce128_c(lo, hi)) atomicAdd(&bad[9], 1ULL);
  }
  if (gl_canon(gl_reduce128_nc(x, y)) != gl_reduce128_c(x, y)) atomicAdd(&bad[10], 1ULL);
  if (gl_canon(gl_add(x, yc)) != gl_add_c(xc, yc) || gl_add(xc, yc) != gl_add_c(xc, yc)) atomicAdd(&bad[11], 1ULL);
  if (gl_canon(gl_sub(x, yc)) != gl_sub_c(xc, yc) || gl_sub(xc, yc) != gl_sub_c(xc, yc)) atomicAdd(&bad[12], 1ULL);
  {
    const uint64_t u = gl_mul_nc(x, y), v = gl_mul_add_nc(y, xc, yc);   // congruent to x y and y x + y
    const uint64_t w = gl_mul_nc(u, v);
    uint64_t l2 = plo + yc, h2 = phi + (l2 < yc);
    const gl_t vv = gl_reduce128_c(l2, h2);
    const uint64_t l3 = prod * vv, h3 = __umul64hi(prod, vv);
    if (gl_canon(w) != gl_reduce128_c(l3, h3)) atomicAdd(&bad[13], 1ULL);
    if (gl_canon(gl_add(w, xc)) != gl_add_c(gl_reduce128_c(l3, h3), xc)) atomicAdd(&bad[13], 1ULL);
    if (gl_canon(gl_sub(w, xc)) != gl_sub_c(gl_reduce128_c(l3, h3), xc)) atomicAdd(&bad[13], 1ULL);
  }
}
void field_selftest(hipStream_t st, const uint64_t *a, const uint64_t *b, uint32_t n, unsigned long long *bad) {
  hipLaunchKernelGGL(field_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, b, n, bad);
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
