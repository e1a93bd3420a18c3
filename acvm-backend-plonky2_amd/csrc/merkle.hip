// merkle.hip -- Keccak-256/25 leaf and node hashing for the Merkle commitments.
//
// Replaces plonky2 0.2.2 hash/merkle_tree.rs MerkleTree::new (+ KeccakHash<25>::
// hash_or_noop / two_to_one) inside `circuit_data.prove`
// (plonky2-backend/src/actions/prove_action.rs:96), SURVEY.md 8a rows P4/P5.
//
// plonky2 transposes the LDE to leaf-major rows and bit-reverses them before
// hashing.  Here the LDE stays column-major per coset ([coset][col][k]): lane
// <-> k, so every column read is a fully coalesced 512 B wave access, and the
// leaf digest of natural row i = 8k + r is stored at [r][k].  plonky2's leaf
// index bitrev(i) = bitrev3(r) * n + bitrev_d(k) means tree level l pairs
// nodes k and k + n/2^l of the same coset: the tree is built level by level
// with unit-stride accesses and no bit reversal; only the cap entries are
// permuted on the host (cap[2*bitrev3(r) + k] for cap_height 4).
// Integer-ALU bound (Keccak-f ~ 8k 32-bit ops per 136 B), not HBM bound.
#include "internal.hpp"

namespace p2 {

// ---- PoseidonHash (hasher 1): Poseidon-Goldilocks, width 12, 8 full + 22 partial rounds, x^7 -----------------
// plonky2 0.2.2 hash/poseidon.rs; the permutation is poseidon.hpp's poseidon_permute_dev (one lane owns one sponge, like
// the Keccak path; the partial rounds' linear layers three at a time).
// hash_n_to_m_no_pad: overwrite-mode sponge, 8 elements per permutation, first 4 words out
template <class F>
__device__ __forceinline__ dig_t poseidon_sponge(uint32_t nwords, F get, const gl_t *prc) {
  gl_t st[12];
#pragma unroll
  for (int i = 0; i < 12; i++) st[i] = 0;
  for (uint32_t off = 0; off < nwords; off += 8) {
#pragma unroll
    for (int w = 0; w < 8; w++)
      if (off + w < nwords) st[w] = get(off + w);
    poseidon_permute_dev(st, prc);
  }
  dig_t d;
  d.w[0] = st[0]; d.w[1] = st[1]; d.w[2] = st[2]; d.w[3] = st[3];
  return d;
}
// hash/hashing.rs compress
__device__ __forceinline__ dig_t poseidon_two_to_one(const dig_t &l, const dig_t &r, const gl_t *prc) {
  gl_t st[12] = {l.w[0], l.w[1], l.w[2], l.w[3], r.w[0], r.w[1], r.w[2], r.w[3], 0, 0, 0, 0};
  poseidon_permute_dev(st, prc);
  dig_t d;
  d.w[0] = st[0]; d.w[1] = st[1]; d.w[2] = st[2]; d.w[3] = st[3];
  return d;
}
template <int H>
__device__ __forceinline__ dig_t node_hash(const dig_t &l, const dig_t &r, const gl_t *prc) {
  if constexpr (H == 1) return poseidon_two_to_one(l, r, prc);
  else return keccak_two_to_one(l, r);
}

template <bool PREFETCH = true, class F>
__device__ __forceinline__ dig_t sponge_hash(uint32_t nwords, F get) {
  uint64_t st[25];
#pragma unroll
  for (int i = 0; i < 25; i++) st[i] = 0;
  uint32_t off = 0;
  // software pipeline: the 17 words of block b+1 are requested before Keccak-f runs on block b,
  // so the loads (one 512 B wave access per column) fly under ~4300 VALU instructions
  if constexpr (PREFETCH) {
    uint64_t nxt[17];
    const bool first_full = nwords >= 17;
    if (first_full) {
#pragma unroll
      for (int w = 0; w < 17; w++) nxt[w] = get(w);
    }
    while (nwords - off >= 17) {
#pragma unroll
      for (int w = 0; w < 17; w++) st[w] ^= nxt[w];
      off += 17;
      if (nwords - off >= 17) {
#pragma unroll
        for (int w = 0; w < 17; w++) nxt[w] = get(off + w);
      }
      keccak_f1600(st);
    }
  } else {
    while (nwords - off >= 17) {
#pragma unroll
      for (int w = 0; w < 17; w++) st[w] ^= get(off + w);
      off += 17;
      keccak_f1600(st);
    }
  }
  const uint32_t rem = nwords - off;
#pragma unroll
  for (int w = 0; w < 17; w++) {
    if ((uint32_t)w < rem) st[w] ^= get(off + w);
    if ((uint32_t)w == rem) st[w] ^= 0x01ULL;
  }
  st[16] ^= 0x8000000000000000ULL;
  keccak_f1600(st);
  return dig_from_state(st);
}

// hash_or_noop: rows of <= 3 elements (Keccak: 25 bytes) / <= 4 elements (Poseidon: a HashOut) are copied
template <int H = 0, bool PREFETCH = true, class F>
__device__ __forceinline__ dig_t hash_or_noop(uint32_t nwords, F get, const gl_t *prc = nullptr) {
  if constexpr (H == 1) {
    if (nwords <= 4) {
      dig_t d;
      d.w[0] = nwords > 0 ? get(0) : 0;
      d.w[1] = nwords > 1 ? get(1) : 0;
      d.w[2] = nwords > 2 ? get(2) : 0;
      d.w[3] = nwords > 3 ? get(3) : 0;
      return d;
    }
    return poseidon_sponge(nwords, get, prc);
  }
  if (nwords * 8 <= 25) {
    dig_t d;
    d.w[0] = nwords > 0 ? get(0) : 0;
    d.w[1] = nwords > 1 ? get(1) : 0;
    d.w[2] = nwords > 2 ? get(2) : 0;
    d.w[3] = 0;
    return d;
  }
  return sponge_hash<PREFETCH>(nwords, get);
}

#ifndef P2_LEAF_WAVES
#define P2_LEAF_WAVES 1
#endif
// The variant with unmaterialised columns absorbs without the prefetch array (most of its words are computed, not
// loaded): 97 VGPRs, 4 waves per SIMD, no scratch.  Measured at 2^20 rows (wires tree, 154 virtual columns): 1.64 ms
// either way with 4 waves (with the prefetch: 128 VGPRs + 20 B of scratch), 1.73 ms unbounded (131 VGPRs, 3 waves),
// 1.64 ms at 5 waves (96 VGPRs + 12 B of scratch) -- the kernel sits at the issue ceiling, not at a latency.
#ifndef P2_LEAFV_WAVES
#define P2_LEAFV_WAVES 4
#endif
#ifndef P2_LEAFV_PREFETCH
#define P2_LEAFV_PREFETCH 0
#endif
// value of column i on this lane's LDE row: read from memory, or -- for a virtual column (VirtCols) -- the
// product of the column's scalar with the unit column's LDE value Lk of the row.  The branch is wave-uniform
// (class and scalar come through scalar loads); gl_mul returns the canonical value the fill kernel would have
// stored, so the digest is the same.
__device__ __forceinline__ gl_t virt_get(const VirtCols &v, uint32_t i, gl_t Lk, const gl_t *base, size_t n) {
  if (i >= v.first) {
    const uint32_t cl = ld_uniform(v.cls + i);  // (scalar loads even in the kernels that store digests inside their loop)
    if (cl < 2u) return cl == 1u ? gl_mul(ld_uniform(v.val + i), Lk) : (gl_t)0;  // (class 3 is materialised like a dense column)
  }
  return base[(size_t)i * n];
}
template <int H, bool V>
__global__ __launch_bounds__(256, V ? P2_LEAFV_WAVES : P2_LEAF_WAVES) void hash_lde_leaves_kernel(const gl_t *__restrict__ lde, uint32_t cols, uint32_t d,
                                                              dig_t *__restrict__ dig, const gl_t *__restrict__ prc, const VirtCols v) {
  const size_t n = (size_t)1 << d;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = blockIdx.y;
  if (k >= n) return;
  const gl_t *base = lde + (size_t)c * cols * n + k;
  if constexpr (V) {
    const gl_t Lk = v.basis ? v.basis[(size_t)(v.coset_first + c * v.coset_stride) * n + k] : (gl_t)0;
    dig[(size_t)c * n + k] = hash_or_noop<H, P2_LEAFV_PREFETCH != 0>(cols, [&](uint32_t i) { return virt_get(v, i, Lk, base, n); }, prc);
  } else {
    dig[(size_t)c * n + k] = hash_or_noop<H>(cols, [&](uint32_t i) { return base[(size_t)i * n]; }, prc);
  }
}

// ---- Keccak leaf hashing with the sponge state in fixed registers (keccak.hpp P2_KF_*) ----------------------------------
template <int I, int N, class F>
__device__ __forceinline__ void kf_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    kf_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ void kf_zero() {
  kf_for<0, 25>([&](auto ic) { P2_KF_SET(decltype(ic)::value, 0u, 0u); });
}
// One sponge step: absorb the next rate block -- 17 full words, or (the last step) the ragged tail of rem < 17 words with the
// original Keccak padding -- and permute.  ONE call site of the 34 KB permutation per kernel: two would not fit the
// 64 KB instruction cache together.
// words [W0, W1) of the rate block: all loads of the group first, then the XORs -- an `asm volatile` is a scheduling barrier
// for hipcc, so a load written next to its XOR is waited for before the next one is even issued (17 dependent HBM latencies
// per block).  G = words per group: what the compiler's share of the register file (P2_KF_BASE registers) can hold.
template <int W0, int W1, class F>
__device__ __forceinline__ void kf_absorb_group(uint32_t off, F get) {
  uint64_t x[W1 - W0];
#pragma unroll
  for (int w = W0; w < W1; w++) x[w - W0] = get(off + w);
  kf_for<W0, W1>([&](auto wc) {
    constexpr int w = decltype(wc)::value;
    const uint64_t xw = x[w - W0];  // (an asm operand may not name a captured variable)
    const uint32_t lo = (uint32_t)xw, hi = (uint32_t)(xw >> 32);
    P2_KF_XOR(w, lo, hi);
  });
}
template <bool SPLIT, class F>
__device__ __forceinline__ void kf_absorb(uint32_t off, uint32_t rem, F get) {
  if (rem >= 17) {
    if constexpr (SPLIT) {  // virtual columns: each word may be a modular product, which needs registers of its own
      kf_absorb_group<0, 6>(off, get);
      kf_absorb_group<6, 12>(off, get);
      kf_absorb_group<12, 17>(off, get);
    } else {
      kf_absorb_group<0, 17>(off, get);
    }
  } else {
    kf_for<0, 17>([&](auto wc) {  // once per leaf: word by word
      constexpr int w = decltype(wc)::value;
      if ((uint32_t)w <= rem) {
        const uint64_t xw = (uint32_t)w < rem ? get(off + w) : (uint64_t)1;  // ... | 0x01 pad
        const uint32_t lo = (uint32_t)xw, hi = (uint32_t)(xw >> 32);
        P2_KF_XOR(w, lo, hi);
      }
    });
    P2_KF_XOR(16, 0u, 0x80000000u);
  }
}
__device__ __forceinline__ dig_t kf_digest() {
  uint32_t l0, h0, l1, h1, l2, h2, l3, h3;
  P2_KF_GET(0, l0, h0);
  P2_KF_GET(1, l1, h1);
  P2_KF_GET(2, l2, h2);
  P2_KF_GET(3, l3, h3);
  dig_t dg;
  dg.w[0] = ((uint64_t)h0 << 32) | l0;
  dg.w[1] = ((uint64_t)h1 << 32) | l1;
  dg.w[2] = ((uint64_t)h2 << 32) | l2;
  dg.w[3] = l3 & 0xFFu;
  (void)h3;
  return dg;
}
// state words 0..24 <- the padded 50-byte message left[25] || right[25] of KeccakHash<25>::two_to_one (the permutation follows)
__device__ __forceinline__ void kf_set_two_to_one(const dig_t &l, const dig_t &r) {
  uint64_t w[7];
  w[0] = l.w[0];
  w[1] = l.w[1];
  w[2] = l.w[2];
  w[3] = (l.w[3] & 0xFFULL) | (r.w[0] << 8);
  w[4] = (r.w[0] >> 56) | (r.w[1] << 8);
  w[5] = (r.w[1] >> 56) | (r.w[2] << 8);
  w[6] = (r.w[2] >> 56) | ((r.w[3] & 0xFFULL) << 8) | (0x01ULL << 16);
  kf_for<0, 7>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const uint64_t x = w[i];
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    P2_KF_SET(i, lo, hi);
  });
  kf_for<7, 25>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    P2_KF_SET(i, 0u, (i == 16 ? 0x80000000u : 0u));
  });
}
// same contract as hash_lde_leaves_kernel<0, V>; rows of <= 3 elements (copied, not hashed) never come here.
// LV = 2 (round 5): the block also builds the first TWO tree levels over its leaves.  Level l pairs nodes k and k + n / 2^l of a
// coset, so the four waves of a block take the leaves k + g * n/4 (g = wave; 64 neighbouring k: every column read is still one
// 512 B access): level 1 node k = H(leaf k, leaf k + n/2) is wave g < 2's own digest with wave g + 2's, level 2 node k is wave
// 0's level-1 digest with wave 1's -- partners pass through LDS, a wave whose digests have been handed on ends (s_barrier
// counts surviving waves only).  Three quarters of a tree's node hashes without a launch of their own, their children never
// read back from memory.  The whole chain runs through ONE permutation call site (two 34 KB copies would not share the 64 KB
// instruction cache): a loop whose body is "prepare the state | permute | if a hash is complete, store it and pick the next".
template <bool V, int LV>
__global__ __launch_bounds__(256) P2_KF_KERNEL_ATTR void hash_lde_leaves_kf_kernel(const gl_t *__restrict__ lde, uint32_t cols, uint32_t d,
                                                                                   dig_t *__restrict__ dig, dig_t *__restrict__ lvl1,
                                                                                   dig_t *__restrict__ lvl2, const VirtCols v) {
  __shared__ dig_t xch1[LV ? 2 : 1][64], xch2[64];
  const size_t n = (size_t)1 << d;
  const uint32_t g = threadIdx.x >> 6, t = threadIdx.x & 63u;
  const size_t k = LV ? (size_t)blockIdx.x * 64 + t + (size_t)g * (n >> 2) : (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = blockIdx.y;
  if (!LV && k >= n) return;
  const gl_t *base = lde + (size_t)c * cols * n + k;
  gl_t Lk = 0;
  if constexpr (V) Lk = v.basis ? v.basis[(size_t)(v.coset_first + c * v.coset_stride) * n + k] : (gl_t)0;
  auto get = [&](uint32_t i) -> gl_t {
    if constexpr (V) return virt_get(v, i, Lk, base, n);
    else return base[(size_t)i * n];
  };
  kf_zero();
  // every word of the plain variant is a load: software pipeline -- the 17 loads of block b + 1 are issued before the permutation
  // of block b and land under its 4 309 instructions (the array lives in hipcc's registers across the asm block; kf_check.py
  // watches them)
  uint64_t x[V ? 1 : 17];
  // words of the rate block at `off`; the last block is ragged: rem < 17 words, then the 0x01 of the padding, then zeros
  auto load_block = [&](uint32_t off) {
    if constexpr (!V) {
      const uint32_t rem = cols - off;
#pragma unroll
#ifdef P2_KF_NOLOAD  /* timing experiment only (scratch/): what the kernel costs without its memory traffic */
      for (int w = 0; w < 17; w++) x[w] = (uint32_t)w < rem ? (uint64_t)(k + off + w) : ((uint32_t)w == rem ? (uint64_t)1 : (uint64_t)0);
#else
      for (int w = 0; w < 17; w++) x[w] = (uint32_t)w < rem ? base[(size_t)(off + w) * n] : ((uint32_t)w == rem ? (uint64_t)1 : (uint64_t)0);
#endif
    }
  };
  load_block(0);
  uint32_t off = 0, stage = 0;  // stage 0: the leaf's sponge; 1, 2: the node of that tree level
  for (;;) {
    bool complete = true;
    // (opaque to the loop optimiser: with `off` a recognisable induction variable hipcc keeps seventeen 64-bit column pointers
    // live across the permutation instead of forming them at the loads -- 30 spilled dwords)
    off = __builtin_amdgcn_readfirstlane(off);
    if (stage == 0) {
      const uint32_t rem = cols - off;
      complete = rem < 17;
      if constexpr (!V) {
        kf_for<0, 17>([&](auto wc) {
          constexpr int w = decltype(wc)::value;
          const uint64_t xw = x[w];
          const uint32_t lo = (uint32_t)xw, hi = (uint32_t)(xw >> 32);
          P2_KF_XOR(w, lo, hi);
        });
        if (complete) P2_KF_XOR(16, 0u, 0x80000000u);
        off += 17;
        if (!complete) load_block(off);
      } else {
        kf_absorb<true>(off, rem, get);
        off += 17;
      }
    }
    P2_KECCAK_FIXED_PERMUTE();  // (the one call site of this kernel)
    if (!complete) continue;
    const dig_t dg = kf_digest();
    if (stage == 0) dig[(size_t)c * n + k] = dg;
    else if (stage == 1) lvl1[(size_t)c * (n >> 1) + k] = dg;
    else lvl2[(size_t)c * (n >> 2) + k] = dg;
    if (stage == (uint32_t)LV) break;
    dig_t r;
    if (stage == 0) {
      if (g >= 2) xch1[g & 1][t] = dg;
      __syncthreads();
      if (g >= 2) break;
      r = xch1[g & 1][t];
    } else {
      if (g == 1) xch2[t] = dg;
      __syncthreads();  // (waves 2 and 3 have ended)
      if (g == 1) break;
      r = xch2[t];
    }
    kf_set_two_to_one(dg, r);
    if constexpr (!V) {
      // the prefetch array is dead from here on; saying so (constants, not the last block's words, go round the loop) lets
      // hipcc give its registers to the node's words above -- otherwise 22 dwords spill
#pragma unroll
      for (int w = 0; w < 17; w++) x[w] = 0;
    }
    stage++;
  }
}
// KeccakHash<25>::two_to_one (keccak_two_to_one of keccak.hpp) on the fixed registers: Keccak-256(left[25] || right[25])[..25].
// PH = code-placement phase of the permutation block (keccak_fixed.inc): 1 for levels with many nodes, 0 when a SIMD holds a
// lone wave (the tails, the small levels).
template <int PH>
__device__ __forceinline__ dig_t kf_two_to_one(const dig_t &l, const dig_t &r) {
  kf_set_two_to_one(l, r);
  if constexpr (PH == 0) P2_KECCAK_FIXED_PERMUTE_PH(0);  // (the macro stringifies its argument: literals only)
  else P2_KECCAK_FIXED_PERMUTE_PH(1);
  return kf_digest();
}
template <int PH>
__global__ __launch_bounds__(256) P2_KF_KERNEL_ATTR void merkle_level_kf_kernel(const dig_t *__restrict__ in, dig_t *__restrict__ out, uint32_t m) {
  const uint32_t half = m >> 1;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = blockIdx.y;
  if (k >= half) return;
  const dig_t l = in[(size_t)c * m + k], r = in[(size_t)c * m + k + half];
  out[(size_t)c * half + k] = kf_two_to_one<PH>(l, r);
}
// Up to three single-lane levels in one launch, for the levels that are latency-bound (at most one wave per SIMD on the chip)
// but still too wide for the 25-lane form: a block of four waves takes 64 neighbouring nodes of its last level and everything
// above them (nodes k + j * mf of every level, as in merkle_coop_kernel); level 1 on all four waves, level 2 on two, level 3 on
// one, children through LDS.  Saves the store -> end of kernel -> launch -> load (3.6 of a level's 11 us) between fused levels.
// grid = cosets * mf / 64, mf = m >> levels a multiple of 64.
__global__ __launch_bounds__(256) P2_KF_KERNEL_ATTR void merkle_levels_kf_kernel(dig_t *lvl, uint32_t cosets, uint32_t m, uint32_t levels) {
  __shared__ dig_t nodes[4][64];
  const uint32_t mf = m >> levels, bpc = mf >> 6;
  const uint32_t c = blockIdx.x / bpc, t = threadIdx.x & 63u, g = threadIdx.x >> 6;
  const uint32_t k = (blockIdx.x % bpc) * 64 + t;
  uint32_t cnt = 1u << levels;
#pragma unroll 1
  for (uint32_t lv = 0; lv < levels; lv++) {
    const uint32_t half = cnt >> 1;
    if (g < half) {
      dig_t l, r;
      if (lv == 0) {
        const dig_t *in = lvl + (size_t)c * m;
        l = in[k + (size_t)g * mf];
        r = in[k + (size_t)(g + half) * mf];
      } else {
        l = nodes[g][t];
        r = nodes[g + half][t];
      }
      const dig_t o = kf_two_to_one<0>(l, r);
      lvl[(size_t)cosets * m + (size_t)c * (m >> 1) + k + (size_t)g * mf] = o;
      nodes[g][t] = o;  // (g < half: read by this lane only; the right children sit at >= half)
    }
    __syncthreads();
    lvl += (size_t)cosets * m;
    m >>= 1;
    cnt = half;
  }
}
// Up to four levels of the top of a Keccak tree per launch, every node hashed by 25 lanes (keccak_f1600_coop): a block of
// four waves (one per SIMD: eight permutations at a time) owns the complete sub-tree under ONE node of the last level it
// computes -- with the (k, k + half) pairing of merkle_level_kernel that is the nodes k0 + j * mf of every level above
// (mf = nodes per coset of that last level) -- so the levels inside a launch need a block barrier only; inputs of the
// first level come from memory, later ones from LDS, every level goes out to memory for the query paths.
// grid = cosets * mf blocks.  m = nodes per coset of the input level, levels in [1, 4].
// mirror (optional, page-locked host memory): the level with cap_per nodes per coset is ALSO stored there, [coset][cap_per] --
// the transcript reads the cap from it after the stream sync, without a copy kernel in between.
__global__ __launch_bounds__(256) void merkle_coop_kernel(dig_t *lvl, uint32_t cosets, uint32_t m, uint32_t levels, dig_t *mirror,
                                                          uint32_t cap_per) {
  __shared__ uint64_t nodes[16][4];
  const uint32_t mf = m >> levels;
  const uint32_t c = blockIdx.x / mf, k0 = blockIdx.x % mf;
  const uint32_t lane = threadIdx.x & 63u, L = lane & 31u, slot = (threadIdx.x >> 6) * 2 + (lane >> 5);
  const KeccakCoopLane cl = keccak_coop_lane(lane);
  uint32_t cnt = 1u << levels;  // this block's nodes of the current input level
  for (uint32_t lv = 0; lv < levels; lv++) {
    const uint32_t half = cnt >> 1;
    if (slot < half) {
      // state words 0..6 = left[25] || right[25] || 0x01 pad, word 16 = the 0x80 that ends the rate (kf_two_to_one):
      // with D[0..3] = left.w, D[4..7] = right.w, word L is (D[L] low part) | (D[L + 1] << 8)
      uint64_t p = 0, q = 0;
      if (L < 7) {
        if (lv == 0) {
          const dig_t *in = lvl + (size_t)c * m;
          const dig_t *l = in + k0 + (size_t)slot * mf, *r = in + k0 + (size_t)(slot + half) * mf;
          p = L < 4 ? l->w[L] : r->w[L - 4];
          if (L >= 3) q = r->w[L - 3];
        } else {
          p = L < 4 ? nodes[slot][L] : nodes[slot + half][L - 4];
          if (L >= 3) q = nodes[slot + half][L - 3];
        }
      }
      uint64_t w = L < 3 ? p : (L == 3 ? (p & 0xFFULL) : (p >> 56));
      if (L >= 3 && L < 7) w |= (L == 6 ? ((q & 0xFFULL) << 8) | (1ULL << 16) : (q << 8));
      if (L >= 7) w = L == 16 ? 0x8000000000000000ULL : 0ULL;
      uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
      keccak_f1600_coop(lo, hi, cl);
      if (L < 4) {
        const uint64_t o = L == 3 ? (uint64_t)(lo & 0xFFu) : (((uint64_t)hi << 32) | lo);
        lvl[(size_t)cosets * m + (size_t)c * (m >> 1) + k0 + (size_t)slot * mf].w[L] = o;
        if (mirror != nullptr && (m >> 1) == cap_per) mirror[(size_t)c * cap_per + k0 + (size_t)slot * mf].w[L] = o;
        nodes[slot][L] = o;  // slot < half is read by this slot only (as its left child); right children sit at >= half
      }
    }
    __syncthreads();
    lvl += (size_t)cosets * m;
    m >>= 1;
    cnt = half;
  }
}

// hash_lde_absorb_kernel with the state in fixed registers between the HBM round trips
template <bool V>
__global__ __launch_bounds__(256) P2_KF_KERNEL_ATTR void hash_lde_absorb_kf_kernel(const gl_t *__restrict__ lde, uint32_t cols, uint32_t d,
                                                                                   uint32_t blk0, uint32_t nblk, int first, int last,
                                                                                   uint64_t *__restrict__ state, dig_t *__restrict__ dig, const VirtCols v) {
  const size_t n = (size_t)1 << d;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = blockIdx.y;
  if (k >= n) return;
  const gl_t *base = lde + (size_t)c * cols * n + k;
  uint64_t *sp = state + (size_t)c * 25 * n + k;
  gl_t Lk = 0;
  if constexpr (V) Lk = v.basis ? v.basis[(size_t)(v.coset_first + c * v.coset_stride) * n + k] : (gl_t)0;
  auto get = [&](uint32_t i) -> gl_t {
    if constexpr (V) return virt_get(v, i, Lk, base, n);
    else return base[(size_t)i * n];
  };
  if (first) kf_zero();
  else
    kf_for<0, 25>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const uint64_t x = sp[(size_t)i * n];
      P2_KF_SET(i, (uint32_t)x, (uint32_t)(x >> 32));
    });
  // nblk full blocks, then (last call only) the ragged tail: cols - 17 * (blk0 + nblk) < 17 words
  const uint32_t steps = nblk + (last ? 1u : 0u);
  for (uint32_t b = 0; b < steps; b++) {
    const uint32_t off = 17 * (blk0 + b);
    kf_absorb<true>(off, b < nblk ? 17u : cols - off, get);  // (groups of 6: this kernel also holds the state pointers)
    P2_KECCAK_FIXED_PERMUTE();
  }
  if (last) {
    dig[(size_t)c * n + k] = kf_digest();
  } else {
    kf_for<0, 25>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      uint32_t lo, hi;  // (locals of the lambda: an asm output may not name a captured variable)
      P2_KF_GET(i, lo, hi);
      sp[(size_t)i * n] = ((uint64_t)hi << 32) | lo;
    });
  }
}

// Leaf hashing for a witness that arrives in column chunks (p2gpu_prove): a Keccak sponge absorbs
// 17 columns per permutation, in column order, so the rate blocks of the columns already on the
// device can be absorbed while later columns are still crossing PCIe.  The 25-word state of every
// row waits in HBM between calls ([coset][25][n], word-major: coalesced); the call with `last` also
// absorbs the ragged tail with the padding and writes the digest.
template <bool V>
__global__ __launch_bounds__(256) void hash_lde_absorb_kernel(const gl_t *__restrict__ lde, uint32_t cols, uint32_t d,
                                                              uint32_t blk0, uint32_t nblk, int first, int last,
                                                              uint64_t *__restrict__ state, dig_t *__restrict__ dig, const VirtCols v) {
  const size_t n = (size_t)1 << d;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = blockIdx.y;
  if (k >= n) return;
  const gl_t *base = lde + (size_t)c * cols * n + k;
  uint64_t *sp = state + (size_t)c * 25 * n + k;
  gl_t Lk = 0;
  if constexpr (V) Lk = v.basis ? v.basis[(size_t)(v.coset_first + c * v.coset_stride) * n + k] : (gl_t)0;
  auto get = [&](uint32_t i) -> gl_t {
    if constexpr (V) return virt_get(v, i, Lk, base, n);
    else return base[(size_t)i * n];
  };
  uint64_t st[25];
#pragma unroll
  for (int i = 0; i < 25; i++) st[i] = first ? 0 : sp[(size_t)i * n];
  for (uint32_t b = 0; b < nblk; b++) {
    const uint32_t i0 = 17 * (blk0 + b);
#pragma unroll
    for (int w = 0; w < 17; w++) st[w] ^= get(i0 + w);
    keccak_f1600(st);
  }
  if (last) {
    const uint32_t off = 17 * (blk0 + nblk), rem = cols - off;  // rem < 17
#pragma unroll
    for (int w = 0; w < 17; w++) {
      if ((uint32_t)w < rem) st[w] ^= get(off + w);
      if ((uint32_t)w == rem) st[w] ^= 0x01ULL;
    }
    st[16] ^= 0x8000000000000000ULL;
    keccak_f1600(st);
    dig[(size_t)c * n + k] = dig_from_state(st);
  } else {
#pragma unroll
    for (int i = 0; i < 25; i++) sp[(size_t)i * n] = st[i];
  }
}

__global__ __launch_bounds__(256) void hash_rows_kernel(const gl_t *__restrict__ rows, size_t n_rows, uint32_t row_len,
                                                        dig_t *__restrict__ dig) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const gl_t *base = rows + r * row_len;
  dig[r] = hash_or_noop(row_len, [&](uint32_t i) { return base[i]; });
}

// leaf (r, kl): ext values at k = bitrev_ab(t) * (npc >> ab) + kl, t < 2^ab, flattened (c0, c1)
template <int H>
__global__ __launch_bounds__(256) void hash_fri_leaves_kernel(const gl_t *__restrict__ vals, uint32_t lg_npc,
                                                              uint32_t ab, dig_t *__restrict__ dig, const gl_t *__restrict__ prc) {
  const uint32_t npc = 1u << lg_npc, per = npc >> ab;
  const uint32_t kl = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t r = blockIdx.y;
  if (kl >= per) return;
  const gl_t *c0 = vals + (size_t)r * 2 * npc, *c1 = c0 + npc;
  dig[(size_t)r * per + kl] = hash_or_noop<H>(2u << ab, [&](uint32_t w) {
    uint32_t t = w >> 1;
    uint32_t k = bitrev32(t, ab) * per + kl;
    return (w & 1) ? c1[k] : c0[k];
  }, prc);
}

// hash_fri_leaves_kernel<1> for the small trees of the later reduction steps: twelve lanes per leaf (poseidon_permute_coop),
// the overwrite-mode sponge of hash_n_to_m_no_pad -- words 8b .. 8b + 7 of the leaf replace state words 0..7 before
// permutation b.  Leaves of 2 << ab >= 8 words (a multiple of the rate).  One block = 16 leaves.
__global__ __launch_bounds__(256) void hash_fri_leaves_coop_poseidon_kernel(const gl_t *__restrict__ vals, uint32_t lg_npc, uint32_t ab,
                                                                            uint32_t cosets, dig_t *__restrict__ dig,
                                                                            const gl_t *__restrict__ prc) {
  const uint32_t npc = 1u << lg_npc, per = npc >> ab;
  const uint32_t lane = threadIdx.x & 63u, i = lane & 15u;
  const uint32_t slot = blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + (lane >> 4);
  if (slot >= cosets * per) return;  // (whole 16-lane groups leave together)
  const uint32_t r = slot / per, kl = slot % per;
  const gl_t *c0 = vals + (size_t)r * 2 * npc, *c1 = c0 + npc;
  const PoseidonCoopLane cl = poseidon_coop_lane(lane);
  gl_t x = 0;
  for (uint32_t b = 0; b < (2u << ab) / 8; b++) {
    if (i < 8) {
      const uint32_t w = 8 * b + i, k = bitrev32(w >> 1, ab) * per + kl;
      x = (w & 1) ? c1[k] : c0[k];
    }
    x = poseidon_permute_coop(x, prc, cl);
  }
  if (i < 4) dig[(size_t)r * per + kl].w[i] = x;
}

template <int H>
__global__ __launch_bounds__(256) void merkle_level_kernel(const dig_t *__restrict__ in, dig_t *__restrict__ out,
                                                           uint32_t m, const gl_t *__restrict__ prc) {
  const uint32_t half = m >> 1;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = blockIdx.y;
  if (k >= half) return;
  const dig_t l = in[(size_t)c * m + k], r = in[(size_t)c * m + k + half];
  out[(size_t)c * half + k] = node_hash<H>(l, r, prc);
}

// merkle_coop_kernel for PoseidonHash: twelve lanes per permutation (poseidon_permute_coop), four per wave, sixteen per block of
// four waves; up to five levels per launch (the first one on all sixteen slots).  two_to_one = the first four words of
// permute(left[4] || right[4] || 0^4) (hashing.rs compress).
__global__ __launch_bounds__(256) void merkle_coop_poseidon_kernel(dig_t *lvl, uint32_t cosets, uint32_t m, uint32_t levels,
                                                                   const gl_t *__restrict__ prc) {
  __shared__ uint64_t nodes[32][4];
  const uint32_t mf = m >> levels;
  const uint32_t c = blockIdx.x / mf, k0 = blockIdx.x % mf;
  const uint32_t lane = threadIdx.x & 63u, i = lane & 15u, slot = (threadIdx.x >> 6) * 4 + (lane >> 4);
  const PoseidonCoopLane cl = poseidon_coop_lane(lane);
  uint32_t cnt = 1u << levels;
  for (uint32_t lv = 0; lv < levels; lv++) {
    const uint32_t half = cnt >> 1;
    if (slot < half) {
      gl_t x = 0;
      if (i < 8) {
        const uint32_t j = slot + (i < 4 ? 0u : half), w = i & 3u;
        x = lv == 0 ? lvl[(size_t)c * m + k0 + (size_t)j * mf].w[w] : nodes[j][w];
      }
      x = poseidon_permute_coop(x, prc, cl);
      if (i < 4) {
        lvl[(size_t)cosets * m + (size_t)c * (m >> 1) + k0 + (size_t)slot * mf].w[i] = x;
        nodes[slot][i] = x;  // slot < half is read by this slot only (as its left child); right children sit at >= half
      }
    }
    __syncthreads();
    lvl += (size_t)cosets * m;
    m >>= 1;
    cnt = half;
  }
}
// levels with at most this many nodes (all cosets together) go to merkle_tail
size_t merkle_tail_from(const gl_t *prc) {
  if (prc) return 16384;      // Poseidon: twelve lanes per node pay up to four waves per SIMD (~27 vs ~75 us per level)
  return (size_t)1024 * 64;   // Keccak: from one wave per SIMD
}
bool merkle_tail(hipStream_t st, dig_t *lvl, uint32_t cosets, uint32_t m, uint32_t cap_per, const gl_t *prc, dig_t *host_mirror) {
  if (m <= cap_per) return false;
  if (prc) {
    while (m > cap_per) {
      if ((size_t)cosets * (m >> 1) > 16384) {
        merkle_level(st, lvl, lvl + (size_t)cosets * m, cosets, m, prc);
        lvl += (size_t)cosets * m;
        m >>= 1;
        continue;
      }
      uint32_t levels = 0;
      while (levels < 5 && (m >> levels) > cap_per) levels++;
      {
        ProfScope ps("merkle_coop_poseidon_kernel", 96.0 * cosets * (double)(m - (m >> levels)));
        hipLaunchKernelGGL(merkle_coop_poseidon_kernel, dim3(cosets * (m >> levels)), dim3(256), 0, st, lvl, cosets, m, levels, prc);
      }
      for (uint32_t q = 0; q < levels; q++) {
        lvl += (size_t)cosets * m;
        m >>= 1;
      }
    }
    return false;
  }
  {
    // Keccak: a level with more than 2 048 nodes (two per wave: one wave per SIMD on the chip) is still cheaper one lane
    // per node (7.5 us); below that the 25-lane form (3.5-5 us per level), four levels per launch
    bool mirrored = false;
    while (m > cap_per) {
      if ((size_t)cosets * (m >> 1) > 2048 && (m >> 1) >= 256) {
        // single-lane levels; the latency-bound ones (<= one wave per SIMD on the chip) up to three per launch
        uint32_t lv = 1;
        if ((size_t)cosets * (m >> 1) <= (size_t)1024 * 64)
          while (lv < 3 && (size_t)cosets * (m >> (lv + 1)) > 2048 && (m >> (lv + 1)) >= 256 && (m >> (lv + 1)) >= cap_per) lv++;
        if (lv == 1) {
          merkle_level(st, lvl, lvl + (size_t)cosets * m, cosets, m, prc);
        } else {
          ProfScope ps("merkle_levels_kf_kernel", 96.0 * cosets * (double)(m - (m >> lv)));
          hipLaunchKernelGGL(merkle_levels_kf_kernel, dim3(cosets * ((m >> lv) >> 6)), dim3(256), 0, st, lvl, cosets, m, lv);
        }
        for (uint32_t i = 0; i < lv; i++) {
          lvl += (size_t)cosets * m;
          m >>= 1;
        }
        continue;
      }
      uint32_t levels = 0;
      while (levels < 4 && (m >> levels) > cap_per) levels++;
      {
        ProfScope ps("merkle_coop_kernel", 96.0 * cosets * (double)(m - (m >> levels)));
        hipLaunchKernelGGL(merkle_coop_kernel, dim3(cosets * (m >> levels)), dim3(256), 0, st, lvl, cosets, m, levels, host_mirror, cap_per);
        if (host_mirror != nullptr && (m >> levels) == cap_per) mirrored = true;  // (a cap this wide ends on a plain level launch)
      }
      for (uint32_t i = 0; i < levels; i++) {
        lvl += (size_t)cosets * m;
        m >>= 1;
      }
    }
    return mirrored;
  }
}

// P2GPU_LEAF_LEVELS=0: the leaf kernels leave every tree level to merkle_level / merkle_tail (A/B measurements)
static bool leaf_levels_on() {
  static const bool on = [] { const char *e = getenv("P2GPU_LEAF_LEVELS"); return !(e && *e == '0'); }();
  return on;
}
// Returns how many tree levels above the leaf digests the launch has ALSO built (0 or 2): lvl1 / lvl2 = storage of the levels with
// n/2 and n/4 nodes per coset ([cosets][n/2], [cosets][n/4]; nullptr: leaves only).
uint32_t hash_lde_leaves(hipStream_t st, const gl_t *lde, uint32_t cols, uint32_t d, uint32_t cosets, dig_t *dig, const gl_t *prc,
                         const VirtCols *virt, dig_t *lvl1, dig_t *lvl2) {
  size_t n = (size_t)1 << d;
  uint32_t threads = n >= 256 ? 256 : 64;
  // Keccak, a hashed leaf (more than 3 elements) and full 256-lane blocks: the fixed-register sponge (profile names = the
  // symbols rocprofv3 shows)
  const bool kf = !prc && cols * 8 > 25 && n >= 256;
  // ... only where it measured faster (profiles/r05_tree_levels.md, 2^20 rows, lone proof): the wires tree of a witness with
  // unmaterialised columns (14 permutations per leaf; grouped-load sponge): 1603 us against 1545 + 57 + 31 for leaves + two level
  // launches.  Not the 2-block leaves of Z / partial products (302 vs 300) or the 1-block leaves of the quotient (204 vs 185:
  // after one permutation half, then three quarters of a block's waves have nothing left to do), and not a dense wires tree,
  // whose prefetching sponge <false, 0> cannot spare the registers (1615 vs 1483 + 59 + 32 through the grouped-load form)
  const bool virt_on = virt && virt->cls && virt->first < cols;
  const bool lv2 = kf && virt_on && lvl1 != nullptr && lvl2 != nullptr && cols > 3 * 17 && leaf_levels_on();
  const double node_bytes = lv2 ? 96.0 * cosets * (double)(n / 2 + n / 4) : 0.0;
  const dim3 grid((uint32_t)((n + threads - 1) / threads), cosets);
  if (virt_on) {  // the wires of a proof with unmaterialised columns (same digests)
    ProfScope ps(prc ? "hash_lde_leaves_kernel<1, true>" : (kf ? (lv2 ? "hash_lde_leaves_kf_kernel<true, 2>" : "hash_lde_leaves_kf_kernel<true, 0>") : "hash_lde_leaves_kernel<0, true>"),
                 (8.0 * cols + 32.0) * cosets * (double)n + node_bytes);
    if (prc) hipLaunchKernelGGL((hash_lde_leaves_kernel<1, true>), grid, dim3(threads), 0, st, lde, cols, d, dig, prc, *virt);
    else if (lv2) hipLaunchKernelGGL((hash_lde_leaves_kf_kernel<true, 2>), grid, dim3(threads), 0, st, lde, cols, d, dig, lvl1, lvl2, *virt);
    else if (kf) hipLaunchKernelGGL((hash_lde_leaves_kf_kernel<true, 0>), grid, dim3(threads), 0, st, lde, cols, d, dig, lvl1, lvl2, *virt);
    else hipLaunchKernelGGL((hash_lde_leaves_kernel<0, true>), grid, dim3(threads), 0, st, lde, cols, d, dig, prc, *virt);
    return lv2 ? 2u : 0u;
  }
  // same spelling as rocprofv3's demangled names (<0> Keccak, <1> Poseidon), so the bench line and profiles/ agree
  ProfScope ps(prc ? "hash_lde_leaves_kernel<1, false>" : (kf ? "hash_lde_leaves_kf_kernel<false, 0>" : "hash_lde_leaves_kernel<0, false>"),
               (8.0 * cols + 32.0) * cosets * (double)n + node_bytes);
  if (prc) hipLaunchKernelGGL((hash_lde_leaves_kernel<1, false>), grid, dim3(threads), 0, st, lde, cols, d, dig, prc, VirtCols());
  else if (kf) hipLaunchKernelGGL((hash_lde_leaves_kf_kernel<false, 0>), grid, dim3(threads), 0, st, lde, cols, d, dig, lvl1, lvl2, VirtCols());
  else hipLaunchKernelGGL((hash_lde_leaves_kernel<0, false>), grid, dim3(threads), 0, st, lde, cols, d, dig, prc, VirtCols());
  return lv2 ? 2u : 0u;
}
void hash_lde_absorb(hipStream_t st, const gl_t *lde, uint32_t cols, uint32_t d, uint32_t cosets, uint32_t blk0,
                     uint32_t nblk, bool first, bool last, uint64_t *state, dig_t *dig, const VirtCols *virt) {
  size_t n = (size_t)1 << d;
  uint32_t threads = n >= 256 ? 256 : 64;
  ProfScope ps("hash_lde_absorb_kernel", (8.0 * 17 * nblk + (first ? 0 : 200) + (last ? 32 + 8.0 * (cols - 17 * (blk0 + nblk)) : 200)) *
                                             cosets * (double)n);
  if (n >= 256) {
    if (virt && virt->cls && virt->first < cols)
      hipLaunchKernelGGL(hash_lde_absorb_kf_kernel<true>, dim3((n + threads - 1) / threads, cosets), dim3(threads), 0, st, lde, cols,
                         d, blk0, nblk, first ? 1 : 0, last ? 1 : 0, state, dig, *virt);
    else
      hipLaunchKernelGGL(hash_lde_absorb_kf_kernel<false>, dim3((n + threads - 1) / threads, cosets), dim3(threads), 0, st, lde, cols,
                         d, blk0, nblk, first ? 1 : 0, last ? 1 : 0, state, dig, VirtCols());
    return;
  }
  if (virt && virt->cls && virt->first < cols)
    hipLaunchKernelGGL(hash_lde_absorb_kernel<true>, dim3((n + threads - 1) / threads, cosets), dim3(threads), 0, st, lde, cols,
                       d, blk0, nblk, first ? 1 : 0, last ? 1 : 0, state, dig, *virt);
  else
    hipLaunchKernelGGL(hash_lde_absorb_kernel<false>, dim3((n + threads - 1) / threads, cosets), dim3(threads), 0, st, lde, cols,
                       d, blk0, nblk, first ? 1 : 0, last ? 1 : 0, state, dig, VirtCols());
}
void hash_rows(hipStream_t st, const gl_t *rows, size_t n_rows, uint32_t row_len, dig_t *dig) {
  hipLaunchKernelGGL(hash_rows_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, st, rows, n_rows, row_len, dig);
}
void hash_fri_leaves(hipStream_t st, const gl_t *vals, uint32_t lg_npc, uint32_t cosets, uint32_t ab, dig_t *dig, const gl_t *prc) {
  uint32_t per = (1u << lg_npc) >> ab;
  uint32_t threads = per >= 256 ? 256 : 64;
  ProfScope ps(prc ? "hash_fri_leaves_kernel<1>" : "hash_fri_leaves_kernel<0>", (16.0 * (1u << ab) + 32.0) * cosets * (double)per);
  if (prc && (2u << ab) >= 8 && (size_t)cosets * per <= 4096)  // latency-bound: 4 permutations of ~12 us instead of ~75
    hipLaunchKernelGGL(hash_fri_leaves_coop_poseidon_kernel, dim3((cosets * per + 15) / 16), dim3(256), 0, st, vals, lg_npc, ab, cosets, dig, prc);
  else if (prc) hipLaunchKernelGGL(hash_fri_leaves_kernel<1>, dim3((per + threads - 1) / threads, cosets), dim3(threads), 0, st, vals, lg_npc, ab, dig, prc);
  else hipLaunchKernelGGL(hash_fri_leaves_kernel<0>, dim3((per + threads - 1) / threads, cosets), dim3(threads), 0, st, vals, lg_npc, ab, dig, prc);
}
void merkle_level(hipStream_t st, const dig_t *in, dig_t *out, uint32_t cosets, uint32_t m, const gl_t *prc) {
  uint32_t half = m >> 1;
  uint32_t threads = half >= 256 ? 256 : 64;
  const bool kf_big = !prc && (size_t)half * cosets >= (size_t)2 * 1024 * 64;
  const bool kf_small = !prc && !kf_big && threads == 256;
  ProfScope ps(prc ? "merkle_level_kernel<1>" : (kf_big ? "merkle_level_kf_kernel<1>" : (kf_small ? "merkle_level_kf_kernel<0>" : "merkle_level_kernel<0>")),
               96.0 * cosets * (double)half);
  if (prc) hipLaunchKernelGGL(merkle_level_kernel<1>, dim3((half + threads - 1) / threads, cosets), dim3(threads), 0, st, in, out, m, prc);
  // >= 2 waves per SIMD on the whole chip (2 * 1024 SIMDs * 64 lanes): throughput placement; below: a SIMD sees a lone wave
  else if ((size_t)half * cosets >= (size_t)2 * 1024 * 64)
    hipLaunchKernelGGL(merkle_level_kf_kernel<1>, dim3((half + threads - 1) / threads, cosets), dim3(threads), 0, st, in, out, m);
  else if (threads == 256)
    hipLaunchKernelGGL(merkle_level_kf_kernel<0>, dim3((half + threads - 1) / threads, cosets), dim3(threads), 0, st, in, out, m);
  else hipLaunchKernelGGL(merkle_level_kernel<0>, dim3((half + threads - 1) / threads, cosets), dim3(threads), 0, st, in, out, m, prc);
}

}  // namespace p2
