// poseidon.hpp -- Poseidon-Goldilocks (width 12, 8 full + 22 partial rounds, x^7): the
// InnerHasher of KeccakGoldilocksConfig, used on the prove path only for
// `public_inputs_hash` (host) and inside PoseidonGate constraint evaluation (device).
// plonky2 0.2.2 hash/poseidon.rs, hash/poseidon_goldilocks.rs, hash/hashing.rs
// hash_n_to_m_no_pad (un-vendored; reference anchor: plonky2-backend/src/lib.rs:13).
//
// The 360 round constants are regenerated the way upstream made them -- `F::rand()` draws
// from `ChaCha8Rng::seed_from_u64(0)` (PCG32 seed expansion, gen_range widening-multiply
// sampling) -- and are pinned by plonky2's permutation test vectors in the test-suite.
#pragma once
#include "gl.hpp"
#include <cstring>

namespace p2 {

constexpr uint32_t POSEIDON_MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
constexpr uint32_t POSEIDON_MDS_DIAG0 = 8;

inline void poseidon_round_constants_host(gl_t out[360]) {
  auto rol = [](uint32_t x, int n) { return (x << n) | (x >> (32 - n)); };
  uint64_t state = 0;
  uint32_t key[8];
  for (int i = 0; i < 8; i++) {
    state = state * 6364136223846793005ULL + 11634580027462260723ULL;
    uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
    key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
  }
  uint32_t blk[16];
  auto block = [&](uint64_t ctr) {
    uint32_t s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    for (int i = 0; i < 8; i++) s[4 + i] = key[i];
    s[12] = (uint32_t)ctr;
    s[13] = (uint32_t)(ctr >> 32);
    s[14] = s[15] = 0;
    uint32_t w[16];
    memcpy(w, s, sizeof w);
    auto qr = [&](int a, int b, int c, int d) {
      w[a] += w[b]; w[d] = rol(w[d] ^ w[a], 16);
      w[c] += w[d]; w[b] = rol(w[b] ^ w[c], 12);
      w[a] += w[b]; w[d] = rol(w[d] ^ w[a], 8);
      w[c] += w[d]; w[b] = rol(w[b] ^ w[c], 7);
    };
    for (int r = 0; r < 4; r++) {
      qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
      qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) blk[i] = w[i] + s[i];
  };
  uint64_t ctr = 0;
  int pos = 16, got = 0;
  while (got < 360) {
    if (pos == 16) { block(ctr++); pos = 0; }
    uint32_t lo32 = blk[pos++];
    if (pos == 16) { block(ctr++); pos = 0; }
    uint32_t hi32 = blk[pos++];
    unsigned __int128 prod = (unsigned __int128)((uint64_t)lo32 | ((uint64_t)hi32 << 32)) * GL_P;
    if ((uint64_t)prod <= GL_P - 1) out[got++] = (uint64_t)(prod >> 64);
  }
}

// one round's linear layer: out[r] = sum_i st[(i + r) % 12] * CIRC[i] + st[r] * DIAG[r]
P2_HD void poseidon_mds(gl_t st[12]) {
  gl_t r[12];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int row = 0; row < 12; row++) {
    unsigned __int128 acc = 0;  // 13 terms < 2^70 each
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 12; i++) acc += (unsigned __int128)st[(i + row) % 12] * POSEIDON_MDS_CIRC[i];
    if (row == 0) acc += (unsigned __int128)st[0] * POSEIDON_MDS_DIAG0;
    r[row] = gl_reduce128((uint64_t)acc, (uint64_t)(acc >> 64));
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < 12; i++) st[i] = r[i];
}
#if defined(__HIPCC__)
// the same linear layer on the device (its outputs are congruent u64s, NOT canonical -- gl_add, the S-box and this layer take
// any u64; poseidon_permute_dev canonicalises the state once after the last round): the MDS entries are < 2^6, so a row is two 64-bit dot products over the
// 32-bit halves of the state (one v_mad_u64_u32 per term and half) and ONE reduction, instead of 13 128-bit
// multiply-accumulates: 30 layers x 12 rows make this the bulk of a permutation
__device__ __forceinline__ void poseidon_mds_dev(gl_t st[12]) {
  gl_t r[12];
#pragma unroll
  for (int row = 0; row < 12; row++) {
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
      const uint32_t k = POSEIDON_MDS_CIRC[i] + ((row == 0 && i == 0) ? POSEIDON_MDS_DIAG0 : 0);
      const gl_t v = st[(i + row) % 12];
      lo += (uint64_t)(uint32_t)v * k;
      hi += (v >> 32) * k;
    }
    const uint64_t l = lo + (hi << 32);
    r[row] = gl_reduce128_nc(l, (hi >> 32) + (l < lo));  // congruent, not canonical: the callers canonicalise once, at the end
  }
#pragma unroll
  for (int i = 0; i < 12; i++) st[i] = r[i];
}
#endif
P2_HD gl_t poseidon_sbox(gl_t x) {
  gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x2, x);
  return gl_mul(x4, x3);
}
// x^7 as SOME u64 congruent to it: for the permutations whose next step is the MDS layer's unreduced dot products (which take
// any u64 and end in one canonical reduction) -- the four canonicalisations of the S-box are 6 % of a permutation
P2_HD uint64_t poseidon_sbox_nc(gl_t x) {
  const uint64_t x2 = gl_mul_nc(x, x), x4 = gl_mul_nc(x2, x2), x3 = gl_mul_nc(x2, x);
  return gl_mul_nc(x4, x3);
}

// ---- the partial rounds, three linear layers at a time --------------------------------------------------------------
// plonky2 speeds the 22 partial rounds up by factoring the MDS matrix into sparse layers with FULL-SIZE coefficients (22
// general products per round instead of 144 small ones) -- a good trade on a CPU, a bad one here: a product by an MDS entry
// (< 2^6) is TWO v_mad_u64_u32 on the 32-bit halves of a word with the sum riding along in the addend, a general product
// with its reduction is 18 VALU (profiles/r06_poseidon.md has the count).  What costs on gfx950 is the REDUCTION of the
// twelve row sums after every layer.  In a partial round only word 0 meets the S-box, so words 1..11 pass from one
// linear layer straight into the next, and small-integer matrices multiply into small-integer matrices: with
// P = diag(0, 1, ..., 1) (drop word 0, it is replaced by the S-box output s)
//     V' = P (M V) + s1 e0,   V'' = P (M V') + s2 e0,   W = M V''
//     u1 = (M V)[0]                              -> s1 = sbox(u1 + c)
//     u2 = (M P M V)[0] + s1 M[0][0]             -> s2 = sbox(u2 + c')
//     W  = (M P M P M) V + s1 (M P M e0) + s2 (M e0)
// M P M P M has entries < 2^23 and row sums < 2^24.2: the two 64-bit sums per row still cannot overflow (2^24.2 * 2^32 +
// the s terms < 2^57).  Three layers cost 14 reductions and 386 multiply-adds instead of 36 and 864; the S-box inputs
// u + c are the ones the plain form feeds, so the PoseidonGate's constraints (gates.hpp) take the same route.
// The 23 linear layers from round 3 (after its S-boxes) to round 25 are 7 such triples and one pair.
struct PoseidonFusedTables {
  uint32_t m[12][12];      // M = circ(POSEIDON_MDS_CIRC) + diag(8, 0, ...)
  uint32_t mpm[12][12];    // M P M
  uint32_t mpmpm[12][12];  // M P M P M
};
constexpr PoseidonFusedTables poseidon_fused_tables() {
  PoseidonFusedTables t{};
  for (int r = 0; r < 12; r++)
    for (int c = 0; c < 12; c++) t.m[r][c] = POSEIDON_MDS_CIRC[(c - r + 12) % 12] + ((r == 0 && c == 0) ? POSEIDON_MDS_DIAG0 : 0);
  for (int r = 0; r < 12; r++)
    for (int c = 0; c < 12; c++) {
      uint64_t a = 0;
      for (int k = 1; k < 12; k++) a += (uint64_t)t.m[r][k] * t.m[k][c];
      t.mpm[r][c] = (uint32_t)a;
    }
  for (int r = 0; r < 12; r++)
    for (int c = 0; c < 12; c++) {
      uint64_t a = 0;
      for (int k = 1; k < 12; k++) a += (uint64_t)t.m[r][k] * t.mpm[k][c];
      t.mpmpm[r][c] = (uint32_t)a;
    }
  return t;
}
inline constexpr PoseidonFusedTables POSEIDON_FUSED = poseidon_fused_tables();
constexpr bool poseidon_fused_fits() {  // every row of M P M P M (+ the two S-box columns) sums below 2^25: sums of 32-bit halves stay below 2^57
  for (int r = 0; r < 12; r++) {
    uint64_t s = POSEIDON_FUSED.mpm[r][0] + POSEIDON_FUSED.m[r][0];
    for (int c = 0; c < 12; c++) s += POSEIDON_FUSED.mpmpm[r][c];
    if (s >= (1ull << 25)) return false;
  }
  return true;
}
static_assert(poseidon_fused_fits(), "fused Poseidon layers: a row sum would overflow the 64-bit half sums");

#if defined(__HIPCC__)
// APPS (3 or 2) linear layers and the APPS - 1 partial S-boxes between them; c1, c2: the word-0 constants of those rounds
// (poseidon_device_constants form).  In: any u64 words; out: congruent u64 words (not canonical).
template <int APPS>
__device__ __forceinline__ void poseidon_fused_dev(gl_t st[12], gl_t c1, gl_t c2) {
  static_assert(APPS == 2 || APPS == 3, "two or three layers");
  uint32_t vl[12], vh[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    vl[i] = (uint32_t)st[i];
    vh[i] = (uint32_t)(st[i] >> 32);
  }
  auto fin = [](uint64_t lo, uint64_t hi) {
    const uint64_t l = lo + (hi << 32);
    return gl_reduce128_nc(l, (hi >> 32) + (l < lo));
  };
  uint64_t lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    lo += (uint64_t)vl[i] * POSEIDON_FUSED.m[0][i];
    hi += (uint64_t)vh[i] * POSEIDON_FUSED.m[0][i];
  }
  const uint64_t s1 = poseidon_sbox_nc(gl_add(fin(lo, hi), c1));
  const uint32_t s1l = (uint32_t)s1, s1h = (uint32_t)(s1 >> 32);
  uint64_t s2 = 0;
  if constexpr (APPS == 3) {
    lo = (uint64_t)s1l * POSEIDON_FUSED.m[0][0];
    hi = (uint64_t)s1h * POSEIDON_FUSED.m[0][0];
#pragma unroll
    for (int i = 0; i < 12; i++) {
      lo += (uint64_t)vl[i] * POSEIDON_FUSED.mpm[0][i];
      hi += (uint64_t)vh[i] * POSEIDON_FUSED.mpm[0][i];
    }
    s2 = poseidon_sbox_nc(gl_add(fin(lo, hi), c2));
  }
  const uint32_t s2l = (uint32_t)s2, s2h = (uint32_t)(s2 >> 32);
#pragma unroll
  for (int row = 0; row < 12; row++) {
    if constexpr (APPS == 3) {
      lo = (uint64_t)s1l * POSEIDON_FUSED.mpm[row][0] + (uint64_t)s2l * POSEIDON_FUSED.m[row][0];
      hi = (uint64_t)s1h * POSEIDON_FUSED.mpm[row][0] + (uint64_t)s2h * POSEIDON_FUSED.m[row][0];
    } else {
      lo = (uint64_t)s1l * POSEIDON_FUSED.m[row][0];
      hi = (uint64_t)s1h * POSEIDON_FUSED.m[row][0];
    }
#pragma unroll
    for (int i = 0; i < 12; i++) {
      const uint32_t k = APPS == 3 ? POSEIDON_FUSED.mpmpm[row][i] : POSEIDON_FUSED.mpm[row][i];
      lo += (uint64_t)vl[i] * k;
      hi += (uint64_t)vh[i] * k;
    }
    st[row] = fin(lo, hi);
  }
}
// The whole permutation, one lane per state (hash/poseidon.rs permute).  prc = the handle's table in the
// poseidon_device_constants form (words 1..11 of a partial round have nothing to add), global memory, L1/L2 resident.
// Rounds are not unrolled (the 12-word state, the row sums and the S-box products already need ~70 VGPRs).
#ifndef P2_POSEIDON_FUSED
#define P2_POSEIDON_FUSED 1  // 0: every round through its own 12 x 12 layer (rounds 1-5; A/B measurements)
#endif
__device__ __forceinline__ void poseidon_permute_dev(gl_t st[12], const gl_t *__restrict__ prc) {
#if P2_POSEIDON_FUSED
#pragma unroll 1
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = poseidon_sbox_nc(gl_add(st[i], prc[12 * r + i]));
    poseidon_mds_dev(st);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) st[i] = poseidon_sbox_nc(gl_add(st[i], prc[36 + i]));
#pragma unroll 1
  for (int r0 = 3; r0 < 24; r0 += 3) {  // layers r0, r0 + 1, r0 + 2 and the partial S-boxes of rounds r0 + 1, r0 + 2; then round r0 + 3's
    poseidon_fused_dev<3>(st, prc[12 * (r0 + 1)], prc[12 * (r0 + 2)]);
    st[0] = poseidon_sbox_nc(gl_add(st[0], prc[12 * (r0 + 3)]));
  }
  poseidon_fused_dev<2>(st, prc[12 * 25], 0);  // layers 24, 25 and round 25's S-box
#pragma unroll 1
  for (int r = 26; r < 30; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = poseidon_sbox_nc(gl_add(st[i], prc[12 * r + i]));
    poseidon_mds_dev(st);
  }
#else
#pragma unroll 1
  for (int r = 0; r < 30; r++) {
    if (r < 4 || r >= 26) {
#pragma unroll
      for (int i = 0; i < 12; i++) st[i] = poseidon_sbox_nc(gl_add(st[i], prc[12 * r + i]));
    } else {
      st[0] = poseidon_sbox_nc(gl_add(st[0], prc[12 * r]));
    }
    poseidon_mds_dev(st);
  }
#endif
#pragma unroll
  for (int i = 0; i < 12; i++) st[i] = gl_canon(st[i]);
}
#endif

#if defined(__HIPCC__)
// ---- ONE Poseidon permutation spread over 12 lanes of a 16-lane group (four permutations per wave) --------------------
// For the latency-bound tops of Poseidon Merkle trees: one lane per node is 25 000 dependent-issue instructions per level (a lone
// wave issues one VALU instruction per 5.3-6.1 cycles: ~60 us); here lane i holds state word i, a round is the constant, the
// S-box (every lane computes it, lane 0 alone keeps it in the partial rounds), 22 ds_bpermute fetches of the other eleven words
// and this row's dot product with the circulant -- ~140 VALU instructions per round instead of ~850.
// nb[k - 1] = byte address (lane * 4) of the lane that holds word (i + k) % 12, k = 1..11.
struct PoseidonCoopLane {
  uint32_t nb[11];
  uint32_t i;   // word index of this lane (12..15: idle lanes, they mimic word 0 and nobody reads them)
};
__device__ __forceinline__ PoseidonCoopLane poseidon_coop_lane(uint32_t lane) {
  PoseidonCoopLane c;
  const uint32_t base = lane & ~15u, i0 = lane & 15u;
  c.i = i0 < 12 ? i0 : 0;
#pragma unroll
  for (int k = 1; k < 12; k++) c.nb[k - 1] = (base + (c.i + k) % 12) * 4;
  return c;
}
__device__ __forceinline__ gl_t poseidon_permute_coop(gl_t x, const gl_t *__restrict__ rc, const PoseidonCoopLane &c) {
  const uint32_t k0 = POSEIDON_MDS_CIRC[0] + (c.i == 0 ? POSEIDON_MDS_DIAG0 : 0u);
  gl_t cur = rc[c.i];
#pragma unroll 1
  for (int r = 0; r < 30; r++) {
    const gl_t nxt = rc[12 * (r < 29 ? r + 1 : 29) + c.i];  // next round's constant: in flight under this round's arithmetic
    x = gl_add(x, cur);
    cur = nxt;
    const gl_t sb = poseidon_sbox_nc(x);
    if (r < 4 || r >= 26 || c.i == 0) x = sb;
    uint64_t lo = (uint64_t)(uint32_t)x * k0, hi = (x >> 32) * k0;
#pragma unroll
    for (int k = 1; k < 12; k++) {
      const uint32_t vl = (uint32_t)__builtin_amdgcn_ds_bpermute((int)c.nb[k - 1], (int)(uint32_t)x);
      const uint32_t vh = (uint32_t)__builtin_amdgcn_ds_bpermute((int)c.nb[k - 1], (int)(uint32_t)(x >> 32));
      lo += (uint64_t)vl * POSEIDON_MDS_CIRC[k];
      hi += (uint64_t)vh * POSEIDON_MDS_CIRC[k];
    }
    const uint64_t l = lo + (hi << 32);
    x = gl_reduce128_nc(l, (hi >> 32) + (l < lo));
  }
  return gl_canon(x);
}
#endif

// Round constants for the DEVICE permutations of the hash paths (not the PoseidonGate, whose wires hold every round's S-box
// inputs as the plain form defines them): in the 22 partial rounds only word 0 meets the S-box, so the constants of words
// 1..11 commute with the linear layer -- M (s + e) = M s + M e -- and can ride along as a known offset o that is settled in
// the first full round after them.  out = rc except: partial round r: out[12r] = rc[12r] + o[0], out[12r + 1..11] = 0, then
// o = M (0, o[1..] + rc[12r + 1..]); round 26: out += o.  Same permutation, 242 modular additions fewer per call for code
// that skips the zeros (one lane per sponge); code that adds all twelve entries stays correct as it is.
inline void poseidon_device_constants(const gl_t rc[360], gl_t out[360]) {
  for (int i = 0; i < 360; i++) out[i] = rc[i];
  gl_t o[12] = {0};
  for (int r = 4; r < 26; r++) {
    gl_t e[12];
    out[12 * r] = gl_add(o[0], rc[12 * r]);
    e[0] = 0;
    for (int i = 1; i < 12; i++) {
      e[i] = gl_add(o[i], rc[12 * r + i]);
      out[12 * r + i] = 0;
    }
    poseidon_mds(e);
    for (int i = 0; i < 12; i++) o[i] = e[i];
  }
  for (int i = 0; i < 12; i++) out[12 * 26 + i] = gl_add(rc[12 * 26 + i], o[i]);
}

inline void poseidon_permute_host(gl_t st[12], const gl_t rc[360]) {
  for (int r = 0; r < 30; r++) {
    for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], rc[12 * r + i]);
    if (r < 4 || r >= 26)
      for (int i = 0; i < 12; i++) st[i] = poseidon_sbox(st[i]);
    else
      st[0] = poseidon_sbox(st[0]);
    poseidon_mds(st);
  }
}
// hash_n_to_m_no_pad (overwrite-mode sponge, rate 8), 4 outputs; [] -> 0^4 without permuting
inline void poseidon_hash_no_pad_host(const gl_t *in, size_t n, gl_t out[4], const gl_t rc[360]) {
  gl_t st[12] = {0};
  for (size_t off = 0; off < n; off += 8) {
    size_t k = n - off < 8 ? n - off : 8;
    for (size_t i = 0; i < k; i++) st[i] = in[off + i];
    poseidon_permute_host(st, rc);
  }
  for (int i = 0; i < 4; i++) out[i] = st[i];
}

}  // namespace p2
