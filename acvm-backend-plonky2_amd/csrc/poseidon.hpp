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
