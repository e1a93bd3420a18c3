// keccak.hpp -- Keccak-f[1600] with the state held in registers (25 x u64 = 50
// VGPRs on gfx950), Keccak-256 for host-side transcript work, and the 25-byte
// truncated digests of plonky2's KeccakHash<25> (hash/keccak.rs in the
// un-vendored plonky2 0.2.2; the reference selects it at
// plonky2-backend/src/lib.rs:13 `type C = KeccakGoldilocksConfig`).
// Original Keccak padding (0x01 .. 0x80), not SHA-3.
//
// Integer-ALU bound: ~24 x (theta 50 + rho/pi 24 rot + chi 75 + iota) 64-bit ops
// per permutation; rotates compile to v_alignbit_b32 pairs.  One lane owns one
// sponge; lanes of a wave hash 64 independent leaves.
#pragma once
#include "gl.hpp"

namespace p2 {

// digests are stored as 4 x u64 (32 B) in device memory; only the first 25
// bytes are significant and the 7 top bytes of word 3 are kept zero.
struct alignas(32) dig_t {
  uint64_t w[4];
};

P2_HD uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 forms: three-input bit ops (v_bitop3_b32: xor3 = 0x96, chi a^(~b&c) = 0xD2) and
// v_alignbit_b32 rotates on the 32-bit halves -- 180 VALU per round instead of ~290.
__device__ __forceinline__ uint64_t kx3(uint64_t a, uint64_t b, uint64_t c) {
  uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0x96);
  uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0x96);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t kchi(uint64_t a, uint64_t b, uint64_t c) {
  uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0xD2);
  uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0xD2);
  return ((uint64_t)hi << 32) | lo;
}
template <int N>
__device__ __forceinline__ uint64_t krot(uint64_t x) {
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  if constexpr (N == 0) {
    return x;
  } else if constexpr (N == 32) {
    return ((uint64_t)lo << 32) | hi;
  } else if constexpr (N < 32) {
    uint32_t nh = __builtin_amdgcn_alignbit(hi, lo, 32 - N), nl = __builtin_amdgcn_alignbit(lo, hi, 32 - N);
    return ((uint64_t)nh << 32) | nl;
  } else {
    uint32_t nh = __builtin_amdgcn_alignbit(lo, hi, 64 - N), nl = __builtin_amdgcn_alignbit(hi, lo, 64 - N);
    return ((uint64_t)nh << 32) | nl;
  }
}
#define P2_KX3(a, b, c) kx3(a, b, c)
#define P2_KCHI(a, b, c) kchi(a, b, c)
#define P2_KROT(x, n) krot<n>(x)
#else
#define P2_KX3(a, b, c) ((a) ^ (b) ^ (c))
#define P2_KCHI(a, b, c) ((a) ^ (~(b) & (c)))
#define P2_KROT(x, n) rotl64(x, n)
#endif

#define P2_KECCAK_ROUND(RC)                                                                                            \
  {                                                                                                                    \
    uint64_t c0 = P2_KX3(P2_KX3(a00, a05, a10), a15, a20), c1 = P2_KX3(P2_KX3(a01, a06, a11), a16, a21),               \
             c2 = P2_KX3(P2_KX3(a02, a07, a12), a17, a22), c3 = P2_KX3(P2_KX3(a03, a08, a13), a18, a23),               \
             c4 = P2_KX3(P2_KX3(a04, a09, a14), a19, a24);                                                             \
    uint64_t r0 = P2_KROT(c0, 1), r1 = P2_KROT(c1, 1), r2 = P2_KROT(c2, 1), r3 = P2_KROT(c3, 1), r4 = P2_KROT(c4, 1);  \
    /* theta: a[x][y] ^= c[x-1] ^ rot(c[x+1], 1) */                                                                    \
    a00 = P2_KX3(a00, c4, r1); a05 = P2_KX3(a05, c4, r1); a10 = P2_KX3(a10, c4, r1); a15 = P2_KX3(a15, c4, r1);        \
    a20 = P2_KX3(a20, c4, r1);                                                                                         \
    a01 = P2_KX3(a01, c0, r2); a06 = P2_KX3(a06, c0, r2); a11 = P2_KX3(a11, c0, r2); a16 = P2_KX3(a16, c0, r2);        \
    a21 = P2_KX3(a21, c0, r2);                                                                                         \
    a02 = P2_KX3(a02, c1, r3); a07 = P2_KX3(a07, c1, r3); a12 = P2_KX3(a12, c1, r3); a17 = P2_KX3(a17, c1, r3);        \
    a22 = P2_KX3(a22, c1, r3);                                                                                         \
    a03 = P2_KX3(a03, c2, r4); a08 = P2_KX3(a08, c2, r4); a13 = P2_KX3(a13, c2, r4); a18 = P2_KX3(a18, c2, r4);        \
    a23 = P2_KX3(a23, c2, r4);                                                                                         \
    a04 = P2_KX3(a04, c3, r0); a09 = P2_KX3(a09, c3, r0); a14 = P2_KX3(a14, c3, r0); a19 = P2_KX3(a19, c3, r0);        \
    a24 = P2_KX3(a24, c3, r0);                                                                                         \
    /* rho + pi: B[y][2x+3y] = rot(A[x][y]) */                                                                         \
    uint64_t b00 = a00, b10 = P2_KROT(a01, 1), b20 = P2_KROT(a02, 62), b05 = P2_KROT(a03, 28), b15 = P2_KROT(a04, 27); \
    uint64_t b16 = P2_KROT(a05, 36), b01 = P2_KROT(a06, 44), b11 = P2_KROT(a07, 6), b21 = P2_KROT(a08, 55),            \
             b06 = P2_KROT(a09, 20);                                                                                   \
    uint64_t b07 = P2_KROT(a10, 3), b17 = P2_KROT(a11, 10), b02 = P2_KROT(a12, 43), b12 = P2_KROT(a13, 25),            \
             b22 = P2_KROT(a14, 39);                                                                                   \
    uint64_t b23 = P2_KROT(a15, 41), b08 = P2_KROT(a16, 45), b18 = P2_KROT(a17, 15), b03 = P2_KROT(a18, 21),           \
             b13 = P2_KROT(a19, 8);                                                                                    \
    uint64_t b14 = P2_KROT(a20, 18), b24 = P2_KROT(a21, 2), b09 = P2_KROT(a22, 61), b19 = P2_KROT(a23, 56),            \
             b04 = P2_KROT(a24, 14);                                                                                   \
    /* chi */                                                                                                          \
    a00 = P2_KCHI(b00, b01, b02); a01 = P2_KCHI(b01, b02, b03); a02 = P2_KCHI(b02, b03, b04);                          \
    a03 = P2_KCHI(b03, b04, b00); a04 = P2_KCHI(b04, b00, b01);                                                        \
    a05 = P2_KCHI(b05, b06, b07); a06 = P2_KCHI(b06, b07, b08); a07 = P2_KCHI(b07, b08, b09);                          \
    a08 = P2_KCHI(b08, b09, b05); a09 = P2_KCHI(b09, b05, b06);                                                        \
    a10 = P2_KCHI(b10, b11, b12); a11 = P2_KCHI(b11, b12, b13); a12 = P2_KCHI(b12, b13, b14);                          \
    a13 = P2_KCHI(b13, b14, b10); a14 = P2_KCHI(b14, b10, b11);                                                        \
    a15 = P2_KCHI(b15, b16, b17); a16 = P2_KCHI(b16, b17, b18); a17 = P2_KCHI(b17, b18, b19);                          \
    a18 = P2_KCHI(b18, b19, b15); a19 = P2_KCHI(b19, b15, b16);                                                        \
    a20 = P2_KCHI(b20, b21, b22); a21 = P2_KCHI(b21, b22, b23); a22 = P2_KCHI(b22, b23, b24);                          \
    a23 = P2_KCHI(b23, b24, b20); a24 = P2_KCHI(b24, b20, b21);                                                        \
    a00 ^= (RC);                                                                                                       \
  }

#if defined(__HIP_DEVICE_COMPILE__)
// Keccak-f with the sponge state in FIXED physical registers v[P2_KF_BASE ...) for the whole life of a kernel: the 24 rounds
// unrolled, registers assigned so that no v_bitop3_b32 has its three sources in one VGPR bank, no compiler-inserted wait
// states (gen_keccak_fixed.py; 11.3 vs 10.3 Gperm/s for the compiler-allocated ordered stream, 9.7 for hipcc's own code).
// A kernel that uses these macros carries P2_KF_KERNEL_ATTR: hipcc then allocates below
// P2_KF_BASE only, and lane i of the state is v[P2_KF_BASE + 2i] (low half), v[P2_KF_BASE + 2i + 1] (high half) between
// the statements below, which are all `asm volatile` (kept in program order).
#include "keccak_fixed.inc"
#define P2_KF_SET(i, lo, hi)                                                                                     \
  asm volatile("v_mov_b32 v%c2, %0\n v_mov_b32 v%c3, %1" ::"v"(lo), "v"(hi), "n"(P2_KF_BASE + 2 * (i)), "n"(P2_KF_BASE + 2 * (i) + 1))
#define P2_KF_XOR(i, lo, hi)                                                                                     \
  asm volatile("v_xor_b32 v%c2, v%c2, %0\n v_xor_b32 v%c3, v%c3, %1" ::"v"(lo), "v"(hi), "n"(P2_KF_BASE + 2 * (i)), "n"(P2_KF_BASE + 2 * (i) + 1))
#define P2_KF_GET(i, lo, hi)                                                                                     \
  asm volatile("v_mov_b32 %0, v%c2\n v_mov_b32 %1, v%c3" : "=v"(lo), "=v"(hi) : "n"(P2_KF_BASE + 2 * (i)), "n"(P2_KF_BASE + 2 * (i) + 1))
// (on gfx950's unified register file hipcc gives a kernel TWICE the number it is asked for -- measured: 18 -> v0..v35,
// 20 -> v0..v39, 22 -> v0..v43 -- so BASE / 2 keeps it below P2_KF_BASE; `make kf-check` verifies it on the built code)
#define P2_KF_KERNEL_ATTR __attribute__((amdgpu_num_vgpr(P2_KF_BASE / 2)))
#else
// host pass of hipcc: the kernels that use these are only parsed
#define P2_KF_KERNEL_ATTR
#define P2_KF_SET(i, lo, hi) ((void)(lo), (void)(hi))
#define P2_KF_XOR(i, lo, hi) ((void)(lo), (void)(hi))
#define P2_KF_GET(i, lo, hi) ((lo) = 0, (hi) = 0)
#define P2_KECCAK_FIXED_PERMUTE() ((void)0)
#define P2_KECCAK_FIXED_PERMUTE_PH(PH) ((void)0)
#endif

// st[x + 5y]
P2_HD void keccak_f1600(uint64_t st[25]) {
  const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
      0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
      0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  uint64_t a00 = st[0], a01 = st[1], a02 = st[2], a03 = st[3], a04 = st[4], a05 = st[5], a06 = st[6], a07 = st[7],
           a08 = st[8], a09 = st[9], a10 = st[10], a11 = st[11], a12 = st[12], a13 = st[13], a14 = st[14],
           a15 = st[15], a16 = st[16], a17 = st[17], a18 = st[18], a19 = st[19], a20 = st[20], a21 = st[21],
           a22 = st[22], a23 = st[23], a24 = st[24];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 2
#endif
  for (int r = 0; r < 24; r++) P2_KECCAK_ROUND(RC[r])
  st[0] = a00; st[1] = a01; st[2] = a02; st[3] = a03; st[4] = a04; st[5] = a05; st[6] = a06; st[7] = a07;
  st[8] = a08; st[9] = a09; st[10] = a10; st[11] = a11; st[12] = a12; st[13] = a13; st[14] = a14; st[15] = a15;
  st[16] = a16; st[17] = a17; st[18] = a18; st[19] = a19; st[20] = a20; st[21] = a21; st[22] = a22; st[23] = a23;
  st[24] = a24;
}

// Keccak-256 over `nwords` little-endian u64 words (every message on the prove
// path is a whole number of words).  Host + device.
P2_HD void keccak256_words(const uint64_t *in, size_t nwords, uint64_t out[4]) {
  uint64_t st[25];
  for (int i = 0; i < 25; i++) st[i] = 0;
  size_t off = 0;
  while (nwords - off >= 17) {
    for (int i = 0; i < 17; i++) st[i] ^= in[off + i];
    keccak_f1600(st);
    off += 17;
  }
  size_t rem = nwords - off;
  for (size_t i = 0; i < rem; i++) st[i] ^= in[off + i];
  st[rem] ^= 0x01ULL;
  st[16] ^= 0x8000000000000000ULL;
  keccak_f1600(st);
  for (int i = 0; i < 4; i++) out[i] = st[i];
}

P2_HD dig_t dig_from_state(const uint64_t st[4]) {
  dig_t d;
  d.w[0] = st[0];
  d.w[1] = st[1];
  d.w[2] = st[2];
  d.w[3] = st[3] & 0xFFULL;
  return d;
}

// KeccakHash<25>::two_to_one: Keccak-256(left[25] || right[25])[..25]
P2_HD dig_t keccak_two_to_one(const dig_t &l, const dig_t &r) {
  uint64_t st[25];
  for (int i = 0; i < 25; i++) st[i] = 0;
  st[0] = l.w[0];
  st[1] = l.w[1];
  st[2] = l.w[2];
  st[3] = (l.w[3] & 0xFFULL) | (r.w[0] << 8);
  st[4] = (r.w[0] >> 56) | (r.w[1] << 8);
  st[5] = (r.w[1] >> 56) | (r.w[2] << 8);
  st[6] = (r.w[2] >> 56) | ((r.w[3] & 0xFFULL) << 8) | (0x01ULL << 16);
  st[16] = 0x8000000000000000ULL;
  keccak_f1600(st);
  return dig_from_state(st);
}

// ---- ONE Keccak-f spread over 25 lanes (lane L = x + 5y of a 32-lane half; two permutations per wave) -----------------
// For the top of a Merkle tree, where a level has fewer nodes than the chip has SIMDs and its cost is the LATENCY of one
// permutation: a lone wave issues one VALU instruction per 5.3-6.1 cycles however independent they are
// (profiles/r03_ubench.txt), so 4 359 instructions are 7.4 us per level; here a round is 16 VALU instructions and three
// ds_bpermute stages (theta column parity | theta row neighbours | the three rho-rotated lanes chi needs, fetched straight
// from where pi takes them): 3.5 us per permutation with <= 2 waves per CU, 4.5 with one per SIMD (the CU's LDS crossbar
// serves ~1 bpermute per 6 cycles: 8.9 with 8 waves) -- scratch/ubench/coop2.hip, checked there against the fixed-register form.
struct KeccakCoopLane {
  uint32_t col[4];       // byte addresses (lane * 4) of the four column mates (y + 1 .. y + 4)
  uint32_t rm, rp;       // row neighbours x - 1, x + 1
  uint32_t b0, b1, b2;   // pi^-1 of (x, y), (x + 1, y), (x + 2, y)
  uint32_t s, swap;      // rho: v_alignbit shift, and whether the halves trade places first (rotation >= 32, or 0)
  uint32_t m0;           // all ones in lane 0 (iota)
};
P2_HD KeccakCoopLane keccak_coop_lane(uint32_t lane) {
  const int rot[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  const uint32_t base = lane & 32u, L0 = lane & 31u, L = L0 < 25 ? L0 : 0;
  const int x = L % 5, y = L / 5;
  KeccakCoopLane c;
  for (int i = 0; i < 4; i++) c.col[i] = (base + (L + 5 * (i + 1)) % 25) * 4;
  c.rm = (base + (x + 4) % 5 + 5 * y) * 4;
  c.rp = (base + (x + 1) % 5 + 5 * y) * 4;
  // B[X][Y] = rot(A[x'][y']) with y' = X and 2x' + 3y' = Y (mod 5), i.e. x' = 3 (Y - 3X) mod 5
  auto src = [&](int X, int Y) { return (uint32_t)((((Y - 3 * X) % 5 + 5) % 5) * 3 % 5 + 5 * X); };
  c.b0 = (base + src(x, y)) * 4;
  c.b1 = (base + src((x + 1) % 5, y)) * 4;
  c.b2 = (base + src((x + 2) % 5, y)) * 4;
  const int r = rot[L];
  c.swap = (r == 0 || r >= 32) ? 1u : 0u;   // (shift 0 of the alignbit pair below IS a swap: rotation 0 swaps twice)
  c.s = (uint32_t)(32 - (r & 31)) & 31u;
  c.m0 = L0 == 0 ? 0xFFFFFFFFu : 0u;
  return c;
}
// (lo, hi) = this lane's 64-bit word of the state; lanes 25..31 of a half carry garbage nobody reads
__device__ __forceinline__ void keccak_f1600_coop(uint32_t &lo, uint32_t &hi, const KeccakCoopLane &c) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
      0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
      0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  auto bp = [](uint32_t addr, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)v); };
  auto x3 = [](uint32_t a, uint32_t b, uint32_t d) { return __builtin_amdgcn_bitop3_b32(a, b, d, 0x96); };
#pragma unroll
  for (int r = 0; r < 24; r++) {
    const uint32_t t0 = bp(c.col[0], lo), t1 = bp(c.col[1], lo), t2 = bp(c.col[2], lo), t3 = bp(c.col[3], lo);
    const uint32_t u0 = bp(c.col[0], hi), u1 = bp(c.col[1], hi), u2 = bp(c.col[2], hi), u3 = bp(c.col[3], hi);
    const uint32_t cl = x3(x3(lo, t0, t1), t2, t3), ch = x3(x3(hi, u0, u1), u2, u3);
    const uint32_t ml = bp(c.rm, cl), mh = bp(c.rm, ch), pl = bp(c.rp, cl), ph = bp(c.rp, ch);
    lo = x3(lo, ml, __builtin_amdgcn_alignbit(pl, ph, 31));
    hi = x3(hi, mh, __builtin_amdgcn_alignbit(ph, pl, 31));
    const uint32_t L_ = c.swap ? hi : lo, H_ = c.swap ? lo : hi;
    const uint32_t rl = __builtin_amdgcn_alignbit(L_, H_, c.s), rh = __builtin_amdgcn_alignbit(H_, L_, c.s);
    const uint32_t a0 = bp(c.b0, rl), a1 = bp(c.b1, rl), a2 = bp(c.b2, rl);
    const uint32_t h0 = bp(c.b0, rh), h1 = bp(c.b1, rh), h2 = bp(c.b2, rh);
    lo = __builtin_amdgcn_bitop3_b32(a0, a1, a2, 0xD2) ^ ((uint32_t)RC[r] & c.m0);
    hi = __builtin_amdgcn_bitop3_b32(h0, h1, h2, 0xD2) ^ ((uint32_t)(RC[r] >> 32) & c.m0);
  }
#else
  (void)lo, (void)hi, (void)c;  // host pass of hipcc: parsed only
#endif
}

// BytesHash<25>::to_vec: 7-byte little-endian chunks -> 4 field elements
P2_HD void dig_to_elems(const dig_t &d, gl_t out[4]) {
  const uint64_t M56 = 0x00FFFFFFFFFFFFFFULL;
  out[0] = d.w[0] & M56;
  out[1] = ((d.w[0] >> 56) | (d.w[1] << 8)) & M56;
  out[2] = ((d.w[1] >> 48) | (d.w[2] << 16)) & M56;
  out[3] = ((d.w[2] >> 40) | ((d.w[3] & 0xFF) << 24)) & 0xFFFFFFFFULL;
}

// KeccakPermutation::permute ("hash onion" with rejection sampling)
// NEED: how many words of the new state the caller will look at (12 = the whole permutation).  The onion produces them four
// at a time, one Keccak-f per layer: the proof-of-work check reads word 7 only, i.e. two layers instead of three (words
// 8..11 of `st` are then left as they were -- NOT a permutation of the state any more, only its first NEED words)
template <int NEED = 12>
P2_HD void keccak_permutation12(gl_t st[12]) {
  uint64_t h[4];
  keccak256_words(st, 12, h);
  int got = 0;
  for (;;) {
    for (int i = 0; i < 4 && got < NEED; i++)
      if (h[i] < GL_P) st[got++] = h[i];
    if (got >= NEED) break;
    uint64_t h2[4];
    keccak256_words(h, 4, h2);
    for (int i = 0; i < 4; i++) h[i] = h2[i];
  }
}

}  // namespace p2
