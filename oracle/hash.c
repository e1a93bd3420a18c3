/* oracle/hash.c -- see hash.h.  TEST INFRASTRUCTURE ONLY. */
#include "hash.h"
#include <string.h>
#include <stdlib.h>

/* ------------------------------------------------------------------------ */
/* Keccak-f[1600] (FIPS-202 permutation; pad byte 0x01 = pre-standard Keccak) */
static const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KPIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};

static inline uint64_t rol64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

void keccak_f1600(uint64_t st[25]) {
  uint64_t bc[5], t;
  for (int r = 0; r < 24; r++) {
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      t = bc[(i + 4) % 5] ^ rol64(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = KPIL[i];
      uint64_t b = st[j];
      st[j] = rol64(t, KROT[i]);
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= KRC[r];
  }
}

void keccak256(const uint8_t *in, size_t len, uint8_t out[32]) {
  uint64_t st[25];
  uint8_t blk[136];
  memset(st, 0, sizeof st);
  while (len >= 136) {
    for (int i = 0; i < 17; i++) {
      uint64_t w;
      memcpy(&w, in + 8 * i, 8);
      st[i] ^= w;
    }
    keccak_f1600(st);
    in += 136;
    len -= 136;
  }
  memset(blk, 0, 136);
  memcpy(blk, in, len);
  blk[len] ^= 0x01;
  blk[135] ^= 0x80;
  for (int i = 0; i < 17; i++) {
    uint64_t w;
    memcpy(&w, blk + 8 * i, 8);
    st[i] ^= w;
  }
  keccak_f1600(st);
  memcpy(out, st, 32);
}

/* hash/keccak.rs KeccakHash<N>::hash_no_pad: Keccak-256 over the canonical
 * little-endian u64 encodings, truncated to N = 25 bytes. */
int g_hasher = 0;
static digest_t from_elems(const gl_t e[4]) {
  digest_t d;
  for (int i = 0; i < 4; i++) memcpy(d.b + 8 * i, &e[i], 8);
  return d;
}
digest_t kh_hash_no_pad(const gl_t *elems, size_t n) {
  if (g_hasher) { /* PoseidonHash::hash_no_pad = hash_n_to_hash_no_pad (hash/hashing.rs) */
    gl_t o[4];
    poseidon_hash_no_pad(elems, n, o);
    return from_elems(o);
  }
  uint8_t h[32];
  digest_t d;
  /* field elements are stored canonical; host is little-endian */
  keccak256((const uint8_t *)elems, 8 * n, h);
  memcpy(d.b, h, DIGEST_BYTES);
  return d;
}
/* hash/hash_types.rs + plonk/config.rs Hasher::hash_or_noop */
digest_t kh_hash_or_noop(const gl_t *elems, size_t n) {
  if (g_hasher) { /* HashOut::from_partial when the leaf has at most 4 elements */
    if (n <= 4) {
      gl_t e[4] = {0, 0, 0, 0};
      for (size_t i = 0; i < n; i++) e[i] = elems[i];
      return from_elems(e);
    }
    return kh_hash_no_pad(elems, n);
  }
  if (8 * n <= DIGEST_BYTES) {
    digest_t d;
    memset(d.b, 0, DIGEST_BYTES);
    memcpy(d.b, elems, 8 * n);
    return d;
  }
  return kh_hash_no_pad(elems, n);
}
digest_t kh_two_to_one(const digest_t *l, const digest_t *r) {
  if (g_hasher) { /* hash/hashing.rs compress: state = l || r || 0^4, one permutation, first 4 elements */
    gl_t st[12];
    memset(st, 0, sizeof st);
    memcpy(st, l->b, 32);
    memcpy(st + 4, r->b, 32);
    poseidon_permute(st);
    return from_elems(st);
  }
  uint8_t buf[2 * 25], h[32];
  digest_t d;
  memcpy(buf, l->b, DIGEST_BYTES);
  memcpy(buf + DIGEST_BYTES, r->b, DIGEST_BYTES);
  keccak256(buf, sizeof buf, h);
  memcpy(d.b, h, DIGEST_BYTES);
  return d;
}
digest_t kh_hash_pad(const gl_t *elems, size_t n) {
  size_t m = n + 1;
  while ((m + 1) % 8 != 0) m++; /* pad10*1 to the sponge RATE (8), pinned by tests/test_reference_proofs.py */
  m++;
  gl_t *p = (gl_t *)calloc(m, sizeof(gl_t));
  if (n) memcpy(p, elems, n * sizeof(gl_t));
  p[n] = 1;
  p[m - 1] = 1;
  digest_t d = kh_hash_no_pad(p, m);
  free(p);
  return d;
}
void digest_to_elems(const digest_t *d, gl_t out[4]) {
  if (g_hasher) { /* HashOut: its four elements */
    memcpy(out, d->b, 32);
    return;
  }
  for (int i = 0; i < 4; i++) {
    uint64_t w = 0;
    int len = (i < 3) ? 7 : 4;
    memcpy(&w, d->b + 7 * i, len);
    out[i] = w; /* < 2^56 < p: canonical */
  }
}

/* hash/keccak.rs KeccakPermutation::permute: serialise the 12-element state,
 * iterate Keccak-256 ("hash onion"), parse LE u64 words, reject words >= p,
 * the first 12 accepted words are the new state. */
void keccak_permutation(gl_t st[12]) {
  uint8_t buf[96], h[32];
  memcpy(buf, st, 96);
  keccak256(buf, 96, h);
  int got = 0;
  for (;;) {
    for (int i = 0; i < 4 && got < 12; i++) {
      uint64_t w;
      memcpy(&w, h + 8 * i, 8);
      if (w < GL_P) st[got++] = w;
    }
    if (got == 12) break;
    uint8_t h2[32];
    keccak256(h, 32, h2);
    memcpy(h, h2, 32);
  }
}

/* ------------------------------------------------------------------------ */
/* Poseidon.  The 360 round constants are not stored: upstream generated them
 * as `F::rand()` draws from `ChaCha8Rng::seed_from_u64(0)` (rand_chacha 0.3 /
 * rand 0.8: seed expanded with PCG32; F::rand = gen_range(0..ORDER) = Lemire
 * widening-multiply sampling).  We regenerate them and pin the result with the
 * known-answer vector of poseidon(0^12) -- see tests/test_oracle_primitives.py. */
static gl_t PRC[360];
static int poseidon_ready = 0;
static const uint64_t MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const uint64_t MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static inline uint32_t rol32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d)                                                                                                 \
  a += b; d ^= a; d = rol32(d, 16);                                                                                    \
  c += d; b ^= c; b = rol32(b, 12);                                                                                    \
  a += b; d ^= a; d = rol32(d, 8);                                                                                     \
  c += d; b ^= c; b = rol32(b, 7);
static void chacha8_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
  uint32_t s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
  for (int i = 0; i < 8; i++) s[4 + i] = key[i];
  s[12] = (uint32_t)counter;
  s[13] = (uint32_t)(counter >> 32);
  s[14] = s[15] = 0;
  uint32_t w[16];
  memcpy(w, s, sizeof w);
  for (int r = 0; r < 4; r++) {
    QR(w[0], w[4], w[8], w[12]) QR(w[1], w[5], w[9], w[13]) QR(w[2], w[6], w[10], w[14]) QR(w[3], w[7], w[11], w[15])
    QR(w[0], w[5], w[10], w[15]) QR(w[1], w[6], w[11], w[12]) QR(w[2], w[7], w[8], w[13]) QR(w[3], w[4], w[9], w[14])
  }
  for (int i = 0; i < 16; i++) out[i] = w[i] + s[i];
}

void poseidon_init(void) {
  if (poseidon_ready) return;
  /* rand_core SeedableRng::seed_from_u64(0): PCG32 stream */
  uint64_t state = 0;
  uint32_t key[8];
  for (int i = 0; i < 8; i++) {
    state = state * 6364136223846793005ULL + 11634580027462260723ULL;
    uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
  }
  uint32_t blk[16];
  uint64_t ctr = 0;
  int pos = 16, got = 0;
  while (got < 360) {
    uint32_t lo32, hi32;
    if (pos == 16) { chacha8_block(key, ctr++, blk); pos = 0; }
    lo32 = blk[pos++];
    if (pos == 16) { chacha8_block(key, ctr++, blk); pos = 0; }
    hi32 = blk[pos++];
    uint64_t v = (uint64_t)lo32 | ((uint64_t)hi32 << 32);
    u128 prod = (u128)v * GL_P;
    uint64_t hi = (uint64_t)(prod >> 64), lo = (uint64_t)prod;
    if (lo <= GL_P - 1) PRC[got++] = hi;
  }
  poseidon_ready = 1;
}
const gl_t *poseidon_round_constants(void) {
  poseidon_init();
  return PRC;
}
static inline gl_t sbox7(gl_t x) {
  gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x2, x);
  return gl_mul(x4, x3);
}
static void mds_layer(gl_t st[12]) {
  gl_t r[12];
  for (int row = 0; row < 12; row++) {
    u128 acc = 0; /* 12 * 2^64 * 41 < 2^128 */
    for (int i = 0; i < 12; i++) acc += (u128)st[(i + row) % 12] * MDS_CIRC[i];
    acc += (u128)st[row] * MDS_DIAG[row];
    r[row] = gl_reduce128(acc);
  }
  memcpy(st, r, sizeof r);
}
void poseidon_permute(gl_t st[12]) {
  poseidon_init();
  for (int r = 0; r < 30; r++) {
    for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], PRC[12 * r + i]);
    if (r < 4 || r >= 26)
      for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
    else
      st[0] = sbox7(st[0]);
    mds_layer(st);
  }
}
/* hash/hashing.rs hash_n_to_m_no_pad (overwrite-mode sponge, rate 8) */
void poseidon_hash_no_pad(const gl_t *in, size_t n, gl_t out[4]) {
  gl_t st[12] = {0};
  for (size_t off = 0; off < n; off += 8) {
    size_t k = n - off < 8 ? n - off : 8;
    for (size_t i = 0; i < k; i++) st[i] = in[off + i];
    poseidon_permute(st);
  }
  for (int i = 0; i < 4; i++) out[i] = st[i];
}

/* ------------------------------------------------------------------------ */
/* iop/challenger.rs */
void ch_init(challenger_t *c) { memset(c, 0, sizeof *c); }
/* the sponge permutation of the configured hasher (Challenger<F, H>: H::Permutation) */
void hasher_permutation(gl_t st[12]) {
  if (g_hasher) poseidon_permute(st);
  else keccak_permutation(st);
}
static void ch_duplex(challenger_t *c) {
  for (int i = 0; i < c->n_in; i++) c->state[i] = c->in[i];
  c->n_in = 0;
  hasher_permutation(c->state);
  for (int i = 0; i < 8; i++) c->out[i] = c->state[i];
  c->n_out = 8;
}
void ch_observe(challenger_t *c, gl_t e) {
  c->n_out = 0;
  c->in[c->n_in++] = e;
  if (c->n_in == 8) ch_duplex(c);
}
void ch_observe_many(challenger_t *c, const gl_t *e, size_t n) {
  for (size_t i = 0; i < n; i++) ch_observe(c, e[i]);
}
void ch_observe_digest(challenger_t *c, const digest_t *d) {
  gl_t e[4];
  digest_to_elems(d, e);
  ch_observe_many(c, e, 4);
}
void ch_observe_cap(challenger_t *c, const digest_t *cap, size_t n) {
  for (size_t i = 0; i < n; i++) ch_observe_digest(c, &cap[i]);
}
void ch_observe_ext(challenger_t *c, ext_t e) {
  ch_observe(c, e.c0);
  ch_observe(c, e.c1);
}
gl_t ch_get(challenger_t *c) {
  if (c->n_in != 0 || c->n_out == 0) ch_duplex(c);
  return c->out[--c->n_out];
}
ext_t ch_get_ext(challenger_t *c) {
  gl_t a = ch_get(c);
  gl_t b = ch_get(c);
  return ext_make(a, b);
}
