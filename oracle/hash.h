/*
 * oracle/hash.h -- Keccak-256/25 hashing (KeccakHash<25>), the Keccak "hash
 * onion" permutation, Poseidon-Goldilocks (InnerHasher) and the Fiat-Shamir
 * Challenger.  TEST INFRASTRUCTURE ONLY (see gl.h).
 *
 * Restates plonky2 0.2.2 (absent from /root/reference):
 *   hash/keccak.rs     KeccakHash<N>, KeccakPermutation
 *   hash/poseidon.rs   Poseidon (width 12, 8 full + 22 partial rounds, x^7)
 *   hash/hashing.rs    hash_n_to_m_no_pad
 *   iop/challenger.rs  Challenger
 * Keccak = original Keccak (pad 0x01), i.e. keccak-hash 0.8.0 / tiny-keccak
 * 2.0.2 (plonky2-backend/Cargo.lock:735,1296).  In-tree anchor for the choice
 * of hasher: plonky2-backend/src/lib.rs:13 (C = KeccakGoldilocksConfig).
 */
#ifndef ORACLE_HASH_H
#define ORACLE_HASH_H
#include "gl.h"

/* Hasher of the Merkle trees, the challenger and the circuit digest: 0 = KeccakHash<25> (the reference's
 * KeccakGoldilocksConfig, plonky2-backend/src/lib.rs:13), 1 = PoseidonHash (PoseidonGoldilocksConfig: digests
 * are 4 field elements = 32 bytes, the challenger permutes with Poseidon).  A process-wide mode: every API
 * entry point sets it from the circuit it is handed before doing anything (the tests use one circuit at a time). */
extern int g_hasher;
#define DIGEST_MAX 32
#define DIGEST_BYTES (g_hasher ? 32 : 25)
typedef struct {
  uint8_t b[DIGEST_MAX];
} digest_t;

void keccak_f1600(uint64_t st[25]);
void keccak256(const uint8_t *in, size_t len, uint8_t out[32]);

/* KeccakHash<25>::hash_no_pad / hash_or_noop / two_to_one */
digest_t kh_hash_no_pad(const gl_t *elems, size_t n);
digest_t kh_hash_or_noop(const gl_t *elems, size_t n);
digest_t kh_two_to_one(const digest_t *l, const digest_t *r);
/* hash_pad: pad10*1 to a multiple of WIDTH=12 then hash_no_pad */
digest_t kh_hash_pad(const gl_t *elems, size_t n);
/* BytesHash<25>::to_vec: 7-byte little-endian chunks -> 4 field elements */
void digest_to_elems(const digest_t *d, gl_t out[4]);

/* KeccakPermutation::permute on 12 field elements */
void keccak_permutation(gl_t st[12]);
/* permutation of the configured hasher (g_hasher) */
void hasher_permutation(gl_t st[12]);

/* Poseidon */
void poseidon_init(void);
const gl_t *poseidon_round_constants(void); /* 360 */
void poseidon_permute(gl_t st[12]);
void poseidon_hash_no_pad(const gl_t *in, size_t n, gl_t out[4]);

/* Challenger<F, KeccakHash<25>> */
typedef struct {
  gl_t state[12];
  gl_t in[8];
  int n_in;
  gl_t out[8];
  int n_out;
} challenger_t;
void ch_init(challenger_t *c);
void ch_observe(challenger_t *c, gl_t e);
void ch_observe_many(challenger_t *c, const gl_t *e, size_t n);
void ch_observe_digest(challenger_t *c, const digest_t *d);
void ch_observe_cap(challenger_t *c, const digest_t *cap, size_t n);
void ch_observe_ext(challenger_t *c, ext_t e);
gl_t ch_get(challenger_t *c);
ext_t ch_get_ext(challenger_t *c);

#endif
