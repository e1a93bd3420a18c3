/*
 * oracle/poly.h -- NTT / LDE / Merkle tree / PolynomialBatch restatement.
 * TEST INFRASTRUCTURE ONLY (see gl.h).
 *
 * Restates plonky2 0.2.2 (absent from /root/reference):
 *   field/src/fft.rs               fft_classic, ifft
 *   field/src/polynomial/mod.rs    PolynomialValues::ifft, PolynomialCoeffs::lde / coset_fft
 *   plonky2/src/fri/oracle.rs      PolynomialBatch::from_values / from_coeffs / get_lde_values
 *   plonky2/src/hash/merkle_tree.rs MerkleTree::new / prove, MerkleCap
 * Call sites in-tree: plonky2-backend/src/actions/prove_action.rs:96 (prove),
 * circuit_translation/mod.rs:81 (build -> constants_sigmas commitment).
 */
#ifndef ORACLE_POLY_H
#define ORACLE_POLY_H
#include "gl.h"
#include "hash.h"

/* in-place radix-2 NTT, natural order in and out; log2 size = lg */
void ntt(gl_t *a, unsigned lg);
void intt(gl_t *a, unsigned lg);
/* values on shift*<w> from coefficients (len 2^lg) and back */
void coset_ntt(gl_t *a, unsigned lg, gl_t shift);
void coset_intt(gl_t *a, unsigned lg, gl_t shift);

typedef struct {
  size_t n_leaves;   /* power of two */
  unsigned cap_h;    /* cap height */
  size_t leaf_len;   /* elements per leaf */
  unsigned n_levels; /* log2(n_leaves) - cap_h + 1 digest levels (level 0 = leaf digests) */
  digest_t **levels; /* levels[l] has n_leaves >> l digests */
  digest_t *cap;     /* = levels[n_levels-1], 2^cap_h entries */
} merkle_t;
/* leaves: row-major [n_leaves][leaf_len] */
void merkle_build(merkle_t *t, const gl_t *leaves, size_t n_leaves, size_t leaf_len, unsigned cap_h);
void merkle_free(merkle_t *t);
/* siblings bottom-up; returns count = log2(n_leaves) - cap_h */
unsigned merkle_prove(const merkle_t *t, size_t idx, digest_t *siblings);
/* verify_merkle_proof_to_cap */
int merkle_verify(const gl_t *leaf, size_t leaf_len, size_t idx, const digest_t *cap, unsigned cap_h,
                  const digest_t *siblings, unsigned n_sib);

typedef struct {
  size_t ncols;
  unsigned d;        /* log2 degree n */
  unsigned rate_bits;
  gl_t *coeffs;      /* [ncols][n] natural order */
  gl_t *leaves;      /* [N][ncols], leaf index = bitrev(natural LDE index) */
  merkle_t tree;
} batch_t;
/* from_values: cols [ncols][n] of evaluations over <w_n> */
void batch_from_values(batch_t *b, const gl_t *vals, size_t ncols, unsigned d, unsigned rate_bits, unsigned cap_h);
/* from_coeffs: takes [ncols][n] coefficient columns */
void batch_from_coeffs(batch_t *b, const gl_t *coeffs, size_t ncols, unsigned d, unsigned rate_bits, unsigned cap_h);
void batch_free(batch_t *b);
void *big_malloc(size_t bytes); /* malloc; >= 4 MB: 2 MB-aligned + MADV_HUGEPAGE (free() releases it) */
/* big_malloc, or -- ORC_SPILL_DIR set and the buffer >= ORC_SPILL_MIN_GB -- a file-backed mapping there; spill_free releases either */
void *spill_malloc(size_t bytes);
void spill_free(void *p);
/* get_lde_values(i, step=1): pointer to the ncols values at natural LDE index i */
static inline const gl_t *batch_lde_row(const batch_t *b, size_t i) {
  return b->leaves + bitrev(i, b->d + b->rate_bits) * b->ncols;
}
/* evaluate coefficient column at an extension point */
ext_t poly_eval_ext(const gl_t *coeffs, size_t n, ext_t x);

#endif
