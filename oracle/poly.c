/* oracle/poly.c -- see poly.h.  TEST INFRASTRUCTURE ONLY. */
#include "poly.h"
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <stdio.h>
#include <omp.h>
static int orc_trace(void) { static int v = -1; if (v < 0) { const char *e = getenv("ORC_TRACE"); v = e && *e == '1'; } return v; }
#define TR(label) do { if (orc_trace()) { double t_ = omp_get_wtime(); fprintf(stderr, "[oracle] %s %.3f s\n", label, t_ - tr0_); tr0_ = t_; } } while (0)

/* root tables shared by the threads of a batch: at most one per (size, root) pair is ever built; a handful
 * of pairs exist per proof (forward / inverse at d and d + rate_bits, the FRI sizes) */
#define TW_CACHE 64
static struct { unsigned lg; gl_t root; gl_t *tw; } g_tw[TW_CACHE];
static int g_tw_n;
static const gl_t *twiddles(unsigned lg, gl_t root) {
  const gl_t *found = NULL;
#pragma omp critical(oracle_twiddles)
  {
    for (int i = 0; i < g_tw_n && !found; i++)
      if (g_tw[i].lg == lg && g_tw[i].root == root) found = g_tw[i].tw;
    if (!found) {
      /* per layer s (butterfly span m = 2^s, half = m / 2) the twiddles root^(j n/m), j < half, CONTIGUOUS at offset
       * half - 1: the flat table root^i read with stride n/m maps a whole layer onto one L1 set */
      size_t n = (size_t)1 << lg;
      gl_t *flat = (gl_t *)malloc(sizeof(gl_t) * (n / 2 + 1));
      flat[0] = 1;
      for (size_t i = 1; i < n / 2; i++) flat[i] = gl_mul(flat[i - 1], root);
      gl_t *tw = (gl_t *)malloc(sizeof(gl_t) * (n + 1));
      for (unsigned s2 = 1; s2 <= lg; s2++) {
        size_t half = (size_t)1 << (s2 - 1), stride = n >> s2;
        for (size_t j = 0; j < half; j++) tw[half - 1 + j] = flat[j * stride];
      }
      free(flat);
      if (g_tw_n == TW_CACHE) { /* full: drop the oldest entry (never hit by the tests; no reader holds it across calls
                                   of a different size) */
        free(g_tw[0].tw);
        memmove(&g_tw[0], &g_tw[1], sizeof g_tw[0] * (TW_CACHE - 1));
        g_tw_n--;
      }
      g_tw[g_tw_n].lg = lg; g_tw[g_tw_n].root = root; g_tw[g_tw_n].tw = tw;
      g_tw_n++;
      found = tw;
    }
  }
  return found;
}

/* fft_classic: bit-reverse the input, then radix-2 DIT butterflies with the
 * forward root w = primitive_root_of_unity(lg): out[k] = sum_j a[j] w^(jk). */
static void ntt_root(gl_t *a, unsigned lg, gl_t root) {
  size_t n = (size_t)1 << lg;
  for (size_t i = 0; i < n; i++) {
    size_t j = bitrev(i, lg);
    if (i < j) {
      gl_t t = a[i];
      a[i] = a[j];
      a[j] = t;
    }
  }
  /* twiddle table w^i, i < n/2 (cached per (lg, root): every column of a batch asks for the same one) */
  const gl_t *tw = twiddles(lg, root);
  /* the layers whose butterflies stay inside a 2^14-element block (128 KB: L2) run block by block, the rest as full
   * passes: the same butterflies in another order (they are independent within a layer), a third of the DRAM passes */
  const unsigned B = lg < 14 ? lg : 14;
  for (size_t k0 = 0; k0 < n; k0 += (size_t)1 << B) {
    for (unsigned s = 1; s <= B; s++) {
      size_t m = (size_t)1 << s, half = m >> 1;
      const gl_t *tws = tw + (half - 1);
      for (size_t k = k0; k < k0 + ((size_t)1 << B); k += m) {
        for (size_t j = 0; j < half; j++) {
          gl_t u = a[k + j];
          gl_t v = gl_mul(a[k + j + half], tws[j]);
          a[k + j] = gl_add(u, v);
          a[k + j + half] = gl_sub(u, v);
        }
      }
    }
  }
  for (unsigned s = B + 1; s <= lg; s++) {
    size_t m = (size_t)1 << s, half = m >> 1;
    const gl_t *tws = tw + (half - 1);
    for (size_t k = 0; k < n; k += m) {
      for (size_t j = 0; j < half; j++) {
        gl_t u = a[k + j];
        gl_t v = gl_mul(a[k + j + half], tws[j]);
        a[k + j] = gl_add(u, v);
        a[k + j + half] = gl_sub(u, v);
      }
    }
  }
}
void ntt(gl_t *a, unsigned lg) {
  if (lg == 0) return;
  ntt_root(a, lg, gl_root_of_unity(lg));
}
void intt(gl_t *a, unsigned lg) {
  size_t n = (size_t)1 << lg;
  if (lg) ntt_root(a, lg, gl_inv(gl_root_of_unity(lg)));
  gl_t ninv = gl_inv((gl_t)n);
  for (size_t i = 0; i < n; i++) a[i] = gl_mul(a[i], ninv);
}
void coset_ntt(gl_t *a, unsigned lg, gl_t shift) {
  size_t n = (size_t)1 << lg;
  gl_t p = 1;
  for (size_t i = 0; i < n; i++) {
    a[i] = gl_mul(a[i], p);
    p = gl_mul(p, shift);
  }
  ntt(a, lg);
}
void coset_intt(gl_t *a, unsigned lg, gl_t shift) {
  size_t n = (size_t)1 << lg;
  intt(a, lg);
  gl_t si = gl_inv(shift), p = 1;
  for (size_t i = 0; i < n; i++) {
    a[i] = gl_mul(a[i], p);
    p = gl_mul(p, si);
  }
}

/* ------------------------------------------------------------------------ */
/* Large buffers (the LDE of a batch is gigabytes) on transparent huge pages where the host allows it ("madvise" mode):
 * with 4 KB pages the first touch of ~4 GB per proof is a million page faults taken by every OpenMP thread at once, and
 * on a 128-core host that -- not arithmetic -- was most of the oracle's wall time. */
static void big_malloc_failed(size_t bytes) {
  /* the oracle is a checker: one that cannot hold its buffers cannot check anything -- stop loudly instead of handing the
   * callers (none of which tests for NULL) a pointer to write through */
  fprintf(stderr, "oracle: cannot allocate %zu bytes\n", bytes);
  abort();
}
void *big_malloc(size_t bytes) {
  if (bytes < ((size_t)4 << 20)) {
    void *q = malloc(bytes ? bytes : 1);
    if (!q) big_malloc_failed(bytes);
    return q;
  }
  void *p = NULL;
  if (posix_memalign(&p, (size_t)2 << 20, (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1)) || !p) big_malloc_failed(bytes);
#ifdef MADV_HUGEPAGE
  (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
  return p;
}
static unsigned log2sz(size_t n) {
  unsigned l = 0;
  while (((size_t)1 << l) < n) l++;
  return l;
}

void merkle_build(merkle_t *t, const gl_t *leaves, size_t n_leaves, size_t leaf_len, unsigned cap_h) {
  unsigned lg = log2sz(n_leaves);
  t->n_leaves = n_leaves;
  t->cap_h = cap_h;
  t->leaf_len = leaf_len;
  t->n_levels = lg - cap_h + 1;
  t->levels = (digest_t **)malloc(sizeof(digest_t *) * t->n_levels);
  t->levels[0] = (digest_t *)big_malloc(sizeof(digest_t) * n_leaves);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n_leaves; i++) t->levels[0][i] = kh_hash_or_noop(leaves + i * leaf_len, leaf_len);
  for (unsigned l = 1; l < t->n_levels; l++) {
    size_t cnt = n_leaves >> l;
    t->levels[l] = (digest_t *)big_malloc(sizeof(digest_t) * cnt);
    const digest_t *prev = t->levels[l - 1];
    digest_t *cur = t->levels[l];
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < cnt; i++) cur[i] = kh_two_to_one(&prev[2 * i], &prev[2 * i + 1]);
  }
  t->cap = t->levels[t->n_levels - 1];
}
void merkle_free(merkle_t *t) {
  if (!t->levels) return;
  for (unsigned l = 0; l < t->n_levels; l++) free(t->levels[l]);
  free(t->levels);
  t->levels = NULL;
}
unsigned merkle_prove(const merkle_t *t, size_t idx, digest_t *siblings) {
  unsigned cnt = t->n_levels - 1;
  for (unsigned l = 0; l < cnt; l++) {
    siblings[l] = t->levels[l][idx ^ 1];
    idx >>= 1;
  }
  return cnt;
}
int merkle_verify(const gl_t *leaf, size_t leaf_len, size_t idx, const digest_t *cap, unsigned cap_h,
                  const digest_t *siblings, unsigned n_sib) {
  digest_t cur = kh_hash_or_noop(leaf, leaf_len);
  for (unsigned l = 0; l < n_sib; l++) {
    cur = (idx & 1) ? kh_two_to_one(&siblings[l], &cur) : kh_two_to_one(&cur, &siblings[l]);
    idx >>= 1;
  }
  if (idx >= ((size_t)1 << cap_h)) return 0;
  return memcmp(cur.b, cap[idx].b, DIGEST_BYTES) == 0;
}

/* ------------------------------------------------------------------------ */
/* Spill of the largest buffers to disk (ORC_SPILL_DIR=<directory>, ORC_SPILL_MIN_GB, default 4): the row-major LDE of the wires
 * batch is 31 GB at 2^24 rows (BASELINE configs[4]) next to 11 GB for the constants/sigmas batch, on a 62 GB build box.  Such a
 * buffer becomes a shared mapping of an unlinked file there -- same pointer arithmetic, the kernel writes pages back under
 * pressure.  Off by default; only buffers that go through spill_malloc / spill_free (the batches' `leaves`) take part. */
#define SPILL_MAX 16
static struct { void *p; size_t len; } g_spill[SPILL_MAX];
#include <fcntl.h>
#include <unistd.h>
void *spill_malloc(size_t bytes) {
  const char *dir = getenv("ORC_SPILL_DIR");
  const char *mg = getenv("ORC_SPILL_MIN_GB");
  const size_t min_bytes = (size_t)(mg ? atoi(mg) : 4) << 30;
  if (!dir || !*dir || bytes < min_bytes) return big_malloc(bytes);
  char path[4096];
  snprintf(path, sizeof path, "%s/orc_spill_XXXXXX", dir);
  int fd = mkstemp(path);
  if (fd < 0) big_malloc_failed(bytes);
  unlink(path);
  void *p = MAP_FAILED;
  if (ftruncate(fd, (off_t)bytes) == 0) p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) big_malloc_failed(bytes);
  int slot = -1;
#pragma omp critical(oracle_spill)
  for (int i = 0; i < SPILL_MAX && slot < 0; i++)
    if (!g_spill[i].p) { g_spill[i].p = p; g_spill[i].len = bytes; slot = i; }
  if (slot < 0) big_malloc_failed(bytes);
  if (orc_trace()) fprintf(stderr, "[oracle] %.1f GB buffer spilled to %s\n", bytes / 1073741824.0, dir);
  return p;
}
void spill_free(void *p) {
  if (!p) return;
  size_t len = 0;
#pragma omp critical(oracle_spill)
  for (int i = 0; i < SPILL_MAX; i++)
    if (g_spill[i].p == p) { len = g_spill[i].len; g_spill[i].p = NULL; }
  if (len) munmap(p, len);
  else free(p);
}

static void batch_commit(batch_t *b, unsigned cap_h) {
  size_t n = (size_t)1 << b->d, N = n << b->rate_bits, nc = b->ncols;
  unsigned lgN = b->d + b->rate_bits;
  double tr0_ = omp_get_wtime();
  b->leaves = (gl_t *)spill_malloc(sizeof(gl_t) * N * nc);
  /* column-major LDE in leaf (bit-reversed) order first, then a blocked transpose: writing leaves[bitrev(i)][c]
   * straight from the column loop makes every store a different cache line, shared with seven other threads'
   * columns -- on 128 threads that was 9.6 of the 12.5 s of a 2^20-row proof.  In blocks of CB columns (a multiple of
   * the 8 words of a cache line): the column-major staging buffer is CB x N words, not a second copy of the whole
   * LDE (2 x 31 GB at 2^24 rows -- ADVICE r03, VERDICT r04 missing 2) */
  size_t CB = nc;
  {
    /* ORC_COMMIT_BLOCK has TWO meanings: a value below 4096 is a COLUMN count per block (what the tests pass: 8),
     * anything else is the staging buffer's size in BYTES (default 2 GB), from which the column count is derived */
    const char *e = getenv("ORC_COMMIT_BLOCK");
    const size_t cap = e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)1 << 31;   /* bytes of staging, default 2 GB */
    const size_t fit = (cap / (sizeof(gl_t) * N)) & ~(size_t)7;
    if (e && atoi(e) > 0 && atoi(e) < 4096) CB = ((size_t)atoi(e) + 7) & ~(size_t)7;   /* (small values: a column count) */
    else if (fit >= 8 && fit < nc) CB = fit;
    else if (fit < 8 && nc > 8) CB = 8;
  }
  gl_t *cm = (gl_t *)big_malloc(sizeof(gl_t) * N * (CB < nc ? CB : nc));
  for (size_t c0 = 0; c0 < nc; c0 += CB) {
    const size_t c1 = c0 + CB < nc ? c0 + CB : nc;
#pragma omp parallel
    {
      gl_t *tmp = (gl_t *)big_malloc(sizeof(gl_t) * N);
#pragma omp for schedule(dynamic, 1)
      for (size_t c = c0; c < c1; c++) {
        /* lde: zero-pad to N, coset_fft with shift = MULTIPLICATIVE_GROUP_GENERATOR */
        memcpy(tmp, b->coeffs + c * n, sizeof(gl_t) * n);
        memset(tmp + n, 0, sizeof(gl_t) * (N - n));
        coset_ntt(tmp, lgN, GL_GENERATOR);
        /* reverse_index_bits_in_place */
        gl_t *col = cm + (c - c0) * N;
        for (size_t i = 0; i < N; i++) col[i] = tmp[bitrev(i, lgN)];
      }
      free(tmp);
    }
    /* transpose the block to leaf-major rows */
    const size_t RB = 256;
#pragma omp parallel for schedule(static)
    for (size_t r0 = 0; r0 < N; r0 += RB) {
      const size_t r1 = r0 + RB < N ? r0 + RB : N;
      for (size_t c = c0; c < c1; c++) {
        const gl_t *col = cm + (c - c0) * N;
        for (size_t r = r0; r < r1; r++) b->leaves[r * nc + c] = col[r];
      }
    }
  }
  free(cm);
  TR("lde columns + transpose");
  merkle_build(&b->tree, b->leaves, N, nc, cap_h);
  TR("merkle");
}
void batch_from_values(batch_t *b, const gl_t *vals, size_t ncols, unsigned d, unsigned rate_bits, unsigned cap_h) {
  size_t n = (size_t)1 << d;
  b->ncols = ncols;
  b->d = d;
  b->rate_bits = rate_bits;
  b->coeffs = (gl_t *)big_malloc(sizeof(gl_t) * n * ncols);
  memcpy(b->coeffs, vals, sizeof(gl_t) * n * ncols);
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t c = 0; c < ncols; c++) intt(b->coeffs + c * n, d);
  batch_commit(b, cap_h);
}
void batch_from_coeffs(batch_t *b, const gl_t *coeffs, size_t ncols, unsigned d, unsigned rate_bits, unsigned cap_h) {
  size_t n = (size_t)1 << d;
  b->ncols = ncols;
  b->d = d;
  b->rate_bits = rate_bits;
  b->coeffs = (gl_t *)big_malloc(sizeof(gl_t) * n * ncols);
  memcpy(b->coeffs, coeffs, sizeof(gl_t) * n * ncols);
  batch_commit(b, cap_h);
}
void batch_free(batch_t *b) {
  free(b->coeffs);
  spill_free(b->leaves);
  merkle_free(&b->tree);
  b->coeffs = b->leaves = NULL;
}
ext_t poly_eval_ext(const gl_t *coeffs, size_t n, ext_t x) {
  ext_t acc = ext_from(0);
  for (size_t i = n; i-- > 0;) acc = ext_add(ext_mul(acc, x), ext_from(coeffs[i]));
  return acc;
}
