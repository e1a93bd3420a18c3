/*
 * oracle/prover.c -- CPU restatement of plonky2 0.2.2 `prover::prove` after
 * witness generation (prove_with_partition_witness), for
 * GenericConfig = KeccakGoldilocksConfig, D = 2.  TEST INFRASTRUCTURE ONLY.
 *
 * Reference call site: plonky2-backend/src/actions/prove_action.rs:96
 *   `circuit_data.prove(witnesses).unwrap()`
 * (test harness twin: circuit_translation/tests/factories/utils.rs:26).
 * The algorithm itself lives in the un-vendored plonky2 crate; this file
 * follows SURVEY.md Appendix C:
 *   C.4 transcript order, C.5 permutation argument, C.7 vanishing polynomial,
 *   C.8 openings, C.9 FRI batching + commit phase, C.10 PoW, C.11 bytes
 * -- all of it except the FRI reduction steps now verified byte for byte
 * against the reference's own proofs (tests/test_reference_proofs.py).
 */
#include "oracle.h"
#include "circuit.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct orc_circuit {
  circuit_t c;
  uint8_t *blob;
};

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int orc_circuit_create(const uint8_t *blob, size_t len, orc_circuit **out) {
  orc_circuit *oc = (orc_circuit *)calloc(1, sizeof *oc);
  oc->blob = (uint8_t *)malloc(len);
  memcpy(oc->blob, blob, len);
  poseidon_init(); /* not thread-safe: force it before any OpenMP region */
  if (len >= 256) g_hasher = ((const uint32_t *)oc->blob)[22] == 1;
  int rc = circuit_load(&oc->c, oc->blob, len);
  if (rc) {
    circuit_free(&oc->c);
    free(oc->blob);
    free(oc);
    return ORC_E_BLOB;
  }
  *out = oc;
  return ORC_OK;
}
int orc_circuit_create_verifier(const uint8_t *blob, size_t len, const uint8_t *cap, const uint8_t *digest,
                                orc_circuit **out) {
  orc_circuit *oc = (orc_circuit *)calloc(1, sizeof *oc);
  /* only the header, gate table and k_is are needed: keep a private copy of the whole blob */
  oc->blob = (uint8_t *)malloc(len);
  memcpy(oc->blob, blob, len);
  poseidon_init();
  if (len >= 256) g_hasher = ((const uint32_t *)oc->blob)[22] == 1;
  int rc = circuit_load_verifier(&oc->c, oc->blob, len, cap, digest);
  if (rc) {
    circuit_free(&oc->c);
    free(oc->blob);
    free(oc);
    return ORC_E_BLOB;
  }
  *out = oc;
  return ORC_OK;
}
void orc_circuit_destroy(orc_circuit *oc) {
  if (!oc) return;
  circuit_free(&oc->c);
  free(oc->blob);
  free(oc);
}
void orc_circuit_cap(const orc_circuit *oc, uint8_t *out) {
  g_hasher = oc->c.hasher == 1;
  size_t ncap = (size_t)1 << oc->c.cap_height;
  for (size_t i = 0; i < ncap; i++) memcpy(out + DIGEST_BYTES * i, oc->c.cs.tree.cap[i].b, DIGEST_BYTES);
}
void orc_circuit_digest(const orc_circuit *oc, uint8_t *out) {
  g_hasher = oc->c.hasher == 1;
  memcpy(out, oc->c.circuit_digest.b, DIGEST_BYTES);
}
const circuit_t *orc_circuit_inner(const orc_circuit *oc) { return &oc->c; }

/* ---- byte buffer ---- */
typedef struct {
  uint8_t *p;
  size_t len, cap;
} buf_t;
static void put(buf_t *b, const void *src, size_t n) {
  if (b->len + n > b->cap) {
    b->cap = (b->len + n) * 2 + 1024;
    b->p = (uint8_t *)realloc(b->p, b->cap);
  }
  memcpy(b->p + b->len, src, n);
  b->len += n;
}
static void put_u64(buf_t *b, uint64_t v) { put(b, &v, 8); }
static void put_ext(buf_t *b, ext_t e) {
  put_u64(b, e.c0);
  put_u64(b, e.c1);
}
static void put_digest(buf_t *b, const digest_t *d) { put(b, d->b, DIGEST_BYTES); }
static void put_merkle_proof(buf_t *b, const merkle_t *t, size_t idx) {
  digest_t sib[64];
  unsigned cnt = merkle_prove(t, idx, sib);
  uint8_t l = (uint8_t)cnt;
  put(b, &l, 1);
  for (unsigned i = 0; i < cnt; i++) put_digest(b, &sib[i]);
}

/* Montgomery batch inversion (field/src/types.rs batch_multiplicative_inverse) */
static void batch_inverse(const gl_t *x, gl_t *out, size_t n, gl_t *scratch) {
  gl_t acc = 1;
  for (size_t i = 0; i < n; i++) {
    scratch[i] = acc;
    acc = gl_mul(acc, x[i]);
  }
  gl_t inv = gl_inv(acc);
  for (size_t i = n; i-- > 0;) {
    out[i] = gl_mul(inv, scratch[i]);
    inv = gl_mul(inv, x[i]);
  }
}

/* ext polynomial NTT helpers: coordinates are transformed independently */
static void ext_coset_ntt(ext_t *a, unsigned lg, gl_t shift) {
  size_t n = (size_t)1 << lg;
  gl_t *t = (gl_t *)malloc(sizeof(gl_t) * n);
  for (int k = 0; k < 2; k++) {
    for (size_t i = 0; i < n; i++) t[i] = k ? a[i].c1 : a[i].c0;
    coset_ntt(t, lg, shift);
    for (size_t i = 0; i < n; i++) {
      if (k) a[i].c1 = t[i];
      else a[i].c0 = t[i];
    }
  }
  free(t);
}

typedef struct {
  size_t n_leaves, leaf_len;
  gl_t *leaves;
  merkle_t tree;
} fri_tree_t;

int orc_prove(const orc_circuit *oc, const uint64_t *wires, const uint64_t *pis, uint32_t n_pi, uint64_t pow_hint,
              uint8_t *proof_out, size_t *proof_len, orc_trace *tr) {
  const circuit_t *c = &oc->c;
  g_hasher = c->hasher == 1;
  const size_t n = c->n, N = c->N;
  const unsigned d = c->d, rb = c->rate_bits, lgN = d + rb, chh = c->cap_height;
  const size_t W = c->num_wires, R = c->num_routed, K = c->num_challenges, QF = c->qdf, NC = c->num_constants;
  const size_t nchunks = (R + QF - 1) / QF, PP = nchunks - 1;
  const size_t ncap = (size_t)1 << chh;
  const size_t nzp = K * (1 + PP);
  if (PP != c->num_pp) return ORC_E_BLOB;
  orc_trace local_tr;
  if (!tr) tr = &local_tr;
  memset(tr, 0, sizeof *tr);
  double t0 = now_s(), t1;

  /* 1. public inputs hash (InnerHasher = Poseidon; [] -> 0^4) */
  gl_t pih[4];
  poseidon_hash_no_pad(pis, n_pi, pih);
  memcpy(tr->pi_hash, pih, sizeof pih);

  /* 2. wires commitment */
  batch_t wb;
  batch_from_values(&wb, wires, W, d, rb, chh);
  challenger_t ch;
  ch_init(&ch);
  ch_observe_digest(&ch, &c->circuit_digest);
  ch_observe_many(&ch, pih, 4);
  ch_observe_cap(&ch, wb.tree.cap, ncap);
  gl_t betas[4], gammas[4], alphas[4];
  for (size_t k = 0; k < K; k++) betas[k] = ch_get(&ch);
  for (size_t k = 0; k < K; k++) gammas[k] = ch_get(&ch);
  memcpy(tr->betas, betas, sizeof betas);
  memcpy(tr->gammas, gammas, sizeof gammas);
  t1 = now_s();
  tr->t_wires = t1 - t0;

  /* 3. partial products and Z (C.5) */
  gl_t *zp = (gl_t *)big_malloc(sizeof(gl_t) * nzp * n);
  {
    gl_t *sub = (gl_t *)malloc(sizeof(gl_t) * n);
    gl_t w = gl_root_of_unity(d);
    sub[0] = 1;
    for (size_t i = 1; i < n; i++) sub[i] = gl_mul(sub[i - 1], w);
    gl_t *cp = (gl_t *)malloc(sizeof(gl_t) * n * nchunks);
    for (size_t k = 0; k < K; k++) {
#pragma omp parallel
      {
        gl_t *num = (gl_t *)malloc(sizeof(gl_t) * R * 4);
        gl_t *den = num + R, *inv = num + 2 * R, *scr = num + 3 * R;
#pragma omp for schedule(static)
        for (size_t i = 0; i < n; i++) {
          gl_t x = sub[i];
          for (size_t j = 0; j < R; j++) {
            gl_t wv = wires[j * n + i];
            gl_t s_id = gl_mul(c->k_is[j], x);
            num[j] = gl_add(gl_add(wv, gl_mul(betas[k], s_id)), gammas[k]);
            den[j] = gl_add(gl_add(wv, gl_mul(betas[k], c->sigmas[j * n + i])), gammas[k]);
          }
          batch_inverse(den, inv, R, scr);
          for (size_t m = 0; m < nchunks; m++) {
            gl_t p = 1;
            for (size_t j = m * QF; j < (m + 1) * QF && j < R; j++) p = gl_mul(p, gl_mul(num[j], inv[j]));
            cp[i * nchunks + m] = p;
          }
        }
        free(num);
      }
      gl_t z = 1;
      for (size_t i = 0; i < n; i++) {
        gl_t acc = z;
        zp[k * n + i] = z;
        for (size_t m = 0; m < nchunks; m++) {
          acc = gl_mul(acc, cp[i * nchunks + m]);
          if (m < PP) zp[(K + k * PP + m) * n + i] = acc;
        }
        z = acc;
      }
    }
    free(cp);
    free(sub);
  }
  batch_t zb;
  batch_from_values(&zb, zp, nzp, d, rb, chh);
  ch_observe_cap(&ch, zb.tree.cap, ncap);
  for (size_t k = 0; k < K; k++) alphas[k] = ch_get(&ch);
  memcpy(tr->alphas, alphas, sizeof alphas);
  t0 = now_s();
  tr->t_zs = t0 - t1;

  /* 4. quotient polynomials (C.7) */
  const size_t NGC = c->num_gate_constraints;
  const size_t nterms = K + K * nchunks + NGC;
  gl_t *qv = (gl_t *)malloc(sizeof(gl_t) * K * N); /* [K][N] */
  {
    gl_t zh[64], zhi[64];
    size_t rate = (size_t)1 << rb;
    gl_t gn = gl_pow(GL_GENERATOR, n);
    gl_t w8 = gl_root_of_unity(rb);
    gl_t p = gn;
    for (size_t i = 0; i < rate; i++) {
      zh[i] = gl_sub(p, 1);
      zhi[i] = gl_inv(zh[i]);
      p = gl_mul(p, w8);
    }
    gl_t *pts = (gl_t *)malloc(sizeof(gl_t) * N);
    gl_t wN = gl_root_of_unity(lgN);
    pts[0] = GL_GENERATOR;
    for (size_t i = 1; i < N; i++) pts[i] = gl_mul(pts[i - 1], wN);
    const size_t next_step = (size_t)1 << rb; /* quotient_degree_bits == rate_bits */
#pragma omp parallel
    {
      gl_t *terms = (gl_t *)malloc(sizeof(gl_t) * (nterms + 2 * NGC + 8));
      gl_t *scratch = terms + nterms;
      /* rows are independent: walked in LEAF order (i = bitrev(leaf)), so that the three big row-major LDE buffers are read
       * front to back (the wires' one may be a 31 GB file mapping, poly.c spill_malloc); the results land at their natural index */
#pragma omp for schedule(static)
      for (size_t leaf = 0; leaf < N; leaf++) {
        const size_t i = bitrev(leaf, lgN);
        gl_t x = pts[i];
        const gl_t *crow = batch_lde_row(&c->cs, i);
        const gl_t *wrow = batch_lde_row(&wb, i);
        const gl_t *zrow = batch_lde_row(&zb, i);
        const gl_t *znext = batch_lde_row(&zb, (i + next_step) % N);
        const gl_t *sig = crow + NC;
        gl_t l0 = gl_mul(zh[i % rate], gl_inv(gl_mul((gl_t)n % GL_P, gl_sub(x, 1))));
        size_t t = 0;
        for (size_t k = 0; k < K; k++) terms[t++] = gl_mul(l0, gl_sub(zrow[k], 1));
        for (size_t k = 0; k < K; k++) {
          for (size_t m = 0; m < nchunks; m++) {
            gl_t prev = m == 0 ? zrow[k] : zrow[K + k * PP + m - 1];
            gl_t next = m == nchunks - 1 ? znext[k] : zrow[K + k * PP + m];
            gl_t np = 1, dp = 1;
            for (size_t j = m * QF; j < (m + 1) * QF && j < R; j++) {
              gl_t s_id = gl_mul(c->k_is[j], x);
              np = gl_mul(np, gl_add(gl_add(wrow[j], gl_mul(betas[k], s_id)), gammas[k]));
              dp = gl_mul(dp, gl_add(gl_add(wrow[j], gl_mul(betas[k], sig[j])), gammas[k]));
            }
            terms[t++] = gl_sub(gl_mul(prev, np), gl_mul(next, dp));
          }
        }
        eval_gate_constraints_base(c, crow, wrow, pih, terms + t, scratch + NGC);
        for (size_t k = 0; k < K; k++) {
          gl_t acc = 0;
          for (size_t q = nterms; q-- > 0;) acc = gl_add(gl_mul(acc, alphas[k]), terms[q]);
          qv[k * N + i] = gl_mul(acc, zhi[i % rate]);
        }
      }
      free(terms);
    }
    free(pts);
  }
  gl_t *qchunks = (gl_t *)malloc(sizeof(gl_t) * K * QF * n);
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t k = 0; k < K; k++) {
    coset_intt(qv + k * N, lgN, GL_GENERATOR);
    /* trim_to_len(quotient_degree = QF * n): identity when QF == 2^rate_bits */
    memcpy(qchunks + k * QF * n, qv + k * N, sizeof(gl_t) * QF * n);
  }
  free(qv);
  batch_t qb;
  batch_from_coeffs(&qb, qchunks, K * QF, d, rb, chh);
  free(qchunks);
  ch_observe_cap(&ch, qb.tree.cap, ncap);
  ext_t zeta = ch_get_ext(&ch);
  tr->zeta[0] = zeta.c0;
  tr->zeta[1] = zeta.c1;
  t1 = now_s();
  tr->t_quotient = t1 - t0;
  int rc = ORC_OK;
  {
    ext_t zn = zeta;
    for (unsigned i = 0; i < d; i++) zn = ext_mul(zn, zn);
    if (ext_eq(zn, ext_from(1))) rc = ORC_E_ZETA_IN_SUBGROUP;
  }

  /* 5. openings (C.8) */
  const batch_t *oracles[4] = {&c->cs, &wb, &zb, &qb};
  const size_t ncs = NC + R;
  const size_t nall = ncs + W + nzp + K * QF;
  ext_t *op0 = (ext_t *)malloc(sizeof(ext_t) * (nall + K)); /* at zeta, FRI order; then zs_next */
  ext_t g_zeta = ext_scale(zeta, gl_root_of_unity(d));
  {
    size_t base = 0;
    for (int o = 0; o < 4; o++) {
      const batch_t *b = oracles[o];
#pragma omp parallel for schedule(dynamic, 4)
      for (size_t j = 0; j < b->ncols; j++) op0[base + j] = poly_eval_ext(b->coeffs + j * n, n, zeta);
      base += b->ncols;
    }
    for (size_t k = 0; k < K; k++) op0[nall + k] = poly_eval_ext(zb.coeffs + k * n, n, g_zeta);
  }
  /* observe_openings(to_fri_openings): batch zeta in FRI order, then zs_next */
  for (size_t j = 0; j < nall + K; j++) ch_observe_ext(&ch, op0[j]);
  t0 = now_s();
  tr->t_openings = t0 - t1;

  /* 6. FRI (C.9) */
  ext_t alpha = ch_get_ext(&ch);
  tr->alpha_fri[0] = alpha.c0;
  tr->alpha_fri[1] = alpha.c1;
  ext_t *final_poly = (ext_t *)calloc(N, sizeof(ext_t)); /* lde: zero padded to N */
  {
    ext_t *apw = (ext_t *)malloc(sizeof(ext_t) * nall);
    apw[0] = ext_from(1);
    for (size_t j = 1; j < nall; j++) apw[j] = ext_mul(apw[j - 1], alpha);
    ext_t *F0 = (ext_t *)malloc(sizeof(ext_t) * n), *F1 = (ext_t *)malloc(sizeof(ext_t) * n);
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < n; p++) {
      ext_t a0 = ext_from(0), a1 = ext_from(0);
      size_t base = 0;
      for (int o = 0; o < 4; o++) {
        const batch_t *b = oracles[o];
        for (size_t j = 0; j < b->ncols; j++) a0 = ext_add(a0, ext_scale(apw[base + j], b->coeffs[j * n + p]));
        base += b->ncols;
      }
      for (size_t k = 0; k < K; k++) a1 = ext_add(a1, ext_scale(apw[k], zb.coeffs[k * n + p]));
      F0[p] = a0;
      F1[p] = a1;
    }
    /* divide_by_linear: q[j-1] = c[j] + z*q[j]; remainder dropped; pad to n */
    ext_t aK = ext_pow(alpha, K);
    ext_t acc0 = ext_from(0), acc1 = ext_from(0);
    for (size_t j = n; j-- > 1;) {
      acc0 = ext_add(ext_mul(acc0, zeta), F0[j]);
      acc1 = ext_add(ext_mul(acc1, g_zeta), F1[j]);
      final_poly[j - 1] = ext_add(ext_mul(aK, acc0), acc1);
    }
    free(apw);
    free(F0);
    free(F1);
  }
  /* lde_final_values = final_poly.lde(rate_bits).coset_fft(g) */
  size_t L = N;
  unsigned lgL = lgN;
  ext_t *coeffs = final_poly;
  ext_t *values = (ext_t *)malloc(sizeof(ext_t) * N);
  memcpy(values, coeffs, sizeof(ext_t) * N);
  ext_coset_ntt(values, lgL, GL_GENERATOR);
  fri_tree_t ft[8];
  gl_t shift = GL_GENERATOR;
  for (uint32_t s = 0; s < c->n_steps; s++) {
    unsigned ab = c->arity_bits[s];
    size_t arity = (size_t)1 << ab;
    fri_tree_t *t = &ft[s];
    t->n_leaves = L / arity;
    t->leaf_len = 2 * arity;
    t->leaves = (gl_t *)malloc(sizeof(gl_t) * 2 * L);
    for (size_t j = 0; j < L; j++) {
      ext_t v = values[bitrev(j, lgL)];
      t->leaves[2 * j] = v.c0;
      t->leaves[2 * j + 1] = v.c1;
    }
    merkle_build(&t->tree, t->leaves, t->n_leaves, t->leaf_len, chh);
    ch_observe_cap(&ch, t->tree.cap, ncap);
    ext_t beta = ch_get_ext(&ch);
    tr->fri_betas[s][0] = beta.c0;
    tr->fri_betas[s][1] = beta.c1;
    size_t L2 = L / arity;
    for (size_t m = 0; m < L2; m++) {
      ext_t acc = ext_from(0);
      for (size_t q = arity; q-- > 0;) acc = ext_add(ext_mul(acc, beta), coeffs[arity * m + q]);
      coeffs[m] = acc; /* in place: m <= arity*m */
    }
    for (unsigned q = 0; q < ab; q++) shift = gl_sqr(shift);
    L = L2;
    lgL -= ab;
    memcpy(values, coeffs, sizeof(ext_t) * L);
    ext_coset_ntt(values, lgL, shift);
  }
  size_t n_final = L >> rb;
  for (size_t j = 0; j < n_final; j++) ch_observe_ext(&ch, coeffs[j]);

  /* PoW (C.10): minimum witness policy */
  gl_t pow_witness;
  {
    gl_t inter[12];
    memcpy(inter, ch.state, sizeof inter);
    for (int i = 0; i < ch.n_in; i++) inter[i] = ch.in[i];
    int pos = ch.n_in;
    if (pow_hint != UINT64_MAX) {
      pow_witness = pow_hint;
    } else {
      uint64_t found = UINT64_MAX;
      for (uint64_t base = 0; found == UINT64_MAX; base += 1 << 14) {
#pragma omp parallel for schedule(static)
        for (uint64_t w = base; w < base + (1 << 14); w++) {
          gl_t st[12];
          memcpy(st, inter, sizeof st);
          st[pos] = w;
          hasher_permutation(st);
          if (c->pow_bits == 0 || (st[7] >> (64 - c->pow_bits)) == 0) {
#pragma omp critical
            if (w < found) found = w;
          }
        }
      }
      pow_witness = found;
    }
    ch_observe(&ch, pow_witness);
    gl_t resp = ch_get(&ch);
    if (c->pow_bits && (resp >> (64 - c->pow_bits)) != 0) rc = rc ? rc : ORC_E_VERIFY;
  }
  tr->pow_witness = pow_witness;

  /* queries */
  size_t qidx[64];
  for (uint32_t q = 0; q < c->num_queries; q++) {
    qidx[q] = (size_t)(ch_get(&ch) % N);
    tr->query_indices[q] = (uint32_t)qidx[q];
  }

  /* 7. serialise (C.11) */
  buf_t out = {0};
  for (size_t i = 0; i < ncap; i++) put_digest(&out, &wb.tree.cap[i]);
  for (size_t i = 0; i < ncap; i++) put_digest(&out, &zb.tree.cap[i]);
  for (size_t i = 0; i < ncap; i++) put_digest(&out, &qb.tree.cap[i]);
  /* OpeningSet order: constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys */
  for (size_t j = 0; j < ncs + W + K; j++) put_ext(&out, op0[j]);
  for (size_t k = 0; k < K; k++) put_ext(&out, op0[nall + k]);
  for (size_t j = ncs + W + K; j < nall; j++) put_ext(&out, op0[j]);
  for (uint32_t s = 0; s < c->n_steps; s++)
    for (size_t i = 0; i < ncap; i++) put_digest(&out, &ft[s].tree.cap[i]);
  for (uint32_t q = 0; q < c->num_queries; q++) {
    size_t x = qidx[q];
    for (int o = 0; o < 4; o++) {
      const batch_t *b = oracles[o];
      put(&out, b->leaves + x * b->ncols, 8 * b->ncols);
      put_merkle_proof(&out, &b->tree, x);
    }
    for (uint32_t s = 0; s < c->n_steps; s++) {
      unsigned ab = c->arity_bits[s];
      size_t li = x >> ab;
      put(&out, ft[s].leaves + li * ft[s].leaf_len, 8 * ft[s].leaf_len);
      put_merkle_proof(&out, &ft[s].tree, li);
      x = li;
    }
  }
  for (size_t j = 0; j < n_final; j++) put_ext(&out, coeffs[j]);
  put_u64(&out, pow_witness);
  for (uint32_t i = 0; i < n_pi; i++) put_u64(&out, pis[i]);
  t1 = now_s();
  tr->t_fri = t1 - t0;
  tr->t_total = tr->t_wires + tr->t_zs + tr->t_quotient + tr->t_openings + tr->t_fri;

  if (out.len > *proof_len) rc = ORC_E_BUFFER;
  else memcpy(proof_out, out.p, out.len);
  *proof_len = out.len;

  free(out.p);
  for (uint32_t s = 0; s < c->n_steps; s++) {
    free(ft[s].leaves);
    merkle_free(&ft[s].tree);
  }
  free(values);
  free(final_poly);
  free(op0);
  free(zp);
  batch_free(&wb);
  batch_free(&zb);
  batch_free(&qb);
  return rc;
}
