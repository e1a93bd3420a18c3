// verify.hip -- proof verification, host code only (no kernel in this file).
//
// The reference's `verify` action (plonky2-backend/src/actions/verify_action.rs:11-17,
// VerifierCircuitData::verify) and the assertion every in-tree test ends with
// (`circuit_data.verify(proof)`, e.g. circuit_translation/tests/factories/utils.rs:26-27) check a
// proof in a few milliseconds of scalar work; so does this file -- it is what SURVEY.md 8(b) calls
// `p2gpu-verify`, and it needs no GPU: a handle made by p2gpu_verifier_create holds only the
// verifier's share of the circuit (parameters, gate table, constants_sigmas cap, circuit digest,
// k_is) -- the counterpart of the VK file written by actions/write_vk_action.rs:65-81.
//
// Checks, in plonky2 0.2.2 order (plonk/verifier.rs verify_with_challenges, fri/verifier.rs
// verify_fri_proof):  shape of the byte string -> transcript replay -> plonk identity at zeta ->
// proof-of-work -> per query: Merkle paths of the four initial oracles, the combined quotient
// value, every arity-16 fold (closed-form barycentric interpolation on the coset s<g_16>), the
// final polynomial.  Proof layout: SURVEY.md C.11; it accepts the reference's own proofs
// (tests/test_reference_proofs.py::test_product_verifier_accepts_reference_proof).
#include "hostproof.hpp"
#include <cstdarg>
#include <cstdio>

using namespace p2;

namespace p2 {

namespace {
struct CollectOut {
  std::vector<ext_t> *v;
  void emit(ext_t c) { v->push_back(c); }
};
}  // namespace

// The verifier's plonk identity at zeta, on the opened values (plonk/verifier.rs
// verify_with_challenges + vanishing_poly.rs eval_vanishing_poly over the extension):
//   vanishing_c(zeta) == Z_H(zeta) * sum_m zeta^(n m) * t_{c,m}(zeta)        for each challenge c.
// `op` = constants, sigmas, wires, zs, partial products, quotient chunks, then zs_next.
// It fails exactly when the witness does not satisfy the circuit (with overwhelming
// probability); upstream only finds that out in its witness generator, which stays in Rust, so
// the prover runs this as its self-check too.
bool plonk_identity_holds(const p2gpu_circuit *c, const std::vector<ext_t> &op, const gl_t *betas, const gl_t *gammas,
                          const gl_t *alphas, ext_t zeta, const gl_t pih[4]) {
  const uint32_t K = c->K, R = c->R, W = c->W, NC = c->NC, QF = c->QF, PP = c->PP, nchunks = c->nchunks;
  const uint32_t ncs = NC + R, nzp = K * (1 + PP), nall = ncs + W + nzp + K * QF;
  const ext_t *o_const = op.data(), *o_sig = o_const + NC, *o_wires = o_sig + R, *o_zs = o_wires + W;
  const ext_t *o_pp = o_zs + K, *o_quot = o_pp + K * PP, *o_zs_next = op.data() + nall;
  ext_t zn = zeta;
  for (uint32_t i = 0; i < c->d; i++) zn = ext_mul(zn, zn);
  const ext_t z_h = ext_sub(zn, ext_from(1));
  const ext_t l0 = ext_mul(z_h, ext_inv(ext_scale(ext_sub(zeta, ext_from(1)), (gl_t)c->n)));
  std::vector<ext_t> terms;
  for (uint32_t k = 0; k < K; k++) terms.push_back(ext_mul(l0, ext_sub(o_zs[k], ext_from(1))));
  for (uint32_t k = 0; k < K; k++)
    for (uint32_t m = 0; m < nchunks; m++) {
      const ext_t prev = m == 0 ? o_zs[k] : o_pp[k * PP + m - 1];
      const ext_t next = m == nchunks - 1 ? o_zs_next[k] : o_pp[k * PP + m];
      ext_t np = ext_from(1), dp = ext_from(1);
      for (uint32_t j = m * QF; j < (m + 1) * QF && j < R; j++) {
        const ext_t s_id = ext_scale(zeta, c->k_is[j]);
        np = ext_mul(np, ext_add(ext_add(o_wires[j], ext_scale(s_id, betas[k])), ext_from(gammas[k])));
        dp = ext_mul(dp, ext_add(ext_add(o_wires[j], ext_scale(o_sig[j], betas[k])), ext_from(gammas[k])));
      }
      terms.push_back(ext_sub(ext_mul(prev, np), ext_mul(next, dp)));
    }
  std::vector<ext_t> gate_terms(c->max_gate_constraints, ext_from(0)), cons;
  ext_t pih_e[4];
  for (int i = 0; i < 4; i++) pih_e[i] = ext_from(pih[i]);
  auto Wf = [&](uint32_t col) { return o_wires[col]; };
  auto LC = [&](uint32_t i) { return o_const[c->num_selectors + i]; };
  for (uint32_t gi = 0; gi < c->num_gates; gi++) {
    const GateDesc &g = c->gates[gi];
    if (!g.num_constraints) continue;
    const ext_t f = gate_filter<ExtOps>(g, gi, c->num_selectors, o_const[g.sel_index]);
    cons.clear();
    CollectOut out{&cons};
    eval_gate<ExtOps, true>(g, Wf, LC, pih_e, c->poseidon_rc_gate, out);
    for (size_t k = 0; k < cons.size() && k < gate_terms.size(); k++)
      gate_terms[k] = ext_add(gate_terms[k], ext_mul(f, cons[k]));
  }
  terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
  for (uint32_t k = 0; k < K; k++) {
    ext_t van = ext_from(0), qz = ext_from(0);
    for (size_t t = terms.size(); t-- > 0;) van = ext_add(ext_scale(van, alphas[k]), terms[t]);
    for (uint32_t m = QF; m-- > 0;) qz = ext_add(ext_mul(qz, zn), o_quot[k * QF + m]);
    if (!ext_eq(van, ext_mul(z_h, qz))) return false;
  }
  return true;
}

}  // namespace p2

namespace {

int reject(const char *fmt, ...) {
  char buf[400];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  set_err("proof rejected: %s", buf);
  return P2GPU_E_VERIFY;
}

}  // namespace

extern "C" {

int p2gpu_verifier_create(const uint8_t *blob, size_t len, p2gpu_circuit **out) try {
  if (!blob || !out) return P2GPU_E_ARG;
  p2gpu_circuit *c = new p2gpu_circuit();
  size_t off = 0;
  const uint8_t *cap_in = nullptr;
  if (int rc = circuit_parse(blob, len, c, &off, &cap_in)) {
    delete c;
    return rc;
  }
  if ((c->flags & 3) != 3 || !cap_in) {
    delete c;
    set_err("a verifier handle needs the circuit digest and the constants_sigmas cap in the blob (flags 0b11)");
    return P2GPU_E_BLOB;
  }
  c->cs.cap.resize((size_t)1 << c->cap_h);
  for (size_t i = 0; i < c->cs.cap.size(); i++) {
    memset(&c->cs.cap[i], 0, sizeof(dig_t));
    memcpy(c->cs.cap[i].w, cap_in + 32 * i, c->hasher ? 32 : 25);
  }
  if (c->hasher) {  // Poseidon digests are field elements: canonical encodings only (as Cursor::digest() asks of a proof)
    bool canon = true;
    for (auto &dg : c->cs.cap)
      for (int i = 0; i < 4; i++) canon &= dg.w[i] < GL_P;
    for (int i = 0; i < 4; i++) canon &= c->circuit_digest.w[i] < GL_P;
    if (!canon) {
      delete c;
      set_err("verifier key holds a non-canonical Poseidon digest word");
      return P2GPU_E_BLOB;
    }
  }
  c->device = -1;  // no device state: only p2gpu_verify and the getters accept this handle
  *out = c;
  return P2GPU_OK;
} P2GPU_CATCH

// header | gate table | cap | k_is: the verifier's share of a circuit blob
int p2gpu_circuit_export_vk(const p2gpu_circuit *c, uint8_t *out, size_t *len) try {
  if (!c || !len) return P2GPU_E_ARG;
  const size_t ncap = (size_t)1 << c->cap_h;
  const size_t need = 256 + 48 * (size_t)c->num_gates + 32 * ncap + 8 * (size_t)c->R;
  if (!out || *len < need) {
    *len = need;
    if (out) set_err("verifier blob needs %zu bytes", need);
    return out ? P2GPU_E_BUFFER : P2GPU_OK;
  }
  if (c->cs.cap.size() != ncap) return P2GPU_E_ARG;
  uint32_t h[64];
  memset(h, 0, sizeof h);
  h[0] = 0x43473250u; h[1] = 1; h[2] = c->d; h[3] = c->W; h[4] = c->R; h[5] = c->NC; h[6] = c->num_selectors;
  h[7] = c->K; h[8] = c->QF; h[9] = c->rate_bits; h[10] = c->cap_h; h[11] = c->pow_bits; h[12] = c->num_queries;
  h[13] = c->n_steps;
  for (int i = 0; i < 8; i++) h[14 + i] = c->arity[i];
  h[22] = c->hasher; h[23] = c->num_gates; h[24] = c->num_pi; h[25] = 3; h[26] = c->PP;
  memcpy(&h[32], c->circuit_digest.w, c->hasher ? 32 : 25);
  memcpy(out, h, sizeof h);
  size_t off = 256;
  for (const GateDesc &G : c->gates) {
    uint32_t g[12];
    memset(g, 0, sizeof g);
    g[0] = G.kind;
    memcpy(&g[1], G.p, 16);
    g[5] = G.sel_index; g[6] = G.group_start; g[7] = G.group_end; g[8] = G.num_constraints; g[9] = G.degree;
    g[10] = G.num_constants;
    memcpy(out + off, g, sizeof g);
    off += sizeof g;
  }
  for (size_t i = 0; i < ncap; i++) {
    memset(out + off, 0, 32);
    memcpy(out + off, c->cs.cap[i].w, c->hasher ? 32 : 25);
    off += 32;
  }
  memcpy(out + off, c->k_is.data(), 8 * (size_t)c->R);
  off += 8 * (size_t)c->R;
  *len = off;
  return P2GPU_OK;
} P2GPU_CATCH

int p2gpu_verify(const p2gpu_circuit *c, const uint8_t *proof, size_t len) try {
  if (!c || !proof) return P2GPU_E_ARG;
  use_hasher(c);
  const uint32_t K = c->K, R = c->R, W = c->W, NC = c->NC, QF = c->QF, PP = c->PP, d = c->d;
  const uint32_t ncs = NC + R, nzp = K * (1 + PP), nq = K * QF, nall = ncs + W + nzp + nq;
  const unsigned lgN = d + c->rate_bits;
  const size_t ncap = (size_t)1 << c->cap_h;
  if (c->cs.cap.size() != ncap) return P2GPU_E_ARG;

  // ---- 1. shape: everything but the queries has a fixed position ----
  Cursor in{proof, len};
  std::vector<dig_t> wires_cap, zs_cap, quot_cap;
  in.digests(wires_cap, ncap);
  in.digests(zs_cap, ncap);
  in.digests(quot_cap, ncap);
  // opening set as serialised: constants, sigmas, wires, zs, zs_next, partial products, quotient
  std::vector<ext_t> op(nall + K);  // kept in the prover's order: ..., zs, pp, quotient | zs_next
  for (uint32_t j = 0; j < ncs + W + K; j++) op[j] = in.ext();
  for (uint32_t k = 0; k < K; k++) op[nall + k] = in.ext();
  for (uint32_t j = ncs + W + K; j < nall; j++) op[j] = in.ext();
  std::vector<std::vector<dig_t>> step_caps(c->n_steps);
  for (auto &sc : step_caps) in.digests(sc, ncap);
  if (!in.ok) return reject("truncated or non-canonical header (caps / openings)");
  const uint32_t oracle_cols[4] = {ncs, W, nzp, nq};
  size_t query_bytes = 0, final_len = c->n;
  const size_t hb = hh_bytes();
  for (int o = 0; o < 4; o++) query_bytes += 8 * (size_t)oracle_cols[o] + 1 + hb * (size_t)(lgN - c->cap_h);
  {
    unsigned lg = lgN;
    for (uint32_t s = 0; s < c->n_steps; s++) {
      const unsigned ab = c->arity[s];
      if (ab < 1 || ab > MAX_ARITY_BITS || lg < ab + c->cap_h || (final_len >> ab) == 0) return reject("bad reduction arity in circuit");
      lg -= ab;
      final_len >>= ab;
      query_bytes += ((size_t)16 << ab) + 1 + hb * (size_t)(lg - c->cap_h);
    }
  }
  const size_t queries_at = in.at;
  const size_t tail_at = queries_at + query_bytes * c->num_queries;
  const size_t want = tail_at + 16 * final_len + 8 + 8 * (size_t)c->num_pi;
  if (want != len) return reject("length %zu, expected %zu for this circuit", len, want);
  in.at = tail_at;
  std::vector<ext_t> final_poly(final_len);
  for (auto &e : final_poly) e = in.ext();
  const uint64_t pow_witness = in.felt();  // canonical, like every other field element (upstream reduces it; a non-canonical encoding is not a second valid proof here)
  std::vector<gl_t> pis(c->num_pi);
  for (auto &v : pis) v = in.felt();
  if (!in.ok) return reject("non-canonical field element in final polynomial / public inputs");

  // ---- 2. transcript (plonk/get_challenges.rs; order: SURVEY.md C.4) ----
  gl_t pih[4];
  poseidon_hash_no_pad_host(pis.data(), pis.size(), pih, c->poseidon_rc);
  Challenger ch;
  ch.observe_digest(c->circuit_digest);
  for (int i = 0; i < 4; i++) ch.observe(pih[i]);
  ch.observe_cap(wires_cap);
  gl_t betas[2], gammas[2], alphas[2];
  for (uint32_t k = 0; k < K; k++) betas[k] = ch.get();
  for (uint32_t k = 0; k < K; k++) gammas[k] = ch.get();
  ch.observe_cap(zs_cap);
  for (uint32_t k = 0; k < K; k++) alphas[k] = ch.get();
  ch.observe_cap(quot_cap);
  const ext_t zeta = ch.get_ext();
  for (uint32_t j = 0; j < nall + K; j++) ch.observe_ext(op[j]);
  const ext_t alpha = ch.get_ext();
  std::vector<ext_t> fold_betas(c->n_steps);
  for (uint32_t s = 0; s < c->n_steps; s++) {
    ch.observe_cap(step_caps[s]);
    fold_betas[s] = ch.get_ext();
  }
  for (auto &e : final_poly) ch.observe_ext(e);
  ch.observe(pow_witness);
  const gl_t pow_response = ch.get();
  std::vector<size_t> qidx(c->num_queries);
  for (auto &x : qidx) x = (size_t)(ch.get() & (c->N - 1));  // N is a power of two: same as % N

  // ---- 3. plonk identity at zeta ----
  {
    ext_t zn = zeta;
    for (uint32_t i = 0; i < d; i++) zn = ext_mul(zn, zn);
    if (zn.c0 == 1 && zn.c1 == 0) return reject("zeta lies in the subgroup");
  }
  if (!plonk_identity_holds(c, op, betas, gammas, alphas, zeta, pih))
    return reject("vanishing(zeta) != Z_H(zeta) * quotient(zeta)");

  // ---- 4. FRI ----
  if (c->pow_bits && (pow_response >> (64 - c->pow_bits)) != 0) return reject("proof-of-work response has too few leading zeros");
  const ext_t opened0 = reduce_with_powers(op.begin(), op.begin() + nall, alpha);        // batch at zeta
  const ext_t opened1 = reduce_with_powers(op.begin() + nall, op.end(), alpha);          // Z at g zeta
  const ext_t g_zeta = ext_scale(zeta, gl_root(d));
  const ext_t alpha_k = ext_pow(alpha, K);
  const std::vector<dig_t> *init_caps[4] = {&c->cs.cap, &wires_cap, &zs_cap, &quot_cap};
  std::vector<gl_t> row(nall);
  std::vector<dig_t> sib;
  in.at = queries_at;
  for (uint32_t qi = 0; qi < c->num_queries; qi++) {
    const size_t x0 = qidx[qi];
    // initial oracles: the row of each tree at leaf x0
    size_t t = 0;
    for (int o = 0; o < 4; o++) {
      const uint8_t *leaf = in.take(8 * (size_t)oracle_cols[o]);
      const uint8_t *plen = in.take(1);
      if (!leaf || !plen || *plen != lgN - c->cap_h) return reject("query %u: malformed initial opening %d", qi, o);
      memcpy(&row[t], leaf, 8 * (size_t)oracle_cols[o]);
      for (uint32_t j = 0; j < oracle_cols[o]; j++)
        if (row[t + j] >= GL_P) return reject("query %u: non-canonical leaf element", qi);
      in.digests(sib, *plen);
      if (!in.ok || !merkle_path_ok(&row[t], oracle_cols[o], x0, sib, *init_caps[o]))
        return reject("query %u: Merkle path of initial oracle %d does not lead to its cap", qi, o);
      t += oracle_cols[o];
    }
    // the queried point: leaf x0 holds natural LDE row bitrev(x0), i.e. g w_N^bitrev(x0), g = GL_GEN
    size_t e = 0;
    for (unsigned b = 0; b < lgN; b++) e |= ((x0 >> b) & 1) << (lgN - 1 - b);
    unsigned lg = lgN;
    gl_t shift = GL_GEN;
    const gl_t x_pt = gl_mul(shift, gl_pow(gl_root(lg), e));
    // fri_combine_initial: (F0(x) - F0(zeta)) / (x - zeta) * alpha^K + (F1(x) - F1(g zeta)) / (x - g zeta)
    ext_t acc0 = ext_from(0), acc1 = ext_from(0);
    for (size_t j = nall; j-- > 0;) acc0 = ext_add(ext_mul(acc0, alpha), ext_from(row[j]));
    for (size_t j = K; j-- > 0;) acc1 = ext_add(ext_mul(acc1, alpha), ext_from(row[ncs + W + j]));
    ext_t cur = ext_mul(ext_sub(acc0, opened0), ext_inv(ext_sub(ext_from(x_pt), zeta)));
    cur = ext_add(ext_mul(cur, alpha_k), ext_mul(ext_sub(acc1, opened1), ext_inv(ext_sub(ext_from(x_pt), g_zeta))));
    // commit-phase folds
    size_t x = x0;
    for (uint32_t s = 0; s < c->n_steps; s++) {
      const unsigned ab = c->arity[s];
      const uint32_t a = 1u << ab;
      ext_t y_leaf[1 << MAX_ARITY_BITS], y_nat[1 << MAX_ARITY_BITS];
      gl_t flat[128];
      for (uint32_t i = 0; i < a; i++) {
        y_leaf[i] = in.ext();
        flat[2 * i] = y_leaf[i].c0;
        flat[2 * i + 1] = y_leaf[i].c1;
      }
      const uint8_t *plen = in.take(1);
      if (!in.ok || !plen || *plen != lg - ab - c->cap_h) return reject("query %u: malformed fold step %u", qi, s);
      in.digests(sib, *plen);
      const size_t leaf_index = x >> ab, within = x & (a - 1);
      if (!ext_eq(y_leaf[within], cur)) return reject("query %u: fold step %u does not continue the previous value", qi, s);
      if (!in.ok || !merkle_path_ok(flat, 2 * (size_t)a, leaf_index, sib, step_caps[s]))
        return reject("query %u: Merkle path of fold step %u does not lead to its cap", qi, s);
      // leaf slot i holds the point s g^bitrev(i), s = shift * w^(e mod 2^(lg-ab))
      for (uint32_t i = 0; i < a; i++) y_nat[bitrev32(i, ab)] = y_leaf[i];
      e &= ((size_t)1 << (lg - ab)) - 1;
      const gl_t s_pt = gl_mul(shift, gl_pow(gl_root(lg), e));
      cur = interpolate_coset(s_pt, ab, y_nat, fold_betas[s]);
      for (unsigned i = 0; i < ab; i++) shift = gl_sqr(shift);
      lg -= ab;
      x = leaf_index;
    }
    // final polynomial at the folded point
    const gl_t x_last = gl_mul(shift, gl_pow(gl_root(lg), e));
    ext_t fe = ext_from(0);
    for (size_t i = final_len; i-- > 0;) fe = ext_add(ext_scale(fe, x_last), final_poly[i]);
    if (!ext_eq(fe, cur)) return reject("query %u: final polynomial disagrees with the folded value", qi);
  }
  if (!in.ok || in.at != tail_at) return reject("query section has the wrong size");
  return P2GPU_OK;
} P2GPU_CATCH

}  // extern "C"
