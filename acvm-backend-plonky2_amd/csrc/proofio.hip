// proofio.hip -- the reference's on-disk proof format, host code only (no kernel in this file).
//
// `plonky2-backend prove` does not write the prover's output as is: it compresses it first,
// `proof.compress(&circuit_digest, &common).to_bytes()` (plonky2-backend/src/actions/
// prove_action.rs:75-78, hex-encoded at :38-42), and `verify` reads that back with
// `CompressedProofWithPublicInputs::from_bytes` + `verify_compressed`
// (actions/verify_action.rs:11-17, noir_and_plonky2_serialization.rs:24-33).  This file is that
// pair on top of the uncompressed bytes p2gpu_prove writes:
//
//   compress   (plonky2 0.2.2 fri/proof.rs FriProof::compress, hash/merkle_proofs.rs
//               compress_merkle_proofs): query indices stored as u32; per tree the Merkle paths of
//               all queries lose every sibling that another path or leaf already determines; a leaf
//               reached by several queries is stored once (the first query's copy, entries sorted
//               by index); each fold step drops the one evaluation the verifier can infer.
//   decompress the inverse: replays the transcript for the challenges, re-derives the dropped
//               evaluations query by query (get_challenges.rs get_inferred_elements) and rebuilds
//               the sibling lists by hashing upwards.
//
// Pinned by the reference's own files: compress(decompress(x)) == x for
// tests/golden/reference/*.proof.hex (tests/test_reference_proofs.py).  Those have no fold step;
// with fold steps the layout follows the same recollection of 0.2.2 as SURVEY.md C.11 and is covered
// by round trips and by both verifiers accepting the decompressed bytes.
#include "hostproof.hpp"
#include <algorithm>
#include <map>
#include <set>

using namespace p2;

namespace {

struct QueryRound {
  std::vector<gl_t> leaf[4];
  std::vector<dig_t> path[4];
  std::vector<std::vector<ext_t>> evals;   // per fold step, leaf order
  std::vector<std::vector<dig_t>> spath;
};

struct ProofData {
  std::vector<dig_t> caps[3];
  std::vector<ext_t> op;  // serialisation order: constants, sigmas, wires, zs, zs_next, partial products, quotient
  std::vector<std::vector<dig_t>> step_caps;
  std::vector<QueryRound> queries;
  std::vector<ext_t> final_poly;
  uint64_t pow_witness = 0;
  std::vector<gl_t> pis;
};

struct Shape {
  uint32_t K, R, W, NC, QF, PP, d, ncs, nzp, nq, nall, cols[4], lgN, cap_h, n_steps, arity[8], queries;
  size_t ncap, final_len;
  explicit Shape(const p2gpu_circuit *c) {
    K = c->K; R = c->R; W = c->W; NC = c->NC; QF = c->QF; PP = c->PP; d = c->d;
    ncs = NC + R; nzp = K * (1 + PP); nq = K * QF; nall = ncs + W + nzp + nq;
    cols[0] = ncs; cols[1] = W; cols[2] = nzp; cols[3] = nq;
    lgN = d + c->rate_bits; cap_h = c->cap_h; n_steps = c->n_steps; queries = c->num_queries;
    ncap = (size_t)1 << cap_h;
    final_len = c->n;
    for (uint32_t s = 0; s < n_steps; s++) {
      arity[s] = c->arity[s];
      final_len >>= arity[s];
    }
  }
  // tree height (in levels below the cap) of the step-s tree
  unsigned step_height(uint32_t s) const {
    unsigned lg = lgN;
    for (uint32_t t = 0; t <= s; t++) lg -= arity[t];
    return lg - cap_h;
  }
};

struct Out {
  std::vector<uint8_t> v;
  void put(const void *p, size_t n) {
    const uint8_t *q = (const uint8_t *)p;
    v.insert(v.end(), q, q + n);
  }
  void u8(uint8_t x) { v.push_back(x); }
  void u32(uint32_t x) { put(&x, 4); }
  void u64(uint64_t x) { put(&x, 8); }
  void ext(ext_t e) { u64(e.c0); u64(e.c1); }
  void dig(const dig_t &d) { put(d.w, hh_bytes()); }
  void path(const std::vector<dig_t> &p) {
    u8((uint8_t)p.size());
    for (auto &d : p) dig(d);
  }
};

int bad(const char *what) {
  set_err("malformed proof: %s", what);
  return P2GPU_E_VERIFY;
}

// ---- the parts both layouts share ----
bool read_head(Cursor &in, const Shape &S, ProofData &P) {
  for (auto &cap : P.caps) in.digests(cap, S.ncap);
  P.op.resize(S.nall + S.K);
  for (auto &e : P.op) e = in.ext();
  P.step_caps.resize(S.n_steps);
  for (auto &sc : P.step_caps) in.digests(sc, S.ncap);
  return in.ok;
}
bool read_tail(Cursor &in, const Shape &S, uint32_t num_pi, ProofData &P) {
  P.final_poly.resize(S.final_len);
  for (auto &e : P.final_poly) e = in.ext();
  P.pow_witness = in.u64();
  P.pis.resize(num_pi);
  for (auto &v : P.pis) v = in.felt();
  return in.ok;
}
void write_head(Out &o, const ProofData &P) {
  for (auto &cap : P.caps)
    for (auto &d : cap) o.dig(d);
  for (auto &e : P.op) o.ext(e);
  for (auto &sc : P.step_caps)
    for (auto &d : sc) o.dig(d);
}
void write_tail(Out &o, const ProofData &P) {
  for (auto &e : P.final_poly) o.ext(e);
  o.u64(P.pow_witness);
  for (auto v : P.pis) o.u64(v);
}
bool read_leaf(Cursor &in, size_t n, std::vector<gl_t> &leaf) {
  leaf.resize(n);
  for (auto &v : leaf) v = in.felt();
  return in.ok;
}
bool read_path(Cursor &in, size_t max_len, std::vector<dig_t> &p) {
  const uint8_t *l = in.take(1);
  if (!l || *l > max_len) return in.ok = false;
  in.digests(p, *l);
  return in.ok;
}

int parse_uncompressed(const p2gpu_circuit *c, const uint8_t *proof, size_t len, ProofData &P) {
  const Shape S(c);
  Cursor in{proof, len};
  if (!read_head(in, S, P)) return bad("caps / openings");
  P.queries.resize(S.queries);
  for (auto &q : P.queries) {
    for (int o = 0; o < 4; o++)
      if (!read_leaf(in, S.cols[o], q.leaf[o]) || !read_path(in, S.lgN - S.cap_h, q.path[o]) ||
          q.path[o].size() != S.lgN - S.cap_h)
        return bad("initial tree opening");
    q.evals.resize(S.n_steps);
    q.spath.resize(S.n_steps);
    for (uint32_t s = 0; s < S.n_steps; s++) {
      q.evals[s].resize((size_t)1 << S.arity[s]);
      for (auto &e : q.evals[s]) e = in.ext();
      if (!read_path(in, S.step_height(s), q.spath[s]) || q.spath[s].size() != S.step_height(s)) return bad("fold step");
    }
  }
  if (!read_tail(in, S, c->num_pi, P) || in.at != len) return bad("final polynomial / length");
  return P2GPU_OK;
}

void write_uncompressed(const ProofData &P, Out &o) {
  write_head(o, P);
  for (auto &q : P.queries) {
    for (int t = 0; t < 4; t++) {
      o.put(q.leaf[t].data(), 8 * q.leaf[t].size());
      o.path(q.path[t]);
    }
    for (size_t s = 0; s < q.evals.size(); s++) {
      for (auto &e : q.evals[s]) o.ext(e);
      o.path(q.spath[s]);
    }
  }
  write_tail(o, P);
}

// ---- transcript: everything the (de)compressor needs from plonk/get_challenges.rs ----
struct Challenges {
  ext_t zeta, alpha;
  std::vector<ext_t> fold_betas;
  std::vector<size_t> indices;
};
Challenges derive_challenges(const p2gpu_circuit *c, const ProofData &P) {
  const Shape S(c);
  Challenges ch;
  gl_t pih[4];
  poseidon_hash_no_pad_host(P.pis.data(), P.pis.size(), pih, c->poseidon_rc);
  Challenger t;
  t.observe_digest(c->circuit_digest);
  for (int i = 0; i < 4; i++) t.observe(pih[i]);
  t.observe_cap(P.caps[0]);
  for (uint32_t k = 0; k < 2 * S.K; k++) (void)t.get();  // betas, gammas
  t.observe_cap(P.caps[1]);
  for (uint32_t k = 0; k < S.K; k++) (void)t.get();      // alphas
  t.observe_cap(P.caps[2]);
  ch.zeta = t.get_ext();
  // to_fri_openings: the zeta batch (everything but zs_next), then zs_next
  const uint32_t head = S.ncs + S.W + S.K;
  for (uint32_t j = 0; j < head; j++) t.observe_ext(P.op[j]);
  for (uint32_t j = head + S.K; j < S.nall + S.K; j++) t.observe_ext(P.op[j]);
  for (uint32_t j = head; j < head + S.K; j++) t.observe_ext(P.op[j]);
  ch.alpha = t.get_ext();
  for (auto &sc : P.step_caps) {
    t.observe_cap(sc);
    ch.fold_betas.push_back(t.get_ext());
  }
  for (auto &e : P.final_poly) t.observe_ext(e);
  t.observe(P.pow_witness);
  (void)t.get();  // PoW response
  ch.indices.resize(S.queries);
  for (auto &x : ch.indices) x = (size_t)(t.get() & (c->N - 1));
  return ch;
}

// ---- hash/merkle_proofs.rs compress_merkle_proofs / decompress_merkle_proofs ----
// heap numbering below the cap: node = (leaf index + 2^height) >> level, `height` levels in total
std::vector<std::vector<dig_t>> compress_paths(unsigned height_total, unsigned cap_h, const std::vector<size_t> &idx,
                                               const std::vector<const std::vector<dig_t> *> &paths) {
  const size_t nl = (size_t)1 << height_total;
  std::set<size_t> known;
  for (size_t x : idx)
    for (unsigned j = 0; j < height_total - cap_h; j++) known.insert((x + nl) >> j);
  std::vector<std::vector<dig_t>> out(idx.size());
  for (size_t q = 0; q < idx.size(); q++) {
    size_t node = idx[q] + nl;
    for (const dig_t &sib : *paths[q]) {
      if (known.insert(node ^ 1).second) out[q].push_back(sib);
      node >>= 1;
    }
  }
  return out;
}
// returns false when a compressed path is too short / too long
bool decompress_paths(unsigned height_total, unsigned cap_h, const std::vector<size_t> &idx,
                      const std::vector<dig_t> &leaf_hash, const std::vector<const std::vector<dig_t> *> &cpaths,
                      std::vector<std::vector<dig_t>> &out) {
  const size_t nl = (size_t)1 << height_total;
  std::map<size_t, dig_t> seen;
  for (size_t q = 0; q < idx.size(); q++) seen[idx[q] + nl] = leaf_hash[q];
  std::vector<size_t> used(idx.size(), 0);
  for (unsigned layer = 0; layer < height_total - cap_h; layer++)
    for (size_t q = 0; q < idx.size(); q++) {
      const size_t node = (idx[q] + nl) >> layer;
      auto it = seen.find(node ^ 1);
      if (it == seen.end()) {
        if (used[q] >= cpaths[q]->size()) return false;
        it = seen.emplace(node ^ 1, (*cpaths[q])[used[q]++]).first;
      }
      const dig_t &cur = seen[node], &sib = it->second;
      seen[node >> 1] = (node & 1) ? node_digest(sib, cur) : node_digest(cur, sib);
    }
  // duplicate queries share one stored path: only the first copy must be used up
  std::set<size_t> first;
  for (size_t q = 0; q < idx.size(); q++)
    if (first.insert(idx[q]).second && used[q] != cpaths[q]->size()) return false;
  out.assign(idx.size(), {});
  for (size_t q = 0; q < idx.size(); q++) {
    size_t node = idx[q] + nl;
    for (unsigned layer = 0; layer < height_total - cap_h; layer++) {
      out[q].push_back(seen[node ^ 1]);
      node >>= 1;
    }
  }
  return true;
}

// fri_combine_initial at the point of leaf index x (fri/verifier.rs)
ext_t combine_initial(const p2gpu_circuit *c, const Shape &S, const ProofData &P, const Challenges &ch,
                      const QueryRound &q, size_t x, gl_t *point_out) {
  // batch order differs from the serialised opening set: zs_next is its own batch
  const uint32_t head = S.ncs + S.W + S.K;
  ext_t opened0 = ext_from(0), opened1 = ext_from(0);
  for (uint32_t j = S.nall + S.K; j-- > head + S.K;) opened0 = ext_add(ext_mul(opened0, ch.alpha), P.op[j]);
  for (uint32_t j = head; j-- > 0;) opened0 = ext_add(ext_mul(opened0, ch.alpha), P.op[j]);
  for (uint32_t j = head + S.K; j-- > head;) opened1 = ext_add(ext_mul(opened1, ch.alpha), P.op[j]);
  size_t e = 0;
  for (unsigned b = 0; b < S.lgN; b++) e |= ((x >> b) & 1) << (S.lgN - 1 - b);
  const gl_t pt = gl_mul(GL_GEN, gl_pow(gl_root(S.lgN), e));
  ext_t acc0 = ext_from(0), acc1 = ext_from(0);
  for (int o = 3; o >= 0; o--)
    for (size_t j = q.leaf[o].size(); j-- > 0;) acc0 = ext_add(ext_mul(acc0, ch.alpha), ext_from(q.leaf[o][j]));
  for (size_t j = S.K; j-- > 0;) acc1 = ext_add(ext_mul(acc1, ch.alpha), ext_from(q.leaf[2][j]));
  const ext_t gz = ext_scale(ch.zeta, gl_root(S.d));
  ext_t cur = ext_mul(ext_sub(acc0, opened0), ext_inv(ext_sub(ext_from(pt), ch.zeta)));
  cur = ext_add(ext_mul(cur, ext_pow(ch.alpha, S.K)), ext_mul(ext_sub(acc1, opened1), ext_inv(ext_sub(ext_from(pt), gz))));
  (void)c;
  *point_out = pt;
  return cur;
}

// one fold: the value that continues into the next step (fri/verifier.rs compute_evaluation)
ext_t fold_value(const Shape &S, uint32_t s, unsigned lg, size_t x, gl_t shift, const std::vector<ext_t> &leaf,
                 ext_t beta) {
  const unsigned ab = S.arity[s];
  ext_t nat[64];
  for (uint32_t i = 0; i < (1u << ab); i++) nat[bitrev32(i, ab)] = leaf[i];
  size_t e = 0;
  for (unsigned b = 0; b < lg; b++) e |= ((x >> b) & 1) << (lg - 1 - b);
  e &= ((size_t)1 << (lg - ab)) - 1;
  return interpolate_coset(gl_mul(shift, gl_pow(gl_root(lg), e)), ab, nat, beta);
}

int do_compress(const p2gpu_circuit *c, const ProofData &P, Out &o) {
  const Shape S(c);
  const Challenges ch = derive_challenges(c, P);
  write_head(o, P);
  for (size_t x : ch.indices) o.u32((uint32_t)x);
  // initial trees: compress the paths tree by tree, then one entry per distinct index
  std::vector<std::vector<dig_t>> cp[4];
  for (int t = 0; t < 4; t++) {
    std::vector<const std::vector<dig_t> *> paths;
    for (auto &q : P.queries) paths.push_back(&q.path[t]);
    cp[t] = compress_paths(S.lgN, S.cap_h, ch.indices, paths);
  }
  std::map<size_t, size_t> first;  // index -> first query with it
  for (size_t q = 0; q < ch.indices.size(); q++) first.emplace(ch.indices[q], q);
  for (auto &kv : first)
    for (int t = 0; t < 4; t++) {
      o.put(P.queries[kv.second].leaf[t].data(), 8 * P.queries[kv.second].leaf[t].size());
      o.path(cp[t][kv.second]);
    }
  // fold steps
  std::vector<size_t> cur = ch.indices;
  unsigned lg = S.lgN;
  for (uint32_t s = 0; s < S.n_steps; s++) {
    const unsigned ab = S.arity[s];
    std::vector<size_t> within(cur.size());
    for (size_t q = 0; q < cur.size(); q++) {
      within[q] = cur[q] & ((1u << ab) - 1);
      cur[q] >>= ab;
    }
    lg -= ab;
    std::vector<const std::vector<dig_t> *> paths;
    for (auto &q : P.queries) paths.push_back(&q.spath[s]);
    const auto cps = compress_paths(lg, S.cap_h, cur, paths);
    std::map<size_t, size_t> firsts;
    for (size_t q = 0; q < cur.size(); q++) firsts.emplace(cur[q], q);
    for (auto &kv : firsts) {
      const auto &ev = P.queries[kv.second].evals[s];
      for (size_t i = 0; i < ev.size(); i++)
        if (i != within[kv.second]) o.ext(ev[i]);
      o.path(cps[kv.second]);
    }
  }
  write_tail(o, P);
  return P2GPU_OK;
}

int do_decompress(const p2gpu_circuit *c, const uint8_t *data, size_t len, ProofData &P) {
  const Shape S(c);
  Cursor in{data, len};
  if (!read_head(in, S, P)) return bad("caps / openings");
  std::vector<size_t> idx(S.queries);
  for (auto &x : idx) {
    const uint8_t *p = in.take(4);
    uint32_t v = 0;
    if (p) memcpy(&v, p, 4);
    x = v;
    if (x >= c->N) in.ok = false;
  }
  if (!in.ok) return bad("query indices");
  // entries sorted by index
  std::set<size_t> distinct(idx.begin(), idx.end());
  struct InitEntry {
    std::vector<gl_t> leaf[4];
    std::vector<dig_t> cpath[4];
  };
  std::map<size_t, InitEntry> init;
  for (size_t x : distinct) {
    InitEntry &e = init[x];
    for (int t = 0; t < 4; t++)
      if (!read_leaf(in, S.cols[t], e.leaf[t]) || !read_path(in, S.lgN - S.cap_h, e.cpath[t])) return bad("initial tree entry");
  }
  struct StepEntry {
    std::vector<ext_t> evals;  // arity - 1 values
    std::vector<dig_t> cpath;
  };
  std::vector<std::map<size_t, StepEntry>> steps(S.n_steps);
  {
    std::vector<size_t> cur = idx;
    for (uint32_t s = 0; s < S.n_steps; s++) {
      std::set<size_t> ds;
      for (auto &x : cur) {
        x >>= S.arity[s];
        ds.insert(x);
      }
      for (size_t x : ds) {
        StepEntry &e = steps[s][x];
        e.evals.resize(((size_t)1 << S.arity[s]) - 1);
        for (auto &v : e.evals) v = in.ext();
        if (!read_path(in, S.step_height(s), e.cpath)) return bad("fold step entry");
      }
    }
  }
  if (!read_tail(in, S, c->num_pi, P) || in.at != len) return bad("final polynomial / length");

  // the stored indices must be the transcript's (CompressedProofWithPublicInputs::decompress
  // takes them from the challenges)
  const Challenges ch = derive_challenges(c, P);
  if (ch.indices != idx) return bad("query indices differ from the transcript's");

  P.queries.assign(S.queries, {});
  for (size_t q = 0; q < idx.size(); q++)
    for (int t = 0; t < 4; t++) P.queries[q].leaf[t] = init[idx[q]].leaf[t];
  for (int t = 0; t < 4; t++) {
    std::vector<dig_t> lh;
    std::vector<const std::vector<dig_t> *> cps;
    for (size_t q = 0; q < idx.size(); q++) {
      lh.push_back(leaf_digest(P.queries[q].leaf[t].data(), P.queries[q].leaf[t].size()));
      cps.push_back(&init[idx[q]].cpath[t]);
    }
    std::vector<std::vector<dig_t>> full;
    if (!decompress_paths(S.lgN, S.cap_h, idx, lh, cps, full)) return bad("compressed Merkle path (initial tree)");
    for (size_t q = 0; q < idx.size(); q++) P.queries[q].path[t] = full[q];
  }
  // fold steps: re-derive the dropped evaluation for the first query that reaches a leaf, reuse it after
  std::vector<std::map<size_t, std::vector<ext_t>>> rebuilt(S.n_steps);
  for (size_t q = 0; q < idx.size(); q++) {
    QueryRound &Q = P.queries[q];
    Q.evals.resize(S.n_steps);
    Q.spath.resize(S.n_steps);
    size_t x = idx[q];
    gl_t pt;
    ext_t cur = combine_initial(c, S, P, ch, Q, x, &pt);
    unsigned lg = S.lgN;
    gl_t shift = GL_GEN;
    bool live = true;  // false once a leaf was already rebuilt: the rest of this query's chain is known
    for (uint32_t s = 0; s < S.n_steps; s++) {
      const unsigned ab = S.arity[s];
      const size_t within = x & ((1u << ab) - 1), leaf_index = x >> ab;
      auto it = rebuilt[s].find(leaf_index);
      if (it == rebuilt[s].end()) {
        if (!live) return bad("fold chain");  // cannot happen: equal leaf at step s implies equal leaves after
        std::vector<ext_t> full = steps[s][leaf_index].evals;
        full.insert(full.begin() + within, cur);
        it = rebuilt[s].emplace(leaf_index, full).first;
      } else {
        live = false;
      }
      Q.evals[s] = it->second;
      if (live) cur = fold_value(S, s, lg, x, shift, Q.evals[s], ch.fold_betas[s]);
      for (unsigned i = 0; i < ab; i++) shift = gl_sqr(shift);
      lg -= ab;
      x = leaf_index;
    }
  }
  {
    std::vector<size_t> cur = idx;
    unsigned lg = S.lgN;
    for (uint32_t s = 0; s < S.n_steps; s++) {
      for (auto &x : cur) x >>= S.arity[s];
      lg -= S.arity[s];
      std::vector<dig_t> lh;
      std::vector<const std::vector<dig_t> *> cps;
      for (size_t q = 0; q < cur.size(); q++) {
        const auto &ev = P.queries[q].evals[s];
        std::vector<gl_t> flat;
        for (auto &e : ev) {
          flat.push_back(e.c0);
          flat.push_back(e.c1);
        }
        lh.push_back(leaf_digest(flat.data(), flat.size()));
        cps.push_back(&steps[s][cur[q]].cpath);
      }
      std::vector<std::vector<dig_t>> full;
      if (!decompress_paths(lg, S.cap_h, cur, lh, cps, full)) return bad("compressed Merkle path (fold step)");
      for (size_t q = 0; q < cur.size(); q++) P.queries[q].spath[s] = full[q];
    }
  }
  return P2GPU_OK;
}

int emit(const Out &o, uint8_t *out, size_t *out_len) {
  if (!out || *out_len < o.v.size()) {
    const bool probe = out == nullptr;
    *out_len = o.v.size();
    if (probe) return P2GPU_OK;
    set_err("output buffer too small: %zu bytes needed", o.v.size());
    return P2GPU_E_BUFFER;
  }
  memcpy(out, o.v.data(), o.v.size());
  *out_len = o.v.size();
  return P2GPU_OK;
}

}  // namespace

extern "C" {

int p2gpu_proof_compress(const p2gpu_circuit *c, const uint8_t *proof, size_t len, uint8_t *out, size_t *out_len) try {
  use_hasher(c);
  if (!c || !proof || !out_len) return P2GPU_E_ARG;
  ProofData P;
  if (int rc = parse_uncompressed(c, proof, len, P)) return rc;
  Out o;
  if (int rc = do_compress(c, P, o)) return rc;
  return emit(o, out, out_len);
} P2GPU_CATCH

int p2gpu_proof_decompress(const p2gpu_circuit *c, const uint8_t *cproof, size_t len, uint8_t *out, size_t *out_len) try {
  use_hasher(c);
  if (!c || !cproof || !out_len) return P2GPU_E_ARG;
  ProofData P;
  if (int rc = do_decompress(c, cproof, len, P)) return rc;
  Out o;
  write_uncompressed(P, o);
  return emit(o, out, out_len);
} P2GPU_CATCH

int p2gpu_verify_compressed(const p2gpu_circuit *c, const uint8_t *cproof, size_t len) try {
  use_hasher(c);
  if (!c || !cproof) return P2GPU_E_ARG;
  ProofData P;
  if (int rc = do_decompress(c, cproof, len, P)) return rc;
  Out o;
  write_uncompressed(P, o);
  return p2gpu_verify(c, o.v.data(), o.v.size());
} P2GPU_CATCH

}  // extern "C"
