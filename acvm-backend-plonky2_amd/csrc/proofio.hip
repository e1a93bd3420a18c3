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


// ---- the reference's verifier-key file (second half) -----------------------------------------
// `plonky2-backend write_vk` stores `verifier_data.to_bytes(&BackendGateSerializer)` (plonky2-backend/src/actions/
// write_vk_action.rs:35-81) and `verify` reads it back with `VerifierCircuitData::from_bytes`
// (noir_and_plonky2_serialization.rs:16-22).  The byte layout lives in the un-vendored crate (plonky2 0.2.2
// util/serialization/mod.rs: write_verifier_only_circuit_data + write_common_circuit_data) and no VK file exists
// in the reference tree: this is a restatement from recollection, **UNPINNED** -- nothing the reference produced
// checks it; what IS pinned is the gate tag order (the list at write_vk_action.rs:39-61, tags count from 0) and
// the bodies of the five custom gates (their `serialize` at add_many_u32.rs:94-97, arithmetic_u32.rs:93-95,
// comparison.rs:104-108, range_check_u32.rs:59-61, subtraction_u32.rs:87-89).  Layout written / read here
// (usize = u64 LE, bool = u8, field = canonical u64 LE, hash = 25 bytes for KeccakHash<25> / 32 for Poseidon):
// (VerifierCircuitData::to_bytes = write_verifier_circuit_data: CommonCircuitData FIRST, VerifierOnlyCircuitData LAST)
//   CircuitConfig: usize num_wires, num_routed_wires, num_constants, security_bits, num_challenges,
//                  max_quotient_degree_factor | bool use_base_arithmetic_gate, zero_knowledge | FriConfig
//   FriConfig: usize rate_bits, cap_height, num_query_rounds | u32 proof_of_work_bits |
//              strategy: u8 0 Fixed(usizes) / 1 ConstantArityBits(usize, usize) / 2 MinSize(u8 flag [usize])
//   FriParams: FriConfig | usizes reduction_arity_bits (len, items) | usize degree_bits | bool hiding
//   SelectorsInfo: usizes selector_indices | usize #groups | (usize start, usize end) each
//   usize quotient_degree_factor, num_gate_constraints, num_constants, num_public_inputs | usize #k_is | field k_is[]
//   usize num_partial_products, num_lookup_polys, num_lookup_selectors | usize #luts (0)
//   usize #gates | per gate: u32 tag | the gate's own serialize
//   usize cap_height | hash cap[2^cap_height] | hash circuit_digest        (verifier_only)
// The configuration constants the blob does not carry are those of CircuitConfig::wide_ecc_config
// (circuit_translation/mod.rs:69 = standard_recursion_config with 234 / 80 wires): num_constants 2,
// security_bits 100, max_quotient_degree_factor 8, use_base_arithmetic_gate, no zero knowledge,
// ConstantArityBits(4, 5).
struct VkGateTag {
  uint32_t tag, kind, base;  // base: BaseSumGate<B>
};
const VkGateTag VK_TAGS[] = {{0, G_ARITHMETIC, 0},     {2, G_BASE_SUM, 2},        {3, G_BASE_SUM, 4},       {4, G_CONSTANT, 0},
                             {10, G_NOOP, 0},          {12, G_POSEIDON, 0},       {13, G_PUBLIC_INPUT, 0},  {14, G_RANDOM_ACCESS, 0},
                             {17, G_COMPARISON, 0},    {18, G_U32_ADD_MANY, 0},   {19, G_U32_ARITHMETIC, 0}, {20, G_U32_RANGE_CHECK, 0},
                             {21, G_U32_SUBTRACTION, 0}};
// Gate::degree() / num_constants() of the kinds the translators emit (gates/*.rs; the custom ones at
// arithmetic_u32.rs:277-279, add_many_u32.rs:278-280, subtraction_u32.rs:222-224, range_check_u32.rs:168-170,
// comparison.rs:325-327)
void vk_gate_shape(uint32_t kind, const uint32_t p[4], uint32_t *degree, uint32_t *nconst) {
  *nconst = 0;
  switch (kind) {
  case G_NOOP: *degree = 0; break;
  case G_CONSTANT: *degree = 1; *nconst = p[0]; break;
  case G_PUBLIC_INPUT: *degree = 1; break;
  case G_ARITHMETIC: *degree = 3; *nconst = 2; break;
  case G_BASE_SUM: *degree = p[0]; break;
  case G_RANDOM_ACCESS: *degree = p[0] + 1; *nconst = p[2]; break;
  case G_POSEIDON: *degree = 7; break;
  case G_COMPARISON: *degree = 1u << ((p[0] + (p[1] ? p[1] : 1) - 1) / (p[1] ? p[1] : 1)); break;
  default: *degree = 4; break;  // the four u32 gates: 1 << limb_bits, limb_bits = 2
  }
}

struct VkOut : Out {
  void usize(uint64_t x) { u64(x); }
  void boolean(bool b) { u8(b ? 1 : 0); }
};
void vk_write_fri_config(VkOut &o, const p2gpu_circuit *c) {
  o.usize(c->rate_bits); o.usize(c->cap_h); o.usize(c->num_queries); o.u32(c->pow_bits);
  // ConstantArityBits(4, 5) when the steps are what that strategy yields for this degree, Fixed(steps) otherwise
  std::vector<uint32_t> want;
  {
    uint32_t dbits = c->d;
    while (dbits > 5 && dbits + c->rate_bits - 4 >= c->cap_h) { want.push_back(4); dbits -= 4; }
  }
  bool constant = want.size() == c->n_steps;
  for (uint32_t i = 0; constant && i < c->n_steps; i++) constant = c->arity[i] == 4;
  if (constant) { o.u8(1); o.usize(4); o.usize(5); }
  else { o.u8(0); o.usize(c->n_steps); for (uint32_t i = 0; i < c->n_steps; i++) o.usize(c->arity[i]); }
}
int vk_write(const p2gpu_circuit *c, VkOut &o) {
  if (c->cs.cap.size() != ((size_t)1 << c->cap_h)) { set_err("handle has no constants_sigmas cap"); return P2GPU_E_ARG; }
  o.usize(c->W); o.usize(c->R); o.usize(2); o.usize(100); o.usize(c->K); o.usize(8);
  o.boolean(true); o.boolean(false);
  vk_write_fri_config(o, c);
  vk_write_fri_config(o, c);
  o.usize(c->n_steps);
  for (uint32_t i = 0; i < c->n_steps; i++) o.usize(c->arity[i]);
  o.usize(c->d);
  o.boolean(false);
  o.usize(c->num_gates);
  for (auto &G : c->gates) o.usize(G.sel_index);
  {
    std::vector<std::pair<uint32_t, uint32_t>> groups;
    for (auto &G : c->gates)
      if (groups.empty() || groups.back().first != G.group_start) groups.push_back({G.group_start, G.group_end});
    o.usize(groups.size());
    for (auto &g : groups) { o.usize(g.first); o.usize(g.second); }
  }
  o.usize(c->QF);
  o.usize(c->max_gate_constraints);
  o.usize(c->NC);
  o.usize(c->num_pi);
  o.usize(c->R);
  for (auto k : c->k_is) o.u64(k);
  o.usize(c->PP); o.usize(0); o.usize(0); o.usize(0);
  o.usize(c->num_gates);
  for (auto &G : c->gates) {
    const VkGateTag *t = nullptr;
    for (auto &e : VK_TAGS)
      if (e.kind == G.kind && (G.kind != G_BASE_SUM || e.base == G.p[0])) t = &e;
    if (!t) { set_err("gate kind %u (BaseSum base %u) is not in the reference's BackendGateSerializer", G.kind, G.p[0]); return P2GPU_E_ARG; }
    o.u32(t->tag);
    switch (G.kind) {
    case G_ARITHMETIC: case G_CONSTANT: case G_U32_ARITHMETIC: case G_U32_SUBTRACTION: case G_U32_RANGE_CHECK: o.usize(G.p[0]); break;
    case G_BASE_SUM: o.usize(G.p[1]); break;
    case G_RANDOM_ACCESS: o.usize(G.p[0]); o.usize(G.p[1]); o.usize(G.p[2]); break;
    case G_U32_ADD_MANY: case G_COMPARISON: o.usize(G.p[0]); o.usize(G.p[1]); break;
    default: break;  // NoopGate, PoseidonGate, PublicInputGate: empty bodies
    }
  }
  // VerifierOnlyCircuitData comes LAST (write_verifier_circuit_data: common, then verifier_only)
  o.usize(c->cap_h);
  for (auto &dg : c->cs.cap) o.dig(dg);
  o.dig(c->circuit_digest);
  return P2GPU_OK;
}

struct VkIn : Cursor {
  uint64_t usize(uint64_t max) {
    const uint64_t v = u64();
    if (v > max) ok = false;
    return ok ? v : 0;
  }
  uint32_t u32() {
    uint32_t v = 0;
    if (const uint8_t *q = take(4)) memcpy(&v, q, 4);
    return v;
  }
  int boolean() {
    const uint8_t *q = take(1);
    if (q && *q > 1) ok = false;
    return q ? *q : 0;
  }
};
struct VkFri {
  uint32_t rate_bits, cap_h, queries, pow_bits;
};
bool vk_read_fri_config(VkIn &in, VkFri &f) {
  f.rate_bits = (uint32_t)in.usize(3); f.cap_h = (uint32_t)in.usize(32); f.queries = (uint32_t)in.usize(64); f.pow_bits = in.u32();
  const uint8_t *tag = in.take(1);
  if (!tag) return false;
  if (*tag == 0) { const uint64_t n = in.usize(8); for (uint64_t i = 0; i < n; i++) in.usize(MAX_ARITY_BITS); }
  else if (*tag == 1) { in.usize(MAX_ARITY_BITS); in.usize(32); }
  else if (*tag == 2) { const uint8_t *some = in.take(1); if (some && *some == 1) in.usize(1u << 24); else if (!some || *some > 1) in.ok = false; }
  else in.ok = false;
  return in.ok;
}
// plonky2 VK bytes -> the verifier blob of p2gpu_verifier_create (header | gate table | cap | k_is, flags 0b11)
int vk_read(const uint8_t *bytes, size_t len, int hasher, Out &blob) {
  VkIn in;
  in.p = bytes; in.len = len;
  const size_t hb = hasher ? 32 : 25;
  auto fail = [](const char *what) { set_err("malformed verifier key: %s", what); return P2GPU_E_BLOB; };
  uint32_t h[64];
  memset(h, 0, sizeof h);
  h[0] = 0x43473250u; h[1] = 1;
  h[3] = (uint32_t)in.usize(4096); h[4] = (uint32_t)in.usize(MAX_ROUTED);
  in.usize(4096);  // config.num_constants
  in.usize(1024);  // security_bits
  h[7] = (uint32_t)in.usize(MAX_CHALLENGES);
  in.usize(64);    // max_quotient_degree_factor
  in.boolean();
  if (in.boolean()) return fail("zero-knowledge circuits are not supported");
  VkFri f0, f1;
  if (!vk_read_fri_config(in, f0) || !vk_read_fri_config(in, f1)) return fail("FRI configuration");
  if (f0.rate_bits != f1.rate_bits || f0.cap_h != f1.cap_h || f0.queries != f1.queries || f0.pow_bits != f1.pow_bits)
    return fail("CircuitConfig.fri_config and FriParams.config disagree");
  h[9] = f0.rate_bits; h[10] = f0.cap_h; h[11] = f0.pow_bits; h[12] = f0.queries;
  const uint32_t n_steps = (uint32_t)in.usize(8);
  h[13] = n_steps;
  for (uint32_t i = 0; i < n_steps; i++) h[14 + i] = (uint32_t)in.usize(MAX_ARITY_BITS);
  h[2] = (uint32_t)in.usize(24);
  if (in.boolean()) return fail("hiding FRI is not supported");
  h[22] = (uint32_t)hasher;
  const uint32_t nsel = (uint32_t)in.usize(MAX_GATES);
  if (!in.ok) return fail("header fields out of range");
  std::vector<uint32_t> sel(nsel);
  for (auto &x : sel) x = (uint32_t)in.usize(4096);
  const uint32_t ngroups = (uint32_t)in.usize(MAX_GATES);
  std::vector<std::pair<uint32_t, uint32_t>> groups(in.ok ? ngroups : 0);
  for (auto &g : groups) { g.first = (uint32_t)in.usize(MAX_GATES); g.second = (uint32_t)in.usize(MAX_GATES); }
  h[8] = (uint32_t)in.usize(64);
  const uint32_t num_gate_constraints = (uint32_t)in.usize(MAX_GATE_CONSTRAINTS);
  h[5] = (uint32_t)in.usize(4096);
  h[24] = (uint32_t)in.usize(1u << 20);
  const uint32_t nk = (uint32_t)in.usize(MAX_ROUTED);
  if (!in.ok || nk != h[4]) return fail("selectors / k_is");
  std::vector<gl_t> k_is(nk);
  for (auto &k : k_is) k = in.felt();
  h[26] = (uint32_t)in.usize(64);
  if (in.usize(0) || in.usize(0) || in.usize(0) || !in.ok) return fail("lookup tables are not supported");
  const uint32_t ngates = (uint32_t)in.usize(MAX_GATES);
  if (!in.ok || ngates != nsel || ngates == 0) return fail("gate count");
  h[23] = ngates;
  h[6] = 0;
  for (auto x : sel) h[6] = std::max(h[6], x + 1);  // selector columns come first in the constants table
  h[25] = 3;
  Out gate_rows;  // the header goes out first, but the digest it carries is read last
  uint32_t max_c = 0;
  for (uint32_t i = 0; i < ngates; i++) {
    const uint32_t tag = in.u32();
    const VkGateTag *t = nullptr;
    for (auto &e : VK_TAGS)
      if (e.tag == tag) t = &e;
    if (!in.ok || !t) return fail("gate tag outside the kinds the translators emit");
    uint32_t g[12];
    memset(g, 0, sizeof g);
    g[0] = t->kind;
    switch (t->kind) {
    case G_ARITHMETIC: case G_CONSTANT: case G_U32_ARITHMETIC: case G_U32_SUBTRACTION: case G_U32_RANGE_CHECK: g[1] = (uint32_t)in.usize(4096); break;
    case G_BASE_SUM: g[1] = t->base; g[2] = (uint32_t)in.usize(64); break;
    case G_RANDOM_ACCESS: g[1] = (uint32_t)in.usize(6); g[2] = (uint32_t)in.usize(1024); g[3] = (uint32_t)in.usize(2); break;
    case G_U32_ADD_MANY: case G_COMPARISON: g[1] = (uint32_t)in.usize(1024); g[2] = (uint32_t)in.usize(1024); break;
    default: break;
    }
    if (!in.ok) return fail("gate parameters");
    g[5] = sel[i];
    g[6] = g[7] = 0;
    for (auto &gr : groups)
      if (gr.first <= i && i < gr.second) { g[6] = gr.first; g[7] = gr.second; }
    // constraint count, degree and constant count follow from the kind; circuit_parse re-validates all of it
    {
      uint64_t wu = 0;
      uint32_t cu = 0;
      if (h[5] < h[6]) return fail("fewer constant columns than selectors");
      if (const char *why = gate_validate(g[0], &g[1], h[3], h[5] - h[6], &wu, &cu)) return fail(why);
    }
    uint32_t deg, nconst;
    vk_gate_shape(g[0], &g[1], &deg, &nconst);
    g[9] = deg; g[10] = nconst;
    g[8] = gate_num_constraints(g[0], &g[1]);
    max_c = std::max(max_c, g[8]);
    gate_rows.put(g, sizeof g);
  }
  if (max_c != num_gate_constraints) return fail("num_gate_constraints does not match the gate set");
  // VerifierOnlyCircuitData: usize cap_height | cap | circuit_digest
  const uint32_t cap_h = (uint32_t)in.usize(16);
  if (!in.ok) return fail("cap height");
  if (cap_h != f0.cap_h) return fail("constants_sigmas_cap height and FriConfig.cap_height disagree");
  const uint8_t *cap = in.take(hb << cap_h);
  const uint8_t *digest = in.take(hb);
  if (!cap || !digest) return fail("truncated (cap / digest)");
  if (in.at != in.len) return fail("trailing bytes");
  memcpy(&h[32], digest, hb);
  blob.put(h, sizeof h);
  blob.put(gate_rows.v.data(), gate_rows.v.size());
  for (size_t i = 0; i < ((size_t)1 << cap_h); i++) {
    uint8_t e[32];
    memset(e, 0, sizeof e);
    memcpy(e, cap + hb * i, hb);
    blob.put(e, 32);
  }
  blob.put(k_is.data(), 8 * k_is.size());
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

// ---- verifier key in the reference's file format (UNPINNED restatement, see vk_write above) ----
int p2gpu_circuit_export_vk_plonky2(const p2gpu_circuit *c, uint8_t *out, size_t *out_len) try {
  if (!c || !out_len) return P2GPU_E_ARG;
  use_hasher(c);
  VkOut o;
  if (int rc = vk_write(c, o)) return rc;
  return emit(o, out, out_len);
} P2GPU_CATCH

int p2gpu_verifier_create_plonky2(const uint8_t *vk, size_t len, int hasher, p2gpu_circuit **out) try {
  if (!vk || !out || hasher < 0 || hasher > 1) return P2GPU_E_ARG;
  Out blob;
  if (int rc = vk_read(vk, len, hasher, blob)) return rc;
  return p2gpu_verifier_create(blob.v.data(), blob.v.size(), out);
} P2GPU_CATCH

}  // extern "C"
