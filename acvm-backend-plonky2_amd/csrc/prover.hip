// prover.hip -- the proof: host pipeline + the `p2gpu_prove*` entry points of libp2gpu.so (see include/p2gpu.h; the circuit
// handle is handle.hip, the exchanges of a sharded proof transport.hip).
//
// Drop-in for the one call `circuit_data.prove(witnesses)` at
// plonky2-backend/src/actions/prove_action.rs:96 (after witness generation):
// plonky2 0.2.2 plonk/prover.rs prove_with_partition_witness restated as a
// sequence of HIP kernel launches on one stream, with the Fiat-Shamir
// transcript (iop/challenger.rs, Keccak duplex) run on the host between
// phases -- each challenge needs only a 16 x 25 B Merkle cap or a few hundred
// opening values back from the device.  Transcript order: SURVEY.md C.4.
// The product path never touches oracle/; without a HIP device every entry
// point fails with P2GPU_E_DEVICE.
#include "circuit.hpp"
#include "prover_internal.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace p2;

namespace p2 {
thread_local Prof *g_prof = nullptr;
}  // namespace p2
namespace {
// P2GPU_TRACE=1: synchronise after every phase and report progress on stderr (debugging aid)
}  // namespace
namespace p2 {
bool trace_on() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("P2GPU_TRACE");
    v = (e && *e && *e != '0') ? 1 : 0;
  }
  return v == 1;
}
}  // namespace p2
namespace {
// P2GPU_HOSTPROF=1: host-side timestamps at the transcript sync points of one proof (no extra
// synchronisation), printed at the end of prove: where the host sits between GPU phases
bool hostprof_on() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("P2GPU_HOSTPROF");
    v = (e && *e && *e != '0') ? 1 : 0;
  }
  return v == 1;
}
struct HostProf {
  std::vector<std::pair<const char *, double>> ev;
  void mark(const char *label);
  void dump();
};
void HostProf::mark(const char *label) {
  if (hostprof_on()) ev.push_back({label, now_ms()});
}
void HostProf::dump() {
  if (!hostprof_on() || ev.empty()) return;
  fprintf(stderr, "[p2gpu hostprof]");
  for (size_t i = 1; i < ev.size(); i++) fprintf(stderr, " %s %+.1fus |", ev[i].first, (ev[i].second - ev[i - 1].second) * 1e3);
  fprintf(stderr, " total %.3f ms\n", ev.back().second - ev.front().second);
  ev.clear();
}
thread_local HostProf g_hp;

}  // namespace
namespace p2 {
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace p2
namespace {



}  // namespace


namespace {

struct EventProf : Prof {  // per-launch timing with HIP events on the launch stream
  p2gpu_circuit *c;
  hipEvent_t a = nullptr, b = nullptr;
  const char *name = nullptr;
  double bytes = 0;
  explicit EventProf(p2gpu_circuit *c_) : c(c_) {}
  hipEvent_t get() {
    hipEvent_t e;
    if (!c->event_pool.empty()) {
      e = c->event_pool.back();
      c->event_pool.pop_back();
    } else {
      (void)hipEventCreate(&e);
    }
    return e;
  }
  // profile = 1 brackets only the launches that move >= 32 MB (the kernels a roofline is about:
  // every event is a marker packet that costs the queue ~3 us, ~190 launches per proof);
  // profile = 2 brackets every launch
  bool active = false;
  void begin(const char *k, double by) override {
    // (all launches of the transform kernel are kept so that its average agrees with rocprofv3's)
    active = c->profile >= 2 || by >= 32e6 || strncmp(k, "ntt_", 4) == 0;
    if (!active) return;
    name = k;
    bytes = by;
    a = get();
    b = get();
    (void)hipEventRecord(a, c->stream);
  }
  void end() override {
    if (!active) return;
    (void)hipEventRecord(b, c->stream);
    c->pending.push_back({name, bytes, a, b});
  }
};
}  // namespace
namespace p2 {
void flush_kstats(p2gpu_circuit *c) {
  for (auto &pe : c->pending) {
    float ms = 0;
    (void)hipEventSynchronize(pe.b);
    (void)hipEventElapsedTime(&ms, pe.a, pe.b);
    auto &s = c->kstats[pe.name];
    s.ms += ms;
    s.launches++;
    s.bytes += pe.bytes;
    c->event_pool.push_back(pe.a);
    c->event_pool.push_back(pe.b);
  }
  c->pending.clear();
}
}  // namespace p2
namespace {

uint32_t brev(uint32_t x, unsigned bits) { return bitrev32(x, bits); }

}  // namespace
namespace p2 {
int pin_exhausted() {
  set_err("internal: pinned staging arena exhausted");
  return P2GPU_E_DEVICE;
}
}  // namespace p2
namespace {

// level offsets of a tree over [cosets][m0] leaf digests reduced to cap_per nodes per coset
}  // namespace
namespace p2 {
void tree_layout(Batch &b, uint32_t cosets, size_t m0, size_t cap_per) {
  b.level_off.clear();
  size_t off = 0;
  for (size_t m = m0;; m >>= 1) {
    b.level_off.push_back(off);
    off += (size_t)cosets * m;
    if (m <= cap_per) break;
  }
}
}  // namespace p2
namespace {

// allocate tree storage for [C][m0] leaf digests reduced to cap_per nodes per coset
}  // namespace
namespace p2 {
int tree_alloc(Batch &b, uint32_t C, size_t m0, size_t cap_per) {
  tree_layout(b, C, m0, cap_per);
  // the last level holds C * cap_per digests (or C * m0 when the leaves already are the cap)
  size_t last_m = m0;
  while (last_m > cap_per) last_m >>= 1;
  const size_t total = b.level_off.back() + (size_t)C * last_m;
  b.ncl = C;
  HIP_TRY(b.dig.alloc(total));
  return 0;
}
}  // namespace p2
namespace {

// device copy of the Poseidon round constants when the circuit's hasher is PoseidonHash, nullptr for Keccak
}  // namespace
namespace p2 {
const gl_t *hprc(const p2gpu_circuit *c) { return c->hasher == 1 ? c->d_prc_hash.p : nullptr; }
}  // namespace p2
namespace {

// The transcript sync points of a proof.  Default: hipStreamSynchronize, which spins on the host (lowest latency: the
// eleven round trips of a lone proof).  Knob "blocking_sync" = 1: record an event created with hipEventBlockingSync and
// sleep on it instead -- a woken thread costs ~10-30 us more per round trip, but a process with several proofs in
// flight no longer burns one CPU per host thread while the GPU works (4 spinning threads per GPU are 32 CPUs on an
// 8-GPU node: more than the 16-CPU cgroup quota of the MI355X boxes, where the spinning would throttle the ranks).
}  // namespace
namespace p2 {
int wait_stream(p2gpu_circuit *c) {
  if (!c->blocking_sync) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
  }
  if (!c->sync_event) HIP_TRY(hipEventCreateWithFlags(&c->sync_event, hipEventBlockingSync | hipEventDisableTiming));
  HIP_TRY(hipEventRecord(c->sync_event, c->stream));
  HIP_TRY(hipEventSynchronize(c->sync_event));
  return 0;
}
}  // namespace p2
namespace {

// levels_done: tree levels above the leaf digests that are in place already (the leaf-hash launch builds two: merkle.hip)
}  // namespace
namespace p2 {
int tree_build(p2gpu_circuit *c, Batch &b, size_t m0, uint32_t levels_done) {
  const uint32_t C = c->C, CL = b.ncl;
  size_t m = m0 >> levels_done;
  const size_t cap_target = ((size_t)1 << c->cap_h) >> c->rate_bits;
  // a tree every rank holds completely (constants/sigmas, FRI steps >= 1) needs no exchange -- except in
  // the one-rank plumbing test, where every tree goes through the transport
  const bool local = CL == C && !(c->shard_world == 1 && sharded(c));
  // the kernel that computes the cap level stores it in page-locked host memory as well when it can (merkle_tail): no
  // copy kernel between the last level and the transcript's sync
  dig_t *mirror = local && m0 > cap_target ? c->pin.take<dig_t>(C * cap_target) : nullptr;
  bool mirrored = false;
  for (size_t l = 1 + levels_done; l < b.level_off.size(); l++) {
    // the rest of the tree in merkle_tail: a few launches of several levels each (Keccak), or one (Poseidon)
    const size_t from = merkle_tail_from(hprc(c));
    if ((size_t)CL * (m >> 1) <= from) {
      mirrored = merkle_tail(c->stream, b.dig.p + b.level_off[l - 1], CL, (uint32_t)m, (uint32_t)cap_target, hprc(c), mirror);
      m = cap_target;
      break;
    }
    merkle_level(c->stream, b.dig.p + b.level_off[l - 1], b.dig.p + b.level_off[l], CL, (uint32_t)m, hprc(c));
    m >>= 1;
  }
  const size_t cap_per = m;
  dig_t *raw = mirrored ? mirror : c->pin.take<dig_t>(C * cap_per);  // [global coset][cap_per]; pinned: the D2H below is a true async copy
  if (!raw) return pin_exhausted();
  if (local) {
    if (!mirrored)
      HIP_TRY(hipMemcpyAsync(raw, b.dig.p + b.level_off.back(), C * cap_per * sizeof(dig_t), hipMemcpyDeviceToHost,
                             c->stream));
    g_hp.mark("enq(cap)");
    if (int rc_ = wait_stream(c)) return rc_;
    g_hp.mark("WAIT(cap)");
  } else {
    // coset r owns whole cap subtrees: exchange the CL * cap_per local roots (the path's only
    // commitment-time collective: 16 x 25 B in total)
    const size_t bytes = (size_t)CL * cap_per * sizeof(dig_t);
    if (int rc = shard_allgather(c, b.dig.p + b.level_off.back(), c->xchg_recv.p, bytes)) return rc;
    dig_t *all = c->pin.take<dig_t>((size_t)c->shard_world * CL * cap_per);
    if (!all) return pin_exhausted();
    HIP_TRY(hipMemcpyAsync(all, c->xchg_recv.p, (size_t)c->shard_world * CL * cap_per * sizeof(dig_t), hipMemcpyDeviceToHost, c->stream));
    if (int rc_ = wait_stream(c)) return rc_;
    shard_assemble_cap(c->shard_world, c->rate_bits, cap_per, all, b.cap);
    return 0;
  }
  shard_assemble_cap(1, c->rate_bits, cap_per, raw, b.cap);
  return 0;
}
}  // namespace p2
namespace {

// coefficients (bit-reversed storage) -> LDE on the 2^rate_bits cosets -> leaf digests -> tree
// the column classes of a batch: only the wires have them (valid for the proof in progress)
const uint32_t *batch_colnz(const p2gpu_circuit *c, const Batch &b) {
  return (&b == &c->wires && c->zero_columns && !c->structured_off && c->wire_nz.p) ? c->wire_nz.p : nullptr;
}
// First wire column whose LDE is not materialised when it is structured (class 0 / 1): no gate reads a wire >=
// gate_wires and the permutation argument stops at R, so the only readers of those LDE columns are the leaf hash
// and the query gather, which recompute val * LDE(unit column) instead (VirtCols).  UINT32_MAX: off.
uint32_t virt_first(const p2gpu_circuit *c) {
  if (!c->virtual_columns || !batch_colnz(c, c->wires)) return UINT32_MAX;
  const uint32_t f = std::max(c->R, c->gate_wires);
  return f < c->W ? f : UINT32_MAX;
}
// hints for the transforms of the wire columns [col0, ...): lde = false: values -> coefficients, true: the LDE
ColHints wire_hints(const p2gpu_circuit *c, uint32_t col0, bool lde) {
  ColHints h;
  h.cls = c->wire_nz.p + col0;
  h.clean = c->wire_clean.p + col0;
  h.val = c->wire_scalar.p + col0;
  h.basis = lde ? c->sparse_lde.p : c->sparse_coeffs.p;
  h.basis_per_coset = lde;
  h.nrows = c->sparse_rows.count;
  h.val_stride = c->W;
  h.basis_stride = lde ? (size_t)c->C * c->n : c->n;
  const uint32_t vf = virt_first(c);
  if (lde && vf != UINT32_MAX) h.virt_first = vf > col0 ? vf - col0 : 0;
  h.dense_hint = col0 == 0 ? c->last_dense : 0;
  return h;
}
// the unmaterialised columns of batch b for its leaf hash (only the wires have any)
VirtCols batch_virt(const p2gpu_circuit *c, const Batch &b) {
  VirtCols v;
  const uint32_t vf = &b == &c->wires ? virt_first(c) : UINT32_MAX;
  if (vf == UINT32_MAX) return v;
  v.cls = c->wire_nz.p;
  v.val = c->wire_scalar.p;
  v.basis = c->sparse_lde.p;
  v.first = vf;
  v.coset_first = b.cm.first;
  v.coset_stride = b.cm.stride;
  return v;
}
// leaf digests of a batch (+ the first two tree levels when the layout has them: the return value)
uint32_t leaf_hash(p2gpu_circuit *c, Batch &b, const VirtCols &v) {
  const bool two = b.level_off.size() >= 3;  // levels with n/2 and n/4 nodes per coset exist
  return hash_lde_leaves(c->stream, b.lde.p, b.cols, c->d, b.ncl, b.dig.p, hprc(c), &v, two ? b.dig.p + b.level_off[1] : nullptr,
                         two ? b.dig.p + b.level_off[2] : nullptr);
}
}  // namespace
namespace p2 {
int batch_commit_from_coeffs(p2gpu_circuit *c, Batch &b) {
  uint32_t lv = 0;
  {
    const uint32_t *nz = batch_colnz(c, b);
    const ColHints h = nz ? wire_hints(c, 0, true) : ColHints();
    ntt_batch(c->stream, c->plan_fwd, b.coeffs.p, b.lde.p, b.cols, b.ncl, c->scale.p, 1, false, b.cm, 0, nz ? &h : nullptr);
    if (nz) column_clean_update(c->stream, nz, b.cols, c->wire_clean.p, true);
  }
  {
    const VirtCols v = batch_virt(c, b);
    lv = leaf_hash(c, b, v);
  }
  TRACE(c, "  lde + leaf hash");
  return tree_build(c, b, c->n, lv);
}
}  // namespace p2
namespace {
// hash + tree of a batch whose LDE is already in place
int batch_commit_from_lde(p2gpu_circuit *c, Batch &b) {
  const VirtCols v = batch_virt(c, b);
  const uint32_t lv = leaf_hash(c, b, v);
  TRACE(c, "  leaf hash");
  return tree_build(c, b, c->n, lv);
}
}  // namespace
namespace p2 {
int batch_commit_from_values(p2gpu_circuit *c, Batch &b, const gl_t *vals_dev) {
  if (&b == &c->wires && c->wires_ntt_done) return c->wires_hash_done ? tree_build(c, b, c->n) : batch_commit_from_lde(c, b);
  {
    // unused wires are zero in every row (wires 80..233 of the 234-wire configuration in circuits without ECC
    // gates): one pass over the witness finds them, and their inverse transform and LDE become stores of zeros
    const uint32_t *nz = batch_colnz(c, b);
    if (nz) {
      column_flags(c->stream, vals_dev, b.cols, c->d, c->sparse_rows, c->wire_nz.p, c->wire_scalar.p, c->W);
      column_clean_update(c->stream, nz, b.cols, c->wire_clean.p, false);
    }
    gl_t ninv = gl_inv((gl_t)c->n);
    const ColHints h = nz ? wire_hints(c, 0, false) : ColHints();
    if (c->shard_intt && sharded(c)) {
      // SURVEY 8(e) steps 1-2 (knob "shard_intt"): rank q transforms only ITS block of the dense columns and the coefficient
      // blocks are all-gathered in place; structured columns are written locally on every rank (no exchange for them).
      // The ranks agree on the blocks without talking: the witness is replicated, so the column classes are too.
      const uint32_t G = (uint32_t)c->shard_world, q = (uint32_t)c->shard_rank, cols = b.cols;
      std::vector<uint32_t> dense;
      dense.reserve(cols);
      if (nz) {
        uint32_t *hc = c->pin.take<uint32_t>(cols);
        if (!hc) return pin_exhausted();
        HIP_TRY(hipMemcpyAsync(hc, nz, 4 * (size_t)cols, hipMemcpyDeviceToHost, c->stream));
        if (int rc_ = wait_stream(c)) return rc_;
        for (uint32_t j = 0; j < cols; j++)
          if (hc[j] == 2u) dense.push_back(j);
      } else {
        for (uint32_t j = 0; j < cols; j++) dense.push_back(j);
      }
      uint32_t lo[8], hi[8];
      size_t off[8], sz[8];
      intt_blocks(dense.data(), (uint32_t)dense.size(), G, lo, hi);
      for (uint32_t p = 0; p < G; p++) {
        off[p] = 8 * (size_t)lo[p] * c->n;
        sz[p] = 8 * (size_t)(hi[p] - lo[p]) * c->n;
      }
      auto part = [&](uint32_t c0, uint32_t c1, bool fill_only) {
        if (c1 <= c0 || (fill_only && !nz)) return;
        ColHints hp = nz ? wire_hints(c, c0, false) : ColHints();
        hp.fill_only = fill_only;
        hp.dense_hint = 0;
        ntt_batch(c->stream, c->plan_inv, vals_dev + (size_t)c0 * c->n, b.coeffs.p + (size_t)c0 * c->n, c1 - c0, 1, nullptr, ninv, false,
                  CosetMap(), 0, nz ? &hp : nullptr);
      };
      if (hi[q] > lo[q]) {
        part(0, lo[q], true);
        part(lo[q], hi[q], false);
        part(hi[q], cols, true);
      } else {
        part(0, cols, true);
      }
      TRACE(c, "  inverse ntt (own block)");
      if (int rc = shard_allgatherv(c, (uint8_t *)b.coeffs.p, off, sz)) return rc;
      TRACE(c, "  coefficient blocks exchanged");
      return batch_commit_from_coeffs(c, b);
    }
    ntt_batch(c->stream, c->plan_inv, vals_dev, b.coeffs.p, b.cols, 1, nullptr, ninv, false, CosetMap(), 0, nz ? &h : nullptr);
  }
  TRACE(c, "  inverse ntt");
  return batch_commit_from_coeffs(c, b);
}
int batch_alloc(p2gpu_circuit *c, Batch &b, uint32_t cols) {
  b.cols = cols;
  b.d = c->d;
  b.ncl = c->C;
  b.cm = CosetMap();
  HIP_TRY(b.coeffs.alloc((size_t)cols * c->n));
  HIP_TRY(b.lde.alloc((size_t)cols * c->N));
  size_t cap_per = ((size_t)1 << c->cap_h) >> c->rate_bits;
  return tree_alloc(b, c->C, c->n, cap_per);
}
}  // namespace p2
namespace {

// proof bytes, written through a cursor into storage that outlives the proof (the caller's buffer when it is large enough,
// else the handle's): a fresh 200 KB vector per proof is an mmap, its page faults and an munmap -- 20 of the serialiser's 45 us
struct Buf {
  uint8_t *base = nullptr;
  size_t len = 0, cap = 0;
  bool overflow = false;
  void put(const void *p, size_t n) {
    if (len + n > cap) { overflow = true; return; }
    memcpy(base + len, p, n);
    len += n;
  }
  void u64(uint64_t x) { put(&x, 8); }
  void ext(ext_t e) {
    u64(e.c0);
    u64(e.c1);
  }
  void dig(const dig_t &d) { put(d.w, hh_bytes()); }
};

// sibling positions (within the digest buffer of a tree) for leaf index `j`
// (plonky2 order) of a tree over [C][m0] leaf digests
void path_positions(const Batch &b, uint32_t C, unsigned lgC, size_t m0, unsigned lgm0, size_t j,
                    std::vector<size_t> &pos) {
  uint32_t r = brev((uint32_t)(j >> lgm0), lgC);
  uint32_t z = (r - b.cm.first) / b.cm.stride;  // local coset (the caller owns r)
  uint32_t k = brev((uint32_t)(j & (m0 - 1)), lgm0);
  size_t m = m0;
  for (size_t l = 0; l + 1 < b.level_off.size(); l++) {
    size_t kk = k & (m - 1);
    size_t sib = kk ^ (m >> 1);
    pos.push_back(b.level_off[l] + (size_t)z * m + sib);
    m >>= 1;
  }
  (void)C;
}


int prove_impl(p2gpu_circuit *c, const gl_t *wires_dev, const uint64_t *pis, uint32_t n_pi, uint8_t *proof_out,
               size_t *proof_len, p2gpu_timings *tm, double h2d_ms) {
  if (n_pi != c->num_pi || (n_pi && !pis)) {
    set_err("expected %u public inputs, got %u", c->num_pi, n_pi);
    return P2GPU_E_ARG;
  }
  for (uint32_t i = 0; i < n_pi; i++)
    if (pis[i] >= GL_P) {
      set_err("public input %u is not a canonical field element", i);
      return P2GPU_E_ARG;
    }
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const uint32_t d = c->d, K = c->K, R = c->R, W = c->W, NC = c->NC, QF = c->QF, PP = c->PP, C = c->C;
  const size_t n = c->n, N = c->N;
  const uint32_t ncs = NC + R, nzp = K * (1 + PP), nq = K * QF;
  const uint32_t nall = ncs + W + nzp + nq;
  EventProf prof(c);
  struct ProfGuard {
    ProfGuard(Prof *p) { g_prof = p; }
    ~ProfGuard() { g_prof = nullptr; }
  } prof_guard(c->profile ? &prof : nullptr);
  double t0 = now_ms(), t1;
  p2gpu_timings T;
  memset(&T, 0, sizeof T);
  T.h2d_ms = h2d_ms;

  gl_t pih[4];  // public_inputs_hash = InnerHasher(Poseidon).hash_no_pad(public_inputs); [] -> 0^4
  poseidon_hash_no_pad_host(pis, n_pi, pih, c->poseidon_rc);

  c->pin.reset();
  use_hasher(c);
  // which structured wire columns go without an LDE in memory is fixed here for the whole proof (structured_off may
  // flip at the openings; the query gather must see what the commitment saw)
  const uint32_t vfirst = virt_first(c);
  const uint32_t *h_cls = nullptr;  // host copies of the column classes / scalars (read back with the openings)
  const gl_t *h_val = nullptr;
  // ---- 1. wires commitment ----
  g_hp.mark("start");
  TRACE(c, "enter");
  if (int rc = batch_commit_from_values(c, c->wires, wires_dev)) return rc;
  TRACE(c, "wires commit");
  Challenger ch;
  ch.observe_digest(c->circuit_digest);
  for (int i = 0; i < 4; i++) ch.observe(pih[i]);
  ch.observe_cap(c->wires.cap);
  gl_t betas[2] = {0, 0}, gammas[2] = {0, 0}, alphas[2] = {0, 0};
  for (uint32_t k = 0; k < K; k++) betas[k] = ch.get();
  for (uint32_t k = 0; k < K; k++) gammas[k] = ch.get();
  t1 = now_ms();
  T.wires_commit_ms = t1 - t0;

  // ---- 2. partial products and Z ----
  {
    ZsArgs a;
    a.wires = wires_dev;
    a.sigmas = c->d_sigmas.p;
    a.k_is = c->d_kis.p;
    a.sub_tw = c->tw_fwd.p;
    a.tw_shift = 0;
    a.d = d; a.R = R; a.QF = QF; a.nchunks = c->nchunks; a.K = K;
    for (uint32_t k = 0; k < 2; k++) { a.betas[k] = betas[k]; a.gammas[k] = gammas[k]; }
    a.cp = c->cp.p;
    a.zp = c->zp_vals.p;
    a.sb = d;
    a.row0 = 0;
    a.rows = (uint32_t)n;
    uint32_t lgG = 0;
    while ((1u << lgG) < (uint32_t)c->shard_world) lgG++;
    if (c->shard_zs && sharded(c) && (1u << lgG) == (uint32_t)c->shard_world && d >= lgG) {
      // SURVEY 8(e) step 5 (knob "shard_zs"): the expensive half -- 160 factors, the batch inversion and the chunk quotients of every
      // row -- split by rows; rank q writes its n / G rows as one contiguous block of the scratch ([rank][column][n / G], zs_idx) and
      // the blocks are all-gathered in place.  The scan and the products along the chunks then run on every rank (two cheap passes).
      a.sb = d - lgG;
      a.rows = (uint32_t)(n >> lgG);
      a.row0 = a.rows * (uint32_t)c->shard_rank;
      zs_chunks(st, a);
      const size_t blk = (size_t)K * (c->nchunks + 1) * a.rows;
      if (int rc = shard_allgather(c, c->cp.p + blk * (size_t)c->shard_rank, c->cp.p, blk * sizeof(gl_t))) return rc;
      zs_scan_finish(st, a, c->scan_tmp.p);
    } else {
      zs_partial_products(st, a, c->scan_tmp.p);
    }
  }
  TRACE(c, "zs_partial_products");
  if (int rc = batch_commit_from_values(c, c->zp, c->zp_vals.p)) return rc;
  ch.observe_cap(c->zp.cap);
  for (uint32_t k = 0; k < K; k++) alphas[k] = ch.get();
  t0 = now_ms();
  T.zs_commit_ms = t0 - t1;

  // ---- 3. quotient ----
  {
    const uint32_t nterms = c->nterms;
    gl_t *ap = c->pin.take<gl_t>((size_t)2 * nterms);
    if (!ap) return pin_exhausted();
    memset(ap, 0, 16 * (size_t)nterms);
    for (uint32_t k = 0; k < K; k++) {
      gl_t a = 1;
      for (uint32_t t = 0; t < nterms; t++) {
        ap[(size_t)k * nterms + t] = a;
        a = gl_mul(a, alphas[k]);
      }
    }
    HIP_TRY(hipMemcpyAsync(c->apow.p, ap, 16 * (size_t)nterms, hipMemcpyHostToDevice, st));
    QuotArgs q;
    memset(&q, 0, sizeof q);
    q.cs_lde = c->cs.lde.p;
    q.wires_lde = c->wires.lde.p;
    q.zp_lde = c->zp.lde.p;
    q.k_is = c->d_kis.p;
    q.tw = c->tw_fwd.p;
    q.apow = c->apow.p;
    q.gates = c->d_gates.p;
    q.host_gates = c->gates.data();
    q.out = c->qvals.p;
    q.tw_shift = 0; q.d = d; q.rate_bits = c->rate_bits; q.W = W; q.R = R; q.NC = NC;
    q.num_selectors = c->num_selectors; q.K = K; q.QF = QF; q.nchunks = c->nchunks; q.PP = PP;
    q.num_gates = c->num_gates; q.nterms = nterms;
    q.coset_first = c->wires.cm.first;
    q.coset_stride = c->wires.cm.stride;
    q.ncosets = c->wires.ncl;
    q.gate_groups = c->gate_groups;
    q.has_poseidon = 0;
    for (auto &g : c->gates)
      if (g.kind == G_POSEIDON) q.has_poseidon = 1;
    for (uint32_t k = 0; k < 2; k++) { q.betas[k] = betas[k]; q.gammas[k] = gammas[k]; }
    for (int i = 0; i < 4; i++) q.pi_hash[i] = pih[i];
    gl_t wN = gl_root(d + c->rate_bits), wC = gl_root(c->rate_bits), gn = gl_pow(GL_GEN, n);
    q.qconst = c->qconst.p;
    (void)wN;
    q.n_inv = gl_inv((gl_t)n);
    q.l0 = c->l0_lde.p;
    // gates of degree <= 4: folded sums on the even cosets, extended to the odd ones (only with every coset on this device)
    q.gate_groups_half = c->gate_groups_half;
    q.nsk = c->half_slots * K;
    q.hsum = c->hsum.p;
    q.use_half = (c->half_gates == 2 || (c->half_gates == 1 && c->half_auto)) && c->half_slots && c->wires.ncl == C && c->wires.cm.stride == 1 &&
                 c->wires.cm.first == 0;
    if (q.use_half) {
      const size_t per = (size_t)4 * q.nsk * n;
      gate_sums_eval(st, q, n >= 64 ? c->sums_groups : 1u);
      ntt_batch(st, c->plan_inv, c->hsum.p, c->htmp_a.p, 4 * q.nsk, 1, nullptr, q.n_inv, false);
      gate_sums_cross(st, c->htmp_a.p, c->inv_scale.p, c->htmp_b.p, d, q.nsk, c->half_cross);
      CosetMap odd;
      odd.first = 1;
      odd.stride = 2;
      ntt_batch(st, c->plan_fwd, c->htmp_b.p, c->hsum.p + per, q.nsk, 4, c->scale.p, 1, true, odd);
      TRACE(c, "gate sums (half domain)");
    }
    {
      quotient_eval(st, q);
    }
    TRACE(c, "quotient_eval");
    // coset_ifft of size N = per-coset inverse transforms + cross-coset butterflies
    {
      ntt_batch(st, c->plan_inv, c->qvals.p, c->qtmp.p, K * c->wires.ncl, 1, nullptr, q.n_inv, false);
    }
    {
      const gl_t *pr = c->qtmp.p;
      if (sharded(c)) {
        // every rank needs all cosets' interpolants for the cross-coset butterflies: all-gather
        // [K][C/world][n] per rank (2 * N * 8 B in total) straight between device buffers
        if (int rc = shard_allgather(c, c->qtmp.p, c->qvals.p, (size_t)K * c->wires.ncl * n * sizeof(gl_t))) return rc;
        pr = c->qvals.p;
      }
      quotient_chunks(st, pr, c->inv_scale.p, c->quot.coeffs.p, d, K, c->rate_bits, gl_inv(wC), gl_inv(gn),
                      gl_inv((gl_t)C), (uint32_t)c->shard_world);
    }
    TRACE(c, "quotient_chunks");
  }
  if (int rc = batch_commit_from_coeffs(c, c->quot)) return rc;
  ch.observe_cap(c->quot.cap);
  ext_t zeta = ch.get_ext();
  t1 = now_ms();
  T.quotient_ms = t1 - t0;
  {
    ext_t zn = zeta;
    for (uint32_t i = 0; i < d; i++) zn = ext_mul(zn, zn);
    if (ext_eq(zn, ext_from(1))) {
      set_err("Opening point is in the subgroup.");
      return P2GPU_E_OPENING_IN_SUBGROUP;
    }
  }

  // ---- 4. openings ----
  const ext_t gzeta = ext_scale(zeta, gl_root(d));
  Batch *oracles[4] = {&c->cs, &c->wires, &c->zp, &c->quot};
  std::vector<ext_t> op(nall + K);
  {
    uint32_t parts = 1;
    while (parts < 16 && (n / (parts * 2)) >= 1024) parts *= 2;
    ext_powers_bitrev2(st, zeta, gzeta, d, c->pw.p, c->pw.p + 2 * n);
    const bool structured = batch_colnz(c, c->wires) != nullptr;
    const ColHints wh = structured ? wire_hints(c, 0, false) : ColHints();
    if (structured) {
      compact_nonzero(st, c->wire_nz.p, c->W, c->wire_nzlist.p);
      if (c->sparse_coeffs.p) eval_columns(st, c->sparse_coeffs.p, 1, d, c->pw.p, parts, c->sparse_partial.p);
    }
    size_t base = 0;
    EvalSegs es;  // the four batches at zeta and Z at g * zeta: one launch
    for (int o = 0; o < 4; o++) {
      if (structured && oracles[o] == &c->wires) {
        es.hinted = es.count;
        es.cls = wh.cls;
        es.val = wh.val;
        es.basis_partial = c->sparse_partial.p;
      }
      es.seg[es.count++] = {oracles[o]->coeffs.p, c->pw.p, c->partial.p + base * parts * 2, oracles[o]->cols};
      base += oracles[o]->cols;
    }
    es.seg[es.count++] = {c->zp.coeffs.p, c->pw.p + 2 * n, c->partial.p + base * parts * 2, K};
    if (sharded(c) && c->shard_world > 1) {
      // SURVEY 8(e) step 8, the openings: every rank holds every coefficient (the inverse transforms are replicated), so rank q
      // evaluates the q-th block of the concatenated columns only and the partial sums (16 x 16 B per column) are all-gathered
      // in place -- 70 KB instead of 7/8 of a 0.09 ms kernel on every rank
      const uint32_t total = (uint32_t)(nall + K), G = (uint32_t)c->shard_world, cpr = (total + G - 1) / G;
      eval_columns_multi(st, es, d, parts, cpr * (uint32_t)c->shard_rank, cpr);
      const size_t blk = (size_t)cpr * parts * 2;
      if (int rc = shard_allgather(c, c->partial.p + blk * (size_t)c->shard_rank, c->partial.p, blk * 8)) return rc;
    } else {
      eval_columns_multi(st, es, d, parts);
    }
    const size_t npart = (size_t)(nall + K) * parts * 2;
    gl_t *part = c->pin.take<gl_t>(npart);
    if (!part) return pin_exhausted();
    HIP_TRY(hipMemcpyAsync(part, c->partial.p, npart * 8, hipMemcpyDeviceToHost, st));
    uint32_t *dense_count = structured ? c->pin.take<uint32_t>(1) : nullptr;
    if (dense_count) HIP_TRY(hipMemcpyAsync(dense_count, c->wire_nzlist.p, 4, hipMemcpyDeviceToHost, st));
    if (vfirst != UINT32_MAX) {
      uint32_t *hc = c->pin.take<uint32_t>(W);
      gl_t *hv = c->pin.take<gl_t>(W);
      if (!hc || !hv) return pin_exhausted();
      HIP_TRY(hipMemcpyAsync(hc, c->wire_nz.p, 4 * (size_t)W, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(hv, c->wire_scalar.p, 8 * (size_t)W, hipMemcpyDeviceToHost, st));
      h_cls = hc;
      h_val = hv;
    }
    g_hp.mark("enq(openings)");
    if (int rc_ = wait_stream(c)) return rc_;
    g_hp.mark("WAIT(openings)");
    // Which wires a circuit leaves unused does not change from proof to proof.  A handle whose witness turned
    // out (almost) fully dense stops looking: the class pass over the witness and the fill launches cost ~0.1 ms
    // at 2^20 rows, what fewer than 5 % structured columns give back; the knob "zero_columns" = 1 makes it look
    // again.  (Every column is then transformed like a dense one, which is always correct.)
    if (dense_count && (uint64_t)(c->W - *dense_count) * 20u < c->W) c->structured_off = true;
    if (dense_count) c->last_dense = *dense_count;
    for (size_t j = 0; j < nall + K; j++) {
      gl_t a0 = 0, a1 = 0;
      for (uint32_t p = 0; p < parts; p++) {
        a0 = gl_add(a0, part[(j * parts + p) * 2]);
        a1 = gl_add(a1, part[(j * parts + p) * 2 + 1]);
      }
      op[j] = ext_make(a0, a1);
    }
  }
  TRACE(c, "openings");
  for (size_t j = 0; j < nall + K; j++) ch.observe_ext(op[j]);
  g_hp.mark("observe(openings)");
  t0 = now_ms();
  T.openings_ms = t0 - t1;

  // ---- 5. FRI ----
  const ext_t alpha = ch.get_ext();
  {
    gl_t *apw = c->pin.take<gl_t>((size_t)2 * nall);
    if (!apw) return pin_exhausted();
    ext_t a = ext_from(1);
    ext_t f0z = ext_from(0), f1z = ext_from(0);
    for (uint32_t j = 0; j < nall; j++) {
      apw[2 * j] = a.c0;
      apw[2 * j + 1] = a.c1;
      f0z = ext_add(f0z, ext_mul(a, op[j]));
      if (j < K) f1z = ext_add(f1z, ext_mul(a, op[nall + j]));
      a = ext_mul(a, alpha);
    }
    HIP_TRY(hipMemcpyAsync(c->ext_apow.p, apw, 16 * (size_t)nall, hipMemcpyHostToDevice, st));
    gl_t *F0 = c->f01.p, *F1 = c->f01.p + 2 * n;
    uint32_t j0 = 0;
    bool reduced = false;
    if (c->shard_reduce && sharded(c)) {
      // SURVEY 8(e) step 8, the FRI batch reduction (knob "shard_reduce"): every rank holds every coefficient, so rank q sums only
      // its block of the 354 concatenated columns (plain loop: structured columns' coefficients are in memory like anybody's) and
      // the partial sums F0_q [2][n] are all-gathered and added -- field addition is exact, the sum does not depend on the split
      const uint32_t G = (uint32_t)c->shard_world, cpr = (nall + G - 1) / G;
      const uint32_t c0 = std::min(cpr * (uint32_t)c->shard_rank, nall), c1 = std::min(c0 + cpr, nall);
      bool first = true;
      uint32_t jo = 0;
      for (int o = 0; o < 4; o++) {
        const uint32_t co = oracles[o]->cols;
        const uint32_t lo = std::max(c0, jo) - jo, hi = std::min(c1, jo + co) > jo ? std::min(c1, jo + co) - jo : 0;
        if (hi > lo) {
          reduce_columns(st, oracles[o]->coeffs.p + (size_t)lo * n, hi - lo, d, c->ext_apow.p, jo + lo, F0, !first);
          first = false;
        }
        jo += co;
      }
      if (first) HIP_TRY(hipMemsetAsync(F0, 0, 16 * n, st));
      if (int rc = shard_allgather(c, F0, c->xchg_recv.p, 16 * n)) return rc;
      sum_parts(st, c->xchg_recv.p, G, 2 * n, F0);
      reduced = true;
    }
    for (int o = reduced ? 4 : 0; o < 4; o++) {
      const bool hw = batch_colnz(c, *oracles[o]) != nullptr;
      const bool unit = hw && c->sparse_coeffs.p != nullptr;
      if (unit) class1_fold(st, wire_hints(c, 0, false), c->W, c->ext_apow.p, j0, c->sparse_partial.p + 32);
      reduce_columns(st, oracles[o]->coeffs.p, oracles[o]->cols, d, c->ext_apow.p, j0, F0, o != 0,
                     hw ? c->wire_nzlist.p : nullptr, unit ? c->sparse_coeffs.p : nullptr, unit ? c->sparse_partial.p + 32 : nullptr);
      j0 += oracles[o]->cols;
    }
    {
      reduce_columns(st, c->zp.coeffs.p, K, d, c->ext_apow.p, 0, F1, false);
    }
    {
      ntt_batch(st, c->plan_fwd, c->f01.p, c->f01v.p, 4, 1, nullptr, 1, false);
    }
    fri_quotient_values(st, c->f01v.p, c->f01v.p + 2 * n, d, c->tw_fwd.p, 0, zeta, gzeta, f0z, f1z, ext_pow(alpha, K),
                        c->fv.p);
    {
      ntt_batch(st, c->plan_inv, c->fv.p, c->fri_coef[0].p, 2, 1, nullptr, gl_inv((gl_t)n), false);
    }
    if (c->n_steps > 0) {  // no reduction step (degree <= 2^5): the values are never committed
      ntt_batch(st, c->plan_fwd, c->fri_coef[0].p, c->fri_vals[0].p, 2, c->fri_trees[0].ncl, c->scale.p, 1, false,
                c->fri_trees[0].cm);
    }
  }
  TRACE(c, "fri final poly lde");
  // the self-check needs only the openings and the challenges: it runs on the host while the GPU is
  // busy with the batch reduction and the FRI LDE enqueued above, off the critical path
  if (c->self_check && !plonk_identity_holds(c, op, betas, gammas, alphas, zeta, pih)) {
    (void)hipStreamSynchronize(st);
    set_err("witness does not satisfy the circuit: the plonk identity fails at zeta (vanishing != Z_H * quotient)");
    return P2GPU_E_UNSATISFIED;
  }
  g_hp.mark("selfcheck");
  std::vector<ext_t> fri_betas;
  uint32_t ds = d;
  gl_t shift = GL_GEN;
  for (uint32_t s = 0; s < c->n_steps; s++) {
    const uint32_t ab = c->arity[s];
    Batch &tr = c->fri_trees[s];
    {
      hash_fri_leaves(st, c->fri_vals[s].p, ds, tr.ncl, ab, tr.dig.p, hprc(c));
    }
    if (int rc = tree_build(c, tr, ((size_t)1 << ds) >> ab)) return rc;
    ch.observe_cap(tr.cap);
    ext_t beta = ch.get_ext();
    fri_betas.push_back(beta);
    fri_fold(st, c->fri_coef[s].p, ds, ab, beta, c->fri_coef[s + 1].p);
    for (uint32_t q = 0; q < ab; q++) shift = gl_sqr(shift);
    ds -= ab;
    if (s + 1 < c->n_steps) {
      fill_coset_scale(st, c->fri_scale.p, shift, gl_root(ds + c->rate_bits), ds, C, 1);
      ntt_batch(st, c->fri_plans[s + 1], c->fri_coef[s + 1].p, c->fri_vals[s + 1].p, 2, C, c->fri_scale.p, 1, false);
    }
  }
  const size_t n_final = (size_t)1 << ds;
  std::vector<ext_t> final_poly(n_final);
  {
    gl_t *raw = c->pin.take<gl_t>(2 * n_final);
    if (!raw) return pin_exhausted();
    HIP_TRY(hipMemcpyAsync(raw, c->fri_coef[c->n_steps].p, 16 * n_final, hipMemcpyDeviceToHost, st));
    g_hp.mark("enq(final_poly)");
    if (int rc_ = wait_stream(c)) return rc_;
    g_hp.mark("WAIT(final_poly)");
    for (size_t j = 0; j < n_final; j++) {
      size_t p = brev((uint32_t)j, ds);
      final_poly[j] = ext_make(raw[p], raw[n_final + p]);
    }
  }
  for (auto &e : final_poly) ch.observe_ext(e);
  TRACE(c, "fri commit phase");

  // PoW: minimum-witness policy (upstream's parallel find_any is not deterministic)
  uint64_t pow_witness = c->pow_hint;
  if (pow_witness == UINT64_MAX) {
    gl_t inter[12];
    memcpy(inter, ch.state, sizeof inter);
    for (int i = 0; i < ch.n_in; i++) inter[i] = ch.in[i];
    // expected minimum witness ~2^pow_bits: start with 2^(pow_bits+1) candidates, then double
    uint64_t batch = 1ull << (c->pow_bits + 1 < 20 ? c->pow_bits + 1 : 20);
    // staging words taken once: the loop may run for thousands of batches at high pow_bits
    unsigned long long *pw = c->pin.take<unsigned long long>(2);
    unsigned long long *allw = sharded(c) ? c->pin.take<unsigned long long>((size_t)c->shard_world) : nullptr;
    if (!pw || (sharded(c) && !allw)) return pin_exhausted();
    for (uint64_t base = 0;; base += batch, batch = batch < (1ull << 22) ? batch * 2 : batch) {
      pw[0] = ~0ull;
      HIP_TRY(hipMemcpyAsync(c->pow_result.p, &pw[0], 8, hipMemcpyHostToDevice, st));
      // sharded: the ranks grind disjoint slices of [base, base + batch) and take the minimum of what
      // they found; every rank ends with the same, globally minimal witness
      const bool split = sharded(c);
      const uint64_t W_ = split ? (uint64_t)c->shard_world : 1, per = (batch + W_ - 1) / W_;
      const uint64_t my0 = base + per * (uint64_t)(split ? c->shard_rank : 0);
      const uint64_t myn = my0 >= base + batch ? 0 : std::min(per, base + batch - my0);
      if (myn) pow_search(st, inter, (uint32_t)ch.n_in, c->pow_bits, my0, myn, c->pow_result.p, hprc(c));
      if (split) {
        if (int rc = shard_allgather(c, c->pow_result.p, c->xchg_recv.p, 8)) return rc;
        HIP_TRY(hipMemcpyAsync(allw, c->xchg_recv.p, 8 * (size_t)c->shard_world, hipMemcpyDeviceToHost, st));
        g_hp.mark("enq(pow)");
        if (int rc_ = wait_stream(c)) return rc_;
        g_hp.mark("WAIT(pow)");
        pw[1] = ~0ull;
        for (int q = 0; q < c->shard_world; q++) pw[1] = std::min(pw[1], allw[q]);
      } else {
        HIP_TRY(hipMemcpyAsync(&pw[1], c->pow_result.p, 8, hipMemcpyDeviceToHost, st));
        g_hp.mark("enq(pow)");
        if (int rc_ = wait_stream(c)) return rc_;
        g_hp.mark("WAIT(pow)");
      }
      const unsigned long long res = pw[1];
      if (res != ~0ull) {
        pow_witness = res;
        break;
      }
      if (base > (1ull << 40)) {
        set_err("proof of work failed");
        return P2GPU_E_DEVICE;
      }
    }
  }
  ch.observe(pow_witness);
  gl_t pow_resp = ch.get();
  if (c->pow_bits && (pow_resp >> (64 - c->pow_bits)) != 0) {
    set_err("proof-of-work witness does not satisfy the leading-zero check");
    return P2GPU_E_ARG;
  }
  T.pow_witness = pow_witness;
  TRACE(c, "pow");
  std::vector<size_t> qidx(c->num_queries);
  for (auto &x : qidx) x = (size_t)(ch.get() % N);

  // ---- query gather: one launch, one D2H ----
  const unsigned lgC = c->rate_bits;
  std::vector<uint64_t> &ptrs = c->h_ptrs;  // (the handle's: its capacity survives the proof -- one prove per handle at a time)
  ptrs.clear();
  ptrs.reserve(c->gather_cap);
  auto push_dig = [&](const Batch &b, size_t pos) {
    const uint64_t base = (uint64_t)(uintptr_t)(b.dig.p + pos);
    for (int w = 0; w < 4; w++) ptrs.push_back(base + 8 * w);
  };
  std::vector<size_t> pos;
  const int world = c->shard_world, me = c->shard_rank;
  size_t per_query = 0;
  std::vector<std::pair<size_t, uint32_t>> virt_fix;  // (slot of the gather, wire column): slot holds LDE(unit column), wants val * it
  for (size_t x : qidx) {
    // every piece of query x lives in coset r = bitrev(top bits of x): one rank owns the query
    const uint32_t rq = brev((uint32_t)(x >> d), lgC);
    const bool mine = (int)(rq % (uint32_t)world) == me;
    const size_t start = ptrs.size();
    for (int o = 0; o < 4; o++) {
      const Batch &b = *oracles[o];
      const uint32_t r = rq, k = brev((uint32_t)(x & (n - 1)), d);
      const uint32_t z = (r - b.cm.first) / b.cm.stride;
      for (uint32_t col = 0; col < b.cols; col++) {
        if (o == 1 && col >= vfirst && h_cls[col] < 2u) {
          // unmaterialised column: class 0 opens to 0, class 1 to val * LDE(unit column)[r][k] (product taken on the host)
          const bool c1 = h_cls[col] == 1u && c->sparse_lde.p;
          if (c1) virt_fix.emplace_back(ptrs.size(), col);
          ptrs.push_back(c1 && mine ? (uint64_t)(uintptr_t)(c->sparse_lde.p + (size_t)r * n + k) : 0);
          continue;
        }
        ptrs.push_back(mine ? (uint64_t)(uintptr_t)(b.lde.p + ((size_t)z * b.cols + col) * n + k) : 0);
      }
      pos.clear();
      if (mine) path_positions(b, C, lgC, n, d, x, pos);
      else pos.assign(b.level_off.size() - 1, 0);
      for (size_t p : pos) {
        if (mine) push_dig(b, p);
        else for (int w = 0; w < 4; w++) ptrs.push_back(0);
      }
    }
    size_t xi = x;
    uint32_t dcur = d;
    for (uint32_t s = 0; s < c->n_steps; s++) {
      const uint32_t ab = c->arity[s];
      const Batch &tr = c->fri_trees[s];
      const size_t npc = (size_t)1 << dcur, per = npc >> ab;  // per-coset leaves
      const size_t li = xi >> ab;                              // leaf index (plonky2 order) in tree s
      const unsigned lgper = dcur - ab;
      const uint32_t r = brev((uint32_t)(li >> lgper), lgC), kl = brev((uint32_t)(li & (per - 1)), lgper);
      const uint32_t z = (r - tr.cm.first) / tr.cm.stride;
      const gl_t *v0 = c->fri_vals[s].p + (size_t)z * 2 * npc;
      for (uint32_t t = 0; t < (1u << ab); t++) {
        size_t k = (size_t)brev(t, ab) * per + kl;
        ptrs.push_back(mine ? (uint64_t)(uintptr_t)(v0 + k) : 0);
        ptrs.push_back(mine ? (uint64_t)(uintptr_t)(v0 + npc + k) : 0);
      }
      pos.clear();
      if (mine) path_positions(tr, C, lgC, per, lgper, li, pos);
      else pos.assign(tr.level_off.size() - 1, 0);
      for (size_t p : pos) {
        if (mine) push_dig(tr, p);
        else for (int w = 0; w < 4; w++) ptrs.push_back(0);
      }
      xi = li;
      dcur -= ab;
    }
    per_query = ptrs.size() - start;
  }
  g_hp.mark("ptrs");
  if (ptrs.size() > c->gather_cap) {
    set_err("internal: gather buffer too small");
    return P2GPU_E_DEVICE;
  }
  gl_t *gathered = c->pin.take<gl_t>(ptrs.size());
  uint64_t *pptrs = c->pin.take<uint64_t>(ptrs.size());
  if (!gathered || !pptrs) return pin_exhausted();
  memcpy(pptrs, ptrs.data(), ptrs.size() * 8);
  HIP_TRY(hipMemcpyAsync(c->gather_ptrs.p, pptrs, ptrs.size() * 8, hipMemcpyHostToDevice, st));
  {
    gather_u64(st, c->gather_ptrs.p, (uint32_t)ptrs.size(), c->gather_out.p);
  }
  g_hp.mark("launch(gather)");
  if (!sharded(c)) {
    HIP_TRY(hipMemcpyAsync(gathered, c->gather_out.p, ptrs.size() * 8, hipMemcpyDeviceToHost, st));
    g_hp.mark("enq(gather)");
    if (int rc_ = wait_stream(c)) return rc_;
    g_hp.mark("WAIT(gather)");
  } else {
    // each rank gathered the queries that fall into its cosets: exchange and pick every query
    // from its owner
    g_hp.mark("enq(gather)");
    if (int rc = shard_allgather(c, c->gather_out.p, c->xchg_recv.p, ptrs.size() * 8)) return rc;
    g_hp.mark("xchg(gather)");
    std::vector<gl_t> all((size_t)world * ptrs.size());
    HIP_TRY(hipMemcpyAsync(all.data(), c->xchg_recv.p, all.size() * 8, hipMemcpyDeviceToHost, st));
    if (int rc_ = wait_stream(c)) return rc_;
    g_hp.mark("WAIT(gather)");
    for (size_t qi = 0; qi < qidx.size(); qi++) {
      const uint32_t owner = brev((uint32_t)(qidx[qi] >> d), lgC) % (uint32_t)world;
      memcpy(&gathered[qi * per_query], &all[(size_t)owner * ptrs.size() + qi * per_query], per_query * 8);
    }
  }

  for (auto &f : virt_fix) gathered[f.first] = gl_mul(h_val[f.second], gathered[f.first]);

  // ---- serialise: plonky2 ProofWithPublicInputs::to_bytes ----
  Buf out;
  {
    const size_t bound = p2gpu_proof_size_bound(c);
    if (*proof_len >= bound) {
      out.base = proof_out;
    } else {
      if (c->h_out.size() < bound) c->h_out.resize(bound);
      out.base = c->h_out.data();
    }
    out.cap = bound;
  }
  for (auto &dg : c->wires.cap) out.dig(dg);
  for (auto &dg : c->zp.cap) out.dig(dg);
  for (auto &dg : c->quot.cap) out.dig(dg);
  // OpeningSet: constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys
  for (size_t j = 0; j < ncs + W + K; j++) out.ext(op[j]);
  for (size_t k = 0; k < K; k++) out.ext(op[nall + k]);
  for (size_t j = ncs + W + K; j < nall; j++) out.ext(op[j]);
  for (uint32_t s = 0; s < c->n_steps; s++)
    for (auto &dg : c->fri_trees[s].cap) out.dig(dg);
  {
    size_t g = 0;
    auto put_path = [&](size_t nsib) {
      uint8_t l = (uint8_t)nsib;
      out.put(&l, 1);
      for (size_t i = 0; i < nsib; i++) {
        dig_t dg;
        for (int w = 0; w < 4; w++) dg.w[w] = gathered[g++];
        out.dig(dg);
      }
    };
    for (size_t qi = 0; qi < qidx.size(); qi++) {
      for (int o = 0; o < 4; o++) {
        const Batch &b = *oracles[o];
        out.put(&gathered[g], 8 * (size_t)b.cols);
        g += b.cols;
        put_path(b.level_off.size() - 1);
      }
      for (uint32_t s = 0; s < c->n_steps; s++) {
        size_t words = 2u << c->arity[s];
        out.put(&gathered[g], 8 * words);
        g += words;
        put_path(c->fri_trees[s].level_off.size() - 1);
      }
    }
  }
  for (auto &e : final_poly) out.ext(e);
  out.u64(pow_witness);
  for (uint32_t i = 0; i < n_pi; i++) out.u64(pis[i]);
  t1 = now_ms();
  T.fri_ms = t1 - t0;
  T.total_ms = T.wires_commit_ms + T.zs_commit_ms + T.quotient_ms + T.openings_ms + T.fri_ms;
  // event pairs are read back lazily (p2gpu_kernel_stats), so profiling adds no synchronisation to
  // the proof itself; bound the backlog
  if (c->profile && c->pending.size() > 16384) flush_kstats(c);
  if (tm) *tm = T;
  if (out.overflow) {
    set_err("internal: proof longer than p2gpu_proof_size_bound");
    return P2GPU_E_DEVICE;
  }
  if (out.len > *proof_len) {
    *proof_len = out.len;
    set_err("proof buffer too small: need %zu bytes", out.len);
    return P2GPU_E_BUFFER;
  }
  if (out.base != proof_out) memcpy(proof_out, out.base, out.len);
  *proof_len = out.len;
  g_hp.mark("serialise");
  g_hp.dump();
  return P2GPU_OK;
}

}  // namespace

extern "C" {

int p2gpu_fill_witness(p2gpu_circuit *c, uint64_t *wires_dev) try {
  if (!c || !wires_dev) return P2GPU_E_ARG;
  if (c->device < 0) { set_err("this is a verifier-only handle (p2gpu_verifier_create): no prover state"); return P2GPU_E_ARG; }
  HIP_TRY(hipSetDevice(c->device));
  fill_witness(c->stream, wires_dev, c->d_row_gate.p, c->d_gates.p, c->d_gconsts.p, c->d_prc.p, c->d, c->NC - c->num_selectors,
               c->W);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return P2GPU_OK;
} P2GPU_CATCH

// A device group proves by running the same entry point on every rank, one host thread each; every rank returns the
// same bytes, rank 0's go to the caller.
extern "C++" {
template <class F>
static int group_prove(p2gpu_circuit *c, uint8_t *proof_out, size_t *proof_len, p2gpu_timings *tm, F f) {
  const size_t cap = *proof_len;
  std::vector<std::vector<uint8_t>> local(c->peer ? 0 : c->group.size() + 1);
  std::vector<std::vector<uint8_t>> &scratch = c->peer ? c->peer->proof_scratch : local;  // one prove per handle at a time
  std::vector<size_t> lens(c->group.size() + 1, cap);
  return group_run(c, [&](p2gpu_circuit *m, int q) {
    if (q == 0) return f(m, q, proof_out, proof_len, tm);
    if (scratch[q].size() < cap) scratch[q].resize(cap);
    return f(m, q, scratch[q].data(), &lens[q], (p2gpu_timings *)nullptr);
  });
}
}  // extern "C++"
static int prove_routed_one(p2gpu_circuit *c, const uint64_t *routed, const uint64_t *pis, uint32_t n_pi, uint8_t *proof_out,
                            size_t *proof_len, p2gpu_timings *tm);
int p2gpu_prove_routed(p2gpu_circuit *c, const uint64_t *routed, const uint64_t *pis, uint32_t n_pi, uint8_t *proof_out,
                       size_t *proof_len, p2gpu_timings *tm) try {
  if (!c || !routed || !proof_out || !proof_len) return P2GPU_E_ARG;
  if (c->device < 0) { set_err("this is a verifier-only handle (p2gpu_verifier_create): no prover state"); return P2GPU_E_ARG; }
  if (!c->group.empty())
    return group_prove(c, proof_out, proof_len, tm, [&](p2gpu_circuit *m, int, uint8_t *po, size_t *pl, p2gpu_timings *t) {
      return prove_routed_one(m, routed, pis, n_pi, po, pl, t);
    });
  return prove_routed_one(c, routed, pis, n_pi, proof_out, proof_len, tm);
} P2GPU_CATCH
static int prove_routed_one(p2gpu_circuit *c, const uint64_t *routed, const uint64_t *pis, uint32_t n_pi, uint8_t *proof_out,
                            size_t *proof_len, p2gpu_timings *tm) try {
  HIP_TRY(hipSetDevice(c->device));
  double t0 = now_ms();
  // only the routed columns cross PCIe; every other column is gate-internal and derived on the GPU
  HIP_TRY(hipMemcpyAsync(c->wires_vals.p, routed, 8 * (size_t)c->R * c->n, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(c->wires_vals.p + (size_t)c->R * c->n, 0, 8 * (size_t)(c->W - c->R) * c->n, c->stream));
  fill_witness(c->stream, c->wires_vals.p, c->d_row_gate.p, c->d_gates.p, c->d_gconsts.p, c->d_prc.p, c->d,
               c->NC - c->num_selectors, c->W);
  HIP_TRY(hipStreamSynchronize(c->stream));
  double h2d = now_ms() - t0;
  return prove_impl(c, c->wires_vals.p, pis, n_pi, proof_out, proof_len, tm, h2d);
} P2GPU_CATCH

int p2gpu_prove_dev(p2gpu_circuit *c, const uint64_t *wires_dev, const uint64_t *pis, uint32_t n_pi, uint8_t *proof_out,
                    size_t *proof_len, p2gpu_timings *tm) try {
  if (!c || !wires_dev || !proof_out || !proof_len) return P2GPU_E_ARG;
  if (c->device < 0) { set_err("this is a verifier-only handle (p2gpu_verifier_create): no prover state"); return P2GPU_E_ARG; }
  if (!c->group.empty())
    // the witness is resident on rank 0's device: the other ranks pull it over the peer link into their own staging
    // buffer (the host-witness entry points shard the upload instead: each rank fetches 1/world of it over its own PCIe link)
    return group_prove(c, proof_out, proof_len, tm, [&](p2gpu_circuit *m, int, uint8_t *po, size_t *pl, p2gpu_timings *t) {
      const uint64_t *w = wires_dev;
      if (m->device != c->device) {
        HIP_TRY(hipSetDevice(m->device));
        HIP_TRY(hipMemcpyPeerAsync(m->wires_vals.p, m->device, wires_dev, c->device, 8 * (size_t)m->W * m->n, m->stream));
        w = m->wires_vals.p;
      }
      return prove_impl(m, w, pis, n_pi, po, pl, t, 0.0);
    });
  return prove_impl(c, wires_dev, pis, n_pi, proof_out, proof_len, tm, 0.0);
} P2GPU_CATCH

// Host-side look at the witness of the full-matrix entry point (p2gpu_prove): the longest SUFFIX of columns that are
// zero outside `row` -- the wires no gate of the circuit uses, which plonky2's build() leaves at zero except for one random
// value in the PublicInputGate row.  They need not cross PCIe (154 of 234 columns, 161 of 245 MB, for a circuit without
// ECC gates): what p2gpu_prove_sparse lets a caller say, found here by looking.  A few host threads read the columns from
// the last one down and stop at the first column that is dense (a dense witness costs a few cache lines); the scan runs
// while the first chunks (columns below the routed-wire count) are already crossing PCIe.
struct HostScan {
  uint32_t ncols = 0;            // columns [ncols, W) are zero outside `row`
  std::vector<uint64_t> tail;    // their values in `row`
};
static void host_scan_suffix(const uint64_t *wires, uint32_t W, size_t n, uint32_t row, uint32_t lo, HostScan *out) {
  // several proofs may be in flight, each with its own scan: a quarter of the CPUs this process may use (the cgroup
  // quota where there is one: the MI355X boxes show 256 hardware threads and grant 16 CPUs), at most 8.  (Round 6 tried
  // twice as many for a scan that finds no other one running: the 154 MB it reads at 2^17 rows take 1.8 ms either way --
  // memory-bound -- and the extra threads cost the pageable upload's staging copy 0.5 ms; gpurun_out/r06_host.)
  static const unsigned T = [] {
    unsigned n = std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      long long q = 0, per = 0;
      if (fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, q / per));
      fclose(f);
    }
    n /= 4;
    return n < 1 ? 1u : (n > 8 ? 8u : n);
  }();
  std::atomic<uint32_t> dense_max{lo};  // columns below this one are not worth looking at any more
  std::vector<uint8_t> sparse(W, 0);
  auto work = [&](unsigned t) {
    for (int64_t j = (int64_t)W - 1 - t; j >= (int64_t)lo; j -= T) {
      if ((uint32_t)j < dense_max.load(std::memory_order_relaxed)) break;
      const uint64_t *p = wires + (size_t)j * n;
      bool zero = true;
      for (size_t b = 0; b < n && zero; b += 2048) {  // 16 KB at a time: early exit on a dense column
        const size_t e = std::min(n, b + 2048);
        uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        size_t i = b;
        for (; i + 4 <= e; i += 4) { a0 |= p[i]; a1 |= p[i + 1]; a2 |= p[i + 2]; a3 |= p[i + 3]; }
        for (; i < e; i++) a0 |= p[i];
        uint64_t acc = a0 | a1 | a2 | a3;
        if (acc && row >= b && row < e) {  // the block holding the special row: look again without it
          acc = 0;
          for (size_t k = b; k < e; k++) acc |= (k == row) ? 0 : p[k];
        }
        zero = acc == 0;
      }
      if (zero) sparse[j] = 1;
      else {
        uint32_t cur = dense_max.load(std::memory_order_relaxed);
        while ((uint32_t)j + 1 > cur && !dense_max.compare_exchange_weak(cur, (uint32_t)j + 1, std::memory_order_relaxed)) {}
      }
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < T; t++) th.emplace_back(work, t);
  work(0);
  for (auto &x : th) x.join();
  uint32_t nc = W;
  while (nc > lo && sparse[nc - 1]) nc--;
  out->ncols = nc;
  out->tail.resize(W - nc);
  for (uint32_t j = nc; j < W; j++) out->tail[j - nc] = row < n ? wires[(size_t)j * n + row] : 0;
}
static bool host_prescan_on() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("P2GPU_HOST_PRESCAN");  // 0: ship the whole matrix as rounds 1-2 did (for A/B measurements)
    v = (e && *e == '0') ? 0 : 1;
  }
  return v == 1;
}

// p2gpu_prove (ncols = W) and p2gpu_prove_sparse (ncols < W: the columns >= ncols are zero except in `row`, where
// column j holds tail[j - ncols]; they are written in HBM instead of crossing PCIe)
static int prove_host(p2gpu_circuit *c, const uint64_t *wires, uint32_t ncols, const uint64_t *tail, uint32_t row,
                      const uint64_t *pis, uint32_t n_pi, uint8_t *proof_out, size_t *proof_len, p2gpu_timings *tm) {
  HIP_TRY(hipSetDevice(c->device));
  // the unused wires of the witness: zeros + one value per column, made on the device (stream-ordered before
  // every consumer below; the host part of the matrix arrives on the copy stream into the columns before them)
  auto make_tail = [&]() -> int {
    if (ncols >= c->W) return 0;
    const uint32_t nt = c->W - ncols;
    gl_t *tv = c->wires_vals.p + (size_t)ncols * c->n;
    HIP_TRY(hipMemsetAsync(tv, 0, 8 * (size_t)nt * c->n, c->stream));
    HIP_TRY(hipMemcpy2DAsync(tv + row, 8 * c->n, tail, 8, 8, nt, hipMemcpyHostToDevice, c->stream));
    return 0;
  };
  // full matrix given: look for the unused-wire suffix on the host while the first chunks upload (see host_scan_suffix).
  // Only where it can pay: the handle still classifies columns (a handle that found a dense witness stopped), and the
  // witness of a sharded proof is split by columns anyway.
  g_hp.mark("host:begin");
  HostScan scan;
  std::thread scan_thread;
  bool scanning = false;
  uint64_t *tail_pinned = nullptr;
  if (ncols == c->W && c->zero_columns && !c->structured_off && c->shard_world == 1 && host_prescan_on() && c->W > c->R) {
    row = c->sparse_row != UINT32_MAX ? c->sparse_row : 0;
    scan_thread = std::thread(host_scan_suffix, wires, c->W, c->n, row, c->R, &scan);
    scanning = true;
  }
  struct Joiner {  // never leave the function with the scan still running
    std::thread &t;
    ~Joiner() { if (t.joinable()) t.join(); }
  } joiner{scan_thread};
  if (int rc = make_tail()) return rc;
  // (Round 6 tried ONE bulk upload + the resident path whenever other proofs are in flight on the device -- their kernels fill
  // the chip anyway: 192 proofs/s against 205 with the chunks, same box, four in flight, pageable witness: the staging copy of
  // 84 MB then sits on the calling thread in one piece in front of the proof instead of under its own transforms.  Removed.)
  // The witness crosses PCIe in column chunks on a copy stream; the inverse transform and the
  // LDE of a chunk run while the next chunk is still in flight (values -> coefficients -> LDE are
  // per-column; only the leaf hash needs every column).
  // The leaf hash is a sponge over the columns in order, 17 per permutation: the rate blocks of the
  // columns that have arrived are absorbed chunk by chunk too (states wait in HBM), so that after
  // the last chunk only its own two permutations and the tree remain.
  const double t0 = now_ms();
  static const uint32_t chunk_cols = [] {
    // rate blocks (17 columns) per upload chunk; the override is for measurements (scratch/chunk_sweep.sh, 2^20 rows, round 3 with the
    // host scan shipping 80 columns: 2 blocks 6.85-7.2 ms lone / 202-203 proofs/s in flight, 3: 7.75-8.05 / 206, 4: 7.5 / 186, 5: 9.6-10 / 203)
    const char *e = getenv("P2GPU_CHUNK_BLOCKS");
    const int b = e ? atoi(e) : 2;
    return (uint32_t)(17 * (b >= 1 && b <= 64 ? b : 2));
  }();
  const uint32_t W = c->W, chunk = chunk_cols;  // two rate blocks
  const size_t n = c->n;
  if (c->shard_world > 1 && ncols < c->W) {
    // sharded proof from the compact witness: every rank uploads the dense columns itself (they are what is left
    // of the matrix once the unused wires are made on the device); no exchange
    HIP_TRY(hipMemcpyAsync(c->wires_vals.p, wires, 8 * (size_t)ncols * n, hipMemcpyHostToDevice, c->stream));
    return prove_impl(c, c->wires_vals.p, pis, n_pi, proof_out, proof_len, tm, now_ms() - t0);
  }
  if (c->shard_world > 1) {
    // Sharded proof, witness in host memory: a rank pulls only ITS block of columns
    // [q * cpr, (q + 1) * cpr) across its own PCIe link -- W / G columns, 31 MB instead of 245 MB at d = 17 and
    // G = 8 -- and the blocks are exchanged GPU to GPU with one in-place all-gather (xGMI on a real node).
    // The inverse transform then runs replicated on every rank: at 0.5 ms it is cheaper than a second
    // exchange of the same 245 MB as coefficients would be.
    const uint32_t G = (uint32_t)c->shard_world, q = (uint32_t)c->shard_rank, cpr = (W + G - 1) / G;
    if (c->wires_vals.count < (size_t)G * cpr * n) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      c->wires_vals.release();
      HIP_TRY(c->wires_vals.alloc((size_t)G * cpr * n));
    }
    const uint32_t c0 = std::min(q * cpr, W), c1 = std::min((q + 1) * cpr, W);
    if (c1 > c0)
      HIP_TRY(hipMemcpyAsync(c->wires_vals.p + (size_t)c0 * n, wires + (size_t)c0 * n, 8 * (size_t)(c1 - c0) * n, hipMemcpyHostToDevice,
                             c->stream));
    {
      // (the exchange belongs to the proof's profile like the ones inside prove_impl: `profile` = 2 counts it)
      EventProf xprof(c);
      struct XGuard {
        XGuard(Prof *p) { g_prof = p; }
        ~XGuard() { g_prof = nullptr; }
      } xguard(c->profile ? &xprof : nullptr);
      if (int rc = shard_allgather(c, c->wires_vals.p + (size_t)q * cpr * n, c->wires_vals.p, 8 * (size_t)cpr * n)) return rc;
    }
    return prove_impl(c, c->wires_vals.p, pis, n_pi, proof_out, proof_len, tm, now_ms() - t0);
  }
  const gl_t ninv = gl_inv((gl_t)n);
  Batch &b = c->wires;
  // (also implies a hashed leaf: more than 3 columns); the chunk-wise sponge is the Keccak one (17-column rate blocks)
  const bool incremental = W > chunk && c->hasher == 0;
  if (incremental && c->hash_state.count < (size_t)b.ncl * 25 * n) {  // also after set_shard(world 1 again): more local cosets
    c->hash_state.release();
    HIP_TRY(c->hash_state.alloc((size_t)b.ncl * 25 * n));
  }
  const uint32_t full_blocks = W / 17;
  uint32_t ci = 0;
  for (uint32_t col0 = 0, nc = 0; col0 < W; col0 += nc, ci++) {
    if (scanning && col0 + chunk > c->R) {
      // the first chunk that reaches beyond the routed wires: the host scan decides what is left to upload.  Columns
      // already enqueued stay as they are (ncols never drops below col0)
      g_hp.mark("host:enq");
      scan_thread.join();
      g_hp.mark("host:WAIT(scan)");
      scanning = false;
      if (scan.ncols < W) {
        ncols = std::max(scan.ncols, col0);
        // the 2-D copy below is asynchronous: its source must outlive this frame AND prove_impl's reset of the pinned arena
        if (!c->tail_stage) HIP_TRY(hipHostMalloc((void **)&c->tail_stage, 8 * (size_t)W, hipHostMallocDefault));
        tail_pinned = c->tail_stage;
        memcpy(tail_pinned, scan.tail.data() + (ncols - scan.ncols), 8 * (size_t)(W - ncols));
        tail = tail_pinned;
        if (int rc = make_tail()) return rc;
      }
    }
    // the chunk that holds the last column coming from the host also takes every column behind it (they are already
    // in HBM: nothing to wait for, and each extra absorb launch is a round trip of the 200 B sponge state per row)
    nc = col0 + chunk >= ncols ? W - col0 : chunk;
    if (ci >= c->copy_events.size()) {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      c->copy_events.push_back(e);
    }
    gl_t *vals = c->wires_vals.p + (size_t)col0 * n;
    const uint32_t nh = col0 < ncols ? std::min(nc, ncols - col0) : 0;  // columns of this chunk that come from the host
    if (nh) {
      g_hp.mark("host:enq");
      HIP_TRY(hipMemcpyAsync(vals, wires + (size_t)col0 * n, 8 * (size_t)nh * n, hipMemcpyHostToDevice, c->copy_stream));
      g_hp.mark("host:h2d");
      HIP_TRY(hipEventRecord(c->copy_events[ci], c->copy_stream));
      HIP_TRY(hipStreamWaitEvent(c->stream, c->copy_events[ci], 0));
    }
    const uint32_t *nz = batch_colnz(c, b) ? c->wire_nz.p + col0 : nullptr;
    uint32_t *cl = nz ? c->wire_clean.p + col0 : nullptr;
    if (nz) {
      column_flags(c->stream, vals, nc, c->d, c->sparse_rows, c->wire_nz.p + col0, c->wire_scalar.p + col0, c->W);
      column_clean_update(c->stream, nz, nc, cl, false);
    }
    const ColHints hi = nz ? wire_hints(c, col0, false) : ColHints(), hl = nz ? wire_hints(c, col0, true) : ColHints();
    ntt_batch(c->stream, c->plan_inv, vals, b.coeffs.p + (size_t)col0 * n, nc, 1, nullptr, ninv, false, CosetMap(), 0,
              nz ? &hi : nullptr);
    ntt_batch(c->stream, c->plan_fwd, b.coeffs.p + (size_t)col0 * n, b.lde.p + (size_t)col0 * n, nc, b.ncl, c->scale.p, 1,
              false, b.cm, W, nz ? &hl : nullptr);
    if (nz) column_clean_update(c->stream, nz, nc, cl, true);
    if (incremental) {
      const bool last = col0 + nc >= W;
      const uint32_t blk0 = col0 / 17;
      const uint32_t nblk = last ? full_blocks - blk0 : chunk / 17;
      const VirtCols v = batch_virt(c, b);
      hash_lde_absorb(c->stream, b.lde.p, W, c->d, b.ncl, blk0, nblk, col0 == 0, last, c->hash_state.p, b.dig.p, &v);
    }
  }
  const double h2d = now_ms() - t0;  // host time spent feeding PCIe (the transforms overlap with it)
  c->wires_ntt_done = true;
  c->wires_hash_done = incremental;
  int rc = prove_impl(c, c->wires_vals.p, pis, n_pi, proof_out, proof_len, tm, h2d);
  c->wires_ntt_done = false;
  c->wires_hash_done = false;
  return rc;
}

int p2gpu_prove(p2gpu_circuit *c, const uint64_t *wires, const uint64_t *pis, uint32_t n_pi, uint8_t *proof_out,
                size_t *proof_len, p2gpu_timings *tm) try {
  if (!c || !wires || !proof_out || !proof_len) return P2GPU_E_ARG;
  if (c->device < 0) { set_err("this is a verifier-only handle (p2gpu_verifier_create): no prover state"); return P2GPU_E_ARG; }
  if (!c->group.empty())
    return group_prove(c, proof_out, proof_len, tm, [&](p2gpu_circuit *m, int, uint8_t *po, size_t *pl, p2gpu_timings *t) {
      return prove_host(m, wires, m->W, nullptr, 0, pis, n_pi, po, pl, t);
    });
  return prove_host(c, wires, c->W, nullptr, 0, pis, n_pi, proof_out, proof_len, tm);
} P2GPU_CATCH

int p2gpu_prove_sparse(p2gpu_circuit *c, const uint64_t *wires, uint32_t ncols, const uint64_t *tail, uint32_t row,
                       const uint64_t *pis, uint32_t n_pi, uint8_t *proof_out, size_t *proof_len, p2gpu_timings *tm) try {
  if (!c || !proof_out || !proof_len || (ncols && !wires)) return P2GPU_E_ARG;
  if (c->device < 0) { set_err("this is a verifier-only handle (p2gpu_verifier_create): no prover state"); return P2GPU_E_ARG; }
  if (ncols > c->W || (ncols < c->W && !tail) || row >= c->n) {
    set_err("p2gpu_prove_sparse: %u dense columns of %u wires, row %u of %zu%s", ncols, c->W, row, c->n, (ncols < c->W && !tail) ? ", no tail values" : "");
    return P2GPU_E_ARG;
  }
  if (!c->group.empty())
    return group_prove(c, proof_out, proof_len, tm, [&](p2gpu_circuit *m, int, uint8_t *po, size_t *pl, p2gpu_timings *t) {
      return prove_host(m, wires, ncols, tail, row, pis, n_pi, po, pl, t);
    });
  return prove_host(c, wires, ncols, tail, row, pis, n_pi, proof_out, proof_len, tm);
} P2GPU_CATCH


}  // extern "C"
