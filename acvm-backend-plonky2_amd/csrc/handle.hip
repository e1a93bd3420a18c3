// handle.hip -- the circuit handle behind include/p2gpu.h: device selection, `p2gpu_circuit_create` (the prover-side half of
// `build()`: tables, plans, the constants / sigmas commitment; plonky2-backend/src/circuit_translation/mod.rs:80-82), knobs,
// sharding configuration, release -- and the stage-level operators the parity tests drive (inverse transform, LDE, row
// hashing, commitment, field self-test).  The proof itself is prover.hip; the exchanges of a sharded proof transport.hip.
#include "prover_internal.hpp"

using namespace p2;

namespace p2 {
void circuit_release(p2gpu_circuit *c) {
  c->tw_fwd.release(); c->tw_inv.release(); c->scale.release(); c->inv_scale.release(); c->d_kis.release();
  c->d_sigmas.release(); c->fri_scale.release(); c->d_gates.release(); c->qconst.release(); c->l0_lde.release();
  c->d_row_gate.release(); c->d_gconsts.release(); c->d_prc.release(); c->d_prc_hash.release(); c->qconst.release();
  c->hsum.release(); c->htmp_a.release(); c->htmp_b.release();
  c->cs.release(); c->wires.release(); c->zp.release(); c->quot.release();
  c->wires_vals.release(); c->zp_vals.release(); c->cp.release(); c->scan_tmp.release();
  c->apow.release(); c->qvals.release(); c->qtmp.release(); c->pw.release(); c->partial.release();
  c->ext_apow.release(); c->f01.release(); c->f01v.release(); c->fv.release();
  for (auto &b : c->fri_coef) b.release();
  for (auto &b : c->fri_vals) b.release();
  for (auto &b : c->fri_trees) b.release();
  c->hash_state.release();
  c->wire_nz.release();
  c->wire_nzlist.release();
  c->wire_clean.release();
  c->wire_scalar.release(); c->sparse_coeffs.release(); c->sparse_lde.release(); c->sparse_partial.release();
  c->pin.release();
  if (c->sync_event) (void)hipEventDestroy(c->sync_event);
  c->sync_event = nullptr;
  if (c->tail_stage) (void)hipHostFree(c->tail_stage);
  c->tail_stage = nullptr;
  c->pow_result.release(); c->gather_ptrs.release(); c->gather_out.release(); c->xchg_recv.release();
  ntt_plan_destroy(c->plan_inv);
  ntt_plan_destroy(c->plan_fwd);
  for (auto *p : c->fri_plans)
    if (p != c->plan_fwd) ntt_plan_destroy(p);
  c->plan_inv = c->plan_fwd = nullptr;
  c->fri_plans.clear();
  for (auto e : c->event_pool) (void)hipEventDestroy(e);
  for (auto e : c->copy_events) (void)hipEventDestroy(e);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
}
}  // namespace p2
namespace {

// hostcore.hip's p2gpu_circuit_destroy calls this for prover handles (it has no HIP code of its own)
void release_for_destroy(p2gpu_circuit *c) {
  // rank 0 of a device group owns the other ranks and the rendezvous
  for (p2gpu_circuit *m : c->group) {
    m->peer = nullptr;
    release_for_destroy(m);
    delete m;
  }
  c->group.clear();
  if (c->peer && c->shard_rank == 0) {
    for (auto e : c->peer->recv_free) if (e) (void)hipEventDestroy(e);
    for (auto e : c->peer->sent) if (e) (void)hipEventDestroy(e);
    delete c->peer;
  }
  c->peer = nullptr;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->rccl_comm) (void)rccl().CommDestroy((ncclComm_t)c->rccl_comm);
  c->rccl_comm = nullptr;
  circuit_release(c);
}
struct ReleaseHook {
  ReleaseHook() { p2::g_circuit_release = release_for_destroy; }
} g_release_hook;

}  // namespace
namespace p2 {
int g_device = -1;
std::vector<int> g_peer_access;  // [a * n + b]: see p2gpu_peer_access
std::vector<int> g_devices;  // p2gpu_init's list; more than one entry: circuit handles are device groups
}  // namespace p2
namespace {

}  // namespace
namespace p2 {
int ensure_device() {
  if (g_device >= 0) return 0;
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt == 0) {
    set_err("no HIP device available (%s); libp2gpu has no CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    return P2GPU_E_DEVICE;
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  g_device = dev;
  // the implicit form of p2gpu_init(&dev, 1): p2gpu_circuit_create_on(dev) and p2gpu_peer_access see a one-device list
  g_devices.assign(1, dev);
  g_peer_access.assign(1, -1);
  return 0;
}
}  // namespace p2
namespace {


}  // namespace


// scratch allocations of the stage-level operators
namespace {
struct Scratch {
  std::vector<void *> ptrs;
  hipStream_t st = nullptr;
  ~Scratch() {
    for (void *p : ptrs) (void)hipFree(p);
    if (st) (void)hipStreamDestroy(st);
  }
  template <class T> T *alloc(size_t n) {
    void *p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return (T *)p;
  }
};
}  // namespace


extern "C" {


int p2gpu_init(const int *device_ids, int n_devices) {
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt == 0) {
    set_err("no HIP device available (%s); libp2gpu has no CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    return P2GPU_E_DEVICE;
  }
  // One id: one process per GPU (replicas, or ranks of an RCCL-sharded proof).  Several ids: THIS process drives them
  // all -- every circuit handle created from now on is a device group, and each prove call is one proof coset-sharded
  // over the group; the count must divide the 2^rate_bits cosets of the circuits (checked at create).
  // The same id may appear more than once: those ranks then share a GPU, which is how the group path is tested on a
  // one-GPU box (functionally identical, no speed-up).
  if (n_devices > 8) {
    set_err("p2gpu_init: %d device ids (at most 8: one per LDE coset)", n_devices);
    return P2GPU_E_ARG;
  }
  std::vector<int> devs;
  for (int i = 0; i < (device_ids ? n_devices : 0); i++) devs.push_back(device_ids[i]);
  if (devs.empty()) devs.push_back(0);
  for (int dev : devs)
    if (dev < 0 || dev >= cnt) {
      set_err("device id %d out of range (%d devices)", dev, cnt);
      return P2GPU_E_ARG;
    }
  // direct peer copies over xGMI between the ranks of a device group.  Whether a pair got peer access is RECORDED
  // (p2gpu_peer_access): without it hipMemcpyPeerAsync still works but stages through host memory, which is a different
  // machine from the "seven concurrent xGMI transfers" the exchange is designed around -- the caller should know
  std::vector<int> peer(devs.size() * devs.size(), -1);
  for (size_t a = 0; a < devs.size(); a++)
    for (size_t b = 0; b < devs.size(); b++)
      if (devs[a] != devs[b]) {
        (void)hipSetDevice(devs[a]);
        const hipError_t pe = hipDeviceEnablePeerAccess(devs[b], 0);
        (void)hipGetLastError();
        peer[a * devs.size() + b] = (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) ? 1 : 0;
      }
  HIP_TRY(hipSetDevice(devs[0]));
  g_device = devs[0];
  g_devices = devs;
  g_peer_access = peer;
  return P2GPU_OK;
}

// matrix_out[a * n + b] for the n ids of the last p2gpu_init: 1 = device a reaches device b's memory directly (peer access
// enabled), 0 = it does not (peer copies are staged through the host), -1 = same device.  Returns n, or < 0.
int p2gpu_peer_access(int *matrix_out, int cap) {
  if (int rc = ensure_device()) return rc;
  const int n = (int)g_devices.size();
  if (matrix_out) {
    if (cap < n * n) { set_err("p2gpu_peer_access: room for %d entries, %d needed", cap, n * n); return P2GPU_E_BUFFER; }
    for (int i = 0; i < n * n; i++) matrix_out[i] = i < (int)g_peer_access.size() ? g_peer_access[i] : -1;
  }
  return n;
}

// Page-locked host memory for the wire matrix (and the proof buffer): hipMemcpyAsync from such a buffer is a true DMA at
// PCIe speed that returns at once; from pageable memory the runtime copies every byte through its own staging buffers on
// the calling thread first (2.3 ms of a 7.4 ms prove at 2^17 gates).  Portable: every device of a group sees it as pinned.
void *p2gpu_host_alloc(size_t bytes) {
  if (ensure_device()) return nullptr;
  void *p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_err("p2gpu_host_alloc: %zu bytes of page-locked memory: %s", bytes, hipGetErrorString(e));
    return nullptr;
  }
  return p;
}
void p2gpu_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int p2gpu_device_info(char *name_out, size_t name_cap, int *cu_count, size_t *hbm_bytes) {
  if (int rc = ensure_device()) return rc;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, g_device));
  if (name_out && name_cap) snprintf(name_out, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return P2GPU_OK;
}


static int shard_layout(p2gpu_circuit *c, int rank, int world);
static int shard_args_ok(p2gpu_circuit *c, int rank, int world);
static int circuit_create_one(const uint8_t *blob, size_t len, int device, p2gpu_circuit **out_c);


// A plain (un-sharded) handle on ONE device of the p2gpu_init list, whatever the length of that list: with several ids
// p2gpu_circuit_create makes device groups (one proof over all of them); this is the other half of BASELINE's metric from the
// same process -- replicas, one or more independent handles per GPU, each driven from its own host thread.
int p2gpu_circuit_create_on(const uint8_t *blob, size_t len, int device_id, p2gpu_circuit **out_c) try {
  if (!blob || !out_c) return P2GPU_E_ARG;
  if (int rc = ensure_device()) return rc;
  if (std::find(g_devices.begin(), g_devices.end(), device_id) == g_devices.end()) {
    set_err("p2gpu_circuit_create_on: device %d is not in the list given to p2gpu_init", device_id);
    return P2GPU_E_ARG;
  }
  const int rc = circuit_create_one(blob, len, device_id, out_c);
  (void)hipSetDevice(g_device);
  return rc;
} P2GPU_CATCH

int p2gpu_circuit_create(const uint8_t *blob, size_t len, p2gpu_circuit **out_c) try {
  if (!blob || !out_c) return P2GPU_E_ARG;
  if (int rc = ensure_device()) return rc;
  if (g_devices.size() <= 1) return circuit_create_one(blob, len, g_device, out_c);
  // a device group: one full handle per device (each commits constants / sigmas for itself: 5 ms), then the coset
  // layout of rank q of n on each and the shared rendezvous
  const int n = (int)g_devices.size();
  std::vector<p2gpu_circuit *> hs(n, nullptr);
  std::vector<int> rcs(n, 0);
  std::vector<std::string> errs(n);
  {
    std::vector<std::thread> th;
    for (int q = 0; q < n; q++)
      th.emplace_back([&, q] {
        rcs[q] = circuit_create_one(blob, len, g_devices[q], &hs[q]);
        if (rcs[q]) errs[q] = p2gpu_last_error();
      });
    for (auto &t : th) t.join();
  }
  (void)hipSetDevice(g_device);
  auto drop = [&](int rc, const std::string &msg) {
    for (auto *h : hs)
      if (h) {
        h->peer = nullptr;
        p2gpu_circuit_destroy(h);
      }
    set_err("%s", msg.c_str());
    return rc;
  };
  for (int q = 0; q < n; q++)
    if (rcs[q]) return drop(rcs[q], errs[q]);
  for (int q = 0; q < n; q++) {
    if (int rc = shard_args_ok(hs[q], q, n)) return drop(rc, p2gpu_last_error());
    if (int rc = shard_layout(hs[q], q, n)) return drop(rc, p2gpu_last_error());
  }
  PeerGroup *pg = new PeerGroup();
  pg->n = n;
  pg->cs = hs;
  pg->recv_free.assign(n, nullptr);
  pg->sent.assign(n, nullptr);
  pg->proof_scratch.resize(n);
  pg->send_ptr.resize(n);
  pg->recv_ptr.resize(n);
  for (int q = 0; q < n; q++) {
    (void)hipSetDevice(hs[q]->device);
    if (hipEventCreateWithFlags(&pg->recv_free[q], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&pg->sent[q], hipEventDisableTiming) != hipSuccess) {
      for (int k = 0; k <= q; k++) {  // the events of the ranks before this one (and this rank's first) exist: give them back
        (void)hipSetDevice(hs[k]->device);
        if (pg->recv_free[k]) (void)hipEventDestroy(pg->recv_free[k]);
        if (pg->sent[k]) (void)hipEventDestroy(pg->sent[k]);
        hs[k]->peer = nullptr;
      }
      (void)hipSetDevice(g_device);
      delete pg;
      return drop(P2GPU_E_DEVICE, "hipEventCreate failed");
    }
    hs[q]->peer = pg;
  }
  (void)hipSetDevice(g_device);
  for (int q = 1; q < n; q++) hs[0]->group.push_back(hs[q]);
  *out_c = hs[0];
  return P2GPU_OK;
} P2GPU_CATCH

static int circuit_create_one(const uint8_t *blob, size_t len, int device, p2gpu_circuit **out_c) try {
  p2gpu_circuit *c = new p2gpu_circuit();
  size_t off = 0;
  const uint8_t *cap_in = nullptr;
  if (int rc = circuit_parse(blob, len, c, &off, &cap_in)) {
    delete c;
    return rc;
  }
  c->device = device;
  auto fail = [&](int rc, const char *msg) {
    set_err("%s", msg);
    circuit_release(c);
    delete c;
    return rc;
  };
  c->nterms = c->K + c->K * c->nchunks + c->max_gate_constraints;
  {
    // heavy gate mixes: split the gates over 4 waves that share a row tile (plonk.hip); greedy
    // balance by an estimate of modmuls per row, group 0 starts with the permutation argument
    // (a PoseidonGate is not part of this: it has its own kernel, plonk.hip poseidon_gate_kernel)
    auto cost = [](const GateDesc &g) { return g.kind == G_POSEIDON ? 0u : 4u * g.num_constraints + 8u; };
    const char *pce = getenv("P2GPU_PERM_COST");  // balance experiments only
    const uint32_t perm_cost = pce ? (uint32_t)atoi(pce) : 8u * c->R + 100u;
    uint32_t total = 0;
    for (auto &g : c->gates) total += cost(g);
    const char *env = getenv("P2GPU_GATE_GROUPS");
    c->gate_groups = env ? (uint32_t)atoi(env) : (total > 2 * perm_cost ? 4u : 1u);
    if (c->gate_groups != 4) c->gate_groups = 1;
    uint32_t load[4] = {perm_cost, 0, 0, 0};
    std::vector<uint32_t> order(c->gates.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cost(c->gates[x]) > cost(c->gates[y]); });
    for (uint32_t gi : order) {
      uint32_t best = 0;
      for (uint32_t q = 1; q < 4; q++)
        if (load[q] < load[best]) best = q;
      c->gates[gi].pad = c->gate_groups == 4 ? best : 0;
      load[best] += cost(c->gates[gi]);
    }
    // Gates of degree <= 4 with enough constraints to pay for two transforms per challenge: evaluated on the even cosets only
    // (plonk.hip gate_sums_kernel).  The degree is the library's own count (gate_degree), not the blob's field.
    const char *he = getenv("P2GPU_HALF_GATES");  // 0: off (A/B measurements); the knob "half_gates" does the same per handle
    uint64_t half_constraints = 0;
    if (!(he && *he == '0') && c->d >= 6 && c->rate_bits == 3)
      for (auto &g : c->gates)
        if (g.kind != G_POSEIDON && g.num_constraints >= 48 && gate_degree(g.kind, g.p) <= 4 && c->half_slots < 64) {
          g.pad |= (++c->half_slots) << 16;
          half_constraints += g.num_constraints;
        }
    // Worth it by default only where it saves more than its five extra launches cost a lone proof: ~77 lane-instructions per
    // constraint on 4n rows at 37.7 T/s against ~100 us -- constraints x n >= 12 M (the heavy mix: from 2^14 gates on).  The knob
    // "half_gates" = 2 takes the route whatever the size (tests).
    c->half_auto = (half_constraints << c->d) >= ((uint64_t)12 << 20);
    if (c->half_slots) {
      // the main kernel without them: a look-up and two products per gate and challenge
      auto cost_main = [&](const GateDesc &g) { return gate_half_slot(g) ? 8u : cost(g); };
      uint32_t rest = 0;
      for (auto &g : c->gates) rest += cost_main(g);
      const char *gge = getenv("P2GPU_GATE_GROUPS_HALF");  // balance experiments only
      c->gate_groups_half = gge ? (atoi(gge) == 4 ? 4u : 1u) : (rest > 2 * perm_cost ? 4u : 1u);
      uint32_t lm[4] = {perm_cost, 0, 0, 0}, ls[4] = {0, 0, 0, 0};
      const char *sge = getenv("P2GPU_SUMS_GROUPS");  // balance experiments only
      c->sums_groups = sge ? (atoi(sge) == 4 ? 4u : 1u) : (c->half_slots >= 2 ? 4u : 1u);
      for (uint32_t gi : order) {
        GateDesc &g = c->gates[gi];
        uint32_t bm = 0, bs = 0;
        for (uint32_t q = 1; q < 4; q++) {
          if (lm[q] < lm[bm]) bm = q;
          if (ls[q] < ls[bs]) bs = q;
        }
        if (c->gate_groups_half == 4) g.pad |= bm << 4;
        lm[bm] += cost_main(g);
        if (gate_half_slot(g)) {
          if (c->sums_groups == 4) g.pad |= bs << 8;
          ls[bs] += cost(g);
        }
      }
    }
  }
  const size_t n = c->n;
  if (len < off + 8 * ((size_t)c->NC * n + (size_t)c->R * n)) return fail(P2GPU_E_BLOB, "blob truncated (tables)");
  const gl_t *k_is = c->k_is.data();
  const gl_t *constants = (const gl_t *)(blob + off);
  off += 8 * (size_t)c->NC * n;
  const gl_t *sigmas = (const gl_t *)(blob + off);

  // ---- device state ----
  auto H = [&](hipError_t e, const char *what) -> int {
    if (e == hipSuccess) return 0;
    set_err("%s: %s", what, hipGetErrorString(e));
    return P2GPU_E_DEVICE;
  };
#define CK(e, what)                                   \
  if (H((e), what)) {                                 \
    std::string keep = last_error_copy();                         \
    circuit_release(c);                               \
    delete c;                                         \
    last_error_restore(keep);                                     \
    return P2GPU_E_DEVICE;                            \
  }
  CK(hipSetDevice(c->device), "hipSetDevice");
  const double t_create0 = now_ms();
  auto mark = [&](const char *label) {  // P2GPU_TRACE=1: where circuit creation spends its time
    if (!trace_on()) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    fprintf(stderr, "[p2gpu] create %-28s +%.2f ms\n", label, now_ms() - t_create0);
  };
  if (poseidon_upload_constants()) return fail(P2GPU_E_DEVICE, "uploading Poseidon constants failed");
  CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate");
  CK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking), "hipStreamCreate");
  mark("streams + poseidon constants");
  hipStream_t st = c->stream;
  const uint32_t d = c->d, K = c->K, C = c->C;
  const uint32_t ncs = c->NC + c->R, nzp = K * (1 + c->PP), nq = K * c->QF;
  const uint32_t nall = ncs + c->W + nzp + nq;
  const size_t half = n >= 2 ? n / 2 : 1;
  CK(c->tw_fwd.alloc(half), "alloc tw");
  CK(c->tw_inv.alloc(half), "alloc tw");
  CK(c->scale.alloc((size_t)C * n), "alloc scale");
  CK(c->inv_scale.alloc((size_t)C * n), "alloc scale");
  CK(c->fri_scale.alloc((size_t)C * n), "alloc scale");
  CK(c->d_kis.alloc(c->R), "alloc kis");
  CK(c->d_sigmas.alloc((size_t)c->R * n), "alloc sigmas");
  CK(c->d_gates.alloc(c->num_gates ? c->num_gates : 1), "alloc gates");
  gl_t wn = gl_root(d), wN = gl_root(d + c->rate_bits);
  c->plan_inv = ntt_plan_create(st, d, 0, true);
  c->plan_fwd = ntt_plan_create(st, d, 1, false);
  if (!c->plan_inv || !c->plan_fwd) return fail(P2GPU_E_DEVICE, "ntt plan allocation failed");
  fill_powers(st, c->tw_fwd.p, wn, (uint32_t)half);
  fill_powers(st, c->tw_inv.p, gl_inv(wn), (uint32_t)half);
  fill_coset_scale(st, c->scale.p, GL_GEN, wN, d, C, 1);
  fill_coset_scale(st, c->inv_scale.p, gl_inv(GL_GEN), gl_inv(wN), d, C, 1);
  if (c->half_slots) {
    const size_t per = (size_t)4 * c->half_slots * K * n;
    CK(c->hsum.alloc(2 * per), "alloc gate sums");
    CK(c->htmp_a.alloc(per), "alloc gate sums");
    CK(c->htmp_b.alloc(per), "alloc gate sums");
    // F[m'][m] = 1/4 sum_j w_8^((2 m' + 1 - 2 m) j): even cosets' interpolants -> odd cosets' coefficient arrays
    const gl_t w8 = gl_root(3), quarter = gl_inv(4);
    for (int mo = 0; mo < 4; mo++)
      for (int m = 0; m < 4; m++) {
        const gl_t base = gl_pow(w8, (uint64_t)((2 * mo + 1 - 2 * m + 16) % 8));
        gl_t acc = 0, pw = 1;
        for (int j = 0; j < 4; j++) {
          acc = gl_add(acc, pw);
          pw = gl_mul(pw, base);
        }
        c->half_cross[4 * mo + m] = gl_mul(acc, quarter);
      }
  }
  CK(c->qconst.alloc(24), "alloc qconst");
  {
    // ZeroPolyOnCoset: Z_H on the LDE coset has period 2^rate_bits
    gl_t qc[24];
    memset(qc, 0, sizeof qc);
    gl_t wC = gl_root(c->rate_bits), gn = gl_pow(GL_GEN, n);
    for (uint32_t r = 0; r < C; r++) {
      qc[r] = gl_mul(GL_GEN, gl_pow(wN, r));
      qc[8 + r] = gl_sub(gl_mul(gn, gl_pow(wC, r)), 1);
      qc[16 + r] = gl_inv(qc[8 + r]);
    }
    CK(hipMemcpy(c->qconst.p, qc, sizeof qc, hipMemcpyHostToDevice), "copy qconst");
    CK(c->l0_lde.alloc((size_t)C * n), "alloc L0 table");
    fill_l0_table(st, c->qconst.p, c->tw_fwd.p, 0, d, C, gl_inv((gl_t)n), c->l0_lde.p);
  }
  mark("root tables + ntt plans");
  CK(hipMemcpyAsync(c->d_kis.p, k_is, 8 * (size_t)c->R, hipMemcpyHostToDevice, st), "copy kis");
  CK(hipMemcpyAsync(c->d_sigmas.p, sigmas, 8 * (size_t)c->R * n, hipMemcpyHostToDevice, st), "copy sigmas");
  if (c->num_gates)
    CK(hipMemcpyAsync(c->d_gates.p, c->gates.data(), sizeof(GateDesc) * c->num_gates, hipMemcpyHostToDevice, st), "copy gates");
  {
    // row -> gate (the one selector column that is not UNUSED holds the index) and the gate
    // constants, for the row-local witness generators (p2gpu_fill_witness)
    std::vector<uint8_t> rg(n, 0);
    for (size_t row = 0; row < n; row++) {
      uint32_t gi = 0;
      for (uint32_t s = 0; s < c->num_selectors; s++) {
        gl_t v = constants[(size_t)s * n + row];
        if (c->num_selectors == 1 || v != 0xFFFFFFFFULL) gi = (uint32_t)v;
      }
      if (gi >= c->num_gates) return fail(P2GPU_E_BLOB, "selector column holds an unknown gate index");
      rg[row] = (uint8_t)gi;
      if (c->sparse_row == UINT32_MAX && c->gates[gi].kind == G_PUBLIC_INPUT) c->sparse_row = (uint32_t)row;
    }
    // the special rows of the column classification: the PublicInputGate row first, then the PoseidonGate rows
    c->sparse_rows = SparseRows();
    if (c->sparse_row != UINT32_MAX) {
      c->sparse_rows.row[c->sparse_rows.count++] = c->sparse_row;
      for (size_t row = 0; row < n && c->sparse_rows.count < MAX_SPARSE_ROWS; row++)
        if (c->gates[rg[row]].kind == G_POSEIDON) c->sparse_rows.row[c->sparse_rows.count++] = (uint32_t)row;
    }
    const uint32_t ngc = c->NC - c->num_selectors;
    CK(c->d_row_gate.alloc(n), "alloc row_gate");
    CK(c->d_gconsts.alloc((size_t)(ngc ? ngc : 1) * n), "alloc gconsts");
    CK(c->d_prc.alloc(360), "alloc prc");
    CK(hipMemcpy(c->d_row_gate.p, rg.data(), n, hipMemcpyHostToDevice), "copy row_gate");
    if (ngc)
      CK(hipMemcpy(c->d_gconsts.p, constants + (size_t)c->num_selectors * n, 8 * (size_t)ngc * n, hipMemcpyHostToDevice), "copy gconsts");
    CK(hipMemcpy(c->d_prc.p, c->poseidon_rc, sizeof c->poseidon_rc, hipMemcpyHostToDevice), "copy prc");
    {
      gl_t hrc[360];
      poseidon_device_constants(c->poseidon_rc, hrc);
      CK(c->d_prc_hash.alloc(360), "alloc prc (hash form)");
      CK(hipMemcpy(c->d_prc_hash.p, hrc, sizeof hrc, hipMemcpyHostToDevice), "copy prc (hash form)");
    }
  }
  mark("sigma/constant uploads, row->gate");
  if (batch_alloc(c, c->cs, ncs) || batch_alloc(c, c->wires, c->W) || batch_alloc(c, c->zp, nzp) || batch_alloc(c, c->quot, nq)) {
    std::string keep = last_error_copy();
    circuit_release(c);
    delete c;
    last_error_restore(keep);
    return P2GPU_E_DEVICE;
  }
  CK(c->wires_vals.alloc((size_t)c->W * n), "alloc wires");
  CK(c->wire_nz.alloc(c->W), "alloc wire flags");
  CK(c->wire_nzlist.alloc(c->W + 1), "alloc wire flags");
  CK(c->wire_clean.alloc(c->W), "alloc wire flags");
  CK(hipMemsetAsync(c->wire_clean.p, 0, sizeof(uint32_t) * c->W, c->stream), "clear wire flags");
  CK(c->wire_scalar.alloc((size_t)MAX_SPARSE_ROWS * c->W), "alloc wire flags");
  if (c->sparse_row != UINT32_MAX) {
    // inverse transform and LDE (all cosets) of the unit column of the PublicInputGate row: what a wire that is
    // zero everywhere but there transforms to, up to its scalar -- and the same for the PoseidonGate rows
    // (sparse_rows: columns that are zero outside these rows are their linear combination, class 3)
    const uint32_t nr = c->sparse_rows.count;
    CK(c->sparse_coeffs.alloc((size_t)nr * n), "alloc sparse basis");
    CK(c->sparse_lde.alloc((size_t)nr * C * n), "alloc sparse basis");
    CK(c->sparse_partial.alloc(16 * 2 + 2), "alloc sparse basis");
    const gl_t one = 1;
    CK(hipMemsetAsync(c->sparse_coeffs.p, 0, 8 * (size_t)nr * n, st), "sparse basis");
    for (uint32_t s = 0; s < nr; s++)
      CK(hipMemcpyAsync(c->sparse_coeffs.p + (size_t)s * n + c->sparse_rows.row[s], &one, 8, hipMemcpyHostToDevice, st), "sparse basis");
    CK(hipStreamSynchronize(st), "sparse basis");  // `one` lives on this stack frame
    ntt_batch(st, c->plan_inv, c->sparse_coeffs.p, c->sparse_coeffs.p, nr, 1, nullptr, gl_inv((gl_t)n), false);
    for (uint32_t s = 0; s < nr; s++)  // [rows][C][n]: one column per launch keeps every row's cosets together
      ntt_batch(st, c->plan_fwd, c->sparse_coeffs.p + (size_t)s * n, c->sparse_lde.p + (size_t)s * C * n, 1, C, c->scale.p, 1, false);
  }
  CK(c->zp_vals.alloc((size_t)nzp * n), "alloc zp");
  CK(c->cp.alloc((size_t)K * (c->nchunks + 1) * n), "alloc cp");  // chunk quotients + the row products behind them (ZsArgs::cp)
  CK(c->scan_tmp.alloc((size_t)K * (n + (n + 255) / 256 + 8)), "alloc scan");
  CK(c->apow.alloc((size_t)2 * c->nterms), "alloc apow");
  CK(c->qvals.alloc((size_t)K * C * n), "alloc qvals");
  CK(c->qtmp.alloc((size_t)K * C * n), "alloc qtmp");
  CK(c->pw.alloc((size_t)4 * n), "alloc pw");
  CK(c->partial.alloc(((size_t)(nall + K) + 8) * 16 * 2), "alloc partial");  // (+8 columns: a sharded proof gathers equal blocks per rank)
  CK(c->ext_apow.alloc((size_t)2 * nall), "alloc ext_apow");
  CK(c->f01.alloc((size_t)4 * n), "alloc f01");
  CK(c->f01v.alloc((size_t)4 * n), "alloc f01v");
  CK(c->fv.alloc((size_t)2 * n), "alloc fv");
  c->fri_coef.resize(c->n_steps + 1);
  c->fri_vals.resize(c->n_steps + 1);
  c->fri_trees.resize(c->n_steps);
  size_t gather_words = 0;
  {
    uint32_t ds = d;
    const size_t cap_per = ((size_t)1 << c->cap_h) >> c->rate_bits;
    for (uint32_t s = 0; s <= c->n_steps; s++) {
      c->fri_plans.push_back(s == 0 ? c->plan_fwd : ntt_plan_create(st, ds, 1, false));
      if (!c->fri_plans.back()) return fail(P2GPU_E_DEVICE, "ntt plan allocation failed");
      CK(c->fri_coef[s].alloc((size_t)2 << ds), "alloc fri coef");
      if (s < c->n_steps) {
        CK(c->fri_vals[s].alloc((size_t)2 * C << ds), "alloc fri vals");
        uint32_t ab = c->arity[s];
        if (ab < 1 || ab > 4 || ds < ab || (((size_t)1 << ds) >> ab) < cap_per) {
          return fail(P2GPU_E_BLOB, "unsupported FRI reduction arity");
        }
        if (tree_alloc(c->fri_trees[s], C, ((size_t)1 << ds) >> ab, cap_per)) {
          std::string keep = last_error_copy();
          circuit_release(c);
          delete c;
          last_error_restore(keep);
          return P2GPU_E_DEVICE;
        }
        gather_words += (2u << ab) + 4 * (size_t)(ds + c->rate_bits);
        ds -= ab;
      }
    }
  }
  gather_words += nall + 4 * 4 * (size_t)(d + c->rate_bits);
  c->gather_cap = gather_words * c->num_queries + 64;
  CK(c->gather_ptrs.alloc(c->gather_cap), "alloc gather");
  CK(c->gather_out.alloc(c->gather_cap), "alloc gather");
  CK(c->pow_result.alloc(1), "alloc pow");
  // pinned host arena for every small transcript transfer of a proof (caps, opening partials, challenge
  // powers, final polynomial, PoW result, query gather): with pageable memory each of those
  // hipMemcpyAsync calls blocks in a staging copy and the following stream sync costs another ~10 us
  {
    size_t n_final = n;
    for (uint32_t s = 0; s < c->n_steps; s++) n_final >>= c->arity[s];
    // per proof: 3 + n_steps trees stage their caps (2^cap_h digests each, twice when sharded: local roots + gathered),
    // the alpha powers of the openings (2 * nall words), the opening partials, the final polynomial, the query gather
    const size_t caps = (size_t)2 * (4 + c->n_steps) * (sizeof(dig_t) << c->cap_h);
    CK(c->pin.alloc(16 * c->gather_cap + 32 * n_final + 16 * (size_t)(nall + K) * 16 + 16 * (size_t)nall + 16 * (size_t)c->nterms +
                    20 * (size_t)c->W + caps + ((size_t)1 << 18)),  // (4 W more: the column classes read back early by shard_intt)
       "alloc pinned staging");
  }

  mark("batch + work buffer allocation");
  // ---- constants_sigmas commitment (the prover-side part of `build()`) ----
  {
    // stage values [constants | sigmas] in the wires buffer region of the cs LDE (reuse cs.lde as scratch)
    gl_t *stage = c->cs.lde.p;
    CK(hipMemcpyAsync(stage, constants, 8 * (size_t)c->NC * n, hipMemcpyHostToDevice, st), "copy constants");
    CK(hipMemcpyAsync(stage + (size_t)c->NC * n, c->d_sigmas.p, 8 * (size_t)c->R * n, hipMemcpyDeviceToDevice, st), "copy sigmas");
    {
      gl_t ninv = gl_inv((gl_t)n);
      ntt_batch(st, c->plan_inv, stage, c->cs.coeffs.p, ncs, 1, nullptr, ninv, false);
    }
    if (int rc = batch_commit_from_coeffs(c, c->cs)) {
      std::string keep = last_error_copy();
      circuit_release(c);
      delete c;
      last_error_restore(keep);
      return rc;
    }
  }
  if (cap_in) {
    for (size_t i = 0; i < c->cs.cap.size(); i++)
      if (memcmp(cap_in + 32 * i, c->cs.cap[i].w, c->hasher ? 32 : 25)) return fail(P2GPU_E_CAP_MISMATCH, "constants_sigmas cap mismatch");
  }
  if (!(c->flags & 1)) {
    // circuit_builder.rs build(): H::hash_no_pad(cap.flatten() || hash_pad([]).to_vec() || [degree_bits])  (pinned: tests/test_reference_proofs.py)
    std::vector<gl_t> parts;
    for (auto &dg : c->cs.cap) {
      gl_t e[4];
      digest_elems(dg, e);
      parts.insert(parts.end(), e, e + 4);
    }
    std::vector<gl_t> pad(8, 0);  // hash_pad([]): pad10*1 to the sponge rate
    pad[0] = 1;
    pad[7] = 1;
    dig_t ds = host_hash_no_pad(pad);
    gl_t e[4];
    digest_elems(ds, e);
    parts.insert(parts.end(), e, e + 4);
    parts.push_back(d);
    c->circuit_digest = host_hash_no_pad(parts);
  }
  CK(hipStreamSynchronize(st), "sync");
  mark("constants_sigmas commitment");
#undef CK
  *out_c = c;
  return P2GPU_OK;
} P2GPU_CATCH



int p2gpu_circuit_set(p2gpu_circuit *c, const char *key, uint64_t value) try {
  if (!c || !key) return P2GPU_E_ARG;
  if (c->device < 0) { set_err("this is a verifier-only handle (p2gpu_verifier_create): no prover state"); return P2GPU_E_ARG; }
  for (p2gpu_circuit *m : c->group)  // a device group: every rank gets the same knobs (the ranks must take the same path)
    if (int rc = p2gpu_circuit_set(m, key, value)) return rc;
  HIP_TRY(hipSetDevice(c->device));
  std::string k(key);
  if (k == "pow_hint") c->pow_hint = value;
  else if (k == "self_check") c->self_check = (int)value;
  else if (k == "shard_exercise") c->shard_exercise = (int)value;
  else if (k == "shard_intt") c->shard_intt = (int)value;
  else if (k == "shard_reduce") c->shard_reduce = (int)value;
  else if (k == "shard_zs") c->shard_zs = (int)value;
  else if (k == "half_gates") c->half_gates = (int)value;
  else if (k == "blocking_sync") c->blocking_sync = (int)value;
  else if (k == "virtual_columns") {
    c->virtual_columns = (int)value;
    // a clean mark set while the knob was on vouches for the coefficients only (the LDE was never written): forget them
    if (c->wire_clean.p) HIP_TRY(hipMemsetAsync(c->wire_clean.p, 0, sizeof(uint32_t) * c->W, c->stream));
  }
  else if (k == "zero_columns") {
    c->zero_columns = (int)value;
    c->structured_off = false;
    // proofs made with the knob off overwrite every column without touching the marks: forget them
    if (c->wire_clean.p) HIP_TRY(hipMemsetAsync(c->wire_clean.p, 0, sizeof(uint32_t) * c->W, c->stream));
  }
  else if (k == "profile") {
    flush_kstats(c);
    c->profile = (int)value;
    c->kstats.clear();
  } else {
    set_err("unknown knob %s", key);
    return P2GPU_E_ARG;
  }
  return P2GPU_OK;
} P2GPU_CATCH

static int shard_layout(p2gpu_circuit *c, int rank, int world) {
  HIP_TRY(hipSetDevice(c->device));
  if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->rccl_comm) {
    (void)rccl().CommDestroy((ncclComm_t)c->rccl_comm);
    c->rccl_comm = nullptr;
  }
  c->shard_rank = rank;
  c->shard_world = world;
  c->shard_fn = nullptr;
  c->shard_ctx = nullptr;
  const uint32_t ncl = c->C / (uint32_t)world;
  const size_t cap_per = ((size_t)1 << c->cap_h) >> c->rate_bits;
  CosetMap cm;
  cm.first = (uint32_t)rank;
  cm.stride = (uint32_t)world;
  // the per-proof oracles and the first FRI tree are sharded by coset; the constants/sigmas oracle
  // and the later (16x smaller) FRI steps stay complete on every rank
  Batch *bs[3] = {&c->wires, &c->zp, &c->quot};
  for (Batch *b : bs) {
    b->ncl = ncl;
    b->cm = cm;
    tree_layout(*b, ncl, c->n, cap_per);
  }
  if (c->n_steps > 0) {
    Batch &t0 = c->fri_trees[0];
    t0.ncl = ncl;
    t0.cm = cm;
    tree_layout(t0, ncl, c->n >> c->arity[0], cap_per);
  }
  // a different number of local cosets changes which part of wires.lde a column's transforms cover
  if (c->wire_clean.p) HIP_TRY(hipMemsetAsync(c->wire_clean.p, 0, sizeof(uint32_t) * c->W, c->stream));
  c->xchg_recv.release();
  // receive side of the largest exchange: quotient interpolants (K * C * n words in total) or the query gather
  HIP_TRY(c->xchg_recv.alloc(std::max((size_t)world * c->gather_cap, (size_t)c->K * c->C * c->n) + 64));
  return P2GPU_OK;
}
static int shard_args_ok(p2gpu_circuit *c, int rank, int world) {
  if (!c || world < 1 || rank < 0 || rank >= world || (c->C % (uint32_t)world) != 0) {
    set_err("bad shard configuration: rank %d of %d (cosets %u)", rank, world, c ? c->C : 0u);
    return P2GPU_E_ARG;
  }
  if (c->device < 0) { set_err("this is a verifier-only handle (p2gpu_verifier_create): no prover state"); return P2GPU_E_ARG; }
  return P2GPU_OK;
}

static int not_a_group(const p2gpu_circuit *c) {
  if (c && c->peer) { set_err("this handle is a device group of one process (p2gpu_init with several ids): its sharding is fixed"); return P2GPU_E_ARG; }
  return P2GPU_OK;
}
int p2gpu_circuit_set_shard(p2gpu_circuit *c, int rank, int world, p2gpu_allgather_fn fn, void *ctx) try {
  if (int rc = not_a_group(c)) return rc;
  if (int rc = shard_args_ok(c, rank, world)) return rc;
  if (world > 1 && !fn) { set_err("a host all-gather callback is needed for world > 1 (or use p2gpu_circuit_set_shard_rccl)"); return P2GPU_E_ARG; }
  if (int rc = shard_layout(c, rank, world)) return rc;
  c->shard_fn = fn;
  c->shard_ctx = ctx;
  return P2GPU_OK;
} P2GPU_CATCH

int p2gpu_shard_unique_id(uint8_t id_out[128]) try {
  if (!id_out) return P2GPU_E_ARG;
  if (int rc = ensure_device()) return rc;
  if (!rccl().ok) { set_err("librccl.so.1 could not be loaded"); return P2GPU_E_DEVICE; }
  ncclUniqueId id;
  static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
  RCCL_TRY(rccl().GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return P2GPU_OK;
} P2GPU_CATCH

int p2gpu_circuit_set_shard_rccl(p2gpu_circuit *c, int rank, int world, const uint8_t id_in[128]) try {
  if (int rc = not_a_group(c)) return rc;
  if (int rc = shard_args_ok(c, rank, world)) return rc;
  if (!id_in) return P2GPU_E_ARG;
  if (!rccl().ok) { set_err("librccl.so.1 could not be loaded"); return P2GPU_E_DEVICE; }
  if (int rc = shard_layout(c, rank, world)) return rc;
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof id);
  ncclComm_t comm = nullptr;
  RCCL_TRY(rccl().CommInitRank(&comm, world, id, rank));
  c->rccl_comm = (void *)comm;
  return P2GPU_OK;
} P2GPU_CATCH

// per-kernel event timings accumulated while "profile" = 1: writes up to `cap`
// entries "name\0" (64 B each) + total ms + launch count; returns the number of entries
int p2gpu_kernel_stats(p2gpu_circuit *c, char *names, double *ms, double *bytes, uint64_t *launches, int cap) try {
  if (!c) return P2GPU_E_ARG;
  if (c->device < 0) { set_err("this is a verifier-only handle (p2gpu_verifier_create): no prover state"); return P2GPU_E_ARG; }
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  flush_kstats(c);
  int i = 0;
  for (auto &kv : c->kstats) {
    if (i >= cap) break;
    snprintf(names + 64 * i, 64, "%s", kv.first.c_str());
    ms[i] = kv.second.ms;
    bytes[i] = kv.second.bytes;
    launches[i] = kv.second.launches;
    i++;
  }
  return i;
} P2GPU_CATCH

// ---- stage-level operators (host buffers) ----
int p2gpu_ifft_batch(const uint64_t *vals, size_t ncols, unsigned d, uint64_t *coeffs_out) try {
  if (int rc = ensure_device()) return rc;
  if (!vals || !coeffs_out || d > 24) return P2GPU_E_ARG;
  Scratch S;
  HIP_TRY(hipStreamCreate(&S.st));
  size_t n = (size_t)1 << d, half = n >= 2 ? n / 2 : 1;
  gl_t *a = S.alloc<gl_t>(ncols * n), *b = S.alloc<gl_t>(ncols * n), *tw = S.alloc<gl_t>(half);
  if (!a || !b || !tw) { set_err("hipMalloc failed"); return P2GPU_E_DEVICE; }
  HIP_TRY(hipMemcpyAsync(a, vals, 8 * ncols * n, hipMemcpyHostToDevice, S.st));
  NttPlan *plan = ntt_plan_create(S.st, d, 0, true);
  if (!plan) { set_err("hipMalloc failed"); return P2GPU_E_DEVICE; }
  ntt_batch(S.st, plan, a, b, (uint32_t)ncols, 1, nullptr, gl_inv((gl_t)n), false);
  bitrev_cols(S.st, b, a, d, (uint32_t)ncols);  // natural-order coefficients for the caller
  hipError_t e1 = hipMemcpyAsync(coeffs_out, a, 8 * ncols * n, hipMemcpyDeviceToHost, S.st);
  hipError_t e2 = hipStreamSynchronize(S.st);
  ntt_plan_destroy(plan);
  HIP_TRY(e1);
  HIP_TRY(e2);
  return P2GPU_OK;
} P2GPU_CATCH

int p2gpu_lde_batch(const uint64_t *coeffs, size_t ncols, unsigned d, unsigned rate_bits, uint64_t *lde_out) try {
  if (int rc = ensure_device()) return rc;
  if (!coeffs || !lde_out || d > 24 || rate_bits > 3) return P2GPU_E_ARG;
  Scratch S;
  HIP_TRY(hipStreamCreate(&S.st));
  size_t n = (size_t)1 << d, half = n >= 2 ? n / 2 : 1;
  uint32_t C = 1u << rate_bits;
  gl_t *a = S.alloc<gl_t>(ncols * n), *b = S.alloc<gl_t>(ncols * n), *tw = S.alloc<gl_t>(half);
  gl_t *scale = S.alloc<gl_t>(C * n), *lde = S.alloc<gl_t>(C * ncols * n);
  if (!a || !b || !tw || !scale || !lde) { set_err("hipMalloc failed"); return P2GPU_E_DEVICE; }
  HIP_TRY(hipMemcpyAsync(a, coeffs, 8 * ncols * n, hipMemcpyHostToDevice, S.st));
  bitrev_cols(S.st, a, b, d, (uint32_t)ncols);
  NttPlan *plan = ntt_plan_create(S.st, d, 1, false);
  if (!plan) { set_err("hipMalloc failed"); return P2GPU_E_DEVICE; }
  fill_coset_scale(S.st, scale, GL_GEN, gl_root(d + rate_bits), d, C, 1);
  ntt_batch(S.st, plan, b, lde, (uint32_t)ncols, C, scale, 1, false);
  // [C][cols][n] -> natural order per column: out[col][C*k + r]
  std::vector<gl_t> tmp(C * ncols * n);
  hipError_t e1 = hipMemcpyAsync(tmp.data(), lde, 8 * tmp.size(), hipMemcpyDeviceToHost, S.st);
  hipError_t e2 = hipStreamSynchronize(S.st);
  ntt_plan_destroy(plan);
  HIP_TRY(e1);
  HIP_TRY(e2);
  for (uint32_t r = 0; r < C; r++)
    for (size_t col = 0; col < ncols; col++)
      for (size_t k = 0; k < n; k++) lde_out[col * C * n + C * k + r] = tmp[((size_t)r * ncols + col) * n + k];
  return P2GPU_OK;
} P2GPU_CATCH

static int field_selftest_words(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *bad_out, size_t words);
// the entry point of rounds 1-3 keeps its contract: EIGHT words (a caller built against the old header passes uint64_t[8])
int p2gpu_field_selftest(const uint64_t *a, const uint64_t *b, size_t n, uint64_t bad_out[8]) { return field_selftest_words(a, b, n, bad_out, 8); }
int p2gpu_field_selftest16(const uint64_t *a, const uint64_t *b, size_t n, uint64_t bad_out[16]) { return field_selftest_words(a, b, n, bad_out, 16); }
static int field_selftest_words(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *bad_out, size_t words) try {
  if (int rc = ensure_device()) return rc;
  if (!a || !b || !bad_out || n == 0 || n > ((size_t)1 << 28)) return P2GPU_E_ARG;
  Scratch S;
  HIP_TRY(hipStreamCreate(&S.st));
  uint64_t *da = S.alloc<uint64_t>(n), *db = S.alloc<uint64_t>(n);
  unsigned long long *bad = S.alloc<unsigned long long>(16);
  if (!da || !db || !bad) { set_err("hipMalloc failed"); return P2GPU_E_DEVICE; }
  HIP_TRY(hipMemcpyAsync(da, a, 8 * n, hipMemcpyHostToDevice, S.st));
  HIP_TRY(hipMemcpyAsync(db, b, 8 * n, hipMemcpyHostToDevice, S.st));
  HIP_TRY(hipMemsetAsync(bad, 0, 128, S.st));
  field_selftest(S.st, da, db, (uint32_t)n, bad);
  HIP_TRY(hipMemcpyAsync(bad_out, bad, 8 * words, hipMemcpyDeviceToHost, S.st));
  HIP_TRY(hipStreamSynchronize(S.st));
  return P2GPU_OK;
} P2GPU_CATCH

int p2gpu_hash_rows(const uint64_t *rows, size_t n_rows, size_t row_len, uint8_t *digests_out) try {
  if (int rc = ensure_device()) return rc;
  if (!rows || !digests_out) return P2GPU_E_ARG;
  Scratch S;
  HIP_TRY(hipStreamCreate(&S.st));
  gl_t *a = S.alloc<gl_t>(n_rows * row_len);
  dig_t *dg = S.alloc<dig_t>(n_rows);
  if (!a || !dg) { set_err("hipMalloc failed"); return P2GPU_E_DEVICE; }
  HIP_TRY(hipMemcpyAsync(a, rows, 8 * n_rows * row_len, hipMemcpyHostToDevice, S.st));
  hash_rows(S.st, a, n_rows, (uint32_t)row_len, dg);
  std::vector<dig_t> h(n_rows);
  HIP_TRY(hipMemcpyAsync(h.data(), dg, sizeof(dig_t) * n_rows, hipMemcpyDeviceToHost, S.st));
  HIP_TRY(hipStreamSynchronize(S.st));
  for (size_t i = 0; i < n_rows; i++) memcpy(digests_out + 25 * i, h[i].w, 25);
  return P2GPU_OK;
} P2GPU_CATCH

int p2gpu_commit_values(const uint64_t *vals, size_t ncols, unsigned d, unsigned rate_bits, unsigned cap_h,
                        uint8_t *cap_out) try {
  if (int rc = ensure_device()) return rc;
  if (!vals || !cap_out || d > 24 || rate_bits > 3 || cap_h < rate_bits || cap_h > rate_bits + d) return P2GPU_E_ARG;
  // a throw-away circuit-like context with just the tables and one batch
  p2gpu_circuit *c = new p2gpu_circuit();
  c->d = d; c->rate_bits = rate_bits; c->cap_h = cap_h; c->n = (size_t)1 << d; c->N = c->n << rate_bits;
  c->C = 1u << rate_bits; c->device = g_device; c->n_steps = 0;
  int rc = P2GPU_OK;
  size_t n = c->n, half = n >= 2 ? n / 2 : 1;
  do {
    if (hipStreamCreate(&c->stream) != hipSuccess || c->tw_fwd.alloc(half) != hipSuccess ||
        c->tw_inv.alloc(half) != hipSuccess || c->scale.alloc((size_t)c->C * n) != hipSuccess ||
        c->wires_vals.alloc(ncols * n) != hipSuccess || c->pin.alloc(((size_t)1 << 16) + 2 * (sizeof(dig_t) << cap_h)) != hipSuccess) {
      set_err("hipMalloc failed");
      rc = P2GPU_E_DEVICE;
      break;
    }
    if ((rc = batch_alloc(c, c->wires, (uint32_t)ncols))) break;
    c->plan_inv = ntt_plan_create(c->stream, d, 0, true);
    c->plan_fwd = ntt_plan_create(c->stream, d, 1, false);
    if (!c->plan_inv || !c->plan_fwd) {
      set_err("hipMalloc failed");
      rc = P2GPU_E_DEVICE;
      break;
    }
    fill_coset_scale(c->stream, c->scale.p, GL_GEN, gl_root(d + rate_bits), d, c->C, 1);
    if (hipMemcpyAsync(c->wires_vals.p, vals, 8 * ncols * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
      set_err("copy failed");
      rc = P2GPU_E_DEVICE;
      break;
    }
    if ((rc = batch_commit_from_values(c, c->wires, c->wires_vals.p))) break;
    for (size_t i = 0; i < c->wires.cap.size(); i++) memcpy(cap_out + 25 * i, c->wires.cap[i].w, 25);
  } while (0);
  circuit_release(c);
  delete c;
  return rc;
} P2GPU_CATCH

}  // extern "C"
