// circuit.hpp -- host-side types shared by the prover pipeline (prover.hip) and the verifier
// (verify.hip): the Fiat-Shamir transcript, the committed-batch bookkeeping and the circuit handle
// behind the opaque `p2gpu_circuit` of include/p2gpu.h.
#pragma once
#include "internal.hpp"
#include "poseidon.hpp"
#include "../../include/p2gpu.h"
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <exception>

struct p2gpu_circuit;
namespace p2 {

// thread-local message behind p2gpu_last_error() (hostcore.hip)
void set_err(const char *fmt, ...);
std::string last_error_copy();
void last_error_restore(const std::string &s);

// ---- the circuit's hasher on the host ------------------------------------------------------------
// 0 = KeccakHash<25> (KeccakGoldilocksConfig, the reference: plonky2-backend/src/lib.rs:13): 25-byte digests,
//     Keccak "hash onion" sponge permutation;
// 1 = PoseidonHash (PoseidonGoldilocksConfig, the north_star's "Poseidon-GL Merkle-tree hashing"): digests are 4
//     field elements (32 bytes), leaves are absorbed 8 elements per Poseidon permutation (overwrite mode,
//     hash/hashing.rs hash_n_to_m_no_pad), nodes are compress(l, r) = permute(l || r || 0^4)[0..4], the
//     challenger permutes with Poseidon.
// Every C-ABI entry point selects the hasher of the handle it was given for the calling thread (use_hasher).
struct HostHasher {
  int kind = 0;
  const gl_t *prc = nullptr;  // the 360 Poseidon round constants of the handle
};
extern thread_local HostHasher g_hh;
inline size_t hh_bytes() { return g_hh.kind ? 32 : 25; }
inline void sponge_permute(gl_t st[12]) {
  if (g_hh.kind) poseidon_permute_host(st, g_hh.prc);
  else keccak_permutation12(st);
}
inline void digest_elems(const dig_t &d, gl_t e[4]) {
  if (g_hh.kind) for (int i = 0; i < 4; i++) e[i] = d.w[i];
  else dig_to_elems(d, e);
}
inline bool dig_eq(const dig_t &a, const dig_t &b) {
  return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2] && (g_hh.kind ? a.w[3] == b.w[3] : (a.w[3] & 0xFF) == (b.w[3] & 0xFF));
}
// H::hash_or_noop of a leaf of field elements
inline dig_t leaf_digest(const gl_t *v, size_t n) {
  dig_t d;
  memset(&d, 0, sizeof d);
  if (g_hh.kind) {
    if (n <= 4) {
      for (size_t i = 0; i < n; i++) d.w[i] = v[i];
    } else {
      poseidon_hash_no_pad_host(v, n, d.w, g_hh.prc);
    }
    return d;
  }
  if (n * 8 <= 25) {
    for (size_t i = 0; i < n; i++) d.w[i] = v[i];
    return d;
  }
  uint64_t h[4];
  keccak256_words(v, n, h);
  return dig_from_state(h);
}
// H::two_to_one
inline dig_t node_digest(const dig_t &l, const dig_t &r) {
  if (!g_hh.kind) return keccak_two_to_one(l, r);
  gl_t st[12] = {l.w[0], l.w[1], l.w[2], l.w[3], r.w[0], r.w[1], r.w[2], r.w[3], 0, 0, 0, 0};
  poseidon_permute_host(st, g_hh.prc);
  dig_t d;
  for (int i = 0; i < 4; i++) d.w[i] = st[i];
  return d;
}

// ---- host transcript (iop/challenger.rs, Challenger<F, H>) ----
struct Challenger {
  gl_t state[12];
  gl_t in[8];
  int n_in = 0;
  gl_t out[8];
  int n_out = 0;
  Challenger() { memset(state, 0, sizeof state); }
  void duplex() {
    for (int i = 0; i < n_in; i++) state[i] = in[i];
    n_in = 0;
    sponge_permute(state);
    for (int i = 0; i < 8; i++) out[i] = state[i];
    n_out = 8;
  }
  void observe(gl_t e) {
    n_out = 0;
    in[n_in++] = e;
    if (n_in == 8) duplex();
  }
  void observe_digest(const dig_t &d) {
    gl_t e[4];
    digest_elems(d, e);
    for (int i = 0; i < 4; i++) observe(e[i]);
  }
  void observe_cap(const std::vector<dig_t> &cap) {
    for (auto &d : cap) observe_digest(d);
  }
  void observe_ext(ext_t e) {
    observe(e.c0);
    observe(e.c1);
  }
  gl_t get() {
    if (n_in != 0 || n_out == 0) duplex();
    return out[--n_out];
  }
  ext_t get_ext() {
    gl_t a = get();
    gl_t b = get();
    return ext_make(a, b);
  }
};

inline dig_t host_hash_no_pad(const std::vector<gl_t> &v) {
  if (g_hh.kind) {
    dig_t d;
    poseidon_hash_no_pad_host(v.data(), v.size(), d.w, g_hh.prc);
    return d;
  }
  uint64_t h[4];
  keccak256_words(v.data(), v.size(), h);
  return dig_from_state(h);
}

template <class T>
struct DBuf {
  T *p = nullptr;
  size_t count = 0;
  hipError_t alloc(size_t n) {
    count = n;
    if (n == 0) return hipSuccess;
    return hipMalloc((void **)&p, n * sizeof(T));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
};

// bump arena in pinned host memory, reset at the start of every proof
struct PinnedArena {
  uint8_t *base = nullptr;
  size_t cap = 0, used = 0;
  hipError_t alloc(size_t bytes) {
    cap = bytes;
    used = 0;
    return hipHostMalloc((void **)&base, bytes, hipHostMallocDefault);
  }
  void release() {
    if (base) (void)hipHostFree(base);
    base = nullptr;
  }
  void reset() { used = 0; }
  template <class T> T *take(size_t n) {
    const size_t off = (used + 63) & ~(size_t)63, bytes = n * sizeof(T);
    if (!base || off + bytes > cap) return nullptr;
    used = off + bytes;
    return (T *)(base + off);
  }
};

// one committed polynomial batch (plonky2 PolynomialBatch), GPU layout
struct Batch {
  uint32_t cols = 0, d = 0;
  uint32_t ncl = 0;   // cosets stored locally (all 2^rate_bits, or this rank's share when sharded)
  CosetMap cm;        // local coset z <-> global coset cm.first + z * cm.stride
  DBuf<gl_t> coeffs;  // [cols][n], bit-reversed positions
  DBuf<gl_t> lde;     // [C][cols][n]
  DBuf<dig_t> dig;    // all tree levels, level l at level_off[l], layout [C][n >> l]
  std::vector<size_t> level_off;
  std::vector<dig_t> cap;  // plonky2 order
  void release() {
    coeffs.release();
    lde.release();
    dig.release();
  }
};

struct KernelStat {
  double ms = 0, bytes = 0;
  uint64_t launches = 0;
};
struct PendingEv {
  const char *name;
  double bytes;
  hipEvent_t a, b;
};


// everything behind the opaque handle
struct CircuitState {
  // parameters
  uint32_t d, W, R, NC, num_selectors, K, QF, rate_bits, cap_h, pow_bits, num_queries, n_steps, arity[8];
  uint32_t num_gates, num_pi, flags, PP, nchunks;
  uint32_t hasher = 0;  // 0 KeccakHash<25>, 1 PoseidonHash
  size_t n, N;
  uint32_t C;  // cosets = 2^rate_bits
  std::vector<GateDesc> gates;
  std::vector<gl_t> k_is;
  uint32_t nterms = 0, max_gate_constraints = 0;
  uint32_t gate_groups = 1;  // quotient kernel: 1, or 4 when the gate set is heavy
  // gates of degree <= 4 evaluated on the even cosets only, their folded sums extended to the odd cosets (plonk.hip gate_sums_kernel)
  uint32_t half_slots = 0;        // number of such gates (0: none worth it)
  uint32_t gate_groups_half = 1;  // waves per row tile of the main kernel when those gates are looked up
  uint32_t sums_groups = 1;       // ... of gate_sums_kernel
  int half_gates = 1;             // knob "half_gates": 0 never, 1 where it pays (half_auto), 2 always
  bool half_auto = false;         // the circuit is large enough for the half-domain route to pay (handle.hip)
  gl_t half_cross[16];            // the even -> odd cross-coset matrix (gate_sums_cross_kernel)
  DBuf<gl_t> hsum, htmp_a, htmp_b;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // H2D of the witness, overlapped with the transforms of earlier columns
  std::vector<hipEvent_t> copy_events;
  bool wires_ntt_done = false;        // set by p2gpu_prove: coefficients + LDE of the wires already enqueued
  bool wires_hash_done = false;       // ... and the leaf digests too (incremental sponge, hash_state)
  DBuf<uint64_t> hash_state;          // [cosets][25][n] sponge states between column chunks (allocated on first use)
  DBuf<uint32_t> wire_nz;             // [W] per proof, class of the wire column: 0 zero in every row, 1 zero except in
                                      // sparse_row, 2 dense (ColHints, internal.hpp); 0 and 1 are not transformed
  DBuf<gl_t> wire_scalar;             // [MAX_SPARSE_ROWS][W] per proof: value of every column in special row s ([0][.]: sparse_row)
  uint32_t sparse_row = UINT32_MAX;   // the PublicInputGate row: build() randomises its unused wires (UINT32_MAX: none)
  SparseRows sparse_rows;             // sparse_row, then the PoseidonGate rows (the public-input hash: wires 80..134 of a
                                      // circuit without ECC gates are non-zero only there and in sparse_row): class 3 columns
  DBuf<gl_t> sparse_coeffs, sparse_lde;  // inverse transform [rows][n] and LDE [rows][C][n] of the special rows' unit columns
  DBuf<gl_t> sparse_partial;          // per proof: [16][2] partial sums of the unit column's opening at zeta, then [2]:
                                      // the class 1 columns' joint coefficient in the FRI batch reduction
  DBuf<uint32_t> wire_nzlist;         // [1 + W]: count, then the indices of the non-zero wire columns
  DBuf<uint32_t> wire_clean;          // [W] across proofs: 1 = wires.coeffs / wires.lde of the column hold zeros already
  int zero_columns = 1;               // knob "zero_columns": do not transform structured wire columns (classes 0 and 1)
  bool structured_off = false;        // a proof on this handle found every wire column dense: stop classifying
  uint32_t last_dense = 0;            // dense wire columns of the previous proof (0: none yet): profile accounting (ColHints::dense_hint)
  uint32_t gate_wires = 0;            // wires [0, gate_wires) are what the gates of the circuit can read (max over the gate table)
  int virtual_columns = 1;            // knob "virtual_columns": structured columns >= max(R, gate_wires) get no LDE in memory (VirtCols)
  // tables
  DBuf<gl_t> tw_fwd, tw_inv, scale, inv_scale, d_kis, d_sigmas, fri_scale, qconst;
  DBuf<gl_t> l0_lde;  // [C][n]: L_0(x) = Z_H(x) / (n (x - 1)) on every LDE point (the quotient kernel's one inversion per row, tabulated)
  DBuf<GateDesc> d_gates;
  DBuf<uint8_t> d_row_gate;   // [n] gate index of every row (from the selector columns)
  DBuf<gl_t> d_gconsts, d_prc; // gate-constant columns [NC - num_selectors][n]; Poseidon round constants
  DBuf<gl_t> d_prc_hash;       // the same constants in the form the hash kernels take (poseidon_device_constants)
  NttPlan *plan_inv = nullptr, *plan_fwd = nullptr;  // size n: values->coeffs (DIF, w^-1), coeffs->values (DIT)
  std::vector<NttPlan *> fri_plans;                   // DIT plans of the FRI step sizes
  // oracles
  Batch cs, wires, zp, quot;
  dig_t circuit_digest;
  gl_t poseidon_rc[360];
  gl_t poseidon_rc_gate[360];  // poseidon_device_constants(poseidon_rc): what eval_poseidon_gate takes
  // work buffers
  DBuf<gl_t> wires_vals, zp_vals, cp, scan_tmp, apow, qvals, qtmp, pw, partial, ext_apow, f01, f01v, fv;
  std::vector<DBuf<gl_t>> fri_coef, fri_vals;
  std::vector<Batch> fri_trees;  // only dig/level_off/cap used
  DBuf<unsigned long long> pow_result;
  PinnedArena pin;
  uint64_t *tail_stage = nullptr;  // pinned [W]: row values of the unused-wire suffix the host scan of p2gpu_prove found
  DBuf<uint64_t> gather_ptrs;
  DBuf<gl_t> gather_out;
  size_t gather_cap = 0;
  std::vector<uint64_t> h_ptrs;  // host scratch that keeps its capacity from proof to proof: the gather's pointer list,
  std::vector<uint8_t> h_out;    // the proof bytes when the caller's buffer is smaller than p2gpu_proof_size_bound
  // coset sharding across ranks (one process per GPU); world = 1: everything local
  int shard_rank = 0, shard_world = 1;
  p2gpu_allgather_fn shard_fn = nullptr;  // host callback transport (tests over gloo)
  void *shard_ctx = nullptr;
  void *rccl_comm = nullptr;              // ncclComm_t: RCCL transport, collectives on `stream`
  int shard_exercise = 0;                 // run the exchange steps even with world = 1 (plumbing test)
  int shard_zs = 0;                       // knob: the chunk quotients of the permutation argument row-sharded + all-gather
  int shard_reduce = 0;                   // knob: column-sharded FRI batch reduction + all-gather / sum of the partial sums
  int shard_intt = 0;                     // knob: column-sharded inverse transforms of the wires / Z-PP + all-gather of the coefficient
                                          // blocks instead of the replicated transform; same bytes either way
  DBuf<gl_t> xchg_recv;
  // ONE process driving several GPUs (p2gpu_init with n > 1 device ids): the handle the caller holds is rank 0 of a
  // group and owns ranks 1..; every prove call fans out over one host thread per rank, and the exchanges of the
  // sharded proof are peer-to-peer copies between the ranks' own streams (PeerGroup, transport.hpp) -- no RCCL, no
  // second process, which is the shape the reference's single `prove` call site (prove_action.rs:96) can drive
  std::vector<p2gpu_circuit *> group;     // ranks 1..world-1 (empty: an ordinary handle)
  struct PeerGroup *peer = nullptr;       // shared by the ranks of a group, owned by rank 0
  // knobs
  uint64_t pow_hint = UINT64_MAX;
  int profile = 0;
  int self_check = 1;
  int blocking_sync = 0;             // knob: sleep (blocking event) instead of spinning at the transcript sync points
  hipEvent_t sync_event = nullptr;
  std::map<std::string, KernelStat> kstats;
  std::vector<PendingEv> pending;
  std::vector<hipEvent_t> event_pool;
};

}  // namespace p2

// closes the function-try-block of a C ABI entry point: no C++ exception (std::bad_alloc from a host
// container, ...) may cross the boundary
#define P2GPU_CATCH                                                   \
  catch (const std::exception &e) {                                   \
    p2::set_err("internal error: %s", e.what());                      \
    return P2GPU_E_DEVICE;                                            \
  }                                                                   \
  catch (...) {                                                       \
    p2::set_err("internal error");                                    \
    return P2GPU_E_DEVICE;                                            \
  }

struct p2gpu_circuit : p2::CircuitState {};

namespace p2 {
extern void (*g_circuit_release)(p2gpu_circuit *);  // hostcore.hip; set by handle.hip
void use_hasher(const p2gpu_circuit *c);  // select the handle's hasher for this thread (hostcore.hip)
void shard_assemble_cap(int world, unsigned rate_bits, size_t cap_per, const dig_t *gathered, std::vector<dig_t> &cap);
// blob header + gate table + (optional) cap + k_is -> the host-side fields of the handle; leaves
// *off at the constants table.  Used by p2gpu_circuit_create and p2gpu_verifier_create.
int circuit_parse(const uint8_t *blob, size_t len, p2gpu_circuit *c, size_t *off, const uint8_t **cap_in);
// per-kind parameter ranges of a gate-table entry (nullptr = acceptable) and its constraint count (hostcore.hip)
const char *gate_validate(uint32_t kind, const uint32_t p[4], uint32_t W, uint32_t gate_consts, uint64_t *wires_used,
                          uint32_t *consts_used);
uint32_t gate_num_constraints(uint32_t kind, const uint32_t p[4]);
uint32_t gate_degree(uint32_t kind, const uint32_t p[4]);
// the verifier's plonk identity at zeta on the opened values (verify.hip); also the prover's self-check
bool plonk_identity_holds(const p2gpu_circuit *c, const std::vector<ext_t> &op, const gl_t *betas, const gl_t *gammas,
                          const gl_t *alphas, ext_t zeta, const gl_t pih[4]);
}  // namespace p2
