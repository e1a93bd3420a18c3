// hostcore.hip -- the kernel-free core of libp2gpu: error text, circuit-blob parsing and validation,
// handle getters and destruction.  Everything here (and verify.hip, proofio.hip) is plain host code
// that never launches a kernel, so the three files also build WITHOUT device code under
// -fsanitize=address,undefined (`make asan` -> libp2gpu_host_asan.so) for the fuzz tests: these are
// the parsers that read untrusted bytes on machines with no GPU (the reference's `verify` action,
// plonky2-backend/src/actions/verify_action.rs:11-17, reader noir_and_plonky2_serialization.rs:16-33).
#include "circuit.hpp"
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

using namespace p2;

namespace {
thread_local std::string g_err;
}  // namespace
namespace p2 {
void set_err(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
thread_local HostHasher g_hh;
void use_hasher(const p2gpu_circuit *c) {
  g_hh.kind = c && c->hasher == 1;
  g_hh.prc = c ? c->poseidon_rc : nullptr;
}
std::string last_error_copy() { return g_err; }
void last_error_restore(const std::string &s) { g_err = s; }
// Merkle cap in plonky2 order from the all-gather of every rank's local subtree roots:
// gathered = [world][C / world local cosets][cap_per], rank q's local coset z being the global coset
// r = q + z * world.  plonky2's leaf index bitrev(8k + r) has bitrev(r) as its top bits, so coset r
// owns the cap entries bitrev(r) * cap_per + bitrev(k); world = 1 is the unsharded case.
void shard_assemble_cap(int world, unsigned rate_bits, size_t cap_per, const dig_t *gathered, std::vector<dig_t> &cap) {
  const uint32_t C = 1u << rate_bits, CL = C / (uint32_t)world;
  unsigned lgp = 0;
  while (((size_t)1 << lgp) < cap_per) lgp++;
  cap.assign((size_t)C * cap_per, dig_t{});
  for (int q = 0; q < world; q++)
    for (uint32_t z = 0; z < CL; z++)
      for (size_t k = 0; k < cap_per; k++)
        cap[(size_t)bitrev32((uint32_t)q + z * (uint32_t)world, rate_bits) * cap_per + bitrev32((uint32_t)k, lgp)] =
            gathered[((size_t)q * CL + z) * cap_per + k];
}
// set by handle.hip (the TU that creates and releases the device state): releases a prover handle's device state
void (*g_circuit_release)(p2gpu_circuit *) = nullptr;
}  // namespace p2

namespace p2 {
// Per-kind parameter ranges and wire/constant bounds of one gate-table entry.  Runs BEFORE anything
// derives a size or an index from the parameters: the blob is untrusted input for the CPU-only
// verifier (p2gpu_verifier_create) as much as for the prover, and the same descriptors drive the
// indices of the quotient / witness kernels.  Returns nullptr when the entry is acceptable.
const char *gate_validate(uint32_t kind, const uint32_t p[4], uint32_t W, uint32_t gate_consts, uint64_t *wires_used,
                          uint32_t *consts_used) {
  uint64_t w = 0;
  uint32_t k = 0;
  auto in = [](uint32_t v, uint32_t lo, uint32_t hi) { return v >= lo && v <= hi; };
  switch (kind) {
  case G_NOOP: break;
  case G_CONSTANT:
    if (!in(p[0], 1, 4096)) return "ConstantGate: num_consts out of range";
    w = p[0]; k = p[0];
    break;
  case G_PUBLIC_INPUT: w = 4; break;
  case G_ARITHMETIC:
    if (!in(p[0], 1, 1024)) return "ArithmeticGate: num_ops out of range";
    w = 4ull * p[0]; k = 2;
    break;
  case G_BASE_SUM:
    if (!in(p[0], 2, 8)) return "BaseSumGate: base must be 2..8";
    if (!in(p[1], 1, 64)) return "BaseSumGate: num_limbs must be 1..64";
    w = 1ull + p[1];
    break;
  case G_RANDOM_ACCESS:
    if (!in(p[0], 1, 6)) return "RandomAccessGate: bits must be 1..6";
    if (!in(p[1], 1, 1024)) return "RandomAccessGate: num_copies out of range";
    if (p[2] > 2) return "RandomAccessGate: more than 2 extra constants";
    w = (2ull + (1ull << p[0])) * p[1] + p[2] + (uint64_t)p[1] * p[0]; k = p[2];
    break;
  case G_POSEIDON: w = 135; break;
  case G_U32_ARITHMETIC:
    if (!in(p[0], 1, 1024)) return "U32ArithmeticGate: num_ops out of range";
    w = 38ull * p[0];
    break;
  case G_U32_ADD_MANY:
    if (!in(p[0], 1, 16)) return "U32AddManyGate: num_addends must be 1..16";
    if (!in(p[1], 1, 1024)) return "U32AddManyGate: num_ops out of range";
    w = ((uint64_t)p[0] + 3 + 18) * p[1];
    break;
  case G_U32_SUBTRACTION:
    if (!in(p[0], 1, 1024)) return "U32SubtractionGate: num_ops out of range";
    w = 21ull * p[0];
    break;
  case G_U32_RANGE_CHECK:
    if (!in(p[0], 1, 1024)) return "U32RangeCheckGate: num_input_limbs out of range";
    w = 17ull * p[0];
    break;
  case G_COMPARISON: {
    if (!in(p[0], 1, 64) || !in(p[1], 1, 64)) return "ComparisonGate: num_bits / num_chunks must be 1..64";
    const uint32_t cb = (p[0] + p[1] - 1) / p[1];
    if (cb > 4) return "ComparisonGate: chunk_bits > 4 unsupported";
    w = 4ull + 5ull * p[1] + cb + 1;
    break;
  }
  default: return "unsupported gate kind in blob";
  }
  if (w > W) return "gate needs more wires than the circuit has";
  if (k > gate_consts) return "gate needs more constants than the circuit has";
  *wires_used = w;
  *consts_used = k;
  return nullptr;
}

// degree of the gate's constraints as polynomials in the wire / constant columns (the library's own count from the formulas
// of gates.hpp, not the blob's `degree` field): what the half-domain evaluation of plonk.hip gate_sums_kernel relies on
uint32_t gate_degree(uint32_t kind, const uint32_t p[4]) {
  switch (kind) {
  case G_NOOP: return 0;
  case G_CONSTANT: case G_PUBLIC_INPUT: return 1;
  case G_ARITHMETIC: return 3;
  case G_BASE_SUM: return p[0] > 1 ? p[0] : 1;
  case G_RANDOM_ACCESS: return p[0] + 1;
  case G_POSEIDON: return 7;
  case G_U32_ARITHMETIC: case G_U32_ADD_MANY: case G_U32_SUBTRACTION: case G_U32_RANGE_CHECK: return 4;  // range4 of a 2-bit limb
  case G_COMPARISON: {
    const uint32_t cb = p[1] ? (p[0] + p[1] - 1) / p[1] : 32;
    return cb >= 31 ? UINT32_MAX : std::max(1u << cb, 3u);
  }
  default: return UINT32_MAX;
  }
}
uint32_t gate_num_constraints(uint32_t kind, const uint32_t p[4]) {
  switch (kind) {
  case G_NOOP: return 0;
  case G_CONSTANT: return p[0];
  case G_PUBLIC_INPUT: return 4;
  case G_ARITHMETIC: return p[0];
  case G_BASE_SUM: return 1 + p[1];
  case G_RANDOM_ACCESS: return p[1] * (p[0] + 2) + p[2];
  case G_POSEIDON: return 123;
  case G_U32_ARITHMETIC: return p[0] * 36;
  case G_U32_ADD_MANY: return p[1] * 21;
  case G_U32_SUBTRACTION: return p[0] * 19;
  case G_U32_RANGE_CHECK: return p[0] * 17;
  case G_COMPARISON: return 6 + 5 * p[1] + (p[0] + p[1] - 1) / p[1];
  }
  return 0;
}

int circuit_parse(const uint8_t *blob, size_t len, p2gpu_circuit *c, size_t *off_out, const uint8_t **cap_in) {
  if (len < 256) { set_err("blob too short"); return P2GPU_E_BLOB; }
  uint32_t h[64];
  memcpy(h, blob, sizeof h);
  if (h[0] != 0x43473250u || h[1] != 1) { set_err("bad blob magic/version"); return P2GPU_E_BLOB; }
  c->d = h[2]; c->W = h[3]; c->R = h[4]; c->NC = h[5]; c->num_selectors = h[6]; c->K = h[7]; c->QF = h[8];
  c->rate_bits = h[9]; c->cap_h = h[10]; c->pow_bits = h[11]; c->num_queries = h[12]; c->n_steps = h[13];
  for (int i = 0; i < 8; i++) c->arity[i] = h[14 + i];
  const uint32_t hasher = h[22];
  c->num_gates = h[23]; c->num_pi = h[24]; c->flags = h[25]; c->PP = h[26];
  auto fail = [&](int rc, const char *msg) {
    set_err("%s", msg);
    return rc;
  };
  if (hasher > 1) return fail(P2GPU_E_BLOB, "unsupported hasher (0 = KeccakHash<25>, 1 = PoseidonHash)");
  c->hasher = hasher;
  if (c->d < 1 || c->d > 24 || c->K < 1 || c->K > 2 || c->rate_bits < 1 || c->rate_bits > 3 || c->cap_h < c->rate_bits ||
      c->cap_h > c->rate_bits + c->d || c->n_steps > 8 || c->R > MAX_ROUTED || c->QF == 0 || c->num_gates > MAX_GATES ||
      c->num_queries > 64 || c->W < c->R || (1u << c->rate_bits) != c->QF || c->pow_bits > 32 || c->W > 4096 ||
      c->NC > 4096 || c->num_selectors > c->NC || c->num_selectors == 0 || c->num_pi > (1u << 20) || c->num_queries == 0)
    return fail(P2GPU_E_BLOB, "unsupported circuit parameters");
  c->n = (size_t)1 << c->d;
  c->N = c->n << c->rate_bits;
  c->C = 1u << c->rate_bits;
  c->nchunks = (c->R + c->QF - 1) / c->QF;
  if (c->nchunks > 16 || c->PP != c->nchunks - 1) return fail(P2GPU_E_BLOB, "unsupported circuit parameters");
  {
    // fri/reduction_strategies.rs: every step must leave at least the cap below it
    uint32_t ds = c->d;
    for (uint32_t s = 0; s < c->n_steps; s++) {
      const uint32_t ab = c->arity[s];
      if (ab < 1 || ab > MAX_ARITY_BITS || ds < ab || ds + c->rate_bits - ab < c->cap_h) return fail(P2GPU_E_BLOB, "unsupported FRI reduction arity");
      ds -= ab;
    }
  }
  size_t off = 256;
  if (len < off + (size_t)c->num_gates * 48) return fail(P2GPU_E_BLOB, "blob truncated (gate table)");
  c->max_gate_constraints = 0;
  c->gate_wires = 0;
  c->gates.clear();
  for (uint32_t i = 0; i < c->num_gates; i++) {
    uint32_t g[12];
    memcpy(g, blob + off, sizeof g);
    off += sizeof g;
    GateDesc G;
    G.kind = g[0];
    memcpy(G.p, &g[1], 16);
    G.sel_index = g[5]; G.group_start = g[6]; G.group_end = g[7]; G.num_constraints = g[8]; G.degree = g[9];
    G.num_constants = g[10]; G.pad = 0;
    if (G.kind >= G_KIND_COUNT) return fail(P2GPU_E_BLOB, "unsupported gate kind in blob");
    uint64_t wires_used = 0;
    uint32_t consts_used = 0;
    if (const char *why = gate_validate(G.kind, G.p, c->W, c->NC - c->num_selectors, &wires_used, &consts_used))
      return fail(P2GPU_E_BLOB, why);
    if (G.num_constraints != gate_num_constraints(G.kind, G.p) || G.num_constraints > MAX_GATE_CONSTRAINTS)
      return fail(P2GPU_E_BLOB, "gate constraint count mismatch");
    if (G.num_constants > c->NC - c->num_selectors || G.num_constants < consts_used || G.degree > 9)
      return fail(P2GPU_E_BLOB, "bad gate constant count / degree");
    if (G.sel_index >= c->num_selectors || G.group_end > c->num_gates || G.group_start > i || i >= G.group_end)
      return fail(P2GPU_E_BLOB, "bad selector info");
    c->max_gate_constraints = std::max(c->max_gate_constraints, G.num_constraints);
    c->gate_wires = std::max(c->gate_wires, (uint32_t)wires_used);
    c->gates.push_back(G);
  }
  *cap_in = nullptr;
  if (c->flags & 2) {
    if (len < off + ((size_t)32 << c->cap_h)) return fail(P2GPU_E_BLOB, "blob truncated (cap)");
    *cap_in = blob + off;
    off += (size_t)32 << c->cap_h;
  }
  if (len < off + 8 * (size_t)c->R) return fail(P2GPU_E_BLOB, "blob truncated (k_is)");
  c->k_is.resize(c->R);
  memcpy(c->k_is.data(), blob + off, 8 * (size_t)c->R);
  off += 8 * (size_t)c->R;
  if (c->flags & 1) {
    memset(&c->circuit_digest, 0, sizeof(dig_t));
    memcpy(c->circuit_digest.w, &h[32], c->hasher ? 32 : 25);
  }
  poseidon_round_constants_host(c->poseidon_rc);
  poseidon_device_constants(c->poseidon_rc, c->poseidon_rc_gate);
  use_hasher(c);
  *off_out = off;
  return P2GPU_OK;
}
}  // namespace p2

extern "C" {

int p2gpu_shard_assemble_cap(int world, unsigned rate_bits, unsigned cap_h, const uint8_t *gathered, uint8_t *cap_out) try {
  if (!gathered || !cap_out || world < 1 || rate_bits > 3 || cap_h < rate_bits || cap_h > 16 || ((1u << rate_bits) % (unsigned)world) != 0)
    return P2GPU_E_ARG;
  const size_t cap_per = ((size_t)1 << cap_h) >> rate_bits, total = (size_t)1 << cap_h;
  std::vector<dig_t> in(total), cap;
  memcpy(in.data(), gathered, total * sizeof(dig_t));
  shard_assemble_cap(world, rate_bits, cap_per, in.data(), cap);
  for (size_t i = 0; i < total; i++) memcpy(cap_out + 25 * i, cap[i].w, 25);
  return P2GPU_OK;
} P2GPU_CATCH

// ---- N2: the prover-side precompute of `builder.build::<C>()` that needs no GPU -----------------------
// (plonky2-backend/src/circuit_translation/mod.rs:80-82, actions/write_vk_action.rs:76): from the gate
// instances (one gate index and the gate's constants per row) and the copy constraints it derives
//   * the selector columns and selector groups (plonky2 gates/selectors.rs selector_polynomials: gates come
//     sorted by (degree, id); one column when max_degree >= max gate degree + #gates - 1, greedy groups of
//     gates[start..start+size) with size + degree(gates[start + size]) < max_degree otherwise; a row of a
//     gate outside the group holds UNUSED = 2^32 - 1),
//   * the sigma polynomials of the permutation argument (plonk/permutation_argument.rs: union-find over the
//     routed cells, every class listed row by row, each cell mapped to the next of its class; as values
//     k_is[col'] * w_n^row'), k_is[j] = g^j,
//   * the FRI reduction arities of ConstantArityBits(4, 5),
// and writes the circuit blob that p2gpu_circuit_create takes (which then commits constants and sigmas on the
// GPU and derives the circuit digest).  Pinned to the reference: the circuits recovered from its own proof
// files come back bit for bit (tests/test_build.py).
int p2gpu_build_blob(const p2gpu_build_params *bp, const p2gpu_gate_decl *gates, uint32_t num_gates, const uint32_t *row_gate,
                     const uint64_t *row_constants, const uint32_t *copies, size_t num_copies, uint8_t *blob_out,
                     size_t *blob_len) try {
  if (!bp || !gates || !row_gate || !blob_len || (num_copies && !copies)) return P2GPU_E_ARG;
  const uint32_t d = bp->degree_bits, W = bp->num_wires, R = bp->num_routed_wires, QF = bp->quotient_degree_factor;
  if (d < 1 || d > 24 || num_gates == 0 || num_gates > (uint32_t)MAX_GATES || R == 0 || R > (uint32_t)MAX_ROUTED || W < R || W > 4096 ||
      QF == 0 || bp->rate_bits < 1 || bp->rate_bits > 3 || (1u << bp->rate_bits) != QF || bp->num_challenges < 1 ||
      bp->num_challenges > 2) {
    set_err("p2gpu_build_blob: unsupported circuit parameters");
    return P2GPU_E_ARG;
  }
  const size_t n = (size_t)1 << d;
  for (uint32_t i = 1; i < num_gates; i++)
    if (gates[i].degree < gates[i - 1].degree) {
      set_err("p2gpu_build_blob: gates must come sorted by (degree, id) as in CommonCircuitData.gates");
      return P2GPU_E_ARG;
    }
  // selector groups
  const uint32_t max_degree = QF + 1;
  std::vector<uint32_t> gstart(num_gates), gend(num_gates), gsel(num_gates);
  uint32_t num_selectors = 0;
  if (gates[num_gates - 1].degree + num_gates - 1 <= max_degree) {
    num_selectors = 1;
    for (uint32_t i = 0; i < num_gates; i++) { gstart[i] = 0; gend[i] = num_gates; gsel[i] = 0; }
  } else {
    uint32_t start = 0;
    while (start < num_gates) {
      uint32_t size = 0;
      while (start + size < num_gates && size + gates[start + size].degree < max_degree) size++;
      if (size == 0) { set_err("p2gpu_build_blob: gate degree %u does not fit max_degree %u", gates[start].degree, max_degree); return P2GPU_E_ARG; }
      for (uint32_t i = start; i < start + size; i++) { gstart[i] = start; gend[i] = start + size; gsel[i] = num_selectors; }
      start += size;
      num_selectors++;
    }
  }
  uint32_t ngc = 0;
  for (uint32_t i = 0; i < num_gates; i++) ngc = std::max(ngc, gates[i].num_constants);
  if (ngc && !row_constants) return P2GPU_E_ARG;
  const uint32_t NC = num_selectors + ngc;
  std::vector<uint32_t> arity;
  for (uint32_t db = d; db > 5 && db + bp->rate_bits - 4 >= bp->cap_height; db -= 4) arity.push_back(4);
  if (arity.size() > 8) return P2GPU_E_ARG;
  const size_t need = 256 + 48 * (size_t)num_gates + 8 * ((size_t)R + (size_t)NC * n + (size_t)R * n);
  if (!blob_out || *blob_len < need) {
    const bool probe = blob_out == nullptr;
    *blob_len = need;
    if (probe) return P2GPU_OK;
    set_err("blob buffer too small: need %zu bytes", need);
    return P2GPU_E_BUFFER;
  }
  // header + gate table
  uint32_t hd[64];
  memset(hd, 0, sizeof hd);
  hd[0] = 0x43473250u; hd[1] = 1; hd[2] = d; hd[3] = W; hd[4] = R; hd[5] = NC; hd[6] = num_selectors; hd[7] = bp->num_challenges;
  hd[8] = QF; hd[9] = bp->rate_bits; hd[10] = bp->cap_height; hd[11] = bp->proof_of_work_bits; hd[12] = bp->num_query_rounds;
  hd[13] = (uint32_t)arity.size();
  for (size_t i = 0; i < arity.size(); i++) hd[14 + i] = arity[i];
  hd[22] = 0; hd[23] = num_gates; hd[24] = bp->num_public_inputs; hd[25] = 0; hd[26] = (R + QF - 1) / QF - 1;
  memcpy(blob_out, hd, sizeof hd);
  size_t off = sizeof hd;
  for (uint32_t i = 0; i < num_gates; i++) {
    uint64_t wu = 0;
    uint32_t cu = 0;
    if (gates[i].kind >= G_KIND_COUNT) { set_err("unsupported gate kind in blob"); return P2GPU_E_ARG; }
    if (const char *why = gate_validate(gates[i].kind, gates[i].p, W, ngc, &wu, &cu)) { set_err("%s", why); return P2GPU_E_ARG; }
    const uint32_t gw[12] = {gates[i].kind, gates[i].p[0], gates[i].p[1], gates[i].p[2], gates[i].p[3], gsel[i], gstart[i], gend[i],
                             gate_num_constraints(gates[i].kind, gates[i].p), gates[i].degree, gates[i].num_constants, 0};
    memcpy(blob_out + off, gw, sizeof gw);
    off += sizeof gw;
  }
  // k_is and the subgroup
  std::vector<gl_t> k_is(R), sub(n);
  k_is[0] = 1;
  for (uint32_t j = 1; j < R; j++) k_is[j] = gl_mul(k_is[j - 1], GL_GEN);
  {
    const gl_t wn = gl_root(d);
    sub[0] = 1;
    for (size_t i = 1; i < n; i++) sub[i] = gl_mul(sub[i - 1], wn);
  }
  memcpy(blob_out + off, k_is.data(), 8 * (size_t)R);
  off += 8 * (size_t)R;
  // constants: selector columns, then the gate constants as given
  gl_t *consts = (gl_t *)(blob_out + off);  // (8-byte aligned: 256 + 48 g + 8 R)
  for (size_t r = 0; r < n; r++) {
    const uint32_t gi = row_gate[r];
    if (gi >= num_gates) { set_err("row %zu holds gate index %u of %u", r, gi, num_gates); return P2GPU_E_ARG; }
    for (uint32_t s = 0; s < num_selectors; s++) consts[(size_t)s * n + r] = (num_selectors == 1 || s == gsel[gi]) ? gi : 0xFFFFFFFFull;
  }
  for (uint32_t k = 0; k < ngc; k++)
    for (size_t r = 0; r < n; r++) {
      const uint64_t v = row_constants[(size_t)k * n + r];
      if (v >= GL_P) { set_err("gate constant (%u, %zu) is not canonical", k, r); return P2GPU_E_ARG; }
      consts[(size_t)(num_selectors + k) * n + r] = v;
    }
  off += 8 * (size_t)NC * n;
  // sigma from the copy constraints
  gl_t *sig = (gl_t *)(blob_out + off);
  {
    const size_t tot = (size_t)R * n;
    std::vector<uint32_t> parent(tot);
    for (size_t x = 0; x < tot; x++) parent[x] = (uint32_t)x;
    auto find = [&](uint32_t x) {
      while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
      return x;
    };
    for (size_t e = 0; e < num_copies; e++) {
      const uint32_t ra = copies[4 * e], ca = copies[4 * e + 1], rb = copies[4 * e + 2], cb = copies[4 * e + 3];
      if (ra >= n || rb >= n || ca >= R || cb >= R) { set_err("copy constraint %zu names a cell outside the routed wires", e); return P2GPU_E_ARG; }
      const uint32_t x = find((uint32_t)((size_t)ca * n + ra)), y = find((uint32_t)((size_t)cb * n + rb));
      if (x != y) parent[std::max(x, y)] = std::min(x, y);
    }
    std::vector<uint32_t> first(tot, UINT32_MAX), last(tot, UINT32_MAX), next(tot);
    for (size_t row = 0; row < n; row++)
      for (uint32_t col = 0; col < R; col++) {
        const uint32_t x = (uint32_t)((size_t)col * n + row), rt = find(x);
        if (first[rt] == UINT32_MAX) first[rt] = x;
        else next[last[rt]] = x;
        last[rt] = x;
      }
    for (size_t x = 0; x < tot; x++) {
      const uint32_t rt = find((uint32_t)x);
      if (last[rt] == x) next[x] = first[rt];
    }
    for (size_t x = 0; x < tot; x++) sig[x] = gl_mul(k_is[next[x] / n], sub[next[x] % n]);
  }
  off += 8 * (size_t)R * n;
  *blob_len = off;
  return P2GPU_OK;
} P2GPU_CATCH

const char *p2gpu_last_error(void) { return g_err.c_str(); }

size_t p2gpu_proof_size_bound(const p2gpu_circuit *c) {
  if (!c) return 0;
  const size_t ncap = (size_t)1 << c->cap_h;
  const size_t ncs = c->NC + c->R, nzp = c->K * (1 + c->PP), nq = c->K * c->QF;
  const size_t hb = c->hasher ? 32 : 25;
  size_t sz = 3 * ncap * hb + 16 * (ncs + c->W + nzp + nq + c->K) + c->n_steps * ncap * hb;
  size_t per_q = 8 * (ncs + c->W + nzp + nq) + 4 * (1 + hb * (size_t)(c->d + c->rate_bits));
  for (uint32_t s = 0; s < c->n_steps; s++) per_q += (16u << c->arity[s]) + 1 + hb * (size_t)(c->d + c->rate_bits);
  size_t n_final = c->n;
  for (uint32_t s = 0; s < c->n_steps; s++) n_final >>= c->arity[s];
  sz += per_q * c->num_queries + 16 * n_final + 8 + 8 * c->num_pi + 64;
  return sz;
}

void p2gpu_circuit_destroy(p2gpu_circuit *c) {
  if (!c) return;
  if (c->device >= 0 && g_circuit_release) g_circuit_release(c);  // a verifier-only handle owns nothing on a device
  delete c;
}

int p2gpu_circuit_cap(const p2gpu_circuit *c, uint8_t *out) {
  if (!c || !out) return P2GPU_E_ARG;
  const size_t hb = c->hasher ? 32 : 25;
  for (size_t i = 0; i < c->cs.cap.size(); i++) memcpy(out + hb * i, c->cs.cap[i].w, hb);
  return P2GPU_OK;
}
int p2gpu_circuit_device(const p2gpu_circuit *c) { return c ? c->device : P2GPU_E_ARG; }
int p2gpu_circuit_hash_bytes(const p2gpu_circuit *c) { return c ? (c->hasher ? 32 : 25) : P2GPU_E_ARG; }
int p2gpu_circuit_digest(const p2gpu_circuit *c, uint8_t *out) {
  if (!c || !out) return P2GPU_E_ARG;
  memcpy(out, c->circuit_digest.w, c->hasher ? 32 : 25);
  return P2GPU_OK;
}

}  // extern "C"
