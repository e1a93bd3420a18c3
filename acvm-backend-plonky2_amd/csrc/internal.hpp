// internal.hpp -- declarations shared by the HIP translation units of libp2gpu.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "gl.hpp"
#include "keccak.hpp"
#include "gates.hpp"

namespace p2 {

constexpr int MAX_GATES = 64;
constexpr uint32_t MAX_GATE_CONSTRAINTS = 4096;  // per gate; every supported kind at W <= 4096 stays below
constexpr uint32_t MAX_ARITY_BITS = 4;  // FRI reduction arity 2^4 at most: one limit for circuit_parse, prover and verifier
constexpr int MAX_CHALLENGES = 2;
constexpr int MAX_ROUTED = 128;

// ---- per-launch profiling hook (HIP events on the launch stream, see prover.hip) ----
struct Prof {
  virtual void begin(const char *kernel, double algorithmic_bytes) = 0;
  virtual void end() = 0;
  virtual ~Prof() {}
};
extern thread_local Prof *g_prof;
struct ProfScope {
  ProfScope(const char *k, double bytes) { if (g_prof) g_prof->begin(k, bytes); }
  ~ProfScope() { if (g_prof) g_prof->end(); }
};

// ---- ntt.hip ----
constexpr int NTT_MAX_ROUNDS = 6;  // rounds of <= 3 layers in one pass: 12 layers are 4, a 2^13 tile's 13 layers are 5
struct NttPass {
  uint32_t s, a, tb, nrounds;
  uint32_t r[NTT_MAX_ROUNDS], tw_off[NTT_MAX_ROUNDS];
  // DIT passes on full 2^12 tiles whose first round has three layers also have a "direct" form (ntt.hip ntt_dit_*_kernel):
  // the first round straight from global memory (strided passes), the last one straight to it, its twiddles shifts only.
  // ftw_off: offset in the plan's table of the folded twiddles of the round before the last (see fold_table_kernel)
  bool direct = false;
  uint32_t ftw_off = 0;
  uint32_t ftw2_off = 0;  // head pass: the folded table in ntt_dit_head2_kernel's lane order
};
struct NttPlan {
  uint32_t d = 0;
  int dit = 0;           // 0: DIF natural in -> bit-reversed out; 1: DIT bit-reversed in -> natural out
  bool inverse = false;  // roots w^-1 (the 1/n factor is the caller's `post`)
  std::vector<NttPass> passes;
  gl_t *ptw = nullptr;   // device: packed per-round twiddle tables
  size_t table_len = 0;
};
// which global cosets a sharded launch covers: local index z <-> global coset first + z * stride
struct CosetMap {
  uint32_t first = 0, stride = 1;
};
NttPlan *ntt_plan_create(hipStream_t st, uint32_t d, int dit, bool inverse);
void ntt_plan_destroy(NttPlan *p);
// src [cols][n] (or [cosets][cols][n] when src_per_coset), dst [cosets][cols][n]; stride_cols != 0: the
// launch covers `cols` columns of a batch that has stride_cols columns per coset (chunked pipelines).
// scale (DIT only): [cosets][n] multiplied into the input; post: multiplied into the output.
// Structured columns of a batch (the unused wires of a witness): a transform is linear, so the zero column maps to
// zeros and v * (unit column of a fixed row) maps to v * (the transform of that unit column, kept by the handle).
constexpr uint32_t MAX_SPARSE_ROWS = 4;  // the PublicInputGate row + up to three PoseidonGate rows
struct SparseRows {
  uint32_t row[MAX_SPARSE_ROWS] = {UINT32_MAX, UINT32_MAX, UINT32_MAX, UINT32_MAX};
  uint32_t count = 0;
};
struct ColHints {
  const uint32_t *cls = nullptr;    // [cols]: 0 zero, 1 = val[c] * unit column of the fixed row, 2 dense,
                                    // 3 = zero outside the handle's special rows: sum_s val[s][c] * unit column of row s
                                    // (written by the fill kernel like class 1, but everybody else treats it as dense)
  const uint32_t *clean = nullptr;  // [cols], optional: 1 = dst already holds the zeros of class 0 column c
  const gl_t *val = nullptr;        // [cols]: scalar of the class 1 columns; class 3: val[s * val_stride + c] for special row s
  const gl_t *basis = nullptr;      // this transform of the unit column: [n], or [all cosets][n] (indexed by GLOBAL coset);
                                    // special row s at basis + s * basis_stride
  bool basis_per_coset = false;
  uint32_t nrows = 1, val_stride = 0;  // INVARIANT the launchers rely on (ntt_batch's fill grid, the profile's byte counts): class 3
                                       // exists only when nrows > 1 (column_class_kernel: with one special row a column that is zero
                                       // elsewhere is class 1) -- a classifier that hands out class 3 with nrows == 1 must also
                                       // widen the fill grid beyond virt_first
  size_t basis_stride = 0;
  uint32_t virt_first = UINT32_MAX; // columns >= virt_first (relative to cls) of class 0 / 1 are not stored at all: their
                                    // consumers recompute val[c] * basis on the fly (VirtCols)
  uint32_t dense_hint = 0;          // profile accounting only: how many of the columns are dense (the handle's previous proof;
                                    // 0 = not known -- the transforms' bytes are then counted for every column)
  bool fill_only = false;           // write the structured columns and return: the dense ones are somebody else's (a rank of a
                                    // sharded proof transforms only its block of the dense columns, prover.hip shard_intt)
};
void ntt_batch(hipStream_t st, const NttPlan *plan, const gl_t *src, gl_t *dst, uint32_t cols, uint32_t cosets,
               const gl_t *scale, gl_t post, bool src_per_coset, CosetMap cm = CosetMap(), uint32_t stride_cols = 0,
               const ColHints *hints = nullptr);
// clean[c] = 1: dst already holds the zeros of zero column c.  `after` = false, enqueued before the transforms:
// non-zero columns lose the mark; `after` = true, enqueued behind them: zero columns gain it.
void column_clean_update(hipStream_t st, const uint32_t *cls, uint32_t cols, uint32_t *clean, bool after);
// class of every column of vals [cols][n] (see ColHints) with respect to the special rows; scalar[s * sstride + c] =
// the column's value in special row s
void column_flags(hipStream_t st, const gl_t *vals, uint32_t cols, uint32_t d, const SparseRows &rows, uint32_t *flags,
                  gl_t *scalar, uint32_t sstride);
void fill_powers(hipStream_t st, gl_t *out, gl_t root, uint32_t count);
void fill_coset_scale(hipStream_t st, gl_t *out, gl_t shift, gl_t wN, uint32_t d, uint32_t cosets, gl_t mult);
void bitrev_cols(hipStream_t st, const gl_t *in, gl_t *out, uint32_t d, uint32_t cols);
// the device forms of the field primitives against the portable code; bad[8] mismatch counters
void field_selftest(hipStream_t st, const uint64_t *a, const uint64_t *b, uint32_t n, unsigned long long *bad);

// ---- merkle.hip ----
// Wire columns whose LDE is never written to memory: a class 0 / class 1 column (ColHints) that no gate and no
// permutation check reads is, on every LDE row, val[c] * basis[coset][k] -- one modular product in the leaf hash
// (and in the query gather) instead of 8 bytes written by the fill kernel and 8 bytes read back by the hash.
struct VirtCols {
  const uint32_t *cls = nullptr;  // [cols] classes of the proof in progress; nullptr: nothing is virtual
  const gl_t *val = nullptr;      // [cols] scalars of the class 1 columns
  const gl_t *basis = nullptr;    // LDE of the unit column, [all cosets][n] indexed by GLOBAL coset (nullptr: no class 1 exists)
  uint32_t first = UINT32_MAX;    // columns >= first whose class is not 2 are virtual
  uint32_t coset_first = 0, coset_stride = 1;
};
// leaf digests of an LDE batch: lde [cosets][cols][n] -> dig [cosets][n]
// prc: the handle's Poseidon round constants in device memory selects PoseidonHash; nullptr = KeccakHash<25>
// lvl1 / lvl2 (optional): storage of the tree levels with n/2 and n/4 nodes per coset; the return value says how many levels
// above the leaves the launch has also built (0 or 2: merkle.hip hash_lde_leaves_kf_kernel<V, 2>)
uint32_t hash_lde_leaves(hipStream_t st, const gl_t *lde, uint32_t cols, uint32_t d, uint32_t cosets, dig_t *dig, const gl_t *prc = nullptr,
                         const VirtCols *virt = nullptr, dig_t *lvl1 = nullptr, dig_t *lvl2 = nullptr);
// row-major rows (stage-level operator)
void hash_rows(hipStream_t st, const gl_t *rows, size_t n_rows, uint32_t row_len, dig_t *dig);
// FRI step leaves: vals [cosets][2][npc] (ext coordinates), leaf = 16 ext values; dig [cosets][npc/16]
void hash_fri_leaves(hipStream_t st, const gl_t *vals, uint32_t lg_npc, uint32_t cosets, uint32_t arity_bits,
                     dig_t *dig, const gl_t *prc = nullptr);
// one tree level: in [cosets][m] -> out [cosets][m/2], out[c][k] = H(in[c][k], in[c][k + m/2])
// incremental variant: absorb the rate blocks [blk0, blk0 + nblk) (17 columns each) into the sponge
// states kept in `state` ([cosets][25][n]); `last` also absorbs the tail + padding and writes digests
void hash_lde_absorb(hipStream_t st, const gl_t *lde, uint32_t cols, uint32_t d, uint32_t cosets, uint32_t blk0,
                     uint32_t nblk, bool first, bool last, uint64_t *state, dig_t *dig, const VirtCols *virt = nullptr);
void merkle_level(hipStream_t st, const dig_t *in, dig_t *out, uint32_t cosets, uint32_t m, const gl_t *prc = nullptr);
// every level below one with m <= 4096 nodes per coset, down to cap_per nodes per coset, in one launch
size_t merkle_tail_from(const gl_t *prc);
// host_mirror (optional, page-locked, [cosets][cap_per]): returns true when the final level was also written there
bool merkle_tail(hipStream_t st, dig_t *lvl, uint32_t cosets, uint32_t m, uint32_t cap_per, const gl_t *prc = nullptr,
                 dig_t *host_mirror = nullptr);

// ---- plonk.hip ----
struct ZsArgs {
  const gl_t *wires;   // [W][n] values
  const gl_t *sigmas;  // [R][n] values
  const gl_t *k_is;    // [R]
  const gl_t *sub_tw;  // w_n^i, i < n/2 (forward table, shift tw_shift)
  uint32_t tw_shift;
  uint32_t d, R, QF, nchunks, K;
  gl_t betas[MAX_CHALLENGES], gammas[MAX_CHALLENGES];
  gl_t *cp;      // chunk quotients (scratch) + the row products behind them: K * (nchunks + 1) columns, column K * nchunks + c =
                 // the row products of challenge c.  Layout [n >> sb][columns][1 << sb] (zs_idx): sb = d is the plain [columns][n];
                 // a rank of a row-sharded computation (knob shard_zs) writes its rows as ONE contiguous block, sb = d - log2(ranks)
  uint32_t sb;   // slice bits of that layout
  uint32_t row0, rows;  // the rows this launch of the chunk kernel computes (all: 0, n)
  gl_t *zp;      // out [K*(1+PP)][n]
};
void zs_partial_products(hipStream_t st, const ZsArgs &a, gl_t *scan_tmp);
// the two halves of it: chunk quotients + row products of rows [a.row0, a.row0 + a.rows) | scan + Z / partial products of all rows
void zs_chunks(hipStream_t st, const ZsArgs &a);
void zs_scan_finish(hipStream_t st, const ZsArgs &a, gl_t *scan_tmp);

struct QuotArgs {
  const gl_t *cs_lde;    // [cosets][NC+R][n]
  const gl_t *wires_lde; // [cosets][W][n]
  const gl_t *zp_lde;    // [cosets][K*(1+PP)][n]
  const gl_t *k_is;      // [R]
  const gl_t *tw;        // w_n^i forward table
  const gl_t *apow;      // [K][nterms] alpha powers
  const GateDesc *gates; // device copy
  const GateDesc *host_gates;  // the same table on the host (launch-time decisions)
  gl_t *out;             // [K][cosets][n]
  uint32_t tw_shift, d, rate_bits, W, R, NC, num_selectors, K, QF, nchunks, PP, num_gates, nterms, has_poseidon;
  uint32_t gate_groups;  // 1, or 4: gates split over the four waves of a 64-row block (GateDesc.pad: gates.hpp gate_group)
  uint32_t gate_groups_half;  // the same for the main kernel when the half-domain gates are only looked up (use_half)
  uint32_t use_half;     // gates of degree <= 4 with a slot: sums from hsum instead of evaluating them (plonk.hip gate_sums_kernel)
  uint32_t nsk;          // half-domain slots x K
  gl_t *hsum;            // [2][4][nsk][n]: folded sums on the even cosets (parity 0) and the odd ones (parity 1)
  uint32_t coset_first, coset_stride, ncosets;  // sharding: grid.y = local coset z, global r = first + z * stride;
                                                // cs_lde/qconst are indexed by r, wires/zp/out by z
  gl_t betas[MAX_CHALLENGES], gammas[MAX_CHALLENGES];
  gl_t pi_hash[4];
  const gl_t *qconst;   // device [3][8]: coset shift g w_N^r | Z_H = g^n w_8^r - 1 | 1 / Z_H (g = GL_GEN)
                        // (in memory, not kernel arguments: they are indexed by blockIdx.y)
  gl_t n_inv;           // 1/n
  const gl_t *l0;       // [all cosets][n] (global coset index): L_0(x) = Z_H(x) / (n (x - 1)), fill_l0_table
};
void quotient_eval(hipStream_t st, const QuotArgs &a);
// folded constraint sums of the half-domain gates on the even cosets -> a.hsum parity 0 (groups: 1 or 4 waves per row tile)
void gate_sums_eval(hipStream_t st, const QuotArgs &a, uint32_t groups);
// in / out [4][cols][n]: per-coset interpolants of the even cosets -> coefficient arrays of the odd cosets (F: CircuitState::half_cross)
void gate_sums_cross(hipStream_t st, const gl_t *in, const gl_t *inv_scale, gl_t *out, uint32_t d, uint32_t cols, const gl_t F[16]);
// out[r][k] = qconst[8 + r] * n_inv / (qconst[r] * w_n^k - 1): per circuit, so that the quotient kernel loads L_0
// instead of inverting x - 1 on every row (~125 modmuls of the ~1 200 a row of the permutation argument costs)
void fill_l0_table(hipStream_t st, const gl_t *qconst, const gl_t *tw, uint32_t tw_shift, uint32_t d, uint32_t cosets, gl_t n_inv, gl_t *out);
// upload the Poseidon round constants used by PoseidonGate evaluation; 0 = ok
int poseidon_upload_constants();
// cross-coset inverse butterflies.  in [K][C][n]: per-coset inverse transforms of
// the quotient values (bit-reversed coefficient storage, C = 2^rate_bits);
// inv_scale [C][n] = (1/s_r)^(bitrev p); out [K*C][n]: chunk polynomials
// Q_m = 7^(-n m) / C * sum_r w_C^(-r m) P_r.
void quotient_chunks(hipStream_t st, const gl_t *in, const gl_t *inv_scale, gl_t *out, uint32_t d, uint32_t K,
                     uint32_t rate_bits, gl_t w_inv, gl_t gn_inv, gl_t rate_inv, uint32_t world = 1);

// ---- witness.hip ----
// row-local generators: derive every gate-internal wire from the gate's input wires, in place
void fill_witness(hipStream_t st, gl_t *wires, const uint8_t *row_gate, const GateDesc *gates, const gl_t *gconsts,
                  const gl_t *prc, uint32_t d, uint32_t ngc, uint32_t num_wires);

// ---- fri.hip ----
// pw[p] = base^(bitrev_d(p)) over the extension: out [2][n]
// out[p] = base^(bitrev_d(p)) (c0 plane, then c1 plane), for two bases in one launch
void ext_powers_bitrev2(hipStream_t st, ext_t base0, ext_t base1, uint32_t d, gl_t *out0, gl_t *out1);
// partial dot products: for each column c of coeffs [cols][n]: sum_p coeffs[c][p] * pw[p]; parts per column
// hints (optional): class 0 columns open to zero; class 1 columns to val[c] times the unit column's partial sums
// (basis_partial [parts][2], left by an earlier eval_columns launch over that one column)
void eval_columns(hipStream_t st, const gl_t *coeffs, uint32_t cols, uint32_t d, const gl_t *pw, uint32_t parts,
                  gl_t *partial /* [cols][parts][2] */, const ColHints *hints = nullptr, const gl_t *basis_partial = nullptr);
// the same for up to six batches of columns in one launch; segment `hinted` (UINT32_MAX: none) carries the column classes
struct EvalSegs {
  struct Seg {
    const gl_t *coeffs;  // [cols][n]
    const gl_t *pw;      // powers of the point, [2][n]
    gl_t *partial;       // [cols][parts][2]
    uint32_t cols;
  } seg[6];
  uint32_t count = 0, hinted = UINT32_MAX;
  const uint32_t *cls = nullptr;
  const gl_t *val = nullptr, *basis_partial = nullptr;
};
// (col0, ncols): the slice of the concatenated columns this launch evaluates (a rank's share of a sharded proof)
void eval_columns_multi(hipStream_t st, const EvalSegs &S, uint32_t d, uint32_t parts, uint32_t col0 = 0, uint32_t ncols = UINT32_MAX);
// acc[2][n] (+)= sum_j apow[j0 + j] * coeffs[j][p]
void reduce_columns(hipStream_t st, const gl_t *coeffs, uint32_t cols, uint32_t d, const gl_t *apow /*[.][2]*/,
                    uint32_t j0, gl_t *acc, bool accumulate, const uint32_t *nzlist = nullptr, const gl_t *basis = nullptr,
                    const gl_t *fold = nullptr);
// nzlist: list[0] = count, list[1..] = indices of the dense (class 2) columns; fold[2]: the class 1 columns' joint
// coefficient of the unit column's polynomial `basis` in the reduction
void compact_nonzero(hipStream_t st, const uint32_t *flags, uint32_t cols, uint32_t *list);
// out[p] = sum_q parts[q * len + p]: the ranks' partial sums of a column-sharded batch reduction
void sum_parts(hipStream_t st, const gl_t *parts, uint32_t nparts, size_t len, gl_t *out);
void class1_fold(hipStream_t st, const ColHints &h, uint32_t cols, const gl_t *apow, uint32_t j0, gl_t *fold);
// final[k] = aK * (F0[k] - f0z) / (x_k - zeta) + (F1[k] - f1z) / (x_k - gzeta), x_k = w_n^k
void fri_quotient_values(hipStream_t st, const gl_t *F0, const gl_t *F1, uint32_t d, const gl_t *tw, uint32_t tw_shift,
                         ext_t zeta, ext_t gzeta, ext_t f0z, ext_t f1z, ext_t aK, gl_t *out /*[2][n]*/);
// coefficient fold in bit-reversed storage: out[q] = sum_t beta^t in[bitrev4(t) * n/16 + q]
void fri_fold(hipStream_t st, const gl_t *in /*[2][n]*/, uint32_t d, uint32_t arity_bits, ext_t beta, gl_t *out);
// proof of work: smallest w in [base, base + count) with leading zeros; result via atomicMin
void pow_search(hipStream_t st, const gl_t state[12], uint32_t pos, uint32_t pow_bits, uint64_t base, uint64_t count,
                unsigned long long *result, const gl_t *prc = nullptr);
// gather: out[i] = *(const u64 *)addr[i] (absolute device addresses; address 0 -> value 0)
void gather_u64(hipStream_t st, const uint64_t *addr, uint32_t count, gl_t *out);

}  // namespace p2
