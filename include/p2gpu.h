/*
 * p2gpu.h -- C ABI of libp2gpu.so: the MI355X (gfx950) implementation of the
 * `prove` hot path of eryxcoop/acvm-backend-plonky2.
 *
 * What each entry point replaces in the reference (there is no FFI today; the
 * seam is a Rust method call into the plonky2 crate):
 *
 *   p2gpu_circuit_create   <- the prover-side part of `builder.build::<C>()`
 *                             plonky2-backend/src/circuit_translation/mod.rs:80-82
 *                             (constants/sigmas LDE + Merkle tree, root tables,
 *                             circuit digest), once per circuit
 *   p2gpu_prove            <- `circuit_data.prove(witnesses).unwrap()`
 *                             plonky2-backend/src/actions/prove_action.rs:91-97,
 *                             after witness generation (input = full wire matrix)
 *   p2gpu_prove_dev        <- same, wire matrix already resident in HBM
 *   p2gpu_prove_sparse     <- same as p2gpu_prove for a witness whose unused wires are given as one value each
 *   p2gpu_fill_witness / p2gpu_prove_routed
 *                          <- the row-local tail of `generate_partial_witness`: the gates' own
 *                             SimpleGenerators (e.g. arithmetic_u32.rs:376-426)
 *   p2gpu_last_error       <- the `anyhow::Error` text the reference unwraps
 *   p2gpu_ifft_batch / p2gpu_lde_batch / p2gpu_commit_values
 *                          <- plonky2 PolynomialValues::ifft,
 *                             PolynomialCoeffs::lde+coset_fft,
 *                             PolynomialBatch::from_values (stage-level operators)
 *
 * Conventions: plain C, little-endian, field elements are canonical u64 < p =
 * 2^64 - 2^32 + 1.  Return 0 = ok, < 0 = error (see P2GPU_E_*); no exception or
 * abort crosses the ABI.  The caller owns every host buffer; the library owns
 * device memory until p2gpu_circuit_destroy.  One in-flight prove per circuit
 * handle; distinct handles may be used from distinct threads.
 *
 * ---------------------------------------------------------------------------
 * Circuit blob (version 1), written once per circuit by the Rust side from
 * CommonCircuitData + ProverOnlyCircuitData:
 *
 *   u32 header[64]:
 *     [0] magic 0x43473250 "P2GC"   [1] version = 1      [2] degree_bits d
 *     [3] num_wires                 [4] num_routed_wires  [5] num_constants (selector + gate-constant columns)
 *     [6] num_selectors             [7] num_challenges    [8] quotient_degree_factor
 *     [9] rate_bits                 [10] cap_height       [11] proof_of_work_bits
 *     [12] num_query_rounds         [13] number of FRI reduction steps
 *     [14..21] reduction_arity_bits [22] hasher: 0 = KeccakHash<25> (the reference's KeccakGoldilocksConfig),
 *                                        1 = PoseidonHash (PoseidonGoldilocksConfig: Poseidon Merkle trees,
 *                                        challenger and circuit digest; digests are 4 field elements = 32 bytes)
 *     [23] num_gates                [24] num_public_inputs
 *     [25] flags: bit0 = circuit_digest present, bit1 = constants_sigmas_cap present
 *     [26] num_partial_products     [32..39] circuit_digest (25 bytes used)
 *   u32 gate[num_gates][12]  in `common.gates` order:
 *     kind, p0, p1, p2, p3, selector_index, group_start, group_end,
 *     num_constraints, degree, num_constants, reserved
 *   (flag bit1) u8 cap[2^cap_height][32]   expected constants_sigmas cap (25 bytes used)
 *   u64 k_is[num_routed_wires]
 *   u64 constants[num_constants][n]        values over the subgroup, column-major
 *   u64 sigmas[num_routed_wires][n]
 *
 * Gate kinds (closed registry, plonky2-backend/src/actions/write_vk_action.rs:35-62):
 *   0 Noop  1 Constant{p0=num_consts}  2 PublicInput  3 Arithmetic{p0=num_ops}
 *   4 BaseSum{p0=B,p1=num_limbs}  5 RandomAccess{p0=bits,p1=copies,p2=extra_consts}
 *   6 Poseidon (PoseidonGate, width 12)  7 U32Arithmetic{p0=num_ops}
 *   8 U32AddMany{p0=num_addends,p1=num_ops}  9 U32Subtraction{p0=num_ops}
 *   10 U32RangeCheck{p0=num_input_limbs}  11 Comparison{p0=num_bits,p1=num_chunks}
 *
 * Proof bytes: plonky2's uncompressed ProofWithPublicInputs::to_bytes layout
 * (SURVEY.md C.11).  Compression (prove_action.rs:75-78) stays in Rust.
 */
#ifndef P2GPU_H
#define P2GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2GPU_OK 0
#define P2GPU_E_BLOB -1               /* malformed / unsupported circuit blob */
#define P2GPU_E_BUFFER -2             /* output buffer too small (proof_len holds the size needed) */
#define P2GPU_E_DEVICE -3             /* HIP error; see p2gpu_last_error */
#define P2GPU_E_OPENING_IN_SUBGROUP -4 /* zeta^n == 1 ("Opening point is in the subgroup.") */
#define P2GPU_E_UNSATISFIED -5        /* witness does not satisfy the circuit (self-check failed) */
#define P2GPU_E_CAP_MISMATCH -6       /* constants_sigmas cap differs from the one in the blob */
#define P2GPU_E_ARG -7
#define P2GPU_E_NOT_INIT -8
#define P2GPU_E_VERIFY -9             /* p2gpu_verify: the proof is not valid for this circuit */

typedef struct p2gpu_circuit p2gpu_circuit;

/* per-phase device time of the last proof (milliseconds, HIP events), phase
 * names follow plonky2's TimingTree labels */
typedef struct {
  double wires_commit_ms;    /* "compute wires commitment" */
  double zs_commit_ms;       /* "compute partial products" + commit */
  double quotient_ms;        /* "compute quotient polys" + commit */
  double openings_ms;        /* "construct the opening set" */
  double fri_ms;             /* "compute opening proofs" */
  double total_ms;           /* entry -> proof bytes on host */
  double h2d_ms;             /* wire matrix upload (0 for p2gpu_prove_dev) */
  uint64_t pow_witness;
} p2gpu_timings;

/* Select the device(s) of this process.  One id (or NULL = {device 0}): one process per GPU -- replicas, or the ranks of a
 * proof sharded over processes (p2gpu_circuit_set_shard*).  Several ids (at most 8, repeats allowed): THIS process drives
 * them all; every handle created afterwards is a device group and each prove call is ONE proof coset-sharded over the
 * group from threads inside the call (the shape `circuit_data.prove(pw)`, prove_action.rs:96, can call).  Out-of-range
 * ids or more than 8: P2GPU_E_ARG. */
int p2gpu_init(const int *device_ids, int n_devices);
/* Page-locked ("pinned") host memory for the wire matrix handed to p2gpu_prove / _routed / _sparse -- what the caller
 * would otherwise hold in a Vec<F> (the witness the reference builds before prove_action.rs:96).  Uploads from it are
 * direct DMA and do not occupy the calling thread; any other host pointer still works (the HIP runtime stages it).
 * NULL on failure (p2gpu_last_error). */
void *p2gpu_host_alloc(size_t bytes);
void p2gpu_host_free(void *p);

int p2gpu_circuit_create(const uint8_t *blob, size_t len, p2gpu_circuit **out);
/* The same on ONE named device of the p2gpu_init list -- a plain handle even when the list has several ids (then
 * p2gpu_circuit_create makes device groups): replicas across the GPUs of a node from one process, one host thread per handle.
 * (Counterpart: the same `circuit_data.prove` call, prove_action.rs:96, issued once per device.) */
int p2gpu_circuit_create_on(const uint8_t *blob, size_t len, int device_id, p2gpu_circuit **out);
void p2gpu_circuit_destroy(p2gpu_circuit *c);
/* bytes of one digest of this circuit's hasher: 25 (KeccakHash<25>) or 32 (PoseidonHash) */
int p2gpu_circuit_hash_bytes(const p2gpu_circuit *c);
/* 2^cap_height x hash_bytes */
int p2gpu_circuit_cap(const p2gpu_circuit *c, uint8_t *out);
/* hash_bytes */
int p2gpu_circuit_digest(const p2gpu_circuit *c, uint8_t *out);
/* HIP device index the handle lives on (< 0: verifier-only handle / bad argument) */
int p2gpu_circuit_device(const p2gpu_circuit *c);
/* upper bound of the proof size in bytes for this circuit */
size_t p2gpu_proof_size_bound(const p2gpu_circuit *c);

/* wires: host pointer, [num_wires][n] column-major.  proof_len: in = capacity,
 * out = bytes used.  pow_hint: UINT64_MAX = grind for the minimum witness. */
int p2gpu_prove(p2gpu_circuit *c, const uint64_t *wires, const uint64_t *public_inputs, uint32_t n_pi,
                uint8_t *proof_out, size_t *proof_len, p2gpu_timings *opt_timings);
/* same, `wires_dev` is a device pointer (e.g. torch tensor data_ptr) on the
 * circuit's device; it is only read. */
int p2gpu_prove_dev(p2gpu_circuit *c, const uint64_t *wires_dev, const uint64_t *public_inputs, uint32_t n_pi,
                    uint8_t *proof_out, size_t *proof_len, p2gpu_timings *opt_timings);
/* Row-local witness generators on the GPU (the reference's SimpleGenerator
 * impls: arithmetic_u32.rs:376-426, add_many_u32.rs:329-378, subtraction_u32.rs:298-343,
 * range_check_u32.rs:198-220, comparison.rs:439-537 + the stock gates'): given a device wire
 * matrix whose gate INPUT wires are set (i.e. after copy-constraint propagation), derive every
 * other wire of each row in place. */
int p2gpu_fill_witness(p2gpu_circuit *c, uint64_t *wires_dev);
/* prove from the ROUTED columns only (host, [num_routed_wires][n]): the non-routed columns are
 * all gate-internal, so they are filled on the GPU instead of crossing PCIe (80 of 234 columns). */
int p2gpu_prove_routed(p2gpu_circuit *c, const uint64_t *routed, const uint64_t *public_inputs, uint32_t n_pi,
                       uint8_t *proof_out, size_t *proof_len, p2gpu_timings *opt_timings);
/* prove from the first `ncols` wire columns (host, [ncols][n]) plus ONE value for each of the others: column
 * j >= ncols is zero in every row but `row`, where it holds tail[j - ncols].  That is what plonky2 leaves in the
 * wires no gate of a circuit uses (circuit_builder.rs randomize_unused_pi_wires: one random value in the
 * PublicInputGate row; 154 of the 234 wires of wide_ecc_config in circuits without ECC gates,
 * circuit_translation/mod.rs:69), so a caller that knows its circuit ships 84 MB instead of 245 MB at 2^17
 * gates.  The columns are written in HBM and the proof is byte-identical to p2gpu_prove on the full matrix --
 * any ncols <= num_wires and any row work (the library classifies the columns it is given, it does not trust
 * the split); tail may be NULL when ncols == num_wires. */
int p2gpu_prove_sparse(p2gpu_circuit *c, const uint64_t *wires, uint32_t ncols, const uint64_t *tail, uint32_t row,
                       const uint64_t *public_inputs, uint32_t n_pi, uint8_t *proof_out, size_t *proof_len,
                       p2gpu_timings *opt_timings);

/* ---- N2: prover-side precompute of `builder.build::<C>()` (host code; circuit_translation/mod.rs:80-82) ----
 * From the gate instances and the copy constraints: selector columns + groups (gates/selectors.rs), the
 * sigma polynomials (plonk/permutation_argument.rs WirePartition), k_is and the FRI arities; output is the
 * circuit blob for p2gpu_circuit_create, which does the GPU part (constants/sigmas commitment, digest).
 *   gates:  in CommonCircuitData.gates order, i.e. sorted by (degree, id); kind / p as in the blob
 *   row_gate[n]: index into `gates` of the gate instance on each row
 *   row_constants[max num_constants][n]: the gate constants of each row, column-major (NULL if no gate has any)
 *   copies[num_copies][4]: (row_a, col_a, row_b, col_b) pairs of routed cells that must be equal
 * blob_out == NULL: only report the size in *blob_len. */
typedef struct {
  uint32_t degree_bits, num_wires, num_routed_wires, num_challenges, quotient_degree_factor, rate_bits, cap_height,
      proof_of_work_bits, num_query_rounds, num_public_inputs;
} p2gpu_build_params;
typedef struct {
  uint32_t kind, p[4], degree, num_constants;
} p2gpu_gate_decl;
int p2gpu_build_blob(const p2gpu_build_params *params, const p2gpu_gate_decl *gates, uint32_t num_gates, const uint32_t *row_gate,
                     const uint64_t *row_constants, const uint32_t *copies, size_t num_copies, uint8_t *blob_out,
                     size_t *blob_len);

/* ---- verification (host code only; needs no GPU) ------------------------------------------------
 * The counterpart of the reference's `verify` action (plonky2-backend/src/actions/verify_action.rs:11-17)
 * and of the `circuit_data.verify(proof)` assertion its tests end with (tests/factories/utils.rs:26-27),
 * on the uncompressed proof bytes p2gpu_prove writes.  Returns P2GPU_OK or P2GPU_E_VERIFY (the failed
 * check is in p2gpu_last_error).  Accepts a prover handle or a verifier-only one. */
int p2gpu_verify(const p2gpu_circuit *c, const uint8_t *proof, size_t len);
/* The verifier's share of a circuit -- header, gate table, constants_sigmas cap, circuit digest, k_is
 * (the blob prefix up to the constants table, flags 0b11) -- like the VK file of
 * actions/write_vk_action.rs:65-81.  out == NULL: only report the size in *len. */
int p2gpu_circuit_export_vk(const p2gpu_circuit *c, uint8_t *out, size_t *len);
/* Handle from such a blob (a full circuit blob with flags 0b11 works too: only its prefix is read).
 * No device is touched; the prove / fill / shard entry points reject it with P2GPU_E_ARG. */
int p2gpu_verifier_create(const uint8_t *blob, size_t len, p2gpu_circuit **out);
/* The same verifier key in the reference's own file format: `VerifierCircuitData::to_bytes(&BackendGateSerializer)`
 * as `write_vk` stores it (plonky2-backend/src/actions/write_vk_action.rs:65-81) and `verify` reads it
 * (noir_and_plonky2_serialization.rs:16-22).  UNPINNED: the layout is plonky2 0.2.2's util/serialization (a crate
 * that is not in the reference tree) restated from recollection -- no VK file exists there to check it against;
 * only the gate tag order (write_vk_action.rs:39-61) and the custom gates' bodies are taken from reference
 * source.  Exact inverse pair: create_plonky2(export_plonky2(c)) verifies what c verifies.
 * hasher: 0 = KeccakGoldilocksConfig (the reference), 1 = PoseidonGoldilocksConfig (the digest width is not
 * in the bytes).  out == NULL: only report the size in *len. */
int p2gpu_circuit_export_vk_plonky2(const p2gpu_circuit *c, uint8_t *out, size_t *len);
int p2gpu_verifier_create_plonky2(const uint8_t *vk, size_t len, int hasher, p2gpu_circuit **out);

/* ---- the reference's on-disk proof format (host code only) ------------------------------------
 * `plonky2-backend prove` writes hex(proof.compress(..).to_bytes()) (prove_action.rs:38-42,75-78) and
 * `verify` reads it back with verify_compressed (verify_action.rs:11-17).  compress / decompress
 * convert between p2gpu_prove's uncompressed bytes and that compressed layout (binary; hex is the
 * caller's business); out == NULL: only report the size in *out_len.  decompress rebuilds what was
 * dropped but does not judge the proof -- p2gpu_verify_compressed = decompress + p2gpu_verify. */
int p2gpu_proof_compress(const p2gpu_circuit *c, const uint8_t *proof, size_t len, uint8_t *out, size_t *out_len);
int p2gpu_proof_decompress(const p2gpu_circuit *c, const uint8_t *cproof, size_t len, uint8_t *out, size_t *out_len);
int p2gpu_verify_compressed(const p2gpu_circuit *c, const uint8_t *cproof, size_t len);

/* optional knobs: "pow_hint" (u64; UINT64_MAX = grind), "self_check" (0/1, default 1: evaluate the
 * verifier's plonk identity at zeta on the host before FRI and return P2GPU_E_UNSATISFIED when the
 * witness does not satisfy the circuit; 0 = emit the proof anyway like upstream), "shard_exercise" (0/1: with world = 1, still run the exchange steps of a sharded proof through the
 * configured transport -- how the RCCL path is tested on a one-GPU machine), "profile" (0 / 1 / 2: time kernel
 * launches with HIP events on the launch stream -- 1 = the launches that move >= 32 MB, 2 = every launch;
 * resets the statistics), "zero_columns" (0/1, default 1: one pass over the witness finds the wire columns that
 * are zero in every row -- the wires no gate of the circuit uses -- and stores zeros instead of running their
 * inverse transform and LDE; the proof bytes do not depend on it), "virtual_columns" (0/1, default 1: structured
 * wire columns that only the leaf hash and the query gather read are never stored -- both recompute them as scalar x
 * LDE(unit column); the proof bytes do not depend on it), "blocking_sync" (0/1, default 0: at the eleven transcript sync
 * points of a proof the calling thread spins in hipStreamSynchronize -- lowest latency; 1 = it sleeps on a blocking event,
 * ~10-30 us later per sync but without burning a CPU per proof in flight: for hosts with fewer CPUs than proving threads),
 * "half_gates" (0/1/2, default 1: the folded constraint sums of gates of degree <= 4 -- the reference's five custom gates,
 * BaseSum<4> -- are evaluated on the four even LDE cosets and interpolated to the odd ones where the circuit is large enough
 * for that to pay (constraints x gates >= 12 M); 2 = whatever the size, 0 = never; exact, the proof bytes do not depend on it), "shard_intt" (0/1, default 0; sharded proofs only: every rank runs the inverse transforms of ITS block of
 * the dense wire / Z-partial-product columns and the coefficient blocks are all-gathered in place, instead of every rank
 * transforming every column -- SURVEY 8(e) steps 1-2; the proof bytes do not depend on it; every rank must set it alike),
 * "shard_zs" (0/1, default 0; sharded proofs only: the chunk quotients of the permutation argument are computed for n / ranks rows
 * per rank and all-gathered in place -- SURVEY 8(e) step 5), "shard_reduce" (0/1, default 0; sharded proofs only: every rank
 * batch-reduces only its block of the opened polynomials for FRI, the partial sums are all-gathered and added -- step 8); same
 * rules: no proof byte depends on them, every rank sets them alike */
int p2gpu_circuit_set(p2gpu_circuit *c, const char *key, uint64_t value);
/* statistics accumulated while "profile" = 1, one entry per kernel symbol:
 * names[64*i] (NUL-terminated), total milliseconds, total algorithmic bytes,
 * launch count.  Returns the number of entries written (<= cap). */
int p2gpu_kernel_stats(p2gpu_circuit *c, char *names, double *ms, double *bytes, uint64_t *launches, int cap);

/* ---- one proof over several GPUs (one process per GPU) --------------------------------------
 * Coset sharding: rank q of `world` keeps the LDE cosets {r : r mod world == q}
 * of the per-proof oracles -- whole Merkle-cap subtrees -- and only three things are exchanged
 * per proof, all through `fn` (an all-gather the host side implements with torch.distributed /
 * RCCL on the device buffers it is handed): the cap entries of each commitment (16 x 25 B), the
 * per-coset quotient interpolants (2 * N * 8 B), and the query openings.  Every rank runs the
 * transcript and returns the same proof bytes.  `fn(ctx, send_dev, recv_dev, bytes)` must gather
 * `bytes` from every rank's send buffer into recv_dev[rank * bytes ...] and return 0 when done. */
typedef int (*p2gpu_allgather_fn)(void *ctx, uint64_t send_dev, uint64_t recv_dev, uint64_t bytes_per_rank);
int p2gpu_circuit_set_shard(p2gpu_circuit *c, int rank, int world, p2gpu_allgather_fn fn, void *ctx);
/* The same with RCCL inside the library (the production transport on a multi-GPU node: xGMI): rank 0
 * obtains a 128-byte id (ncclGetUniqueId), the host side broadcasts it to the other ranks by any means
 * (torch.distributed, MPI, a file), every rank then calls p2gpu_circuit_set_shard_rccl with its rank.
 * The collectives (ncclAllGather) are enqueued on the circuit's own HIP stream -- no host synchronisation
 * around them.  RCCL is resolved with dlopen("librccl.so.1") at this point, not at link time: the copy
 * the process already uses (torch's) is taken when there is one. */
int p2gpu_shard_unique_id(uint8_t id_out[128]);
/* Host-only piece of the sharded commitment, exposed for callers that run the exchange themselves and for
 * the CPU-side tests: the Merkle cap in plonky2 order (2^cap_h x 25 B) from the all-gather of every rank's
 * local subtree roots, gathered = [world][2^rate_bits / world local cosets][2^(cap_h - rate_bits)] digests
 * of 32 B (25 used); rank q's local coset z is the global coset q + z * world. */
int p2gpu_shard_assemble_cap(int world, unsigned rate_bits, unsigned cap_h, const uint8_t *gathered, uint8_t *cap_out);
int p2gpu_circuit_set_shard_rccl(p2gpu_circuit *c, int rank, int world, const uint8_t id[128]);

/* stage-level operators (host buffers in/out; used by the parity tests) */
/* values [ncols][2^d] -> coefficients [ncols][2^d], natural order */
int p2gpu_ifft_batch(const uint64_t *vals, size_t ncols, unsigned d, uint64_t *coeffs_out);
/* coefficients [ncols][2^d] -> LDE values [ncols][2^(d+rate_bits)] on the coset g<w> (g = 14293326489335486720, plonky2's coset shift), natural order */
int p2gpu_lde_batch(const uint64_t *coeffs, size_t ncols, unsigned d, unsigned rate_bits, uint64_t *lde_out);
/* PolynomialBatch::from_values: commit value columns, return the Merkle cap (2^cap_h x 25 B) */
int p2gpu_commit_values(const uint64_t *vals, size_t ncols, unsigned d, unsigned rate_bits, unsigned cap_h,
                        uint8_t *cap_out);
/* KeccakHash<25>::hash_no_pad of `n_rows` rows of `row_len` elements (row-major) -> n_rows x 25 B */
int p2gpu_hash_rows(const uint64_t *rows, size_t n_rows, size_t row_len, uint8_t *digests_out);
/* Self-test of the device field arithmetic (the carry-chain forms of csrc/gl.hpp and the NTT's power-of-two
 * multipliers) against the portable code the host and the oracle run, on the n pairs (a[i], b[i]) of arbitrary
 * 64-bit words.  bad_out[k] = mismatches of: 0 canon, 1 add, 2 sub, 3 reduce128, 4 mul, 5 mul_add,
 * 6 x * 2^e for e = 1..95, 7 the 160-bit accumulator -- the EIGHT words p2gpu_field_selftest writes (its contract since
 * round 1); p2gpu_field_selftest16 writes SIXTEEN: those, then 8 mul_nc, 9 mul_add_nc, 10 reduce128_nc (the congruent-word
 * forms: any u64 in, some congruent u64 out), 11 add / 12 sub with a non-canonical first operand, 13 chains of those; 14, 15
 * unused.  All zero on a correct build. */
int p2gpu_field_selftest(const uint64_t *a, const uint64_t *b, size_t n, uint64_t bad_out[8]);
int p2gpu_field_selftest16(const uint64_t *a, const uint64_t *b, size_t n, uint64_t bad_out[16]);

const char *p2gpu_last_error(void);
/* name/arch of the device in use, and peak numbers the bench prints */
int p2gpu_device_info(char *name_out, size_t name_cap, int *cu_count, size_t *hbm_bytes);
/* Peer access between the devices of the last p2gpu_init (device groups, section "multi-GPU" of DESIGN.md):
 * matrix_out[a * n + b] = 1: device a reaches device b's memory directly (hipDeviceEnablePeerAccess succeeded: peer copies
 * travel over xGMI), 0: it does not (peer copies are staged through host memory), -1: a and b are the same device.
 * matrix_out may be NULL.  Returns n (>= 1), P2GPU_E_BUFFER when cap < n * n, or another error code.
 * (No reference counterpart: plonky2's prover is single-device; prove_action.rs:96 is the one call this serves.) */
int p2gpu_peer_access(int *matrix_out, int cap);

#ifdef __cplusplus
}
#endif
#endif
