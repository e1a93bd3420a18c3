/*
 * oracle/oracle.h -- public C API of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  A CPU restatement of the `prove` hot path of
 * eryxcoop/acvm-backend-plonky2:
 *   plonky2-backend/src/actions/prove_action.rs:91-97
 *     circuit_data.prove(witnesses)  ->  plonky2 0.2.2 prover::prove
 * and of the matching verifier (plonky2-backend/src/actions/verify_action.rs:11-17).
 * Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may
 * load this library; the product (libp2gpu.so) never does.
 *
 * PARITY STATUS: pinned, bit-exact, to the two proofs the reference ships
 * (plonky2-backend/example_programs/basic_{if,div}/proofs/basic_{if,div}.proof; fixtures
 * tests/golden/reference/, derivation tests/golden/reference_proofs.py):
 * orc_prove reproduces their bytes from the circuit + witness recovered from
 * them, orc_verify accepts them (tests/test_reference_proofs.py).  plonky2
 * 0.2.2 itself (fork github.com/brweisz/plonky2, unpinned path dependency:
 * plonky2-backend/Cargo.toml:14,29-32, Cargo.lock:898-965) is not vendored
 * under /root/reference and there is no Rust toolchain, so what those two
 * 2^3-row proofs cannot reach stays pinned only to the reference's in-tree
 * sources/tests: FRI reduction steps (arity-16 fold), RandomAccessGate and the
 * five custom gates (tests/test_oracle_gates.py restates their gate tests).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_circuit orc_circuit;

/* intermediate values of one proof, for stage-level parity checks */
typedef struct {
  uint64_t pi_hash[4];
  uint64_t betas[4], gammas[4], alphas[4];
  uint64_t zeta[2], alpha_fri[2];
  uint64_t fri_betas[8][2];
  uint64_t pow_witness;
  uint32_t query_indices[64];
  double t_wires, t_zs, t_quotient, t_openings, t_fri, t_total; /* seconds */
} orc_trace;

#define ORC_OK 0
#define ORC_E_BLOB -1
#define ORC_E_BUFFER -2
#define ORC_E_VERIFY -3
#define ORC_E_ZETA_IN_SUBGROUP -4

int orc_circuit_create(const uint8_t *blob, size_t len, orc_circuit **out);
/* verifier-only handle (VerifierCircuitData): cap = 2^cap_height x 25 B, no commitment work */
int orc_circuit_create_verifier(const uint8_t *blob, size_t len, const uint8_t *cap, const uint8_t *digest,
                                orc_circuit **out);
void orc_circuit_destroy(orc_circuit *c);
/* copies 2^cap_height digests of 25 bytes each */
void orc_circuit_cap(const orc_circuit *c, uint8_t *out);
void orc_circuit_digest(const orc_circuit *c, uint8_t *out /* 25 B (Keccak) or 32 B (Poseidon hasher) */);
/* hasher of the stage-level helpers (orc_commit_values, orc_merkle_cap, orc_challenger_squeeze): 0 Keccak, 1 Poseidon */
void orc_set_hasher(int hasher);

/* wires: [num_wires][n] column-major canonical u64.  pow_hint: UINT64_MAX =
 * search for the minimum witness, otherwise use the given witness. */
int orc_prove(const orc_circuit *c, const uint64_t *wires, const uint64_t *public_inputs, uint32_t n_pi,
              uint64_t pow_hint, uint8_t *proof_out, size_t *proof_len, orc_trace *trace);
int orc_verify(const orc_circuit *c, const uint8_t *proof, size_t len, orc_trace *trace);

/* row-local witness generators: fill every wire a gate's own generator derives
 * from the gate's input wires; wires [num_wires][n] in place */
int orc_fill_witness(const orc_circuit *c, uint64_t *wires);

/* the GPU-free part of build(): selectors, sigma, k_is, FRI arities -> circuit blob (build.c).  Same argument
 * meaning as the product's p2gpu_build_blob (include/p2gpu.h). */
typedef struct {
  uint32_t degree_bits, num_wires, num_routed_wires, num_challenges, quotient_degree_factor, rate_bits, cap_height,
      proof_of_work_bits, num_query_rounds, num_public_inputs;
} orc_build_params;
typedef struct {
  uint32_t kind, p[4], degree, num_constants;
} orc_gate_decl;
int orc_build_blob(const orc_build_params *bp, const orc_gate_decl *gates, uint32_t num_gates, const uint32_t *row_gate,
                   const uint64_t *row_constants, const uint32_t *copies, size_t num_copies, uint8_t *out, size_t *len);

/* stage-level entry points for parity tests */
void orc_ntt(uint64_t *a, unsigned lg, int inverse);
void orc_coset_lde(const uint64_t *coeffs, unsigned d, unsigned rate_bits, uint64_t *out /* 2^(d+rate_bits), natural */);
void orc_keccak256(const uint8_t *in, size_t len, uint8_t out[32]);
void orc_keccak_permutation(uint64_t st[12]);
void orc_poseidon_permute(uint64_t st[12]);
void orc_poseidon_round_constants(uint64_t out[360]);
void orc_poseidon_hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4]);
/* commit value columns [ncols][n]; outputs cap (2^cap_h x 25 B) */
void orc_commit_values(const uint64_t *vals, size_t ncols, unsigned d, unsigned rate_bits, unsigned cap_h, uint8_t *cap);
/* Merkle cap over row-major leaves */
void orc_merkle_cap(const uint64_t *leaves, size_t n_leaves, size_t leaf_len, unsigned cap_h, uint8_t *cap);
/* unfiltered gate constraints on one row; returns num_constraints */
int orc_gate_eval(uint32_t kind, const uint32_t params[4], const uint64_t *wires, const uint64_t *consts,
                  const uint64_t pi_hash[4], uint64_t *out);
/* same through the extension-field evaluator; wires/consts/out are (c0,c1) pairs */
int orc_gate_eval_ext(uint32_t kind, const uint32_t params[4], const uint64_t *wires, const uint64_t *consts,
                      const uint64_t pi_hash[4], uint64_t *out);
uint64_t orc_gl_mul(uint64_t a, uint64_t b);
uint64_t orc_gl_inv(uint64_t a);
uint64_t orc_gl_pow(uint64_t a, uint64_t e);
/* challenger transcript helper for tests: observe `n` elements then squeeze `m` */
void orc_challenger_squeeze(const uint64_t *obs, size_t n, uint64_t *out, size_t m);
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
