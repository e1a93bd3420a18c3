/*
 * oracle/circuit.h -- circuit blob parsing + gate constraint evaluators.
 * TEST INFRASTRUCTURE ONLY (see gl.h).
 *
 * The blob is the build-defined export of plonky2's CommonCircuitData +
 * ProverOnlyCircuitData that a Rust shim would write once per circuit
 * (SURVEY.md 8(b)); layout documented in include/p2gpu.h.
 *
 * Gate evaluators restate:
 *   in-tree (citable) custom gates, plonky2-backend/src/plonky2_ecdsa/biguint/gates/
 *     arithmetic_u32.rs:289-348   U32ArithmeticGate
 *     add_many_u32.rs:151-192     U32AddManyGate
 *     subtraction_u32.rs:234-270  U32SubtractionGate
 *     range_check_u32.rs:95-117   U32RangeCheckGate
 *     comparison.rs:337-414       ComparisonGate
 *   stock plonky2 0.2.2 gates (absent):
 *     NoopGate, ConstantGate, PublicInputGate, ArithmeticGate, BaseSumGate<B>,
 *     RandomAccessGate, PoseidonGate
 * The closed set of gate kinds is the registry at
 * plonky2-backend/src/actions/write_vk_action.rs:35-62.
 */
#ifndef ORACLE_CIRCUIT_H
#define ORACLE_CIRCUIT_H
#include "gl.h"
#include "hash.h"
#include "poly.h"

enum {
  G_NOOP = 0,
  G_CONSTANT = 1,      /* p0 = num_consts */
  G_PUBLIC_INPUT = 2,
  G_ARITHMETIC = 3,    /* p0 = num_ops */
  G_BASE_SUM = 4,      /* p0 = base B, p1 = num_limbs */
  G_RANDOM_ACCESS = 5, /* p0 = bits, p1 = num_copies, p2 = num_extra_constants */
  G_POSEIDON = 6,      /* PoseidonGate (width 12); no parameters */
  G_U32_ARITHMETIC = 7,  /* p0 = num_ops */
  G_U32_ADD_MANY = 8,    /* p0 = num_addends, p1 = num_ops */
  G_U32_SUBTRACTION = 9, /* p0 = num_ops */
  G_U32_RANGE_CHECK = 10, /* p0 = num_input_limbs */
  G_COMPARISON = 11,     /* p0 = num_bits, p1 = num_chunks */
  G_KIND_COUNT
};

#define BLOB_MAGIC 0x43473250u
#define BLOB_HEADER_WORDS 64
#define BLOB_GATE_WORDS 12

typedef struct {
  uint32_t kind, p[4];
  uint32_t sel_index, group_start, group_end;
  uint32_t num_constraints, degree, num_constants;
} gate_t;

typedef struct {
  uint32_t d, num_wires, num_routed, num_constants, num_selectors, num_challenges, qdf;
  uint32_t rate_bits, cap_height, pow_bits, num_queries, n_steps, arity_bits[8];
  uint32_t hasher, num_gates, num_pi, flags, num_pp;
  uint8_t digest_in[32];
  gate_t *gates;
  uint32_t num_gate_constraints;
  const gl_t *k_is;      /* [num_routed] */
  const gl_t *constants; /* [num_constants][n] */
  const gl_t *sigmas;    /* [num_routed][n] */
  const uint8_t *cap_in; /* optional expected constants_sigmas cap (32 B stride) */
  size_t n, N;
  /* derived by circuit_load (plonky2 `build`): oracle 0 and the digest */
  batch_t cs;
  digest_t circuit_digest;
  digest_t *vcap; /* verifier-only mode: cap supplied by the caller */
} circuit_t;

/* returns 0 ok; <0 error.  Keeps pointers into blob (caller keeps it alive). */
int circuit_parse(circuit_t *c, const uint8_t *blob, size_t len);
/* parse + constants_sigmas commitment + circuit digest */
int circuit_load(circuit_t *c, const uint8_t *blob, size_t len);
/* verifier-only: parse + take the constants_sigmas cap (2^cap_height x 25 B) and digest as given
 * (VerifierOnlyCircuitData), no commitment is computed */
int circuit_load_verifier(circuit_t *c, const uint8_t *blob, size_t len, const uint8_t *cap, const uint8_t *digest);
void circuit_free(circuit_t *c);

uint32_t gate_num_constraints(uint32_t kind, const uint32_t p[4]);
uint32_t gate_degree(uint32_t kind, const uint32_t p[4]);
uint32_t gate_num_constants(uint32_t kind, const uint32_t p[4]);
uint32_t gate_num_wires(uint32_t kind, const uint32_t p[4]);

/* unfiltered constraint evaluation of one gate; out has num_constraints slots */
void gate_eval_base(const gate_t *g, const gl_t *wires, const gl_t *local_consts, const gl_t *pi_hash, gl_t *out);
void gate_eval_ext(const gate_t *g, const ext_t *wires, const ext_t *local_consts, const ext_t *pi_hash, ext_t *out);

/* selectors.rs compute_filter */
gl_t gate_filter_base(const circuit_t *c, uint32_t gate_idx, gl_t s);
ext_t gate_filter_ext(const circuit_t *c, uint32_t gate_idx, ext_t s);

/* evaluate_gate_constraints: out[num_gate_constraints] = sum_g filter_g * c_g */
void eval_gate_constraints_base(const circuit_t *c, const gl_t *consts_row /*num_constants*/, const gl_t *wires,
                                const gl_t *pi_hash, gl_t *out, gl_t *scratch);
void eval_gate_constraints_ext(const circuit_t *c, const ext_t *consts_row, const ext_t *wires, const ext_t *pi_hash,
                               ext_t *out, ext_t *scratch);
#endif
