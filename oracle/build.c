/* oracle/build.c -- TEST INFRASTRUCTURE: CPU restatement of the GPU-free part of
 * `builder.build::<C>()` (plonky2-backend/src/circuit_translation/mod.rs:80-82; plonky2 0.2.2
 * gates/selectors.rs selector_polynomials, plonk/permutation_argument.rs WirePartition::get_sigma_*,
 * fri/reduction_strategies.rs ConstantArityBits) -- the checker for the product's p2gpu_build_blob.
 * Written independently of it: classes are collected by sorting (root, row, column) triples instead of
 * threading linked lists, selectors are filled group by group.  Output: the circuit blob of include/p2gpu.h.
 * PARITY STATUS: pinned to the reference through the circuits recovered from its two shipped proofs
 * (tests/golden/reference_proofs.py): their constants and sigma columns come back bit for bit
 * (tests/test_build.py). */
#include <stdlib.h>
#include <string.h>
#include "gl.h"
#include "oracle.h"

typedef struct { uint32_t root, row, col; } cell_t;
static int cell_cmp(const void *a, const void *b) {
  const cell_t *x = a, *y = b;
  if (x->root != y->root) return x->root < y->root ? -1 : 1;
  if (x->row != y->row) return x->row < y->row ? -1 : 1;
  return x->col < y->col ? -1 : (x->col > y->col);
}
static uint32_t uf_find(uint32_t *p, uint32_t x) {
  while (p[x] != x) x = p[x] = p[p[x]];
  return x;
}
static uint32_t gate_constraints(uint32_t kind, const uint32_t p[4]) {
  switch (kind) {
  case 1: return p[0];
  case 2: return 4;
  case 3: return p[0];
  case 4: return 1 + p[1];
  case 5: return p[1] * (p[0] + 2) + p[2];
  case 6: return 123;
  case 7: return p[0] * 36;
  case 8: return p[1] * 21;
  case 9: return p[0] * 19;
  case 10: return p[0] * 17;
  case 11: return 6 + 5 * p[1] + (p[0] + p[1] - 1) / p[1];
  default: return 0;
  }
}

int orc_build_blob(const orc_build_params *bp, const orc_gate_decl *gates, uint32_t ng, const uint32_t *row_gate,
                   const uint64_t *row_constants, const uint32_t *copies, size_t ncopies, uint8_t *out, size_t *len) {
  const uint32_t d = bp->degree_bits, R = bp->num_routed_wires, QF = bp->quotient_degree_factor;
  const size_t n = (size_t)1 << d;
  /* selectors: group index and [start, end) per gate */
  uint32_t gsel[64], gs[64], ge[64], nsel = 0;
  if (ng == 0 || ng > 64) return ORC_E_BLOB;
  if (gates[ng - 1].degree + ng - 1 <= QF + 1) {
    nsel = 1;
    for (uint32_t i = 0; i < ng; i++) { gsel[i] = 0; gs[i] = 0; ge[i] = ng; }
  } else {
    for (uint32_t start = 0; start < ng; nsel++) {
      uint32_t end = start;
      while (end < ng && (end - start) + gates[end].degree < QF + 1) end++;
      if (end == start) return ORC_E_BLOB;
      for (uint32_t i = start; i < end; i++) { gsel[i] = nsel; gs[i] = start; ge[i] = end; }
      start = end;
    }
  }
  uint32_t ngc = 0;
  for (uint32_t i = 0; i < ng; i++) if (gates[i].num_constants > ngc) ngc = gates[i].num_constants;
  const uint32_t NC = nsel + ngc;
  uint32_t arity[8], nar = 0;
  for (uint32_t db = d; db > 5 && db + bp->rate_bits - 4 >= bp->cap_height && nar < 8; db -= 4) arity[nar++] = 4;
  const size_t need = 256 + 48 * (size_t)ng + 8 * ((size_t)R + (size_t)NC * n + (size_t)R * n);
  if (!out || *len < need) { *len = need; return out ? ORC_E_BUFFER : ORC_OK; }
  uint32_t h[64];
  memset(h, 0, sizeof h);
  h[0] = 0x43473250u; h[1] = 1; h[2] = d; h[3] = bp->num_wires; h[4] = R; h[5] = NC; h[6] = nsel; h[7] = bp->num_challenges; h[8] = QF;
  h[9] = bp->rate_bits; h[10] = bp->cap_height; h[11] = bp->proof_of_work_bits; h[12] = bp->num_query_rounds; h[13] = nar;
  for (uint32_t i = 0; i < nar; i++) h[14 + i] = arity[i];
  h[23] = ng; h[24] = bp->num_public_inputs; h[26] = (R + QF - 1) / QF - 1;
  memcpy(out, h, sizeof h);
  size_t off = sizeof h;
  for (uint32_t i = 0; i < ng; i++) {
    uint32_t g[12] = {gates[i].kind, gates[i].p[0], gates[i].p[1], gates[i].p[2], gates[i].p[3], gsel[i], gs[i], ge[i],
                      gate_constraints(gates[i].kind, gates[i].p), gates[i].degree, gates[i].num_constants, 0};
    memcpy(out + off, g, sizeof g);
    off += sizeof g;
  }
  uint64_t *kis = (uint64_t *)(out + off);
  for (uint32_t j = 0; j < R; j++) kis[j] = j ? gl_mul(kis[j - 1], GL_GENERATOR) : 1;
  off += 8 * (size_t)R;
  uint64_t *cst = (uint64_t *)(out + off);
  for (uint32_t s = 0; s < nsel; s++)
    for (size_t r = 0; r < n; r++) {
      const uint32_t gi = row_gate[r];
      if (gi >= ng) return ORC_E_BLOB;
      cst[(size_t)s * n + r] = (nsel == 1 || gsel[gi] == s) ? gi : 0xFFFFFFFFull;
    }
  if (ngc) memcpy(cst + (size_t)nsel * n, row_constants, 8 * (size_t)ngc * n);
  off += 8 * (size_t)NC * n;
  /* sigma: class members sorted by (row, column); each maps to its successor, the last to the first */
  uint64_t *sig = (uint64_t *)(out + off);
  const size_t tot = (size_t)R * n;
  uint32_t *par = malloc(4 * tot);
  cell_t *cells = malloc(sizeof(cell_t) * tot);
  uint64_t *sub = malloc(8 * n);
  if (!par || !cells || !sub) { free(par); free(cells); free(sub); return ORC_E_BUFFER; }
  for (size_t x = 0; x < tot; x++) par[x] = (uint32_t)x;
  for (size_t e = 0; e < ncopies; e++) {
    const uint32_t a = uf_find(par, (uint32_t)((size_t)copies[4 * e + 1] * n + copies[4 * e]));
    const uint32_t b = uf_find(par, (uint32_t)((size_t)copies[4 * e + 3] * n + copies[4 * e + 2]));
    if (a != b) par[a > b ? a : b] = a > b ? b : a;
  }
  for (size_t x = 0; x < tot; x++) { cells[x].root = uf_find(par, (uint32_t)x); cells[x].col = (uint32_t)(x / n); cells[x].row = (uint32_t)(x % n); }
  qsort(cells, tot, sizeof(cell_t), cell_cmp);
  const uint64_t wn = gl_root_of_unity(d);
  for (size_t i = 0; i < n; i++) sub[i] = i ? gl_mul(sub[i - 1], wn) : 1;
  for (size_t i = 0; i < tot;) {
    size_t j = i;
    while (j < tot && cells[j].root == cells[i].root) j++;
    for (size_t k = i; k < j; k++) {
      const cell_t *nx = &cells[k + 1 < j ? k + 1 : i];
      sig[(size_t)cells[k].col * n + cells[k].row] = gl_mul(kis[nx->col], sub[nx->row]);
    }
    i = j;
  }
  free(par); free(cells); free(sub);
  off += 8 * tot;
  *len = off;
  return ORC_OK;
}
