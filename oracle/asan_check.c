/* asan_check.c -- sanitizer driver for the oracle (test infrastructure; SURVEY.md section 5's sanitizer row).
 * Built by `make asan` with -fsanitize=address,undefined together with the oracle's sources:
 *   asan_check <circuit.blob> <wires.bin> <iterations> <seed>
 * proves the circuit, verifies the proof, re-creates the handle as a verifier-only one, then feeds
 * `iterations` mutated proofs (bit flips, non-canonical words, truncations, trailing bytes) to
 * orc_verify.  Survival under the sanitizers (and no accepted mutant) is the result; exit 0 / 2. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static uint64_t st;
static uint64_t rnd(void) {
  uint64_t z = (st += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static void *slurp(const char *p, size_t *n) {
  FILE *f = fopen(p, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long l = ftell(f);
  fseek(f, 0, SEEK_SET);
  void *b = malloc(l > 0 ? (size_t)l : 1);
  if (b && fread(b, 1, (size_t)l, f) != (size_t)l) { free(b); b = NULL; }
  fclose(f);
  *n = (size_t)l;
  return b;
}
int main(int argc, char **argv) {
  if (argc < 5) return 1;
  size_t bl, wl;
  uint8_t *blob = slurp(argv[1], &bl);
  uint64_t *wires = slurp(argv[2], &wl);
  long iters = atol(argv[3]);
  st = strtoull(argv[4], NULL, 0);
  if (!blob || !wires) return 1;
  orc_circuit *c = NULL;
  if (orc_circuit_create(blob, bl, &c)) { fprintf(stderr, "create failed\n"); return 1; }
  size_t cap = 1u << 21, len = cap;
  uint8_t *proof = malloc(cap);
  orc_trace tr;
  if (orc_prove(c, wires, NULL, 0, UINT64_MAX, proof, &len, &tr) || orc_verify(c, proof, len, &tr)) {
    fprintf(stderr, "prove/verify failed\n");
    return 1;
  }
  uint8_t capb[25 * 64], dig[25];
  orc_circuit_cap(c, capb);
  orc_circuit_digest(c, dig);
  orc_circuit *v = NULL;
  if (orc_circuit_create_verifier(blob, bl, capb, dig, &v) || orc_verify(v, proof, len, &tr)) return 1;
  long accepted = 0, rejected = 0;
  uint8_t *m = malloc(len + 64);
  for (long it = 0; it < iters; it++) {
    size_t n = len;
    memcpy(m, proof, len);
    switch (rnd() % 5) {
    case 0: n = rnd() % len; break;
    case 1: n = len + 1 + rnd() % 63; for (size_t i = len; i < n; i++) m[i] = (uint8_t)rnd(); break;
    case 2: m[rnd() % len] ^= (uint8_t)(1u << (rnd() % 8)); break;
    case 3: { uint64_t x = rnd() & 1 ? 0xFFFFFFFF00000001ull : 0xFFFFFFFFFFFFFFFFull; memcpy(m + 8 * (rnd() % (len / 8)), &x, 8); break; }
    default: { size_t a = rnd() % len; for (size_t i = a; i < len && i < a + 1 + rnd() % 32; i++) m[i] = (uint8_t)rnd(); }
    }
    if (n == len && !memcmp(m, proof, len)) continue;
    if (orc_verify(v, m, n, &tr) == 0) accepted++;
    else rejected++;
  }
  printf("oracle: proof %zu bytes ok; mutants rejected %ld, ACCEPTED %ld\n", len, rejected, accepted);
  orc_circuit_destroy(v);
  orc_circuit_destroy(c);
  free(m); free(proof); free(blob); free(wires);
  return accepted ? 2 : 0;
}
