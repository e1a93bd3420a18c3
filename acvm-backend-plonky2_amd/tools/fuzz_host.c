/* fuzz_host.c -- mutation fuzzer for the host-only parsers of libp2gpu (plain C on the C ABI).
 *
 * The verifier side of the library eats untrusted bytes on a machine without a GPU: the verifier-key
 * blob (p2gpu_verifier_create -> circuit_parse), uncompressed proofs (p2gpu_verify,
 * p2gpu_proof_compress) and the reference's compressed on-disk format (p2gpu_proof_decompress,
 * p2gpu_verify_compressed; reference reader: plonky2-backend/src/actions/verify_action.rs:11-17,
 * noir_and_plonky2_serialization.rs:16-33).  Every mutated input must come back as an error code --
 * or as "accepted" only when the mutation did not change the bytes.  Built twice by the tests: against
 * the normal library, and against a host-only -fsanitize=address,undefined build (make asan).
 *
 *   fuzz_host <vk.blob> <proof.bin> <compressed.bin> <iterations> <seed>
 * exit 0 = survived (counts on stdout), 1 = usage / io, 2 = a mutated proof was ACCEPTED.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/p2gpu.h"

static uint64_t rng_state;
static uint64_t rnd(void) { /* splitmix64 */
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static uint8_t *slurp(const char *path, size_t *len) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t *buf = malloc(n > 0 ? (size_t)n : 1);
  if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) {
    free(buf);
    buf = NULL;
  }
  fclose(f);
  *len = (size_t)n;
  return buf;
}

static const uint32_t EDGE[] = {0, 1, 2, 3, 4, 7, 8, 16, 31, 32, 63, 64, 65, 127, 128, 255, 256, 1023, 4095, 4096, 4097,
                                65535, 65536, 1u << 20, 1u << 24, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFEu, 0xFFFFFFFFu};

/* returns a malloc'd mutated copy; *out_len may shrink (truncation) or grow (padding) */
static uint8_t *mutate(const uint8_t *src, size_t len, size_t *out_len, size_t structured_prefix) {
  size_t n = len;
  const unsigned kind = (unsigned)(rnd() % 8);
  if (kind == 0 && len > 0) n = (size_t)(rnd() % len);          /* truncate */
  if (kind == 1) n = len + 1 + (size_t)(rnd() % 64);             /* trailing bytes */
  uint8_t *m = malloc(n ? n : 1);
  memcpy(m, src, n < len ? n : len);
  for (size_t i = len; i < n; i++) m[i] = (uint8_t)rnd();
  if (n == 0) {
    *out_len = 0;
    return m;
  }
  const size_t pre = structured_prefix && structured_prefix < n ? structured_prefix : n;
  switch (kind) {
  case 2: /* one bit */
    m[rnd() % n] ^= (uint8_t)(1u << (rnd() % 8));
    break;
  case 3: { /* a few random bytes */
    unsigned k = 1 + (unsigned)(rnd() % 8);
    while (k--) m[rnd() % n] = (uint8_t)rnd();
    break;
  }
  case 4: /* an aligned u32 of the structured prefix (header / gate table) := edge value */
  case 5: {
    unsigned k = 1 + (unsigned)(rnd() % 3);
    while (k--) {
      size_t w = (size_t)(rnd() % (pre / 4 ? pre / 4 : 1));
      uint32_t v = EDGE[rnd() % (sizeof EDGE / sizeof EDGE[0])];
      if (4 * w + 4 <= n) memcpy(m + 4 * w, &v, 4);
    }
    break;
  }
  case 6: { /* an aligned u64 := non-canonical / extreme field element */
    static const uint64_t V[] = {0xFFFFFFFF00000001ull, 0xFFFFFFFF00000000ull, 0xFFFFFFFFFFFFFFFFull, 0, 1, 0xFFFFFFFF00000002ull};
    size_t w = (size_t)(rnd() % (n / 8 ? n / 8 : 1));
    uint64_t v = V[rnd() % 6];
    if (8 * w + 8 <= n) memcpy(m + 8 * w, &v, 8);
    break;
  }
  case 7: { /* a random run */
    size_t a = (size_t)(rnd() % n), l = 1 + (size_t)(rnd() % 64);
    for (size_t i = a; i < n && i < a + l; i++) m[i] = (uint8_t)rnd();
    break;
  }
  default: break;
  }
  *out_len = n;
  return m;
}

int main(int argc, char **argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s <vk.blob> <proof.bin> <compressed.bin> <iterations> <seed>\n", argv[0]);
    return 1;
  }
  size_t blob_len, proof_len, comp_len;
  uint8_t *blob = slurp(argv[1], &blob_len), *proof = slurp(argv[2], &proof_len), *comp = slurp(argv[3], &comp_len);
  const long iters = atol(argv[4]);
  rng_state = strtoull(argv[5], NULL, 0);
  if (!blob || !proof || !comp) {
    fprintf(stderr, "cannot read inputs\n");
    return 1;
  }
  p2gpu_circuit *good = NULL;
  if (p2gpu_verifier_create(blob, blob_len, &good) || p2gpu_verify(good, proof, proof_len) ||
      p2gpu_verify_compressed(good, comp, comp_len)) {
    fprintf(stderr, "the unmutated inputs are not accepted: %s\n", p2gpu_last_error());
    return 1;
  }
  const size_t out_cap = 4 * (proof_len + comp_len) + (1u << 16);
  uint8_t *out = malloc(out_cap);
  /* header (256 B) + gate table are the structured part of the blob; the number of gates is word 23 */
  uint32_t ng;
  memcpy(&ng, blob + 4 * 23, 4);
  const size_t blob_prefix = 256 + 48 * (size_t)(ng < 64 ? ng : 64);
  long blobs_ok = 0, blobs_rejected = 0, proofs_rejected = 0, unchanged = 0, accepted_mutants = 0, vks_ok = 0, vks_rejected = 0;
  /* the same key in the reference's VK file format (p2gpu_verifier_create_plonky2) */
  size_t vk_len = 0;
  (void)p2gpu_circuit_export_vk_plonky2(good, NULL, &vk_len);
  uint8_t *vk = malloc(vk_len + 1);
  if (p2gpu_circuit_export_vk_plonky2(good, vk, &vk_len)) {
    fprintf(stderr, "export_vk_plonky2 failed: %s\n", p2gpu_last_error());
    return 2;
  }
  for (long it = 0; it < iters; it++) {
    const unsigned target = (unsigned)(rnd() % 4);
    size_t n;
    if (target == 3) {
      uint8_t *m = mutate(vk, vk_len, &n, vk_len);
      p2gpu_circuit *h = NULL;
      if (p2gpu_verifier_create_plonky2(m, n, 0, &h) == 0) {
        vks_ok++;
        size_t ol = 0;
        (void)p2gpu_verify(h, proof, proof_len);
        (void)p2gpu_verify_compressed(h, comp, comp_len);
        (void)p2gpu_circuit_export_vk_plonky2(h, NULL, &ol);
        p2gpu_circuit_destroy(h);
      } else {
        vks_rejected++;
      }
      free(m);
    } else if (target == 0) {
      uint8_t *m = mutate(blob, blob_len, &n, blob_prefix);
      p2gpu_circuit *h = NULL;
      if (p2gpu_verifier_create(m, n, &h) == 0) {
        blobs_ok++;
        size_t ol = out_cap;
        (void)p2gpu_verify(h, proof, proof_len);
        (void)p2gpu_proof_compress(h, proof, proof_len, out, &ol);
        ol = out_cap;
        (void)p2gpu_proof_decompress(h, comp, comp_len, out, &ol);
        (void)p2gpu_verify_compressed(h, comp, comp_len);
        ol = 0;
        (void)p2gpu_circuit_export_vk(h, NULL, &ol);
        (void)p2gpu_proof_size_bound(h);
        p2gpu_circuit_destroy(h);
      } else {
        blobs_rejected++;
      }
      free(m);
    } else if (target == 1) {
      uint8_t *m = mutate(proof, proof_len, &n, 0);
      const int same = n == proof_len && memcmp(m, proof, n) == 0;
      size_t ol = out_cap;
      const int rc = p2gpu_verify(good, m, n);
      (void)p2gpu_proof_compress(good, m, n, out, &ol);
      if (same) unchanged++;
      else if (rc == 0) accepted_mutants++;
      else proofs_rejected++;
      free(m);
    } else {
      uint8_t *m = mutate(comp, comp_len, &n, 0);
      const int same = n == comp_len && memcmp(m, comp, n) == 0;
      size_t ol = out_cap;
      const int rc = p2gpu_verify_compressed(good, m, n);
      (void)p2gpu_proof_decompress(good, m, n, out, &ol);
      if (same) unchanged++;
      else if (rc == 0) accepted_mutants++;
      else proofs_rejected++;
      free(m);
    }
  }
  printf("iterations %ld: mutated blobs parsed %ld / rejected %ld, mutated VK files parsed %ld / rejected %ld, mutated proofs rejected %ld, no-op mutations %ld, ACCEPTED MUTANTS %ld\n",
         iters, blobs_ok, blobs_rejected, vks_ok, vks_rejected, proofs_rejected, unchanged, accepted_mutants);
  free(vk);
  p2gpu_circuit_destroy(good);
  free(out);
  free(blob);
  free(proof);
  free(comp);
  return accepted_mutants ? 2 : 0;
}
