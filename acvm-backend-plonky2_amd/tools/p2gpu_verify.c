/*
 * p2gpu-verify -- stand-alone verifier over the C ABI (include/p2gpu.h), plain C, no GPU needed:
 *
 *     p2gpu-verify <vk.blob> <proof.bin>
 *
 * The counterpart of `plonky2-backend verify -k <vk> -p <proof>`
 * (plonky2-backend/src/argument_parsing.rs:49-53 -> actions/verify_action.rs:11-17) on the
 * uncompressed proof bytes p2gpu-prove writes.  <vk.blob> is the verifier's share of the circuit
 * (p2gpu_circuit_export_vk, or `p2gpu-prove --vk <file>`).  A proof file that consists of hex digits
 * only is taken to be in the reference's own format -- hex of the COMPRESSED proof
 * (prove_action.rs:38-42,75-78) -- and goes through verify_compressed like verify_action.rs:14-16.
 * Exit code 0 = proof accepted,
 * 1 = usage/IO, 2 = bad blob, 3 = proof rejected (the failed check is printed).
 */
#include "../../include/p2gpu.h"
#include <stdio.h>
#include <ctype.h>
#include <stdlib.h>

static void *slurp(const char *path, size_t *len) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  rewind(f);
  void *buf = malloc(n > 0 ? (size_t)n : 1);
  if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) {
    free(buf);
    buf = NULL;
  }
  fclose(f);
  *len = (size_t)n;
  return buf;
}

int main(int argc, char **argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s <vk.blob> <proof.bin>\n", argv[0]);
    return 1;
  }
  size_t vk_len, proof_len;
  uint8_t *vk = slurp(argv[1], &vk_len);
  uint8_t *proof = slurp(argv[2], &proof_len);
  if (!vk || !proof) {
    fprintf(stderr, "cannot read inputs\n");
    return 1;
  }
  p2gpu_circuit *c = NULL;
  /* "P2GC": this library's verifier blob; anything else is read as the reference's VK file
   * (VerifierCircuitData::to_bytes through BackendGateSerializer, KeccakGoldilocksConfig) */
  const int own = vk_len >= 4 && vk[0] == 0x50 && vk[1] == 0x32 && vk[2] == 0x47 && vk[3] == 0x43;
  int rc = own ? p2gpu_verifier_create(vk, vk_len, &c) : p2gpu_verifier_create_plonky2(vk, vk_len, 0, &c);
  if (rc) {
    fprintf(stderr, "p2gpu_verifier_create: %d: %s\n", rc, p2gpu_last_error());
    return 2;
  }
  int is_hex = proof_len > 0;
  size_t digits = 0;
  for (size_t i = 0; i < proof_len && is_hex; i++) {
    if (isxdigit(proof[i])) digits++;
    else if (!isspace(proof[i])) is_hex = 0;
  }
  if (is_hex && digits % 2 == 0) {
    uint8_t *bin = malloc(digits / 2 + 1);
    size_t n = 0;
    int hi = -1;
    for (size_t i = 0; i < proof_len; i++) {
      if (!isxdigit(proof[i])) continue;
      int v = isdigit(proof[i]) ? proof[i] - '0' : (tolower(proof[i]) - 'a' + 10);
      if (hi < 0) hi = v;
      else {
        bin[n++] = (uint8_t)(hi << 4 | v);
        hi = -1;
      }
    }
    rc = p2gpu_verify_compressed(c, bin, n);
    free(bin);
  } else {
    rc = p2gpu_verify(c, proof, proof_len);
  }
  if (rc) fprintf(stderr, "p2gpu_verify: %d: %s\n", rc, p2gpu_last_error());
  else fprintf(stderr, "proof accepted (%zu bytes)\n", proof_len);
  p2gpu_circuit_destroy(c);
  free(vk);
  free(proof);
  return rc ? 3 : 0;
}
