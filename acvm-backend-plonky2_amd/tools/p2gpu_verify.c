/*
 * p2gpu-verify -- stand-alone verifier over the C ABI (include/p2gpu.h), plain C, no GPU needed:
 *
 *     p2gpu-verify <vk.blob> <proof.bin>
 *
 * The counterpart of `plonky2-backend verify -k <vk> -p <proof>`
 * (plonky2-backend/src/argument_parsing.rs:49-53 -> actions/verify_action.rs:11-17) on the
 * uncompressed proof bytes p2gpu-prove writes.  <vk.blob> is the verifier's share of the circuit
 * (p2gpu_circuit_export_vk, or `p2gpu-prove --vk <file>`).  Exit code 0 = proof accepted,
 * 1 = usage/IO, 2 = bad blob, 3 = proof rejected (the failed check is printed).
 */
#include "../../include/p2gpu.h"
#include <stdio.h>
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
  int rc = p2gpu_verifier_create(vk, vk_len, &c);
  if (rc) {
    fprintf(stderr, "p2gpu_verifier_create: %d: %s\n", rc, p2gpu_last_error());
    return 2;
  }
  rc = p2gpu_verify(c, proof, proof_len);
  if (rc) fprintf(stderr, "p2gpu_verify: %d: %s\n", rc, p2gpu_last_error());
  else fprintf(stderr, "proof accepted (%zu bytes)\n", proof_len);
  p2gpu_circuit_destroy(c);
  free(vk);
  free(proof);
  return rc ? 3 : 0;
}
