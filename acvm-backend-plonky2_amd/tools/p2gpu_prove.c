/*
 * p2gpu-prove -- stand-alone caller of the C ABI (include/p2gpu.h), plain C, no Python:
 *
 *     p2gpu-prove <circuit.blob> <wires.bin> <proof.bin> [public_inputs.bin] [--routed] [--vk <vk.blob>]
 *                 [--reference-format] [--timing]
 *
 * The counterpart of `plonky2-backend prove -b <acir> -w <witness> -o <proof>`
 * (plonky2-backend/src/argument_parsing.rs:36-41 -> actions/prove_action.rs:27-43) below the
 * translation layer: the circuit blob is what the Rust side exports once per circuit and
 * wires.bin is the dense witness matrix [num_wires][n] of little-endian u64 (with --sparse <ncols> <row>: its
 * first ncols columns followed by the value each other column holds in <row>, p2gpu_prove_sparse; with --routed only
 * the [num_routed_wires][n] routed columns; the rest is derived on the GPU).  Writes the
 * uncompressed ProofWithPublicInputs bytes; --vk also writes the verifier's share of the circuit for
 * p2gpu-verify (the reference's `write_vk`, actions/write_vk_action.rs:65-81); --reference-format
 * writes what the reference's CLI writes instead: hex text of the COMPRESSED proof
 * (prove_action.rs:38-42,75-78).  --timing prints one JSON line on stdout with the wall time of every stage of this
 * process -- reading the inputs, p2gpu_init, p2gpu_circuit_create, the first prove, a second prove on the warm handle --
 * i.e. what one invocation of the reference's CLI pays, which re-translates and re-builds per call (prove_action.rs:27-43);
 * bench.py reports it as `cold_process`.  Exit code 0 = ok, 1 = usage/IO, 2 = library error.
 */
#include "../../include/p2gpu.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

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
  int routed = 0, npos = 0, ref_format = 0, sparse = 0, timing = 0;
  const double t_start = now_ms();
  unsigned long sp_cols = 0, sp_row = 0;
  const char *pos[4] = {0, 0, 0, 0}, *vk_path = NULL;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--routed")) routed = 1;
    else if (!strcmp(argv[i], "--reference-format")) ref_format = 1;
    else if (!strcmp(argv[i], "--timing")) timing = 1;
    else if (!strcmp(argv[i], "--sparse") && i + 2 < argc) {
      sparse = 1;
      sp_cols = strtoul(argv[++i], NULL, 10);
      sp_row = strtoul(argv[++i], NULL, 10);
    }
    else if (!strcmp(argv[i], "--vk") && i + 1 < argc) vk_path = argv[++i];
    else if (npos < 4) pos[npos++] = argv[i];
  }
  if (npos < 3) {
    fprintf(stderr, "usage: %s <circuit.blob> <wires.bin> <proof.bin> [public_inputs.bin] [--routed] [--sparse <ncols> <row>] [--vk <vk.blob>] [--reference-format]\n", argv[0]);
    return 1;
  }
  size_t blob_len, wires_len, pi_len = 0;
  uint8_t *blob = slurp(pos[0], &blob_len);
  uint64_t *wires = slurp(pos[1], &wires_len);
  uint64_t *pis = pos[3] ? slurp(pos[3], &pi_len) : NULL;
  if (!blob || !wires || (pos[3] && !pis)) {
    fprintf(stderr, "cannot read inputs\n");
    return 1;
  }
  /* the wire matrix must have exactly the shape the blob header announces (p2gpu_prove reads
   * 8 * cols * 2^d bytes from this pointer): header words [2] degree_bits, [3] num_wires, [4] num_routed, [24] #PI */
  uint32_t magic = 0;
  if (blob_len >= 4) memcpy(&magic, blob, 4);
  if (blob_len >= 256 && magic == 0x43473250u) { /* anything else: let p2gpu_circuit_create report it */
    uint32_t h[64];
    memcpy(h, blob, sizeof h);
    const uint64_t cols = routed ? h[4] : h[3];
    if (sparse) { /* [ncols][2^d] dense columns, then one value for each of the num_wires - ncols others */
      if (sp_cols > h[3] || h[2] >= 40 || wires_len != 8ull * (sp_cols * (1ull << h[2]) + (h[3] - sp_cols))) {
        fprintf(stderr, "%s: %zu bytes, --sparse %lu needs [%lu][2^%u] + %lu u64\n", pos[1], wires_len, sp_cols, sp_cols, h[2],
                (unsigned long)h[3] - sp_cols);
        return 1;
      }
    } else if (h[2] < 40 && wires_len != 8ull * cols * (1ull << h[2])) {
      fprintf(stderr, "%s: %zu bytes, but the circuit needs %s[%u][2^%u] u64 = %llu bytes\n", pos[1], wires_len,
              routed ? "routed" : "wires", (unsigned)cols, h[2], (unsigned long long)(8ull * cols * (1ull << h[2])));
      return 1;
    }
    if (pi_len != 8ull * h[24]) {
      fprintf(stderr, "public inputs: %zu bytes, the circuit has %u public inputs\n", pi_len, h[24]);
      return 1;
    }
  }
  const double t_read = now_ms();
  p2gpu_circuit *c = NULL;
  int rc = p2gpu_init(NULL, 0); /* device 0: HIP runtime start-up + code-object load happen here and in the first launches */
  if (rc) {
    fprintf(stderr, "p2gpu_init: %d: %s\n", rc, p2gpu_last_error());
    return 2;
  }
  const double t_init = now_ms();
  rc = p2gpu_circuit_create(blob, blob_len, &c);
  const double t_create = now_ms();
  if (rc) {
    fprintf(stderr, "p2gpu_circuit_create: %d: %s\n", rc, p2gpu_last_error());
    return 2;
  }
  if (vk_path) {
    size_t vk_len = 0;
    /* with --reference-format the key goes out as the reference's VK file too (UNPINNED layout, include/p2gpu.h) */
    int (*export_vk)(const p2gpu_circuit *, uint8_t *, size_t *) = ref_format ? p2gpu_circuit_export_vk_plonky2 : p2gpu_circuit_export_vk;
    export_vk(c, NULL, &vk_len);
    uint8_t *vk = malloc(vk_len);
    FILE *vf = fopen(vk_path, "wb");
    if (!vk || export_vk(c, vk, &vk_len) || !vf || fwrite(vk, 1, vk_len, vf) != vk_len) {
      fprintf(stderr, "cannot write %s\n", vk_path);
      return 1;
    }
    fclose(vf);
    free(vk);
  }
  size_t cap = p2gpu_proof_size_bound(c), len = cap;
  uint8_t *proof = malloc(cap);
  p2gpu_timings t;
  if (sparse) {
    uint32_t h[64];
    memcpy(h, blob, sizeof h);
    rc = p2gpu_prove_sparse(c, wires, (uint32_t)sp_cols, wires + sp_cols * (1ull << h[2]), (uint32_t)sp_row, pis, (uint32_t)(pi_len / 8),
                            proof, &len, &t);
  } else
    rc = routed ? p2gpu_prove_routed(c, wires, pis, (uint32_t)(pi_len / 8), proof, &len, &t)
                : p2gpu_prove(c, wires, pis, (uint32_t)(pi_len / 8), proof, &len, &t);
  if (rc) {
    fprintf(stderr, "p2gpu_prove: %d: %s\n", rc, p2gpu_last_error());
    p2gpu_circuit_destroy(c);
    return 2;
  }
  const double t_prove = now_ms();
  double warm_ms = -1;
  if (timing && !sparse && !routed) { /* the same call again on the warm handle: what a resident service pays per proof */
    size_t len2 = cap;
    uint8_t *proof2 = malloc(cap);
    p2gpu_timings t2;
    const double w0 = now_ms();
    if (proof2 && p2gpu_prove(c, wires, pis, (uint32_t)(pi_len / 8), proof2, &len2, &t2) == 0) warm_ms = now_ms() - w0;
    free(proof2);
  }
  const double t_warm = now_ms();
  FILE *f = fopen(pos[2], "wb");
  if (!f) {
    fprintf(stderr, "cannot write %s\n", pos[2]);
    return 1;
  }
  if (ref_format) {
    size_t clen = 0;
    p2gpu_proof_compress(c, proof, len, NULL, &clen);
    uint8_t *comp = malloc(clen ? clen : 1);
    rc = p2gpu_proof_compress(c, proof, len, comp, &clen);
    if (rc) {
      fprintf(stderr, "p2gpu_proof_compress: %d: %s\n", rc, p2gpu_last_error());
      return 2;
    }
    for (size_t i = 0; i < clen; i++) fprintf(f, "%02x", comp[i]);
    fprintf(stderr, "compressed: %zu -> %zu bytes, written as hex\n", len, clen);
    free(comp);
  } else if (fwrite(proof, 1, len, f) != len) {
    fprintf(stderr, "cannot write %s\n", pos[2]);
    return 1;
  }
  fclose(f);
  fprintf(stderr, "proof: %zu bytes; wires %.2f ms, zs %.2f ms, quotient %.2f ms, openings %.2f ms, fri %.2f ms, h2d %.2f ms\n", len,
          t.wires_commit_ms, t.zs_commit_ms, t.quotient_ms, t.openings_ms, t.fri_ms, t.h2d_ms);
  const double t_write = now_ms();
  p2gpu_circuit_destroy(c);
  if (timing)
    printf("{\"read_inputs_ms\": %.3f, \"p2gpu_init_ms\": %.3f, \"circuit_create_ms\": %.3f, \"first_prove_ms\": %.3f, "
           "\"second_prove_ms\": %.3f, \"write_proof_ms\": %.3f, \"destroy_ms\": %.3f, \"process_ms\": %.3f, "
           "\"cold_process_ms\": %.3f, \"h2d_ms\": %.3f, \"proof_bytes\": %zu}\n",
           t_read - t_start, t_init - t_read, t_create - t_init, t_prove - t_create, warm_ms, t_write - t_warm, now_ms() - t_write,
           now_ms() - t_start, t_prove - t_read, t.h2d_ms, len);
  free(proof);
  free(blob);
  free(wires);
  free(pis);
  return 0;
}
