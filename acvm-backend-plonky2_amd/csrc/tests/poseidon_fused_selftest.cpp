// poseidon_fused_selftest.cpp -- host-side check of poseidon.hpp's fused partial layers (POSEIDON_FUSED: M, M P M, M P M P M) and of
// the poseidon_device_constants form they run on: the permutation computed the way the device computes it -- rounds 0-2 plain, the 23
// linear layers between the S-boxes of round 3 and those of round 26 as seven triples and one pair, integer row sums over the 32-bit
// halves of the words, one reduction per row -- against poseidon_permute_host (plain rounds, the form the oracle and plonky2's
// vectors pin), on the vectors given on stdin and on random / extreme states.  Also: every half-row sum stays below 2^58.
// Built and run by tests/test_oracle_primitives.py with g++ (no GPU, no HIP).
//   stdin: lines of 12 decimal words; stdout: per line the 12 output words of the fused form; exit code 1 on any mismatch.
#include <cstdio>
#include <cstdlib>
#include "../poseidon.hpp"

using namespace p2;

static gl_t rc[360], rcd[360];
static uint64_t max_half_sum = 0;

static gl_t fin(unsigned __int128 lo, unsigned __int128 hi) {
  if ((uint64_t)(lo >> 64) || (uint64_t)(hi >> 64)) { printf("half sum beyond 64 bits\n"); exit(2); }
  if ((uint64_t)lo > max_half_sum) max_half_sum = (uint64_t)lo;
  if ((uint64_t)hi > max_half_sum) max_half_sum = (uint64_t)hi;
  const unsigned __int128 v = lo + (hi << 32);
  return gl_reduce128((uint64_t)v, (uint64_t)(v >> 64));
}
template <int APPS>
static void fused(gl_t st[12], gl_t c1, gl_t c2) {
  unsigned __int128 lo = 0, hi = 0;
  for (int i = 0; i < 12; i++) { lo += (unsigned __int128)(uint32_t)st[i] * POSEIDON_FUSED.m[0][i]; hi += (unsigned __int128)(st[i] >> 32) * POSEIDON_FUSED.m[0][i]; }
  const gl_t s1 = poseidon_sbox(gl_add(fin(lo, hi), c1));
  gl_t s2 = 0;
  if (APPS == 3) {
    lo = (unsigned __int128)(uint32_t)s1 * POSEIDON_FUSED.m[0][0]; hi = (unsigned __int128)(s1 >> 32) * POSEIDON_FUSED.m[0][0];
    for (int i = 0; i < 12; i++) { lo += (unsigned __int128)(uint32_t)st[i] * POSEIDON_FUSED.mpm[0][i]; hi += (unsigned __int128)(st[i] >> 32) * POSEIDON_FUSED.mpm[0][i]; }
    s2 = poseidon_sbox(gl_add(fin(lo, hi), c2));
  }
  gl_t w[12];
  for (int r = 0; r < 12; r++) {
    if (APPS == 3) {
      lo = (unsigned __int128)(uint32_t)s1 * POSEIDON_FUSED.mpm[r][0] + (unsigned __int128)(uint32_t)s2 * POSEIDON_FUSED.m[r][0];
      hi = (unsigned __int128)(s1 >> 32) * POSEIDON_FUSED.mpm[r][0] + (unsigned __int128)(s2 >> 32) * POSEIDON_FUSED.m[r][0];
    } else {
      lo = (unsigned __int128)(uint32_t)s1 * POSEIDON_FUSED.m[r][0];
      hi = (unsigned __int128)(s1 >> 32) * POSEIDON_FUSED.m[r][0];
    }
    for (int i = 0; i < 12; i++) {
      const uint32_t k = APPS == 3 ? POSEIDON_FUSED.mpmpm[r][i] : POSEIDON_FUSED.mpm[r][i];
      lo += (unsigned __int128)(uint32_t)st[i] * k;
      hi += (unsigned __int128)(st[i] >> 32) * k;
    }
    w[r] = fin(lo, hi);
  }
  for (int i = 0; i < 12; i++) st[i] = w[i];
}
static void permute_fused(gl_t st[12]) {
  for (int r = 0; r < 3; r++) {
    for (int i = 0; i < 12; i++) st[i] = poseidon_sbox(gl_add(st[i], rcd[12 * r + i]));
    poseidon_mds(st);
  }
  for (int i = 0; i < 12; i++) st[i] = poseidon_sbox(gl_add(st[i], rcd[36 + i]));
  for (int r0 = 3; r0 < 24; r0 += 3) {
    fused<3>(st, rcd[12 * (r0 + 1)], rcd[12 * (r0 + 2)]);
    st[0] = poseidon_sbox(gl_add(st[0], rcd[12 * (r0 + 3)]));
  }
  fused<2>(st, rcd[12 * 25], 0);
  for (int r = 26; r < 30; r++) {
    for (int i = 0; i < 12; i++) st[i] = poseidon_sbox(gl_add(st[i], rcd[12 * r + i]));
    poseidon_mds(st);
  }
}
static uint64_t sm = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  uint64_t z = (sm += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
int main() {
  poseidon_round_constants_host(rc);
  poseidon_device_constants(rc, rcd);
  long bad = 0;
  unsigned long long x[12];
  while (scanf("%llu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu", &x[0], &x[1], &x[2], &x[3], &x[4], &x[5], &x[6], &x[7], &x[8], &x[9], &x[10], &x[11]) == 12) {
    gl_t a[12], b[12];
    for (int i = 0; i < 12; i++) a[i] = b[i] = (gl_t)x[i];
    poseidon_permute_host(a, rc);
    permute_fused(b);
    for (int i = 0; i < 12; i++) { bad += a[i] != b[i]; printf("%llu%c", (unsigned long long)b[i], i == 11 ? '\n' : ' '); }
  }
  const gl_t edge[] = {0, 1, GL_P - 1, GL_P - 2, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, 0x7FFFFFFFFFFFFFFFull};
  for (int it = 0; it < 20000; it++) {
    gl_t a[12], b[12];
    for (int i = 0; i < 12; i++) {
      const uint64_t r = rnd();
      a[i] = b[i] = (it & 1) ? r % GL_P : ((r & 3) ? edge[(r >> 8) % 8] : (r >> 2) % GL_P);
    }
    poseidon_permute_host(a, rc);
    permute_fused(b);
    for (int i = 0; i < 12; i++) bad += a[i] != b[i];
  }
  printf("mismatches %ld max_half_sum_bits %d\n", bad, 64 - __builtin_clzll(max_half_sum | 1));
  return bad ? 1 : 0;
}
