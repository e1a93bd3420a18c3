// phi_selftest.cpp -- host-side check of glphi.hpp (the signed a + b*2^32 arithmetic under the NTT
// butterfly networks): every network shape the kernels instantiate, against a from-the-definition DFT
// in canonical arithmetic, on random AND extreme inputs (0, 1, p - 1, 2^64 - 1 as non-canonical input,
// 2^32 - 1, 2^32, ...), plus the run-time magnitude of every component against the compile-time
// bounds.  Built and run by tests/test_ntt_phi.py with g++ (no GPU, no HIP).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../glphi.hpp"

using namespace p2;

static uint64_t st = 0x1234567;
static uint64_t rnd() {
  uint64_t z = (st += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static const uint64_t EDGE[] = {0, 1, 2, GL_P - 1, GL_P - 2, GL_P, GL_P + 1, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFull, 0x100000000ull,
                                0xFFFFFFFF00000000ull, 0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFEFFFFFFFFull};
static uint64_t pick(int mode) {
  if (mode == 0) return rnd();
  if (mode == 1) return EDGE[rnd() % (sizeof EDGE / sizeof EDGE[0])];
  return (rnd() & 1) ? rnd() : EDGE[rnd() % (sizeof EDGE / sizeof EDGE[0])];
}
static long failures = 0;
#define CHECK(c, ...) do { if (!(c)) { if (failures < 20) { printf(__VA_ARGS__); printf("\n"); } failures++; } } while (0)

static gl_t canon(uint64_t x) { return x >= GL_P ? x - GL_P : x; }
static uint64_t absu(int64_t x) { return x < 0 ? (uint64_t)0 - (uint64_t)x : (uint64_t)x; }

template <int LOGR, int DIT, bool INV, bool MUL>
static void test_net(int iters) {
  constexpr int R = 1 << LOGR;
  constexpr PhiBound in = MUL ? phi_bound_mul() : phi_bound_from();
  typedef PhiNet<LOGR, DIT, INV, in.ma, in.mb> Net;
  int norms = 0;
  for (int l = 0; l < 4; l++) for (int i = 0; i < 16; i++) norms += Net::sched.pre[l][i] + Net::sched.mid[l][i];
  // w_R = 2^(192 / R * ...) : w_16 = 2^12, i.e. w_{2^L} = 2^(96 >> (L - 1)) ; inverse = 2^(192 - e)
  const int e1 = LOGR ? ((96 >> (LOGR - 1)) % 192) : 0;
  gl_t w = gl_pow(2, INV ? (192 - e1) % 192 : e1);
  for (int it = 0; it < iters; it++) {
    const int mode = it % 3;
    uint64_t x[R], tw[R];
    phi_t v[R];
    gl_t xin[R];
    for (int i = 0; i < R; i++) {
      x[i] = pick(mode);
      tw[i] = pick(mode);
      if (MUL) {
        v[i] = phi_mul_u64(x[i], tw[i]);
        xin[i] = gl_mul(canon(x[i]), canon(tw[i]));
        CHECK(absu(v[i].a) <= in.ma && absu(v[i].b) <= in.mb, "phi_mul_u64 bound");
      } else {
        v[i] = phi_from(x[i]);
        xin[i] = canon(x[i]);
      }
      CHECK(canon(phi_to_u64(v[i])) == xin[i], "conversion round trip");
    }
    Net::run(v);
    for (int i = 0; i < R; i++)
      CHECK(absu(v[i].a) <= Net::sched.out.ma && absu(v[i].b) <= Net::sched.out.mb, "output exceeds the compile-time bound (LOGR %d DIT %d)", LOGR, DIT);
    // reference: y[k] = sum_j xin'[j] w^(jk); DIT takes bit-reversed input, DIF gives bit-reversed output
    for (int k = 0; k < R; k++) {
      gl_t acc = 0;
      for (int j = 0; j < R; j++) {
        const int src = DIT ? (int)bitrev32(j, LOGR) : j;
        acc = gl_add(acc, gl_mul(xin[src], gl_pow(w, (uint64_t)j * k)));
      }
      const int dst = DIT ? k : (int)bitrev32(k, LOGR);
      const uint64_t got = phi_to_u64(v[dst]);
      CHECK(canon(got) == acc, "DFT mismatch LOGR %d DIT %d INV %d MUL %d out %d: got %llu want %llu", LOGR, DIT, (int)INV, (int)MUL, k,
            (unsigned long long)canon(got), (unsigned long long)acc);
    }
  }
  printf("net LOGR=%d DIT=%d INV=%d MUL=%d: %d normalisations, output bound 2^%.1f / 2^%.1f\n", LOGR, DIT, (int)INV, (int)MUL, norms,
         __builtin_log2((double)Net::sched.out.ma), __builtin_log2((double)Net::sched.out.mb));
}

template <int E>
static void test_pow2_one() {
  for (int it = 0; it < 2000; it++) {
    phi_t v;
    constexpr int room = 60 - (E % 32);  // |a|, |b| < 2^room: the shifted, phi-rotated value stays inside int64
    v.a = (int64_t)(rnd() >> (64 - room)) - ((int64_t)1 << (room - 1));
    v.b = (int64_t)(rnd() >> (64 - room)) - ((int64_t)1 << (room - 1));
    if (it % 7 == 0 && room >= 34) v = phi_from(pick(1));
    const gl_t want = gl_mul(canon(phi_to_u64(v)), gl_pow(2, E));
    CHECK(canon(phi_to_u64(phi_mul_pow2<E>(v))) == want, "phi_mul_pow2<%d>", E);
    CHECK(canon(phi_to_u64(phi_norm(phi_mul_pow2<E>(v)))) == want, "phi_norm after phi_mul_pow2<%d>", E);
  }
}
template <int E>
static void test_pow2_all() {
  if constexpr (E < 192) {
    test_pow2_one<E>();
    test_pow2_all<E + 4>();
  }
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  // conversions at the extremes of what phi_to_u64 accepts
  for (int it = 0; it < 200000; it++) {
    phi_t v;
    const int sh = (int)(rnd() % 62);
    v.a = (int64_t)(rnd() >> 1 >> sh) * ((rnd() & 1) ? 1 : -1);
    v.b = (int64_t)(rnd() >> 2 >> (rnd() % 61)) * ((rnd() & 1) ? 1 : -1);
    if (it % 5 == 0) { v.a = (rnd() & 1) ? INT64_MAX : INT64_MIN; }
    if (it % 11 == 0) { v.b = ((rnd() & 1) ? 1 : -1) * (int64_t)(PHI_MAX); }
    // reference through 128-bit integers
    __int128 V = (__int128)v.a + (__int128)v.b * ((__int128)1 << 32);
    __int128 P = (__int128)GL_P;
    __int128 m = V % P;
    if (m < 0) m += P;
    CHECK(canon(phi_to_u64(v)) == (uint64_t)m, "phi_to_u64(%lld, %lld)", (long long)v.a, (long long)v.b);
    const phi_t nv = phi_norm(v);
    CHECK(canon(phi_to_u64(nv)) == (uint64_t)m, "phi_norm value");
    const PhiBound nb = phi_bound_norm(PhiBound{absu(v.a), absu(v.b)});
    CHECK(absu(nv.a) <= nb.ma && absu(nv.b) <= nb.mb, "phi_norm bound");
  }
  test_pow2_all<0>();
  test_pow2_one<12>(); test_pow2_one<60>(); test_pow2_one<124>(); test_pow2_one<188>(); test_pow2_one<191>(); test_pow2_one<31>(); test_pow2_one<33>();
  test_net<4, 1, false, true>(iters);  test_net<4, 1, false, false>(iters);
  test_net<3, 1, false, true>(iters);  test_net<3, 1, false, false>(iters);
  test_net<2, 1, false, true>(iters);  test_net<2, 1, false, false>(iters);
  test_net<1, 1, false, true>(iters);  test_net<1, 1, false, false>(iters);
  test_net<4, 0, true, false>(iters);  test_net<3, 0, true, false>(iters);
  test_net<2, 0, true, false>(iters);  test_net<1, 0, true, false>(iters);
  test_net<4, 1, true, true>(iters);   test_net<4, 0, false, false>(iters);
  test_net<3, 1, true, false>(iters);  test_net<3, 0, false, false>(iters);
  printf("failures: %ld\n", failures);
  return failures ? 1 : 0;
}
