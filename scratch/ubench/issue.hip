// issue.hip -- VALU issue-rate micro-benchmark for gfx950: how many cycles does one SIMD need per
// wave64 instruction, for the f32 FMA the microarchitecture guide quotes (2 cycles) and for the
// integer instructions the prover's kernels are made of?
//
// Every lane runs 8 independent dependency chains of one instruction (inline asm, so the compiler
// cannot fold or reorder anything); a launch puts exactly K waves on every SIMD (256 CUs x K blocks
// of 256 threads).  Two clocks are reported:
//   * cyc/instr (s_memtime): shader cycles between the first and last instruction of a wave,
//     divided by the wave-instructions ALL K waves of that SIMD issued in that time -- independent
//     of the DVFS clock;
//   * lane-instr/s (HIP events): wall-clock throughput of the whole chip, which includes the clock
//     the chip actually sustains under that load.
// Output is committed as profiles/rNN_ubench.txt; bench.py's roofline.issue peak comes from it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

enum { FMA_F32, PK_FMA_F32, ADD_U32, XOR_B32, BITOP3, ALIGNBIT, ADD3, LSHL_ADD_U64, LSHLREV_B64, ADDCO_PAIR, CMP64_CNDMASK,
       MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MUL_U24, NMODES };
static const char *NAMES[NMODES] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_xor_b32", "v_bitop3_b32", "v_alignbit_b32",
                                    "v_add3_u32", "v_lshl_add_u64", "v_lshlrev_b64", "v_add_co+v_addc_co", "v_cmp_lt_u64+v_cndmask",
                                    "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24"};
static const int NINSTR[NMODES] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 1, 1, 1, 1};

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint64_t *cyc, uint64_t seed, int iters) {
  uint64_t a[8];
  uint32_t lo[8], hi[8];
  float f[8];
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v pf[8];
  for (int i = 0; i < 8; i++) {
    a[i] = seed * (i + 3) + threadIdx.x;
    lo[i] = (uint32_t)a[i];
    hi[i] = (uint32_t)(a[i] >> 32) | 1;
    f[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i);
    pf[i] = float2v{f[i], f[i] + 0.5f};
  }
  uint64_t m = seed | 1;
  uint32_t m32 = (uint32_t)seed | 1;
  float fm = 0.999f, fa = 1e-4f;
  float2v pfm = float2v{0.999f, 0.998f}, pfa = float2v{1e-4f, 2e-4f};
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fm), "v"(fa));
      if (MODE == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pf[i]) : "v"(pfm), "v"(pfa));
      if (MODE == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(m32));
      if (MODE == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(lo[i]) : "v"(m32));
      if (MODE == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(lo[i]) : "v"(m32), "v"(hi[i]));
      if (MODE == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(lo[i]) : "v"(hi[i]));
      if (MODE == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(m32), "v"(hi[i]));
      if (MODE == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(m));
      if (MODE == LSHLREV_B64) asm volatile("v_lshlrev_b64 %0, 13, %0" : "+v"(a[i]));
      if (MODE == ADDCO_PAIR)
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo[i]), "+v"(hi[i]) : "v"(m32), "v"(m32) : "vcc");
      if (MODE == CMP64_CNDMASK)
        asm volatile("v_cmp_lt_u64 vcc, %1, %0\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a[i]), "+v"(m), "+v"(lo[i]) : "v"(m32) : "vcc");
      if (MODE == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a[i]) : "v"(lo[i]), "v"(m32) : "s20", "s21");
      if (MODE == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(m32));
      if (MODE == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(m32));
      if (MODE == MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(lo[i]) : "v"(m32));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  uint64_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i] ^ lo[i] ^ ((uint64_t)hi[i] << 32) ^ (uint64_t)__float_as_uint(f[i]) ^ (uint64_t)__float_as_uint(pf[i].x + pf[i].y);
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int M>
static void run(uint64_t *out, uint64_t *cyc, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 4096;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, cyc, 12345ull, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, cyc, 12345ull, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  std::vector<uint64_t> h((size_t)blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  const double wave_instr = (double)iters * 8 * NINSTR[M];  // per wave
  // K waves share the SIMD for the whole interval: cycles per wave-instruction of the SIMD
  const double cyc_per_instr = med / (wave_instr * waves_per_simd);
  const double lane_instr = wave_instr * 64.0 * blocks * 4;
  const double rate = lane_instr / (ms * 1e-3);
  printf("%-24s K=%d  %6.2f cyc/wave-instr/SIMD  %8.3f ms  %6.2f T lane-instr/s  (eff. clock if that cyc count: %.2f GHz)\n", NAMES[M],
         waves_per_simd, cyc_per_instr, ms, rate / 1e12, med / (ms * 1e-3) / 1e9);
}

int main() {
  uint64_t *out, *cyc;
  hipMalloc(&out, 8ull * 256 * 8 * 256);
  hipMalloc(&cyc, 8ull * 256 * 8 * 4);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("# device %s, %d CUs, clockRate %d kHz; peak if 2 cyc/wave-instr: %.1f T lane-instr/s, if 4: %.1f T (at 2.4 GHz)\n", p.gcnArchName,
         p.multiProcessorCount, p.clockRate, 256 * 4 * 32 * 2.4e9 / 1e12, 256 * 4 * 16 * 2.4e9 / 1e12);
  for (int K : {1, 2, 4, 8}) {
    run<FMA_F32>(out, cyc, K);
    run<PK_FMA_F32>(out, cyc, K);
    run<ADD_U32>(out, cyc, K);
    run<XOR_B32>(out, cyc, K);
    run<BITOP3>(out, cyc, K);
    run<ALIGNBIT>(out, cyc, K);
    run<ADD3>(out, cyc, K);
    run<LSHL_ADD_U64>(out, cyc, K);
    run<LSHLREV_B64>(out, cyc, K);
    run<ADDCO_PAIR>(out, cyc, K);
    run<CMP64_CNDMASK>(out, cyc, K);
    run<MAD_U64_U32>(out, cyc, K);
    run<MUL_LO_U32>(out, cyc, K);
    run<MUL_HI_U32>(out, cyc, K);
    run<MUL_U24>(out, cyc, K);
    printf("\n");
  }
  return 0;
}
