#include <hip/hip_runtime.h>
__global__ void k(unsigned *o, const unsigned *a) {
  int i = threadIdx.x;
  unsigned x = a[i], y = a[i + 64], z = a[i + 128];
  o[i] = __builtin_amdgcn_bitop3_b32(x, y, z, 0x96) + __builtin_amdgcn_bitop3_b32(x, y, z, 0xD2) + __builtin_amdgcn_alignbit(x, y, 7);
}
