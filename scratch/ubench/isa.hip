#include <hip/hip_runtime.h>
#include "../../acvm-backend-plonky2_amd/csrc/gl.hpp"
using namespace p2;
__global__ void k_add(gl_t *o, const gl_t *a, const gl_t *b) { int i = threadIdx.x; o[i] = gl_add(a[i], b[i]); }
__global__ void k_sub(gl_t *o, const gl_t *a, const gl_t *b) { int i = threadIdx.x; o[i] = gl_sub(a[i], b[i]); }
__global__ void k_mul(gl_t *o, const gl_t *a, const gl_t *b) { int i = threadIdx.x; o[i] = gl_mul(a[i], b[i]); }
