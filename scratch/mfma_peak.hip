// Pure MFMA issue-rate probe: 8 waves per workgroup (2 per SIMD), 8 independent accumulators.
// mode 0: constant small-integer operands; mode 1: pseudo-random operand bits (realistic toggling).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode) {
  v8i a[4], b[2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) {
    unsigned h = hash(threadIdx.x * 131 + i * 17 + j + blockIdx.x * 7919);
    // keep exponents moderate so that products stay finite: clear the top exponent bit of each byte
    a[i][j] = mode ? (int)(h & 0xBFBFBFBFu) : (i + j);
  }
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) {
    unsigned h = hash(threadIdx.x * 977 + i * 29 + j + 12345);
    b[i][j] = mode ? (int)(h & 0xBFBFBFBFu) : (i * j);
  }
  v16f acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i >> 1], b[i & 1], acc[i], 0, 0, 0, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float *out; (void)hipMalloc(&out, 4096 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode : {0, 1, 0, 1}) for (int iters : {20000, 100000}) {
    int wg = 512;
    k<<<wg, 512>>>(out, iters, mode); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<<<wg, 512>>>(out, iters, mode); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 32 * 32 * 64 * 8.0 * iters * 8 * wg;
    printf("mode %d iters %d: %.3f ms  %.0f TFLOP/s\n", mode, iters, ms, fl / ms / 1e9);
  }
  return 0;
}
