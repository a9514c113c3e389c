// Pure MFMA issue-rate probe: 8 waves per workgroup (2 per SIMD), 8 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float *out, int iters) {
  v8i a = {(int)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b = {7, 6, 5, 4, 3, 2, 1, (int)threadIdx.x};
  v16f acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i], 0, 0, 0, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float *out; hipMalloc(&out, 4096 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg : {256, 512, 1024}) for (int iters : {2000, 20000}) {
    k<<<wg, 512>>>(out, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<wg, 512>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 32 * 32 * 64 * 8.0 * iters * 8 * wg;
    printf("wg %d iters %d: %.3f ms  %.0f TFLOP/s\n", wg, iters, ms, fl / ms / 1e9);
  }
  return 0;
}
