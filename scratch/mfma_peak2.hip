// MFMA shape probe on random operand bits: 32x32x64 vs 16x16x128 (f8f6f4), 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int SHAPE>
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode) {
  v8i a[4], b[2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) {
    unsigned h = hash(threadIdx.x * 131 + i * 17 + j + blockIdx.x * 7919);
    a[i][j] = mode ? (int)(h & 0xBFBFBFBFu) : (i + j);
  }
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) {
    unsigned h = hash(threadIdx.x * 977 + i * 29 + j + 12345);
    b[i][j] = mode ? (int)(h & 0xBFBFBFBFu) : (i * j);
  }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    v16f acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i >> 1], b[i & 1], acc[i], 0, 0, 0, 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    v4f acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[(i >> 1) & 3], b[i & 1], acc[i], 0, 0, 0, 0, 0, 0);
    }
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float *out; (void)hipMalloc(&out, 4096 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int wg = 512;
  for (int round = 0; round < 2; ++round)
  for (int shape : {32, 16}) for (int mode : {0, 1}) {
    int iters = 40000;
    auto launch = [&] { if (shape == 32) k<32><<<wg, 512>>>(out, iters, mode); else k<16><<<wg, 512>>>(out, iters / 2, mode); };
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double fl = shape == 32 ? 2.0 * 32 * 32 * 64 * 8.0 * iters * 8 * wg : 2.0 * 16 * 16 * 128 * 32.0 * (iters / 2) * 8 * wg;
    printf("shape %dx%d mode %d: %.3f ms  %.0f TFLOP/s\n", shape, shape, mode, ms, fl / ms / 1e9);
  }
  return 0;
}
