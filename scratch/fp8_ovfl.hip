// Does MODE.FP16_OVFL make v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 saturate like the reference's SATFINITE cast?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
__global__ void k(const float *in, unsigned *out_e4, unsigned *out_e5, int n, int ovfl) {
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
  int i = threadIdx.x;
  if (i < n) {
    out_e4[i] = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], in[i], 0, false) & 0xff;
    out_e5[i] = __builtin_amdgcn_cvt_pk_bf8_f32(in[i], in[i], 0, false) & 0xff;
  }
}
int main() {
  float h[] = {0.f, 1.f, 448.f, 449.f, 464.f, 465.f, 480.f, 500.f, 1e30f, INFINITY, -INFINITY, NAN, -500.f, 57344.f, 57345.f, 61440.f, 65536.f, -1e30f, 1e-10f, -449.f};
  const int n = sizeof(h) / sizeof(float);
  float *d; unsigned *o4, *o5, r4[2][64], r5[2][64];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o4, 256); hipMalloc(&o5, 256);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int ov = 0; ov < 2; ++ov) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o4, o5, n, ov);
    hipMemcpy(r4[ov], o4, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r5[ov], o5, n * 4, hipMemcpyDeviceToHost);
  }
  for (int i = 0; i < n; ++i)
    printf("%12g  e4m3 plain %02x ovfl %02x | e5m2 plain %02x ovfl %02x\n", h[i], r4[0][i], r4[1][i], r5[0][i], r5[1][i]);
  return 0;
}
