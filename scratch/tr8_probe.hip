// What does ds_read_b64_tr_b8 deliver?  Lane l supplies LDS address l * STRIDE (+ offset); LDS byte a holds
// (a & 0xff) in pass 0 and (a >> 8) in pass 1, so the two passes give the 16-bit source address of every
// result byte of every lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out, int stride, int pass) {
  __shared__ unsigned char lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = pass ? (i >> 8) : (i & 0xff);
  __syncthreads();
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned addr = (unsigned)(size_t)(lds_u8 *)lds + threadIdx.x * stride;   // LDS address space offset
  v2i r;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[threadIdx.x * 2] = r[0];
  out[threadIdx.x * 2 + 1] = r[1];
}
int main(int argc, char **argv) {
  int stride = argc > 1 ? atoi(argv[1]) : 64;
  unsigned *d, h[2][128];
  if (hipMalloc(&d, 512) != hipSuccess) return 1;
  for (int p = 0; p < 2; ++p) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride, p);
    if (hipMemcpy(h[p], d, 512, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  }
  printf("stride %d: lane -> source (row = addr / stride, col = addr %% stride) of its 8 result bytes\n", stride);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int b = 0; b < 8; ++b) {
      unsigned lo = (h[0][l * 2 + b / 4] >> (8 * (b % 4))) & 0xff, hi = (h[1][l * 2 + b / 4] >> (8 * (b % 4))) & 0xff;
      unsigned a = lo | (hi << 8);
      printf(" (%2u,%2u)", a / stride, a % stride);
    }
    printf("\n");
  }
  return 0;
}
