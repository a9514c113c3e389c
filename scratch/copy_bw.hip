// Streaming copy variants: what HBM rate is reachable for 235 MB in + 235 MB out?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copyk(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (; i + (UNROLL - 1) * 256 < n; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(v[u], out + i + u * 256); else out[i + u * 256] = v[u]; }
  }
}
int main() {
  const size_t bytes = 234881024;  // 8192 x 14336 x 2
  const size_t n = bytes / 16;
  u32x4 *a, *b; (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes);
  (void)hipMemset(a, 1, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 20; ++i) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.1f us  %.0f GB/s\n", name, ms / 50 * 1e3, 2.0 * bytes / (ms / 50) / 1e6);
  };
  for (int rep = 0; rep < 2; ++rep) {
    run("hipMemcpyDtoD", [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    run("u1 full grid", [&] { copyk<1, false><<<n / 256, 256>>>(a, b, n); });
    run("u4 full grid", [&] { copyk<4, false><<<n / 1024, 256>>>(a, b, n); });
    run("u4 grid 2048", [&] { copyk<4, false><<<2048, 256>>>(a, b, n); });
    run("u4 grid 4096", [&] { copyk<4, false><<<4096, 256>>>(a, b, n); });
    run("u8 grid 2048", [&] { copyk<8, false><<<2048, 256>>>(a, b, n); });
    run("u4 nt full grid", [&] { copyk<4, true><<<n / 1024, 256>>>(a, b, n); });
    run("u4 nt grid 2048", [&] { copyk<4, true><<<2048, 256>>>(a, b, n); });
    run("u8 nt grid 4096", [&] { copyk<8, true><<<4096, 256>>>(a, b, n); });
  }
  return 0;
}
