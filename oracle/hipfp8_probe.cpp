// TEST INFRASTRUCTURE.  Host-side probe of ROCm's own OCP FP8 conversions
// (hip/hip_fp8.h, software path, saturating) used to pin oracle/te_oracle.c's
// float -> FP8 including the saturation the torch CPU cast does not do.
#include <hip/hip_fp8.h>
#include <cstdint>
#include <cstddef>

extern "C" {
void probe_e4m3(const float *in, uint8_t *out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    __hip_fp8_e4m3 v(in[i]);   // default: __HIP_SATFINITE
    out[i] = v.__x;
  }
}
void probe_e5m2(const float *in, uint8_t *out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    __hip_fp8_e5m2 v(in[i]);
    out[i] = v.__x;
  }
}
}
