// Activation math shared by activation.hip and scaled_activation.hip (device code only).
#pragma once
#include <hip/hip_runtime.h>

namespace te {

// exp through the hardware 2^x and reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each): the library expf /
// tanhf cost 20-40 instructions per value, which made gelu -> FP8 compute-bound (138 us against 82 us
// for relu on the same tensor).  tanh never appears as such: with p = sigmoid(2u),
//   0.5 (1 + tanh u) = p        and        1 - tanh^2 u = 4 p (1 - p),
// so gelu and dgelu (ref math.h:21-32) are evaluated without the cancellation of 1 + tanh(u) for
// very negative u - p keeps its RELATIVE accuracy there (2^+big -> rcp -> tiny, saturating to 0).
__device__ __forceinline__ float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}


// clamped SwiGLU (GPT-OSS): act(x) = c * sigmoid(alpha c) with c = min(x, limit); its derivative is 0 above
// the limit; the linear half is clamped to [-limit, limit] and shifted by glu_offset, and its gradient is 0
// where the clamp was active (ref math.h:80-98, vectorized_pointwise.h:436-438,543-546)
__device__ __forceinline__ float clamped_act(float x, float limit, float alpha) {
  const float c = fminf(x, limit);
  return c * sigmoid_f(alpha * c);
}
__device__ __forceinline__ float clamped_dact(float x, float limit, float alpha) {
  const float c = fminf(x, limit);
  const float s = sigmoid_f(alpha * c);
  return x <= limit ? alpha * c * s * (1.f - s) + s : 0.f;
}

}  // namespace te
