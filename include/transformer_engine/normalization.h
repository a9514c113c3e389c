/* LayerNorm / RMSNorm with (optionally) FP8 output - the producer side of LayerNormLinear
 * (norm + quantize fusion).
 * Mirrors ref common/include/transformer_engine/normalization.h; every function only ENQUEUES on
 * `stream`.  Workspace protocol as in the reference: a call with an EMPTY workspace tensor only
 * writes the required shape/dtype into it and returns. */
#ifndef TE_AMD_NORMALIZATION_H_
#define TE_AMD_NORMALIZATION_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref normalization.h:44.  z = gamma' * (x - mu) * rsigma + beta, gamma' = gamma (+1 when
 * zero_centered_gamma); mu [N], rsigma [N] fp32.  z may be FP32/BF16/FP16, FP8 with per-tensor
 * scaling (scale in, amax out = max|pre-scale value|, scale_inv = 1/scale; a present columnwise
 * buffer receives the byte transpose) or MXFP8 (compact scales). */
void nvte_layernorm_fwd(const NVTETensor x, const NVTETensor gamma, const NVTETensor beta, const float epsilon,
                        NVTETensor z, NVTETensor mu, NVTETensor rsigma, NVTETensor workspace,
                        const int multiprocessorCount, const bool zero_centered_gamma, hipStream_t stream);
/* ref normalization.h:76 */
void nvte_layernorm_bwd(const NVTETensor dz, const NVTETensor x, const NVTETensor mu, const NVTETensor rsigma,
                        const NVTETensor gamma, NVTETensor dx, NVTETensor dgamma, NVTETensor dbeta,
                        NVTETensor workspace, const int multiprocessorCount, const bool zero_centered_gamma,
                        hipStream_t stream);
/* ref normalization.h:107.  z = gamma' * x * rsigma, rsigma = rsqrt(mean(x^2) + epsilon). */
void nvte_rmsnorm_fwd(const NVTETensor x, const NVTETensor gamma, const float epsilon, NVTETensor z,
                      NVTETensor rsigma, NVTETensor workspace, const int multiprocessorCount,
                      const bool zero_centered_gamma, hipStream_t stream);
/* ref normalization.h:138 */
void nvte_rmsnorm_bwd(const NVTETensor dz, const NVTETensor x, const NVTETensor rsigma, const NVTETensor gamma,
                      NVTETensor dx, NVTETensor dgamma, NVTETensor workspace, const int multiprocessorCount,
                      const bool zero_centered_gamma, hipStream_t stream);
/* ref normalization.h:161.  dx += add */
void nvte_rmsnorm_bwd_add(const NVTETensor dz, const NVTETensor x, const NVTETensor add, const NVTETensor rsigma,
                          const NVTETensor gamma, NVTETensor dx, NVTETensor dgamma, NVTETensor workspace,
                          const int multiprocessorCount, const bool zero_centered_gamma, hipStream_t stream);
/* ref normalization.h:170-186.  There is one backend here, so the cuDNN switches pick no other kernels.
 * As in the reference (normalization/common.cpp:226: only its cuDNN backend honours it), a zero-centered
 * gamma is formed as (gamma + 1) rounded to the WEIGHT dtype when both
 * nvte_enable_zero_centered_gamma_in_weight_dtype(true) and the cuDNN switch of that pass are on;
 * otherwise gamma + 1 is formed in fp32. */
void nvte_enable_cudnn_norm_fwd(bool enable);
void nvte_enable_cudnn_norm_bwd(bool enable);
void nvte_enable_zero_centered_gamma_in_weight_dtype(bool enable);

#ifdef __cplusplus
}
#endif
#endif
