/* Multi-tensor kernels of the FP8 weight-update path: one launch walks lists of tensors.
 * Replaces the part of reference .../include/transformer_engine/multi_tensor.h that an
 * "FP8 primary weights" optimizer step calls; the other fused optimizers
 * (plain Adam, SGD, L2 norm) are outside the FP8 linear path and are not provided.
 *
 * tensor_lists is [num_tensor_lists][num_tensors_per_list]; list 0 gives the element counts and
 * the data lists must match it element for element.  chunk_size is the number of elements one
 * workgroup processes.  Zero-sized tensors are an error, as in the reference
 * (multi_tensor_apply.cuh:78-80).
 */
#ifndef TE_AMD_MULTI_TENSOR_H_
#define TE_AMD_MULTI_TENSOR_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref multi_tensor.h:243 (multi_tensor/scale.cu:150).  lists: in, out (fp32 / fp16 / bf16 each).
 * out = float(in) * scale; *is_infinite (int32) is set to 1 when any input is Inf or NaN. */
void nvte_multi_tensor_scale_cuda(int chunk_size, NVTETensor is_infinite, NVTETensor **tensor_lists,
                                  const size_t num_tensor_lists, const size_t num_tensors_per_list,
                                  float scale, hipStream_t stream);
/* ref multi_tensor.h:259 (scale.cu:161).  Same with the factor read from a device fp32 tensor. */
void nvte_multi_tensor_scale_tensor_cuda(int chunk_size, NVTETensor is_infinite,
                                         NVTETensor **tensor_lists, const size_t num_tensor_lists,
                                         const size_t num_tensors_per_list, NVTETensor scale,
                                         hipStream_t stream);

/* ref multi_tensor.h:278 (multi_tensor/compute_scale.cu:107).  lists: amax, scale, scale_inv
 * (fp32).  Per element: scale = max_fp8 / max(amax, epsilon) with the recipe's special cases
 * (recipe_common.cuh:14-49; an infinite quotient becomes FLT_MAX, power-of-two rounding on
 * request), scale_inv = 1 / scale.  noop_flag is accepted and not read, as in the reference. */
void nvte_multi_tensor_compute_scale_and_scale_inv_cuda(int chunk_size, NVTETensor noop_flag,
                                                        NVTETensor **tensor_lists,
                                                        const size_t num_tensor_lists,
                                                        const size_t num_tensors_per_list,
                                                        float max_fp8, int force_pow_2_scales,
                                                        float epsilon, hipStream_t stream);
/* ref multi_tensor.h:295 (compute_scale.cu:122).  lists: amax (bf16), scale_inv (uint8 / e8m0).
 * scale_inv = e8m0 exponent of amax / 448 rounded up to a power of two. */
void nvte_multi_tensor_compute_scale_inv_e8m0_cuda(int chunk_size, NVTETensor **tensor_lists,
                                                   const size_t num_tensor_lists,
                                                   const size_t num_tensors_per_list,
                                                   hipStream_t stream);

/* ref multi_tensor.h:144 (multi_tensor/adam.cu:1005).  lists: g (fp32/fp16/bf16), p_fp8 (bytes),
 * m, v, p_master (fp32), scale, amax, scale_inv (fp32 [1] each; their data pointers may be null).
 * Adam on the fp32 master copy (mode 0: L2 term added to the gradient, mode 1: decoupled weight
 * decay; bias_correction 1 divides by 1 - beta^step), then p_fp8 = cast(p_master * scale),
 * amax = max(amax, max |p_master|), scale_inv = 1 / scale. */
void nvte_multi_tensor_adam_fp8_cuda(int chunk_size, NVTETensor noop_flag, NVTETensor **tensor_lists,
                                     const size_t num_tensor_lists, const size_t num_tensors_per_list,
                                     const float lr, const float beta1, const float beta2,
                                     const float epsilon, const int step, const int mode,
                                     const int bias_correction, const float weight_decay,
                                     const enum NVTEDType fp8_dtype, hipStream_t stream);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif
