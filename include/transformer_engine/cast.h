/* Quantize / dequantize entry points.
 * Replaces reference transformer_engine/common/include/transformer_engine/cast.h
 * (definitions in common/cast/cast.cu:21-89).  Dispatch rule (ref
 * common/cast/dispatch/quantize.cuh:63-93): the OUTPUT tensor's scaling mode picks
 * the kernel - per-tensor scaling with a columnwise buffer -> cast+transpose,
 * rowwise only -> flat cast, NVTE_MXFP8_1D_SCALING -> block-scaled cast.
 */
#ifndef TE_AMD_CAST_H_
#define TE_AMD_CAST_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref cast.h:90 */
void nvte_quantize(const NVTETensor input, NVTETensor output, hipStream_t stream);
/* ref cast.h:115.  noop = fp32[1] on device; value 1.0f => kernel leaves outputs untouched. */
void nvte_quantize_noop(const NVTETensor input, NVTETensor output, NVTETensor noop,
                        hipStream_t stream);
/* ref cast.h:125 */
void nvte_quantize_v2(const NVTETensor input, NVTETensor output,
                      const NVTEQuantizationConfig quant_config, hipStream_t stream);
/* ref cast.h:145.  Two-call protocol: a call whose workspace has no data pointer only
 * fills in the workspace shape/dtype and returns (ref transpose.h:48-49). */
void nvte_quantize_dbias(const NVTETensor input, NVTETensor output, NVTETensor dbias,
                         NVTETensor workspace, hipStream_t stream);
/* ref cast.h:419 */
void nvte_dequantize(const NVTETensor input, NVTETensor output, hipStream_t stream);
/* ref cast.h:440 */
void nvte_multi_tensor_quantize(const NVTETensor *inputs, NVTETensor *outputs,
                                const NVTEQuantizationConfig quant_config, const size_t num_tensors,
                                hipStream_t stream);

/* ref cast.h:102.  Grouped quantize: every member tensor of `input` is quantized with ITS entry of
 * output's per-tensor `scale` array (FP8 per-tensor scaling); rowwise bytes keep the flat layout,
 * columnwise bytes are each member's transpose at the member's own offset.  Shapes are read on the
 * device (first_dims / last_dims / tensor_offsets), honours the config's noop flag.  amax and
 * scale_inv are NOT touched (nvte_group_compute_amax_with_config / _scale_from_amax own them).
 * MXFP8 grouped output (ref cast/dispatch/quantize.cuh:470 -> cast/mxfp8/group_quantize_mxfp8.cuh):
 * same-shape and varying-first-dim groups, member first dims and the last dim multiples of 128; the
 * group is quantized as ONE [sum first, last] tensor as in the reference (rowwise scales at the
 * member's global rows); columnwise output in the reference layout or, if
 * nvte_amd_set_grouped_columnwise_transposed() was set on `output`, per-member transposed (amd_ext.h);
 * compact or GEMM-swizzled scales per output's with_gemm_swizzled_scales. */
void nvte_group_quantize(const NVTEGroupedTensor input, NVTEGroupedTensor output,
                         const NVTEQuantizationConfig quant_config, hipStream_t stream);
/* ref cast.h:167.  Grouped cast + dbias: output as nvte_group_quantize (FP8 per-tensor scaling with
 * output.scale[i] per member, or MXFP8), dbias [num_tensors, last] = column sums of every member, in the
 * input dtype.  Same-shape and varying-first-dim groups (the reference: "not yet supported for grouped tensors
 * with a varying last dimension").  Two-call protocol: an empty workspace only receives its shape. */
void nvte_group_quantize_dbias(const NVTEGroupedTensor input, NVTEGroupedTensor output, NVTEGroupedTensor dbias,
                               NVTETensor workspace, const NVTEQuantizationConfig quant_config, hipStream_t stream);
/* ref cast.h:214,262,310,358,406.  output = cast(input * dact(act_input)), dbias = its column sums per member. */
void nvte_group_quantize_dbias_dgelu(const NVTEGroupedTensor input, const NVTEGroupedTensor act_input,
                                     NVTEGroupedTensor output, NVTEGroupedTensor dbias, NVTETensor workspace,
                                     hipStream_t stream);
void nvte_group_quantize_dbias_dsilu(const NVTEGroupedTensor input, const NVTEGroupedTensor act_input,
                                     NVTEGroupedTensor output, NVTEGroupedTensor dbias, NVTETensor workspace,
                                     hipStream_t stream);
void nvte_group_quantize_dbias_drelu(const NVTEGroupedTensor input, const NVTEGroupedTensor act_input,
                                     NVTEGroupedTensor output, NVTEGroupedTensor dbias, NVTETensor workspace,
                                     hipStream_t stream);
void nvte_group_quantize_dbias_dqgelu(const NVTEGroupedTensor input, const NVTEGroupedTensor act_input,
                                      NVTEGroupedTensor output, NVTEGroupedTensor dbias, NVTETensor workspace,
                                      hipStream_t stream);
void nvte_group_quantize_dbias_dsrelu(const NVTEGroupedTensor input, const NVTEGroupedTensor act_input,
                                      NVTEGroupedTensor output, NVTEGroupedTensor dbias, NVTETensor workspace,
                                      hipStream_t stream);
/* ref cast.h:429 */
void nvte_group_dequantize(const NVTEGroupedTensor input, NVTEGroupedTensor output, hipStream_t stream);

/* ref cast.h:190-384.  output = cast(input * act'(act_input)), dbias[c] = sum_r of that product (in the
 * input dtype), ONE pass over the data.  Two-call workspace protocol as nvte_quantize_dbias. */
void nvte_quantize_dbias_dgelu(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                               NVTETensor dbias, NVTETensor workspace, hipStream_t stream);
void nvte_quantize_dbias_dsilu(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                               NVTETensor dbias, NVTETensor workspace, hipStream_t stream);
void nvte_quantize_dbias_drelu(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                               NVTETensor dbias, NVTETensor workspace, hipStream_t stream);
void nvte_quantize_dbias_dqgelu(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                NVTETensor dbias, NVTETensor workspace, hipStream_t stream);
void nvte_quantize_dbias_dsrelu(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                NVTETensor dbias, NVTETensor workspace, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
