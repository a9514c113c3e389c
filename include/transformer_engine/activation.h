/* Activations fused with the cast to FP8: the step either side of a
 * Linear in an MLP (FC1 -> act -> FC2).  Mirrors ref common/include/transformer_engine/activation.h;
 * the math is ref common/util/math.h:21-134, the gated layout ref util/vectorized_pointwise.h:410-560
 * (input [N, 2H]: activation half first, gate half second).
 *
 * Every OUTPUT may be FP32/BF16/FP16 (row-wise) or FP8 with per-tensor scaling: scale in,
 * amax (of the pre-scale value) and scale_inv out, and - when the output carries a column-wise
 * buffer - the transpose written by the same kernel.  MXFP8 outputs are not implemented here. */
#ifndef TE_AMD_ACTIVATION_H_
#define TE_AMD_ACTIVATION_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref activation.h:54-138: output = act(input) */
void nvte_gelu(const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_silu(const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_relu(const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_qgelu(const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_srelu(const NVTETensor input, NVTETensor output, hipStream_t stream);
/* ref activation.h:160-260: output = grad * act'(input) */
void nvte_dgelu(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_dsilu(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_drelu(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_dqgelu(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_dsrelu(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream);
/* ref activation.h:288-439: input [N, 2H] -> output [N, H] = act(input[:, :H]) * input[:, H:] */
void nvte_glu(const NVTETensor input, NVTETensor output, hipStream_t stream);     /* sigmoid gate */
void nvte_geglu(const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_swiglu(const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_reglu(const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_qgeglu(const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_sreglu(const NVTETensor input, NVTETensor output, hipStream_t stream);
/* ref activation.h:299,450-587: grad [N, H], input [N, 2H] -> output [N, 2H] =
 * [ act'(a) * grad * b , act(a) * grad ] */
void nvte_dglu(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_dgeglu(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_dswiglu(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_dreglu(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_dqgeglu(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream);
void nvte_dsreglu(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream);

/* ref activation.h:346-370.  out[N, H] = act(in[:, :H]) * (clamp(in[:, H:], -limit, limit) + glu_linear_offset)
 * with act(x) = c * sigmoid(alpha * c), c = min(x, limit) (GPT-OSS).  The form without the offset is the
 * reference's deprecated one (offset 1).  Outputs as the other gated activations (high precision,
 * FP8 per-tensor scaling with transpose, MXFP8). */
void nvte_clamped_swiglu(const NVTETensor input, NVTETensor output, float limit, float alpha,
                         hipStream_t stream);
void nvte_clamped_swiglu_v2(const NVTETensor input, NVTETensor output, float limit, float alpha,
                            float glu_linear_offset, hipStream_t stream);
/* ref activation.h:487-512.  out[N, 2H]: [:, :H] = grad * act'(a) * (clamp(g) + offset) with act' = 0 for
 * a > limit; [:, H:] = grad * act(a) where |g| <= limit, else 0. */
void nvte_clamped_dswiglu(const NVTETensor grad, const NVTETensor input, NVTETensor output, float limit,
                          float alpha, hipStream_t stream);
void nvte_clamped_dswiglu_v2(const NVTETensor grad, const NVTETensor input, NVTETensor output, float limit,
                             float alpha, float glu_linear_offset, hipStream_t stream);

/* Activations multiplied by a per-row scale (MoE router probabilities), ref activation.h:372-406 (forward),
 * :513-552 (backward), :590-615 (SReLU); kernels ref common/activation/scaled_activation.cu.
 *   out[N, H] = Act(input) * act_scales[:, None], all math in fp32, one cast at the store.
 * input [N, 2H] for the gated forms: [act | gate], or with glu_interleave_size = S (a positive multiple of 32
 * that divides H; 0 = not interleaved) blocks [act[0:S] | gate[0:S] | act[S:2S] | ...].  input, act_scales,
 * output / grad_input and grad_act_scales are FP32 / BF16 / FP16, each with its own dtype (no FP8 here, as in
 * the reference).  Backward: grad_input has input's shape; grad_act_scales may be NULL, otherwise it receives
 * sum(grad * Act(input), dim=-1) of shape [N].  The clamped forms take nvte_clamped_swiglu_v2's limit,
 * alpha and glu_linear_offset. */
void nvte_scaled_swiglu(const NVTETensor input, const NVTETensor act_scales, NVTETensor output,
                        int64_t glu_interleave_size, hipStream_t stream);
void nvte_scaled_clamped_swiglu(const NVTETensor input, const NVTETensor act_scales, NVTETensor output, float limit,
                                float alpha, float glu_linear_offset, int64_t glu_interleave_size,
                                hipStream_t stream);
void nvte_scaled_dswiglu(const NVTETensor grad, const NVTETensor input, const NVTETensor act_scales,
                         NVTETensor grad_input, NVTETensor grad_act_scales, int64_t glu_interleave_size,
                         hipStream_t stream);
void nvte_scaled_clamped_dswiglu(const NVTETensor grad, const NVTETensor input, const NVTETensor act_scales,
                                 NVTETensor grad_input, NVTETensor grad_act_scales, float limit, float alpha,
                                 float glu_linear_offset, int64_t glu_interleave_size, hipStream_t stream);
void nvte_scaled_srelu(const NVTETensor input, const NVTETensor act_scales, NVTETensor output, hipStream_t stream);
void nvte_scaled_dsrelu(const NVTETensor grad, const NVTETensor input, const NVTETensor act_scales,
                        NVTETensor grad_input, NVTETensor grad_act_scales, hipStream_t stream);

/* Grouped forms (ref activation.h:65,86,107,128,149): the unary activation of every member of a grouped
 * tensor, cast as output's scaling mode asks - FP8 per-tensor scaling with output.scale[i] per member (amax /
 * scale_inv untouched, as nvte_group_quantize) or MXFP8 (rows of every member and the last dim multiples of
 * 128).  Member shapes are read on the device.  Same-shape and varying-first-dim groups. */
void nvte_group_gelu(const NVTEGroupedTensor input, NVTEGroupedTensor output, hipStream_t stream);
void nvte_group_silu(const NVTEGroupedTensor input, NVTEGroupedTensor output, hipStream_t stream);
void nvte_group_relu(const NVTEGroupedTensor input, NVTEGroupedTensor output, hipStream_t stream);
void nvte_group_qgelu(const NVTEGroupedTensor input, NVTEGroupedTensor output, hipStream_t stream);
void nvte_group_srelu(const NVTEGroupedTensor input, NVTEGroupedTensor output, hipStream_t stream);
/* ref activation.h:173,198,223,248,273 (output is declared NVTETensor there; it is a grouped handle):
 * output = cast(grad * act'(input)) per member. */
void nvte_group_dgelu(const NVTEGroupedTensor grad, const NVTEGroupedTensor input, NVTETensor output, hipStream_t stream);
void nvte_group_dsilu(const NVTEGroupedTensor grad, const NVTEGroupedTensor input, NVTETensor output, hipStream_t stream);
void nvte_group_drelu(const NVTEGroupedTensor grad, const NVTEGroupedTensor input, NVTETensor output, hipStream_t stream);
void nvte_group_dqgelu(const NVTEGroupedTensor grad, const NVTEGroupedTensor input, NVTETensor output, hipStream_t stream);
void nvte_group_dsrelu(const NVTEGroupedTensor grad, const NVTEGroupedTensor input, NVTETensor output, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
