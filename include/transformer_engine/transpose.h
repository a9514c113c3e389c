/* Cast+transpose and transpose entry points.
 * Replaces reference transformer_engine/common/include/transformer_engine/transpose.h
 * (definitions: common/transpose/cast_transpose.cu:348, transpose.cu:292,
 * multi_cast_transpose.cu:335, cast_transpose_fusion.cu:1260).
 */
#ifndef TE_AMD_TRANSPOSE_H_
#define TE_AMD_TRANSPOSE_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref transpose.h:32 */
void nvte_cast_transpose(const NVTETensor input, NVTETensor output, hipStream_t stream);
/* ref transpose.h:40 */
void nvte_transpose(const NVTETensor input, NVTETensor transposed_output, hipStream_t stream);
/* ref transpose.h:60 */
void nvte_cast_transpose_dbias(const NVTETensor input, NVTETensor output, NVTETensor dbias,
                               NVTETensor workspace, hipStream_t stream);
/* ref transpose.h:79 (def. transpose/transpose_fusion.cu:498).  input FP8 [N, H] -> transposed_output
 * FP8 [H, N] (same bytes) + dbias[H] = sum over N of float(input) * scale_inv, where scale_inv is the
 * OUTPUT tensor's (as the reference reads it, transpose_fusion.cu:469), else the input's, else 1.
 * Two-call workspace protocol: an empty workspace only gets its shape [ceil(N/128), H] fp32. */
void nvte_fp8_transpose_dbias(const NVTETensor input, NVTETensor transposed_output, NVTETensor dbias,
                              NVTETensor workspace, hipStream_t stream);
/* ref transpose.h:93 */
void nvte_multi_cast_transpose(size_t num_tensors, const NVTETensor *input_list,
                               NVTETensor *output_list, hipStream_t stream);

/* ref transpose.h:118-222: the older names of nvte_quantize_dbias_d<act> for outputs that carry both
 * layouts (cast + transpose + dbias of input * act'(act_input) in one pass). */
void nvte_cast_transpose_dbias_dgelu(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                     NVTETensor dbias, NVTETensor workspace, hipStream_t stream);
void nvte_cast_transpose_dbias_dsilu(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                     NVTETensor dbias, NVTETensor workspace, hipStream_t stream);
void nvte_cast_transpose_dbias_drelu(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                     NVTETensor dbias, NVTETensor workspace, hipStream_t stream);
void nvte_cast_transpose_dbias_dqgelu(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                      NVTETensor dbias, NVTETensor workspace, hipStream_t stream);
void nvte_cast_transpose_dbias_dsrelu(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                      NVTETensor dbias, NVTETensor workspace, hipStream_t stream);
/* ref transpose.h:242-318: gated backward (see activation.h) with cast + transpose of the [N, 2H] result */
void nvte_dgeglu_cast_transpose(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                hipStream_t stream);
void nvte_dswiglu_cast_transpose(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                 hipStream_t stream);
void nvte_dreglu_cast_transpose(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                hipStream_t stream);
void nvte_dqgeglu_cast_transpose(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                 hipStream_t stream);
void nvte_dsreglu_cast_transpose(const NVTETensor input, const NVTETensor act_input, NVTETensor output,
                                 hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
