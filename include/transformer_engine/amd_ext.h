/* Extensions with NO counterpart in the reference.  They exist because this
 * library is also bound from languages that cannot catch C++ exceptions or build
 * std::vector arguments (ctypes, cgo, JNI ...).
 */
#ifndef TE_AMD_EXT_H_
#define TE_AMD_EXT_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Error mode.  0 (default) = reference behaviour: a failed check throws
 * std::runtime_error through the C entry point (ref common/util/logging.h:38-51).
 * 1 = "record": the entry point catches the exception, stores its text in a
 * thread-local slot and returns normally; the caller polls nvte_amd_last_error(). */
void nvte_amd_set_error_mode(int mode);
int nvte_amd_get_error_mode(void);
/* Text of the last recorded error on this thread, or NULL.  Valid until the next
 * nvte_* call on the same thread. */
const char *nvte_amd_last_error(void);
void nvte_amd_clear_error(void);
/* Address of a process-wide count of recorded errors (all threads).  A foreign-function binding reads it
 * before and after a call - a plain memory read - and only asks for nvte_amd_last_error() when it moved:
 * polling the thread-local slot through a second foreign call after EVERY call doubled the per-call cost of
 * the ctypes binding. */
const unsigned long long *nvte_amd_error_counter(void);

/* Plain-C twin of nvte_delayed_scaling_recipe_amax_and_scale_update_after_reduction
 * (ref recipe.h:71-74, whose std::vector arguments make the symbol C++-mangled). */
void nvte_amd_delayed_scaling_after_reduction(const NVTETensor amax_reduction_buffer,
                                              const NVTETensor *amax_histories,
                                              const NVTETensor *scales, size_t num_tensors,
                                              const char *amax_compute_algo,
                                              enum NVTEDType fp8_dtype, float margin,
                                              hipStream_t stream);

/* The same bulk update with ONE FP8 format per tensor, so that the forward (E4M3) and backward (E5M2 under
 * the HYBRID format) recipe states of a step share one launch: the reference issues one call per direction
 * (ref pytorch/quantization.py:646-701, recipe.h:71-74); a layer's two states are independent, so updating
 * both behind the backward pass gives the values of the two separate calls. */
void nvte_amd_delayed_scaling_after_reduction_multi(const NVTETensor amax_reduction_buffer,
                                                    const NVTETensor *amax_histories,
                                                    const NVTETensor *scales,
                                                    const enum NVTEDType *fp8_dtypes,
                                                    size_t num_tensors, const char *amax_compute_algo,
                                                    float margin, hipStream_t stream);

/* MXFP8 "transposed columnwise" layout (no counterpart in the reference; OPTIONAL since round 3).
 * The reference keeps the columnwise-quantized MXFP8 payload in the input's own [rows, cols]
 * layout and lets cuBLASLt read it as a non-TN operand; nvte_cublas_gemm* here read that layout
 * K-major through the transposing LDS loads (the default, nothing to set).  An MXFP8 output tensor
 * can instead ask nvte_quantize* to store its columnwise payload transposed (K-contiguous):
 *   columnwise_data       [cols, rows]              (FP8 bytes, blocks of 32 along the LAST dim)
 *   columnwise_scale_inv  [pad128(cols), pad4(ceil(rows/32))]   (e8m0; GEMM-swizzled if the
 *                          tensor's with_gemm_swizzled_scales flag is set -- for that flag the
 *                          bytes are identical to the reference's swizzled columnwise scales)
 * nvte_cublas_gemm* consume either layout for the NN / NT layouts (dgrad / wgrad).  Default 0 =
 * reference layout. */
void nvte_amd_set_columnwise_transposed(NVTETensor tensor, int transposed);
int nvte_amd_get_columnwise_transposed(const NVTETensor tensor);

/* The same for a grouped MXFP8 tensor (no counterpart in the reference): columnwise_data holds every
 * member's payload transposed ([last, first_i] at the member's element offset) and
 * columnwise_scale_inv the members' [pad128(last), pad4(first_i / 32)] scale matrices back to back
 * (member i at the sum of the padded sizes before it - the offset rule of ref
 * gemm/cublaslt_grouped_gemm.cu:1117-1198).  nvte_group_quantize writes this layout when the flag is set;
 * nvte_grouped_gemm* takes it or the reference's un-transposed columnwise members (read K-major, members
 * with one last dimension) for the NN / NT layouts (dgrad / wgrad).  Member first dims must be
 * multiples of 128 for grouped MXFP8 tensors (the reference states the same for the logical first dim,
 * cast/mxfp8/group_quantize_mxfp8.cuh:842, and its swizzled-scale offsets assume it per member). */
void nvte_amd_set_grouped_columnwise_transposed(NVTEGroupedTensor tensor, int transposed);
int nvte_amd_get_grouped_columnwise_transposed(const NVTEGroupedTensor tensor);

/* A list of independent GEMMs in ONE persistent launch (no counterpart in the reference, whose list form
 * nvte_multi_tensor_gemm, gemm/cublaslt_gemm.cu:1052, spreads the list over streams and shares one
 * transa / transb): entry i computes D[i] = op(A[i]) op(B[i]) (+ D[i] when accumulate[i]) exactly as
 * nvte_cublas_gemm(A[i], B[i], D[i], NULL, NULL, transa[i], transb[i], grad, workspace, accumulate[i], ...)
 * does.  Meant for the dgrad + wgrad pair of a layer: the tiles of all entries form one list that the
 * persistent workgroups walk, cut per XCD by cost, so a GEMM whose tile count is not a multiple of the
 * workgroup count (wgrad of Llama-3-8B's FFN: 3.5 rounds) leaves no idle half round.  Taken when 2 <= num_gemms
 * <= 8, every operand is FP8 with per-tensor scaling, the FP8 formats and the (high-precision) output dtype
 * agree across the list and workspace holds nvte_get_grouped_gemm_setup_workspace_size(num_gemms) bytes;
 * any other list runs as the nvte_cublas_gemm calls above, one after the other.  accumulate may be NULL. */
void nvte_amd_multi_gemm(const NVTETensor *A, const NVTETensor *B, NVTETensor *D, const int *transa,
                         const int *transb, int num_gemms, int grad, NVTETensor workspace, const int *accumulate,
                         hipStream_t stream);

/* Library identification: "transformerengine_amd <version> gfx950". */
const char *nvte_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif
