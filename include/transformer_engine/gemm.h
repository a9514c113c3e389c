/* Scaled FP8 / MXFP8 / bf16 GEMM entry points.
 * Replaces reference .../include/transformer_engine/gemm.h; the definitions it
 * stands in for are common/gemm/cublaslt_gemm.cu:803-968 (C entry points),
 * :315-799 (cublas_gemm) and :102-307 (operand canonicalisation).  The names keep
 * "cublas" because callers bind them by name; the implementation is a hand-written
 * MFMA kernel (csrc/gemm_fp8.hip), no BLAS library is linked.
 *
 * Convention (ref cublaslt_gemm.cu:324-335): tensors are row-major; the GEMM is
 * stated column-major, so with A = [A0, A1], B = [B0, B1]
 *     m = transa ? A0 : A1,  k = transa ? A1 : A0,  n = transb ? B1 : B0
 * and D is row-major [n, m] with  D[j][i] = alpha * sum_k opA[i][k] * opB[j][k] + beta * C[j][i].
 */
#ifndef TE_AMD_GEMM_H_
#define TE_AMD_GEMM_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref gemm.h:23 */
typedef void *NVTEMatmulConfig;

/* ref gemm.h:31-58 */
enum NVTEMatmulConfigAttribute {
  kNVTEMatmulConfigBiasTensor = 0,           /* NVTETensor */
  kNVTEMatmulConfigDBiasTensor = 1,          /* NVTETensor */
  kNVTEMatmulConfigWithGELUEpilogue = 2,     /* uint8 (bool) */
  kNVTEMatmulConfigWithDGELUEpilogue = 3,    /* uint8 (bool) */
  kNVTEMatmulConfigEpilogueAuxTensor = 4,    /* NVTETensor */
  kNVTEMatmulConfigUseSplitAccumulator = 5,  /* uint8 (bool); hint only: MFMA always accumulates in fp32 */
  kNVTEMatmulConfigSMCount = 6,              /* int; hint: CUs the kernel may occupy (0 = all) */
  kNVTEMatmulConfigNumAttributes
};

/* ref gemm.h:93  */ NVTEMatmulConfig nvte_create_matmul_config(void);
/* ref gemm.h:106 */ void nvte_get_matmul_config_attribute(NVTEMatmulConfig config,
                                                           enum NVTEMatmulConfigAttribute attr, void *buf,
                                                           size_t size_in_bytes, size_t *size_written);
/* ref gemm.h:116 */ void nvte_set_matmul_config_attribute(NVTEMatmulConfig config,
                                                           enum NVTEMatmulConfigAttribute attr,
                                                           const void *buf, size_t size_in_bytes);
/* ref gemm.h:120 */ void nvte_destroy_matmul_config(NVTEMatmulConfig config);

/* ref gemm.h:178 (deprecated form, still what tex.generic_gemm's older callers bind) */
void nvte_cublas_gemm(const NVTETensor A, const NVTETensor B, NVTETensor D, const NVTETensor bias,
                      NVTETensor pre_gelu_out, bool transa, bool transb, bool grad,
                      NVTETensor workspace, bool accumulate, bool use_split_accumulator,
                      int math_sm_count, hipStream_t stream);

/* ref gemm.h:200.  alpha / beta are HOST pointers.  C must be D (ref cublaslt_gemm.cu:845). */
void nvte_cublas_gemm_v2(int transa, int transb, const float *alpha, const NVTETensor A,
                         const NVTETensor B, const float *beta, const NVTETensor C, NVTETensor D,
                         NVTETensor workspace, NVTEMatmulConfig config, hipStream_t stream);

/* ref gemm.h:231 */
void nvte_cublas_gemm_scaled(const NVTETensor A, const NVTETensor B, NVTETensor D,
                             const NVTETensor bias, NVTETensor pre_gelu_out, bool transa,
                             bool transb, bool grad, NVTETensor workspace, float alpha, float beta,
                             bool use_split_accumulator, int math_sm_count, hipStream_t stream);

/* ref gemm.h:295.  A list of independent GEMMs (GroupedLinear's legacy path). */
void nvte_multi_tensor_gemm(const NVTETensor *A, const NVTETensor *B, NVTETensor *D,
                            const NVTETensor *bias, NVTETensor *pre_gelu_out, const int num_gemms,
                            bool transa, bool transb, bool grad, NVTETensor *workspace,
                            bool accumulate, bool use_split_accumulator, int math_sm_count,
                            hipStream_t stream);

/* ref gemm.h:485.  No handle exists here; kept so callers that pre-warm can still call it. */
void nvte_cublas_handle_init(void);

#ifdef __cplusplus
}
#endif
#endif
