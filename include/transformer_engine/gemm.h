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

/* ---- grouped GEMM (ref gemm.h:60-92, 122-152, 301-468) ---- */
typedef void *NVTEGroupedMatmulConfig;
enum NVTEGroupedMatmulConfigAttribute {
  kNVTEGroupedMatmulConfigAvgM = 0,               /* int64 hint (stored, unused) */
  kNVTEGroupedMatmulConfigAvgN = 1,               /* int64 hint (stored, unused) */
  kNVTEGroupedMatmulConfigAvgK = 2,               /* int64 hint (stored, unused) */
  kNVTEGroupedMatmulConfigSMCount = 3,            /* int: number of CUs to use (0 = all) */
  kNVTEGroupedMatmulConfigUseSplitAccumulator = 4, /* uint8 (stored; FP32 accumulation always) */
  kNVTEGroupedMatmulConfigNumAttributes
};
/* ref gemm.h:123 */ NVTEGroupedMatmulConfig nvte_create_grouped_matmul_config(void);
/* ref gemm.h:136 */ void nvte_get_grouped_matmul_config_attribute(NVTEGroupedMatmulConfig config,
                                                                 enum NVTEGroupedMatmulConfigAttribute attr,
                                                                 void *buf, size_t size_in_bytes,
                                                                 size_t *size_written);
/* ref gemm.h:147 */ void nvte_set_grouped_matmul_config_attribute(NVTEGroupedMatmulConfig config,
                                                                 enum NVTEGroupedMatmulConfigAttribute attr,
                                                                 const void *buf, size_t size_in_bytes);
/* ref gemm.h:152 */ void nvte_destroy_grouped_matmul_config(NVTEGroupedMatmulConfig config);

/* ref gemm.h:312.  Bytes of workspace_setup: the per-group descriptors + tile prefix the device-side
 * setup kernel writes for the persistent grouped kernel. */
size_t nvte_get_grouped_gemm_setup_workspace_size(size_t num_tensors);
/* ref gemm.h:324 */ void nvte_convert_int32_to_int64(const int32_t *src, int64_t *dst, size_t n, hipStream_t stream);
/* ref gemm.h:337 */ void nvte_convert_int32_to_int64_with_multiplier(const int32_t *src, int64_t *dst, size_t n,
                                                                    int64_t multiplier, hipStream_t stream);
/* ref gemm.h:352 */ void nvte_compute_grouped_tensor_offsets(const int64_t *first_dims, int64_t *offsets,
                                                            size_t n_groups, int64_t last_dim, hipStream_t stream);

/* ref gemm.h:388.  D[i] = alpha[i] * op(A[i]) op(B[i]) + beta[i] * C[i] for every member i, ONE
 * persistent launch; member shapes are read on the device.  FP8 per-tensor scaling operands
 * (scale_inv[i] per member) or MXFP8 operands (both A and B; ref cublaslt_grouped_gemm.cu:283-300; member
 * i's e8m0 scale matrix at the sum of the padded sizes before it, :1117-1198; compact or swizzled;
 * layouts that contract over a member's first dim read the reference's un-transposed columnwise members
 * K-major, or the transposed columnwise storage of amd_ext.h), BF16/FP16/FP32 C and D.  C may be NULL (beta
 * is then ignored) or D.  Per-tensor members without columnwise data are read K-major the same way. */
void nvte_grouped_gemm(const NVTEGroupedTensor A, int transa, const NVTEGroupedTensor B, int transb,
                       const NVTEGroupedTensor C, NVTEGroupedTensor D, const NVTETensor alpha,
                       const NVTETensor beta, NVTETensor workspace_setup, NVTETensor workspace_cublas,
                       NVTEGroupedMatmulConfig config, hipStream_t stream);
/* NO counterpart in the reference.  nvte_grouped_gemm for callers that cannot allocate device memory
 * or carry a workspace: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  Same arguments minus the two workspaces: the per-member records + tile prefix live in a
 * static device array of this library (about 26 KiB per GPU, held until the process ends) whose
 * address is resolved once per device (at nvte_cublas_handle_init() or at the first call on that
 * device).  The call allocates nothing.  Limits: at most nvte_amd_grouped_gemm_noalloc_max_tensors()
 * (256) member tensors, more is an error; every call on a GPU shares that GPU's array, so two calls
 * that overlap on one GPU (different streams) race - serialise them or use nvte_grouped_gemm.  As for
 * every entry point here, `stream` must belong to the CURRENT device (the array's address is that device's). */
void nvte_amd_grouped_gemm_noalloc(const NVTEGroupedTensor A, int transa, const NVTEGroupedTensor B, int transb,
                                   const NVTEGroupedTensor C, NVTEGroupedTensor D, const NVTETensor alpha,
                                   const NVTETensor beta, NVTEGroupedMatmulConfig config, hipStream_t stream);
size_t nvte_amd_grouped_gemm_noalloc_max_tensors(void);
/* ref gemm.h:415.  A given as a list of (same-shape or not) per-expert weight tensors. */
void nvte_grouped_gemm_with_discrete_inputA(const NVTETensor *A_list, size_t num_a_tensors, int transa,
                                            const NVTEGroupedTensor B, int transb, const NVTEGroupedTensor C,
                                            NVTEGroupedTensor D, const NVTETensor alpha, const NVTETensor beta,
                                            NVTETensor workspace_setup, NVTETensor workspace_cublas,
                                            NVTEGroupedMatmulConfig config, hipStream_t stream);
/* ref gemm.h:445.  C / D given as lists (per-expert wgrad buffers). */
void nvte_grouped_gemm_with_discrete_out(const NVTEGroupedTensor A, int transa, const NVTEGroupedTensor B,
                                         int transb, const NVTETensor *C_list, size_t num_c_tensors,
                                         NVTETensor *D_list, size_t num_d_tensors, const NVTETensor alpha,
                                         const NVTETensor beta, NVTETensor workspace_setup,
                                         NVTETensor workspace_cublas, NVTEGroupedMatmulConfig config,
                                         hipStream_t stream);
/* ref gemm.h:458.  output[row, col] += bias[group(row)][col] */
void nvte_grouped_bias_add(const NVTEGroupedTensor output, const NVTEGroupedTensor bias, hipStream_t stream);
/* ref gemm.h:467.  output[row, col] += bias[group(row)][col] * scale[row] */
void nvte_grouped_scaled_bias_add(const NVTEGroupedTensor output, const NVTEGroupedTensor bias,
                                  const NVTETensor scale, hipStream_t stream);

#ifdef __cplusplus
}  /* extern "C" */

namespace transformer_engine {

/* ref gemm.h:490-572.  Owns an NVTEMatmulConfig. */
class MatmulConfigWrapper {
 public:
  MatmulConfigWrapper() : config_{nvte_create_matmul_config()} {}
  MatmulConfigWrapper(const MatmulConfigWrapper &) = delete;
  MatmulConfigWrapper &operator=(const MatmulConfigWrapper &) = delete;
  MatmulConfigWrapper(MatmulConfigWrapper &&other) noexcept : config_{other.config_} { other.config_ = nullptr; }
  MatmulConfigWrapper &operator=(MatmulConfigWrapper &&other) noexcept {
    if (this != &other) {
      reset();
      config_ = other.config_;
      other.config_ = nullptr;
    }
    return *this;
  }
  ~MatmulConfigWrapper() { reset(); }
  operator NVTEMatmulConfig() const noexcept { return config_; }

  void set_bias_tensor(NVTETensor t) { put(kNVTEMatmulConfigBiasTensor, &t, sizeof(t)); }
  void set_dbias_tensor(NVTETensor t) { put(kNVTEMatmulConfigDBiasTensor, &t, sizeof(t)); }
  void set_with_gelu_epilogue(bool v) { put_u8(kNVTEMatmulConfigWithGELUEpilogue, v); }
  void set_with_dgelu_epilogue(bool v) { put_u8(kNVTEMatmulConfigWithDGELUEpilogue, v); }
  void set_epilogue_aux_tensor(NVTETensor t) { put(kNVTEMatmulConfigEpilogueAuxTensor, &t, sizeof(t)); }
  void set_use_split_accumulator(bool v) { put_u8(kNVTEMatmulConfigUseSplitAccumulator, v); }
  void set_sm_count(int v) { put(kNVTEMatmulConfigSMCount, &v, sizeof(v)); }

 private:
  void put(NVTEMatmulConfigAttribute attr, const void *buf, size_t n) {
    nvte_set_matmul_config_attribute(config_, attr, buf, n);
  }
  void put_u8(NVTEMatmulConfigAttribute attr, bool flag) {
    const uint8_t v = flag ? 1 : 0;
    put(attr, &v, sizeof(v));
  }
  void reset() noexcept {
    if (config_ != nullptr) nvte_destroy_matmul_config(config_);
    config_ = nullptr;
  }
  NVTEMatmulConfig config_ = nullptr;
};

/* ref gemm.h:574-640.  Owns an NVTEGroupedMatmulConfig. */
class GroupedMatmulConfigWrapper {
 public:
  GroupedMatmulConfigWrapper() : config_{nvte_create_grouped_matmul_config()} {}
  GroupedMatmulConfigWrapper(const GroupedMatmulConfigWrapper &) = delete;
  GroupedMatmulConfigWrapper &operator=(const GroupedMatmulConfigWrapper &) = delete;
  GroupedMatmulConfigWrapper(GroupedMatmulConfigWrapper &&other) noexcept : config_{other.config_} {
    other.config_ = nullptr;
  }
  GroupedMatmulConfigWrapper &operator=(GroupedMatmulConfigWrapper &&other) noexcept {
    if (this != &other) {
      reset();
      config_ = other.config_;
      other.config_ = nullptr;
    }
    return *this;
  }
  ~GroupedMatmulConfigWrapper() { reset(); }
  operator NVTEGroupedMatmulConfig() const noexcept { return config_; }

  void set_avg_m(int64_t v) { put(kNVTEGroupedMatmulConfigAvgM, &v, sizeof(v)); }
  void set_avg_n(int64_t v) { put(kNVTEGroupedMatmulConfigAvgN, &v, sizeof(v)); }
  void set_avg_k(int64_t v) { put(kNVTEGroupedMatmulConfigAvgK, &v, sizeof(v)); }
  void set_sm_count(int v) { put(kNVTEGroupedMatmulConfigSMCount, &v, sizeof(v)); }
  void set_use_split_accumulator(bool flag) {
    const uint8_t v = flag ? 1 : 0;
    put(kNVTEGroupedMatmulConfigUseSplitAccumulator, &v, sizeof(v));
  }

 private:
  void put(NVTEGroupedMatmulConfigAttribute attr, const void *buf, size_t n) {
    nvte_set_grouped_matmul_config_attribute(config_, attr, buf, n);
  }
  void reset() noexcept {
    if (config_ != nullptr) nvte_destroy_grouped_matmul_config(config_);
    config_ = nullptr;
  }
  NVTEGroupedMatmulConfig config_ = nullptr;
};

}  // namespace transformer_engine
#endif
#endif
