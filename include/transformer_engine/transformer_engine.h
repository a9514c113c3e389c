/* MI355X-native FP8 linear hot path: tensor-handle C ABI.
 *
 * Every declaration here replaces the entry point of the same name in the
 * reference's  transformer_engine/common/include/transformer_engine/transformer_engine.h
 * (line numbers quoted per declaration, "ref:" prefix).  Enum VALUES are ABI and
 * are identical to the reference's.  The only substitution is the stream type:
 * the reference takes cudaStream_t (ref :14), this library takes hipStream_t, which
 * is what PyTorch-ROCm hands out.
 *
 * Error convention (ref common/util/logging.h:38-51): a failed check throws
 * std::runtime_error out of the C entry point.  Callers that cannot catch C++
 * exceptions (ctypes, cgo, ...) switch to the "record" mode declared in amd_ext.h.
 */
#ifndef TE_AMD_TRANSFORMER_ENGINE_H_
#define TE_AMD_TRANSFORMER_ENGINE_H_

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ref :25-38 */
enum NVTEDType {
  kNVTEByte = 0,
  kNVTEInt16 = 1,
  kNVTEInt32 = 2,
  kNVTEInt64 = 3,
  kNVTEFloat32 = 4,
  kNVTEFloat16 = 5,
  kNVTEBFloat16 = 6,
  kNVTEFloat8E4M3 = 7,
  kNVTEFloat8E5M2 = 8,
  kNVTEFloat8E8M0 = 9,
  kNVTEFloat4E2M1 = 10,
  kNVTENumTypes
};

/* ref :43-48.  Passed and returned BY VALUE. */
struct NVTEShape {
  size_t data[15];
  size_t ndim;
};

/* ref :54-61.  Non-owning view of one buffer. */
struct NVTEBasicTensor {
  void *data_ptr;
  enum NVTEDType dtype;
  struct NVTEShape shape;
};

/* ref :66-94 */
enum NVTETensorParam {
  kNVTERowwiseData = 0,
  kNVTEColumnwiseData = 1,
  kNVTEScale = 2,
  kNVTEAmax = 3,
  kNVTERowwiseScaleInv = 4,
  kNVTEColumnwiseScaleInv = 5,
  kNVTEColumnwiseAmax = 6,
  kNVTEWithGEMMSwizzledScales = 7,
  kNVTERowScaledNVFP4 = 8,   /* accepted and stored; NVFP4 kernels are out of scope */
  kNVTENVFP4E4M3Max = 9,     /* accepted and stored; NVFP4 kernels are out of scope */
  kNVTENumTensorParams
};

/* ref :99-119 */
enum NVTEScalingMode {
  NVTE_DELAYED_TENSOR_SCALING = 0, /* unquantized, or FP8 with ONE fp32 scale per tensor */
  NVTE_MXFP8_1D_SCALING = 1,       /* one e8m0 scale per 32 consecutive elements */
  NVTE_BLOCK_SCALING_1D = 2,       /* not implemented (out of scope) */
  NVTE_BLOCK_SCALING_2D = 3,       /* not implemented (out of scope) */
  NVTE_NVFP4_1D_SCALING = 4,       /* not implemented (out of scope) */
  NVTE_INVALID_SCALING = 100
};

/* ref :136.  Opaque handle = 1-based slot index in a process-wide arena. 0 = invalid. */
typedef void *NVTETensor;

/* ref :148 */ NVTETensor nvte_create_tensor(enum NVTEScalingMode scaling_mode);
/* ref :162 */ void nvte_create_tensors(enum NVTEScalingMode scaling_mode, NVTETensor *tensors, size_t N);
/* ref :171 */ void nvte_destroy_tensor(NVTETensor tensor);
/* ref :182 */ void nvte_destroy_tensors(NVTETensor *tensors, size_t N);
/* ref :190 */ void *nvte_tensor_data(const NVTETensor tensor);
/* ref :198 */ void *nvte_tensor_columnwise_data(const NVTETensor tensor);
/* ref :208 */ struct NVTEShape nvte_make_shape(const size_t *data, size_t ndim);
/* ref :216 */ struct NVTEShape nvte_tensor_shape(const NVTETensor tensor);
/* ref :224 */ struct NVTEShape nvte_tensor_columnwise_shape(const NVTETensor tensor);
/* ref :232 */ size_t nvte_tensor_ndims(const NVTETensor tensor);
/* ref :241 */ size_t nvte_tensor_size(const NVTETensor tensor, const size_t dim);
/* ref :249 */ size_t nvte_tensor_size_bytes(const NVTETensor tensor);
/* ref :257 */ size_t nvte_tensor_numel(const NVTETensor tensor);
/* ref :265 */ size_t nvte_tensor_element_size(const NVTETensor tensor);
/* ref :273 */ size_t nvte_tensor_element_size_bits(const NVTETensor tensor);
/* ref :281 */ enum NVTEDType nvte_tensor_type(const NVTETensor tensor);
/* ref :289 */ float *nvte_tensor_amax(const NVTETensor tensor);
/* ref :297 */ float *nvte_tensor_scale(const NVTETensor tensor);
/* ref :305 */ float *nvte_tensor_scale_inv(const NVTETensor tensor);
/* ref :313 */ struct NVTEShape nvte_tensor_scale_inv_shape(const NVTETensor tensor);
/* ref :320 */ void nvte_zero_tensor(const NVTETensor tensor, hipStream_t stream);
/* ref :330 */ void nvte_set_tensor_param(NVTETensor *tensor, enum NVTETensorParam param_name,
                                          const struct NVTEBasicTensor *param);
/* ref :340 */ struct NVTEBasicTensor nvte_get_tensor_param(const NVTETensor tensor,
                                                            enum NVTETensorParam param_name);
/* ref :349 */ void nvte_set_tensor_param_v2(NVTETensor tensor, enum NVTETensorParam param,
                                             const void *buf, size_t size_in_bytes);
/* ref :363 */ void nvte_get_tensor_param_v2(const NVTETensor tensor, enum NVTETensorParam param,
                                             void *buf, size_t size_in_bytes, size_t *size_written);
/* ref :372 */ enum NVTEScalingMode nvte_tensor_scaling_mode(const NVTETensor tensor);

/* ref :377-384 */
#define NVTE_TENSOR_PACK_MAX_SIZE 10
struct NVTETensorPack {
  NVTETensor tensors[NVTE_TENSOR_PACK_MAX_SIZE];
  size_t size;
};
/* ref :388 */ void nvte_tensor_pack_create(struct NVTETensorPack *pack);
/* ref :392 */ void nvte_tensor_pack_destroy(struct NVTETensorPack *pack);

/* ref :395 */
typedef void *NVTEQuantizationConfig;

/* ref :400-440 */
enum NVTEQuantizationConfigAttribute {
  kNVTEQuantizationConfigForcePow2Scales = 0,                /* uint8 (bool)  */
  kNVTEQuantizationConfigAmaxEpsilon = 1,                    /* float         */
  kNVTEQuantizationConfigNoopTensor = 2,                     /* NVTETensor    */
  kNVTEQuantizationConfigFloat8BlockScaleTensorFormat = 3,   /* int, stored only */
  kNVTEQuantizationConfigRNGState = 4,                       /* NVTETensor, stored only */
  kNVTEQuantizationConfigNVFP42DQuantization = 5,            /* uint8, stored only */
  kNVTEQuantizationConfigStochasticRounding = 6,             /* uint8, stored only */
  kNVTEQuantizationConfigUseFastMath = 7,                    /* uint8, stored only */
  kNVTEQuantizationConfigNVFP44Over6Mode = 8,                /* uint8, stored only */
  kNVTEQuantizationConfigNVFP44Over6ErrUseFastMath = 9,      /* uint8, stored only */
  kNVTEQuantizationConfigNumAttributes
};

/* ref :445 */ NVTEQuantizationConfig nvte_create_quantization_config(void);
/* ref :458 */ void nvte_get_quantization_config_attribute(NVTEQuantizationConfig config,
                                                           enum NVTEQuantizationConfigAttribute attr,
                                                           void *buf, size_t size_in_bytes,
                                                           size_t *size_written);
/* ref :469 */ void nvte_set_quantization_config_attribute(NVTEQuantizationConfig config,
                                                           enum NVTEQuantizationConfigAttribute attr,
                                                           const void *buf, size_t size_in_bytes);
/* ref :477 */ void nvte_destroy_quantization_config(NVTEQuantizationConfig config);

/* ref :483.  Returns 0 by default: callers then materialise the pre-transposed
 * ("columnwise") FP8 copy and all three Linear GEMMs reach the MFMA kernel with
 * K-contiguous operands.  Since round 3 the GEMM also reads row-wise payloads of
 * non-TN operands K-major at 95-98 % of that rate: with
 * NVTE_AMD_NON_TN_FP8_GEMM=1 in the environment this returns 1 and a framework
 * may skip the transposes (casts at 3 instead of 4 bytes per element). */
int nvte_is_non_tn_fp8_gemm_supported(void);

/* ref :494 */ void nvte_memset(void *ptr, int value, size_t size_in_bytes, hipStream_t stream);

/* ---- grouped-tensor offsets helpers (ref transformer_engine.h:496-547) ---- */
/* output[0] = 0, output[i+1] = sum_{j<=i} split_sizes[j] * stride; device int64 arrays. */
/* ref :509 */ void nvte_splits_to_offsets(const int64_t *split_sizes, int64_t *output, size_t num_splits,
                                           int64_t stride, hipStream_t stream);
/* output[i] = sum_{j<i} first_dims[j] * last_dims[j]; int32/int64 1-D tensors, output has N+1 entries. */
/* ref :527 */ void nvte_splits_to_offsets_2d(const NVTETensor first_dims, const NVTETensor last_dims,
                                              NVTETensor output, hipStream_t stream);
/* several scaled inclusive scans of one split_sizes tensor (optional leading zero per output). */
/* ref :545 */ void nvte_splits_to_offsets_multi(NVTETensor split_sizes, NVTETensor *outputs,
                                                 const int64_t *strides, const int *include_leading_zero,
                                                 size_t num_outputs, hipStream_t stream);

/* ---- grouped tensors (ref transformer_engine.h:549-662) ----
 * A collection of num_tensors matrices with one dtype and scaling mode, stored back to back in
 * flat 1-D buffers.  first_dims / last_dims / tensor_offsets are DEVICE int64 arrays (absent when
 * the dimension is uniform; tensor_offsets has num_tensors + 1 entries) and are only ever read on
 * the device, so every grouped entry point is safe under HIP graph capture. */
typedef void *NVTEGroupedTensor;

enum NVTEGroupedTensorParam {
  kNVTEGroupedRowwiseData = 0,
  kNVTEGroupedColumnwiseData = 1,
  kNVTEGroupedScale = 2,
  kNVTEGroupedAmax = 3,
  kNVTEGroupedRowwiseScaleInv = 4,
  kNVTEGroupedColumnwiseScaleInv = 5,
  kNVTEGroupedColumnwiseAmax = 6,
  kNVTEGroupedFirstDims = 7,
  kNVTEGroupedLastDims = 8,
  kNVTEGroupedTensorOffsets = 9,
  kNVTEGroupedWithGEMMSwizzledScales = 10,
  kNVTENumGroupedTensorParams
};

/* ref :589 */ NVTEGroupedTensor nvte_create_grouped_tensor(NVTEScalingMode scaling_mode, size_t num_tensors,
                                                            NVTEShape logical_shape);
/* ref :600 */ void nvte_destroy_grouped_tensor(NVTEGroupedTensor tensor);
/* ref :610 */ void nvte_set_grouped_tensor_param(NVTEGroupedTensor tensor, enum NVTEGroupedTensorParam param,
                                                  const void *buf, size_t size_in_bytes);
/* ref :625 */ void nvte_get_grouped_tensor_param(const NVTEGroupedTensor tensor,
                                                  enum NVTEGroupedTensorParam param, void *buf,
                                                  size_t size_in_bytes, size_t *size_written);
/* ref :635 */ size_t nvte_grouped_tensor_num_tensors(const NVTEGroupedTensor tensor);
/* ref :644 */ NVTEDType nvte_grouped_tensor_type(const NVTEGroupedTensor tensor);
/* ref :653 */ NVTEScalingMode nvte_grouped_tensor_scaling_mode(const NVTEGroupedTensor tensor);
/* ref :662 */ NVTEShape nvte_get_grouped_tensor_logical_shape(const NVTEGroupedTensor tensor);

#ifdef __cplusplus
} /* extern "C" */

/* C++ callers also get namespace transformer_engine (DType, TensorWrapper, QuantizationConfigWrapper, ...),
 * ref transformer_engine.h:670-1515: a header-only layer over the entry points above. */
#include "cxx_wrappers.h"
#endif

#endif /* TE_AMD_TRANSFORMER_ENGINE_H_ */
