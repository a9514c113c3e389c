/* Header-only C++ layer over the tensor-handle C ABI: namespace transformer_engine.
 *
 * Replaces the C++ half of the reference's transformer_engine.h (ref :670-1515): DType, the dtype
 * predicates, TensorWrapper, MultiTensorWrapper, GroupedTensorWrapper, Float8BlockScaleTensorFormat and
 * QuantizationConfigWrapper - the types the reference's framework shims (pytorch/csrc, jax/csrc) and its
 * C++ tests are written against.  Same class names, member names, argument order and defaults, so such a
 * translation unit compiles against include/ unchanged; the bodies are this library's own and only call
 * the C entry points declared in transformer_engine.h.
 *
 * Included at the end of transformer_engine.h when compiled as C++; do not include directly.
 */
#ifndef TE_AMD_CXX_WRAPPERS_H_
#define TE_AMD_CXX_WRAPPERS_H_

#ifndef TE_AMD_TRANSFORMER_ENGINE_H_
#error "include <transformer_engine/transformer_engine.h> instead"
#endif

#include <utility>
#include <vector>

namespace transformer_engine {

/* ref :678-691.  Same numeric values as NVTEDType. */
enum class DType {
  kByte = 0,
  kInt16 = 1,
  kInt32 = 2,
  kInt64 = 3,
  kFloat32 = 4,
  kFloat16 = 5,
  kBFloat16 = 6,
  kFloat8E4M3 = 7,
  kFloat8E5M2 = 8,
  kFloat8E8M0 = 9,
  kFloat4E2M1 = 10,
  kNumTypes
};

/* ref :698-716 */
inline bool is_fp8_dtype(const DType t) { return t == DType::kFloat8E4M3 || t == DType::kFloat8E5M2; }
inline bool is_fp4_dtype(const DType t) { return t == DType::kFloat4E2M1; }
inline bool is_high_precision_dtype(const DType t) {
  return t == DType::kFloat32 || t == DType::kBFloat16 || t == DType::kFloat16;
}

namespace detail {
/* Shapes arrive either as NVTEShape or as std::vector<size_t>. */
inline NVTEShape to_shape(const NVTEShape &s) { return s; }
inline NVTEShape to_shape(const std::vector<size_t> &s) { return nvte_make_shape(s.data(), s.size()); }
inline NVTEBasicTensor view(void *ptr, DType type, const NVTEShape &shape) {
  NVTEBasicTensor b;
  b.data_ptr = ptr;
  b.dtype = static_cast<NVTEDType>(type);
  b.shape = shape;
  return b;
}
}  // namespace detail

/* ref :721-1109.  Owns an NVTETensor HANDLE (never the buffers it points at). */
class TensorWrapper {
 public:
  static constexpr size_t defaultData = 1;
  static constexpr NVTEShape defaultShape = {{defaultData, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 1};
  static constexpr NVTEShape emptyShape = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 1};

  /* ref :738-766.  A scale_inv left at its default (no pointer, shape {1}) means "absent". */
  TensorWrapper(void *dptr, const NVTEShape &shape, const DType dtype, float *amax_dptr = nullptr,
                float *scale_dptr = nullptr, float *scale_inv_dptr = nullptr,
                NVTEShape scale_inv_shape = defaultShape,
                const NVTEScalingMode scaling_mode = NVTE_DELAYED_TENSOR_SCALING)
      : tensor_(nvte_create_tensor(scaling_mode)) {
    put(kNVTERowwiseData, detail::view(dptr, dtype, shape));
    put(kNVTEAmax, detail::view(amax_dptr, DType::kFloat32, amax_dptr ? defaultShape : emptyShape));
    put(kNVTEScale, detail::view(scale_dptr, DType::kFloat32, scale_dptr ? defaultShape : emptyShape));
    const bool default_sinv = scale_inv_dptr == nullptr && scale_inv_shape.ndim == 1 &&
                              scale_inv_shape.data[0] == defaultData;
    put(kNVTERowwiseScaleInv,
        detail::view(scale_inv_dptr, DType::kFloat32, default_sinv ? emptyShape : scale_inv_shape));
  }

  /* ref :775-786 */
  TensorWrapper(void *dptr, const std::vector<size_t> &shape, const DType dtype, float *amax_dptr = nullptr,
                float *scale_dptr = nullptr, float *scale_inv_dptr = nullptr,
                const std::vector<size_t> &scale_inv_shape = {1},
                const NVTEScalingMode scaling_mode = NVTE_DELAYED_TENSOR_SCALING)
      : TensorWrapper(dptr, detail::to_shape(shape), dtype, amax_dptr, scale_dptr, scale_inv_dptr,
                      detail::to_shape(scale_inv_shape), scaling_mode) {}

  /* ref :788: an empty tensor of the given scaling mode */
  explicit TensorWrapper(const NVTEScalingMode scaling_mode = NVTE_DELAYED_TENSOR_SCALING)
      : tensor_(nvte_create_tensor(scaling_mode)) {}

  ~TensorWrapper() { nvte_destroy_tensor(tensor_); }

  TensorWrapper(const TensorWrapper &) = delete;
  TensorWrapper &operator=(const TensorWrapper &) = delete;
  TensorWrapper(TensorWrapper &&other) noexcept : tensor_(other.tensor_) { other.tensor_ = nullptr; }
  TensorWrapper &operator=(TensorWrapper &&other) noexcept {
    if (this != &other) {
      nvte_destroy_tensor(tensor_);
      tensor_ = other.tensor_;
      other.tensor_ = nullptr;
    }
    return *this;
  }

  /* ---- setters (ref :823-880) ---- */
  template <typename ShapeType>
  TensorWrapper &set_parameter(const NVTETensorParam param, void *dptr, DType type,
                               const ShapeType &shape) noexcept {
    put(param, detail::view(dptr, type, detail::to_shape(shape)));
    return *this;
  }
  template <typename ShapeType>
  TensorWrapper &set_rowwise_data(void *dptr, DType type, const ShapeType &shape) noexcept {
    return set_parameter(kNVTERowwiseData, dptr, type, shape);
  }
  template <typename ShapeType>
  TensorWrapper &set_columnwise_data(void *dptr, DType type, const ShapeType &shape) noexcept {
    return set_parameter(kNVTEColumnwiseData, dptr, type, shape);
  }
  template <typename ShapeType>
  TensorWrapper &set_scale(void *dptr, DType type, const ShapeType &shape) noexcept {
    return set_parameter(kNVTEScale, dptr, type, shape);
  }
  template <typename ShapeType>
  TensorWrapper &set_amax(void *dptr, DType type, const ShapeType &shape) noexcept {
    return set_parameter(kNVTEAmax, dptr, type, shape);
  }
  template <typename ShapeType>
  TensorWrapper &set_rowwise_scale_inv(void *dptr, DType type, const ShapeType &shape) noexcept {
    return set_parameter(kNVTERowwiseScaleInv, dptr, type, shape);
  }
  template <typename ShapeType>
  TensorWrapper &set_columnwise_scale_inv(void *dptr, DType type, const ShapeType &shape) noexcept {
    return set_parameter(kNVTEColumnwiseScaleInv, dptr, type, shape);
  }
  template <typename ShapeType>
  TensorWrapper &set_columnwise_amax(void *dptr, DType type, const ShapeType &shape) noexcept {
    return set_parameter(kNVTEColumnwiseAmax, dptr, type, shape);
  }
  void set_with_gemm_swizzled_scales(bool flag) { put_flag(kNVTEWithGEMMSwizzledScales, flag); }
  void set_row_scaled_nvfp4(bool flag) { put_flag(kNVTERowScaledNVFP4, flag); }
  void set_nvfp4_e4m3_max(int value) {
    nvte_set_tensor_param_v2(tensor_, kNVTENVFP4E4M3Max, &value, sizeof(value));
  }

  /* ---- getters (ref :884-930) ---- */
  NVTEBasicTensor get_parameter(const NVTETensorParam param) const noexcept {
    NVTEBasicTensor out;
    nvte_get_tensor_param_v2(tensor_, param, &out, sizeof(out), nullptr);
    return out;
  }
  NVTEBasicTensor get_rowwise_data() const noexcept { return get_parameter(kNVTERowwiseData); }
  NVTEBasicTensor get_columnwise_data() const noexcept { return get_parameter(kNVTEColumnwiseData); }
  NVTEBasicTensor get_scale() const noexcept { return get_parameter(kNVTEScale); }
  NVTEBasicTensor get_amax() const noexcept { return get_parameter(kNVTEAmax); }
  NVTEBasicTensor get_rowwise_scale_inv() const noexcept { return get_parameter(kNVTERowwiseScaleInv); }
  NVTEBasicTensor get_columnwise_scale_inv() const noexcept { return get_parameter(kNVTEColumnwiseScaleInv); }
  NVTEBasicTensor get_columnwise_amax() const noexcept { return get_parameter(kNVTEColumnwiseAmax); }
  bool get_with_gemm_swizzled_scales() const { return get_flag(kNVTEWithGEMMSwizzledScales); }
  bool get_row_scaled_nvfp4() const { return get_flag(kNVTERowScaledNVFP4); }
  int get_nvfp4_e4m3_max() const {
    int value = 0;
    nvte_get_tensor_param_v2(tensor_, kNVTENVFP4E4M3Max, &value, sizeof(value), nullptr);
    return value;
  }

  /* ---- queries (ref :932-1090).  A moved-from wrapper answers like an empty tensor. ---- */
  NVTETensor data() const noexcept { return tensor_; }
  const NVTEShape shape() const noexcept { return tensor_ ? nvte_tensor_shape(tensor_) : emptyShape; }
  const NVTEShape columnwise_shape() const noexcept {
    return tensor_ ? nvte_tensor_columnwise_shape(tensor_) : emptyShape;
  }
  size_t size(const size_t dim) const { return tensor_ ? nvte_tensor_size(tensor_, dim) : 0; }
  size_t ndim() const noexcept { return tensor_ ? nvte_tensor_ndims(tensor_) : 0; }
  size_t numel() const noexcept { return tensor_ ? nvte_tensor_numel(tensor_) : 0; }
  size_t element_size() const noexcept { return tensor_ ? nvte_tensor_element_size(tensor_) : 0; }
  size_t element_size_bits() const noexcept { return tensor_ ? nvte_tensor_element_size_bits(tensor_) : 0; }
  size_t bytes() const noexcept { return tensor_ ? nvte_tensor_size_bytes(tensor_) : 0; }
  DType dtype() const noexcept {
    return tensor_ ? static_cast<DType>(nvte_tensor_type(tensor_)) : DType::kNumTypes;
  }
  void *dptr() const noexcept { return tensor_ ? nvte_tensor_data(tensor_) : nullptr; }
  void *columnwise_dptr() const noexcept { return tensor_ ? nvte_tensor_columnwise_data(tensor_) : nullptr; }
  float *amax() const noexcept { return tensor_ ? nvte_tensor_amax(tensor_) : nullptr; }
  float *scale() const noexcept { return tensor_ ? nvte_tensor_scale(tensor_) : nullptr; }
  float *scale_inv() const noexcept { return tensor_ ? nvte_tensor_scale_inv(tensor_) : nullptr; }
  const NVTEShape scale_inv_shape() const noexcept {
    return tensor_ ? nvte_tensor_scale_inv_shape(tensor_) : emptyShape;
  }
  NVTEScalingMode scaling_mode() const noexcept {
    return tensor_ ? nvte_tensor_scaling_mode(tensor_) : NVTE_DELAYED_TENSOR_SCALING;
  }
  void zero_(hipStream_t stream) { nvte_zero_tensor(tensor_, stream); }

 private:
  void put(NVTETensorParam param, const NVTEBasicTensor &b) noexcept {
    nvte_set_tensor_param_v2(tensor_, param, &b, sizeof(b));
  }
  void put_flag(NVTETensorParam param, bool flag) {
    const uint8_t v = flag ? 1 : 0;
    nvte_set_tensor_param_v2(tensor_, param, &v, sizeof(v));
  }
  bool get_flag(NVTETensorParam param) const {
    uint8_t v = 0;
    nvte_get_tensor_param_v2(tensor_, param, &v, sizeof(v), nullptr);
    return v != 0;
  }
  NVTETensor tensor_ = nullptr;
};

/* ref :1112-1175.  A batch of handles created and destroyed with one call each. */
class MultiTensorWrapper {
 public:
  MultiTensorWrapper() = default;
  explicit MultiTensorWrapper(size_t num_tensors, NVTEScalingMode scaling_mode = NVTE_DELAYED_TENSOR_SCALING)
      : tensors_(num_tensors) {
    if (!tensors_.empty()) nvte_create_tensors(scaling_mode, tensors_.data(), tensors_.size());
  }
  ~MultiTensorWrapper() { release(); }
  MultiTensorWrapper(const MultiTensorWrapper &) = delete;
  MultiTensorWrapper &operator=(const MultiTensorWrapper &) = delete;
  MultiTensorWrapper(MultiTensorWrapper &&) noexcept = default;
  MultiTensorWrapper &operator=(MultiTensorWrapper &&other) noexcept {
    if (this != &other) {
      release();
      tensors_ = std::move(other.tensors_);
    }
    return *this;
  }
  size_t size() const noexcept { return tensors_.size(); }
  bool empty() const noexcept { return tensors_.empty(); }
  NVTETensor operator[](size_t i) const noexcept { return tensors_[i]; }
  NVTETensor *data() noexcept { return tensors_.data(); }
  const NVTETensor *data() const noexcept { return tensors_.data(); }
  operator NVTETensor *() noexcept { return tensors_.data(); }
  auto begin() noexcept { return tensors_.begin(); }
  auto end() noexcept { return tensors_.end(); }
  auto begin() const noexcept { return tensors_.begin(); }
  auto end() const noexcept { return tensors_.end(); }

 private:
  void release() noexcept {
    if (!tensors_.empty()) nvte_destroy_tensors(tensors_.data(), tensors_.size());
    tensors_.clear();
  }
  std::vector<NVTETensor> tensors_;
};

/* ref :1180-1396 */
class GroupedTensorWrapper {
 public:
  static constexpr size_t defaultData = 1;
  static constexpr NVTEShape defaultShape = {{defaultData, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 1};
  static constexpr NVTEShape emptyShape = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 1};

  GroupedTensorWrapper(const size_t num_tensors, const NVTEShape &logical_shape,
                       const NVTEScalingMode scaling_mode = NVTE_DELAYED_TENSOR_SCALING)
      : tensor_(nvte_create_grouped_tensor(scaling_mode, num_tensors, logical_shape)) {}
  GroupedTensorWrapper(const size_t num_tensors, const std::vector<size_t> &logical_shape,
                       const NVTEScalingMode scaling_mode = NVTE_DELAYED_TENSOR_SCALING)
      : GroupedTensorWrapper(num_tensors, detail::to_shape(logical_shape), scaling_mode) {}
  ~GroupedTensorWrapper() { nvte_destroy_grouped_tensor(tensor_); }
  GroupedTensorWrapper(const GroupedTensorWrapper &) = delete;
  GroupedTensorWrapper &operator=(const GroupedTensorWrapper &) = delete;
  GroupedTensorWrapper(GroupedTensorWrapper &&other) noexcept : tensor_(other.tensor_) { other.tensor_ = nullptr; }
  GroupedTensorWrapper &operator=(GroupedTensorWrapper &&other) noexcept {
    if (this != &other) {
      nvte_destroy_grouped_tensor(tensor_);
      tensor_ = other.tensor_;
      other.tensor_ = nullptr;
    }
    return *this;
  }

  template <typename ShapeType>
  GroupedTensorWrapper &set_parameter(const NVTEGroupedTensorParam param, void *dptr, DType type,
                                      const ShapeType &shape) noexcept {
    const NVTEBasicTensor b = detail::view(dptr, type, detail::to_shape(shape));
    nvte_set_grouped_tensor_param(tensor_, param, &b, sizeof(b));
    return *this;
  }
#define TE_AMD_GROUPED_SETTER(name, param)                                                          \
  template <typename ShapeType>                                                                     \
  GroupedTensorWrapper &name(void *dptr, DType type, const ShapeType &shape) noexcept {             \
    return set_parameter(param, dptr, type, shape);                                                 \
  }
  TE_AMD_GROUPED_SETTER(set_rowwise_data, kNVTEGroupedRowwiseData)
  TE_AMD_GROUPED_SETTER(set_columnwise_data, kNVTEGroupedColumnwiseData)
  TE_AMD_GROUPED_SETTER(set_scale, kNVTEGroupedScale)
  TE_AMD_GROUPED_SETTER(set_amax, kNVTEGroupedAmax)
  TE_AMD_GROUPED_SETTER(set_rowwise_scale_inv, kNVTEGroupedRowwiseScaleInv)
  TE_AMD_GROUPED_SETTER(set_columnwise_scale_inv, kNVTEGroupedColumnwiseScaleInv)
  TE_AMD_GROUPED_SETTER(set_columnwise_amax, kNVTEGroupedColumnwiseAmax)
  TE_AMD_GROUPED_SETTER(set_first_dims, kNVTEGroupedFirstDims)
  TE_AMD_GROUPED_SETTER(set_last_dims, kNVTEGroupedLastDims)
  TE_AMD_GROUPED_SETTER(set_tensor_offsets, kNVTEGroupedTensorOffsets)
#undef TE_AMD_GROUPED_SETTER
  void set_with_gemm_swizzled_scales(bool flag) {
    const uint8_t v = flag ? 1 : 0;
    nvte_set_grouped_tensor_param(tensor_, kNVTEGroupedWithGEMMSwizzledScales, &v, sizeof(v));
  }

  NVTEBasicTensor get_parameter(const NVTEGroupedTensorParam param) const noexcept {
    NVTEBasicTensor out;
    nvte_get_grouped_tensor_param(tensor_, param, &out, sizeof(out), nullptr);
    return out;
  }
  NVTEBasicTensor get_rowwise_data() const noexcept { return get_parameter(kNVTEGroupedRowwiseData); }
  NVTEBasicTensor get_columnwise_data() const noexcept { return get_parameter(kNVTEGroupedColumnwiseData); }
  NVTEBasicTensor get_scale() const noexcept { return get_parameter(kNVTEGroupedScale); }
  NVTEBasicTensor get_amax() const noexcept { return get_parameter(kNVTEGroupedAmax); }
  NVTEBasicTensor get_rowwise_scale_inv() const noexcept { return get_parameter(kNVTEGroupedRowwiseScaleInv); }
  NVTEBasicTensor get_columnwise_scale_inv() const noexcept { return get_parameter(kNVTEGroupedColumnwiseScaleInv); }
  NVTEBasicTensor get_columnwise_amax() const noexcept { return get_parameter(kNVTEGroupedColumnwiseAmax); }
  NVTEBasicTensor get_first_dims() const noexcept { return get_parameter(kNVTEGroupedFirstDims); }
  NVTEBasicTensor get_last_dims() const noexcept { return get_parameter(kNVTEGroupedLastDims); }
  NVTEBasicTensor get_tensor_offsets() const noexcept { return get_parameter(kNVTEGroupedTensorOffsets); }
  bool get_with_gemm_swizzled_scales() const {
    uint8_t v = 0;
    nvte_get_grouped_tensor_param(tensor_, kNVTEGroupedWithGEMMSwizzledScales, &v, sizeof(v), nullptr);
    return v != 0;
  }

  NVTEGroupedTensor data() const noexcept { return tensor_; }
  size_t num_tensors() const noexcept { return tensor_ ? nvte_grouped_tensor_num_tensors(tensor_) : 0; }
  DType dtype() const noexcept {
    return tensor_ ? static_cast<DType>(nvte_grouped_tensor_type(tensor_)) : DType::kNumTypes;
  }
  NVTEScalingMode scaling_mode() const noexcept {
    return tensor_ ? nvte_grouped_tensor_scaling_mode(tensor_) : NVTE_DELAYED_TENSOR_SCALING;
  }
  const NVTEShape logical_shape() const noexcept {
    return tensor_ ? nvte_get_grouped_tensor_logical_shape(tensor_) : emptyShape;
  }

 private:
  NVTEGroupedTensor tensor_ = nullptr;
};

/* ref :1398-1405 (stored by the config; FP8 block scaling itself is out of scope) */
enum class Float8BlockScaleTensorFormat { GEMM_READY = 0, COMPACT = 1, INVALID };

/* ref :1409-1513 */
class QuantizationConfigWrapper {
 public:
  QuantizationConfigWrapper() : config_{nvte_create_quantization_config()} {}
  QuantizationConfigWrapper(const QuantizationConfigWrapper &) = delete;
  QuantizationConfigWrapper &operator=(const QuantizationConfigWrapper &) = delete;
  QuantizationConfigWrapper(QuantizationConfigWrapper &&other) noexcept : config_{other.config_} {
    other.config_ = nullptr;
  }
  QuantizationConfigWrapper &operator=(QuantizationConfigWrapper &&other) noexcept {
    if (this != &other) {
      reset();
      config_ = other.config_;
      other.config_ = nullptr;
    }
    return *this;
  }
  ~QuantizationConfigWrapper() { reset(); }

  operator NVTEQuantizationConfig() const noexcept { return config_; }

  void set_force_pow_2_scales(bool v) { put_u8(kNVTEQuantizationConfigForcePow2Scales, v); }
  void set_amax_epsilon(float v) {
    nvte_set_quantization_config_attribute(config_, kNVTEQuantizationConfigAmaxEpsilon, &v, sizeof(v));
  }
  void set_noop_tensor(NVTETensor t) {
    nvte_set_quantization_config_attribute(config_, kNVTEQuantizationConfigNoopTensor, &t, sizeof(t));
  }
  void set_float8_block_scale_tensor_format(Float8BlockScaleTensorFormat format) {
    const int v = static_cast<int>(format);
    nvte_set_quantization_config_attribute(config_, kNVTEQuantizationConfigFloat8BlockScaleTensorFormat, &v,
                                           sizeof(v));
  }
  void set_rng_state(NVTETensor t) {
    nvte_set_quantization_config_attribute(config_, kNVTEQuantizationConfigRNGState, &t, sizeof(t));
  }
  void set_nvfp4_2d_quantization(bool v) { put_u8(kNVTEQuantizationConfigNVFP42DQuantization, v); }
  void set_stochastic_rounding(bool v) { put_u8(kNVTEQuantizationConfigStochasticRounding, v); }
  void set_use_fast_math(bool v) { put_u8(kNVTEQuantizationConfigUseFastMath, v); }
  void set_nvfp4_4over6_err_use_fast_math(bool v) { put_u8(kNVTEQuantizationConfigNVFP44Over6ErrUseFastMath, v); }

 private:
  void put_u8(NVTEQuantizationConfigAttribute attr, bool flag) {
    const uint8_t v = flag ? 1 : 0;
    nvte_set_quantization_config_attribute(config_, attr, &v, sizeof(v));
  }
  void reset() noexcept {
    if (config_ != nullptr) nvte_destroy_quantization_config(config_);
    config_ = nullptr;
  }
  NVTEQuantizationConfig config_ = nullptr;
};

}  // namespace transformer_engine

#endif /* TE_AMD_CXX_WRAPPERS_H_ */
