// Internal host-side types of the MI355X FP8-linear library.
//
// Counterpart of the reference's common/common.h:243-450 (SimpleTensor / Tensor) and
// common/util/logging.h:38-59 (NVTE_CHECK / NVTE_ERROR), written for this library's
// own needs: a tensor descriptor is seven non-owning buffer views plus a scaling
// mode; a handle is a 1-based slot number in a chunked arena (tensor_api.cpp).
#pragma once

#include <hip/hip_runtime_api.h>

#include <array>
#include <cstddef>
#include <cstdint>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "transformer_engine/transformer_engine.h"

namespace te {

// ---------------------------------------------------------------- errors
template <typename T>
inline void append_to(std::ostringstream &os, const T &v) {
  os << v;
}
inline void append_to(std::ostringstream &os, const NVTEShape &s) {
  os << "(";
  for (size_t i = 0; i < s.ndim; ++i) os << (i ? "," : "") << s.data[i];
  os << ")";
}
template <typename T>
inline void append_to(std::ostringstream &os, const std::vector<T> &v) {
  os << "(";
  for (size_t i = 0; i < v.size(); ++i) os << (i ? "," : "") << v[i];
  os << ")";
}
template <typename... Ts>
inline std::string cat(const Ts &...parts) {
  std::ostringstream os;
  (append_to(os, parts), ...);
  return os.str();
}

[[noreturn]] inline void fail(const char *file, int line, const char *func, const std::string &msg) {
  throw std::runtime_error(cat(file, ":", line, " in function ", func, ": ", msg));
}

}  // namespace te

// Same surface as the reference's macros (ref common/util/logging.h:38-59): message
// pieces are streamed together, the throw carries "file:line in function f: msg".
#define NVTE_ERROR(...) ::te::fail(__FILE__, __LINE__, __func__, ::te::cat(__VA_ARGS__))
#define NVTE_CHECK(cond, ...)                                                          \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      ::te::fail(__FILE__, __LINE__, __func__,                                         \
                 ::te::cat("Assertion failed: " #cond ". ", ##__VA_ARGS__));           \
    }                                                                                  \
  } while (0)
#define NVTE_CHECK_HIP(expr)                                                           \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      ::te::fail(__FILE__, __LINE__, __func__, ::te::cat("HIP error: ", hipGetErrorString(e_))); \
    }                                                                                  \
  } while (0)

namespace te {

// ---------------------------------------------------------------- dtypes
using DType = NVTEDType;

inline bool is_fp8(DType t) { return t == kNVTEFloat8E4M3 || t == kNVTEFloat8E5M2; }
inline bool is_high_precision(DType t) {
  return t == kNVTEFloat32 || t == kNVTEFloat16 || t == kNVTEBFloat16;
}
inline size_t dtype_bits(DType t) {
  switch (t) {
    case kNVTEByte: return 8;
    case kNVTEInt16: return 16;
    case kNVTEInt32: return 32;
    case kNVTEInt64: return 64;
    case kNVTEFloat32: return 32;
    case kNVTEFloat16: return 16;
    case kNVTEBFloat16: return 16;
    case kNVTEFloat8E4M3: return 8;
    case kNVTEFloat8E5M2: return 8;
    case kNVTEFloat8E8M0: return 8;
    case kNVTEFloat4E2M1: return 4;
    default: NVTE_ERROR("Invalid dtype ", static_cast<int>(t));
  }
}
const char *dtype_name(DType t);
const char *scaling_mode_name(NVTEScalingMode m);
// Largest finite magnitude of an FP8 format (ref common/utils.cuh:1029-1037).
inline float fp8_max_of(DType t) {
  if (t == kNVTEFloat8E4M3) return 448.f;
  if (t == kNVTEFloat8E5M2) return 57344.f;
  NVTE_ERROR("Expected FP8 dtype, but got ", dtype_name(t));
}

inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }
inline size_t round_up(size_t a, size_t b) { return ceil_div(a, b) * b; }

// ---------------------------------------------------------------- buffer view
// "Absent" is encoded exactly like the reference does (common/common.h:251,269):
// null pointer AND shape == {0}.
struct Buf {
  void *ptr = nullptr;
  NVTEShape shape{{0}, 1};
  DType dtype = kNVTEFloat32;

  Buf() = default;
  explicit Buf(const NVTEBasicTensor &b) : ptr(b.data_ptr), shape(b.shape), dtype(b.dtype) {}
  NVTEBasicTensor as_basic() const { return NVTEBasicTensor{ptr, dtype, shape}; }

  size_t numel() const {
    size_t n = 1;
    for (size_t i = 0; i < shape.ndim; ++i) n *= shape.data[i];
    return n;
  }
  bool present() const { return !(ptr == nullptr && shape.ndim == 1 && shape.data[0] == 0); }
  size_t bytes() const { return (numel() * dtype_bits(dtype) + 7) / 8; }
  void reset() { *this = Buf(); }
};

// A descriptor may carry a shape without a pointer (workspace queries, shape-only scale swizzles); a
// launcher that is about to hand the pointer to a kernel calls this first: a null pointer with a
// non-empty shape must end as an error here, not as a memory fault on the GPU.
#define NVTE_REQUIRE_DATA(buf, what) \
  NVTE_CHECK((buf).ptr != nullptr || (buf).numel() == 0, what, " has a shape but no data pointer")

inline bool same_shape(const NVTEShape &a, const NVTEShape &b) {
  if (a.ndim != b.ndim) return false;
  for (size_t i = 0; i < a.ndim; ++i)
    if (a.data[i] != b.data[i]) return false;
  return true;
}

// ---------------------------------------------------------------- tensor descriptor
struct Tensor {
  Buf data;             // rowwise payload
  Buf columnwise_data;  // per-tensor scaling: the TRANSPOSE; MXFP8: same layout, column blocks
  Buf amax;
  Buf columnwise_amax;
  Buf scale;
  Buf scale_inv;
  Buf columnwise_scale_inv;
  NVTEScalingMode scaling_mode = NVTE_DELAYED_TENSOR_SCALING;
  bool with_gemm_swizzled_scales = false;
  // amd_ext.h: the MXFP8 columnwise payload is stored TRANSPOSED ([last, prod(leading)]) with
  // columnwise scales indexed (column, 32-row block), i.e. as the rowwise form of the transpose.
  bool columnwise_transposed = false;
  bool row_scaled_nvfp4 = false;  // stored only
  int nvfp4_e4m3_max = 448;       // stored only
  uint32_t slot = 0;              // arena slot (handle - 1)
  bool live = false;

  void reset(NVTEScalingMode mode) {
    data.reset();
    columnwise_data.reset();
    amax.reset();
    columnwise_amax.reset();
    scale.reset();
    scale_inv.reset();
    columnwise_scale_inv.reset();
    scaling_mode = mode;
    with_gemm_swizzled_scales = false;
    columnwise_transposed = false;
    row_scaled_nvfp4 = false;
    nvfp4_e4m3_max = 448;
  }

  bool has_data() const { return data.present(); }
  bool has_columnwise_data() const { return columnwise_data.present(); }
  DType dtype() const { return (!has_data() && has_columnwise_data()) ? columnwise_data.dtype : data.dtype; }

  // Logical shape (ref common/common.h:386-418): for per-tensor scaling a tensor that
  // only carries the transposed copy reports the UN-transposed shape.
  NVTEShape shape() const;
  // Collapse to [prod(leading dims), last dim]; a 0-d tensor is 1x1 (ref :425-435).
  std::array<size_t, 2> flat2d() const;
  size_t rows() const { return flat2d()[0]; }
  size_t cols() const { return flat2d()[1]; }
  size_t numel() const {
    NVTEShape s = shape();
    size_t n = 1;
    for (size_t i = 0; i < s.ndim; ++i) n *= s.data[i];
    return n;
  }
};

inline bool is_tensor_scaling(NVTEScalingMode m) { return m == NVTE_DELAYED_TENSOR_SCALING; }
inline bool is_mxfp8(NVTEScalingMode m) { return m == NVTE_MXFP8_1D_SCALING; }

// Handle <-> descriptor.  lookup() returns nullptr for the null handle;
// lookup_checked() throws for a null or dead handle.
Tensor *lookup(NVTETensor h);
Tensor *lookup_checked(NVTETensor h);

// Validators (ref common/transformer_engine.cpp:44-259).
void check_noop(const Tensor &t, const char *name);
void check_scale_shapes(const Tensor &t, const char *name);
void check_input(const Tensor &t, const char *name, bool check_scale_shapes_too = true);
void check_output(const Tensor &t, const char *name, bool allow_empty = false);

// Stream-ordered fill by a kernel of this library (tensor_api.cpp).  Used instead of hipMemsetAsync
// wherever a launch may be captured into a HIP graph: a memset node followed by a kernel node with a
// large argument block was seen to write a 16-byte pattern of that kernel's arguments instead of the
// fill value from the second replay on (ROCm 7.2, gfx950; scratch/dbg_splitk_graph.py).
void fill_async(void *ptr, int value, size_t bytes, hipStream_t stream);
inline void zero_async(void *ptr, size_t bytes, hipStream_t stream) { fill_async(ptr, 0, bytes, stream); }

// ---------------------------------------------------------------- grouped tensor descriptor
// ref common/common.h:452-600.  All payload buffers are flat 1-D; member i is the
// [first_i, last_i] matrix at element offset tensor_offsets[i] (or i * M * N when uniform).
struct GroupedTensor {
  Buf data, columnwise_data, scale_inv, columnwise_scale_inv, amax, columnwise_amax, scale;
  Buf first_dims, last_dims, tensor_offsets;  // device int64 arrays, absent when uniform
  NVTEScalingMode scaling_mode = NVTE_DELAYED_TENSOR_SCALING;
  size_t num_tensors = 0;
  NVTEShape logical_shape{{0}, 1};
  bool with_gemm_swizzled_scales = false;
  // MXFP8 extension (amd_ext.h): columnwise_data holds each member's payload TRANSPOSED ([last, first_i] at
  // the member's own offset) with scales [pad128(last), pad4(first_i / 32)] per member, back to back
  bool columnwise_transposed = false;
  bool live = false;

  bool has_data() const { return data.present(); }
  bool has_columnwise_data() const { return columnwise_data.present(); }
  bool all_same_first_dim() const { return !first_dims.present(); }
  bool all_same_last_dim() const { return !last_dims.present(); }
  bool all_same_shape() const { return all_same_first_dim() && all_same_last_dim(); }
  bool varying_both_dims() const { return !all_same_first_dim() && !all_same_last_dim(); }
  DType dtype() const { return (!has_data() && has_columnwise_data()) ? columnwise_data.dtype : data.dtype; }
  size_t common_first_dim() const;  // throws when the first dim varies
  size_t common_last_dim() const;   // throws when the last dim varies
};

GroupedTensor *lookup_grouped(NVTEGroupedTensor h);          // nullptr for null / dead handles
GroupedTensor *lookup_grouped_checked(NVTEGroupedTensor h);  // throws instead
// ref common/transformer_engine.cpp:261-395
void check_grouped_input(const GroupedTensor &t, const char *name);
void check_grouped_output(const GroupedTensor &t, const char *name, bool allow_empty = false);

// Member shapes as the kernels take them: uniform values from the host side of the descriptor,
// per-member arrays as device pointers (null = uniform).
struct GroupShapes {
  const int64_t *first_dims = nullptr;
  const int64_t *last_dims = nullptr;
  const int64_t *offsets = nullptr;  // num_tensors + 1 entries, or null
  int64_t uniform_first = 0;
  int64_t uniform_last = 0;
  unsigned num_tensors = 0;
};
GroupShapes group_shapes_of(const GroupedTensor &t);

struct GroupedMatmulConfig {
  int64_t avg_m = 0, avg_n = 0, avg_k = 0;
  bool has_avg_m = false, has_avg_n = false, has_avg_k = false;
  int sm_count = 0;
  bool use_split_accumulator = false;
};

// ---------------------------------------------------------------- configs
struct QuantConfig {
  bool force_pow_2_scales = false;
  float amax_epsilon = 0.f;
  NVTETensor noop_tensor = nullptr;
  int float8_block_scale_tensor_format = 0;
  NVTETensor rng_state = nullptr;
  bool nvfp4_2d_quantization = false;
  bool stochastic_rounding = false;
  bool use_fast_math = false;
  uint8_t nvfp4_4over6_mode = 0;
  bool nvfp4_4over6_err_use_fast_math = false;
};

struct MatmulConfig {
  NVTETensor bias_tensor = nullptr;
  NVTETensor dbias_tensor = nullptr;
  bool with_gelu_epilogue = false;
  bool with_dgelu_epilogue = false;
  NVTETensor epilogue_aux_tensor = nullptr;
  bool use_split_accumulator = false;
  int sm_count = 0;
};

// noop flag pointer out of an optional config (ref common/cast/dispatch/quantize.cuh:46-51)
const float *noop_ptr_of(const QuantConfig *cfg);

// ---------------------------------------------------------------- API guard
int error_mode();
void record_error(const char *api, const char *what);
void api_enter(const char *api);  // clears this thread's recorded error; trace hook

template <typename F>
inline auto api_guard(const char *name, F &&body) -> decltype(body()) {
  api_enter(name);
  if (error_mode() == 0) return body();
  try {
    return body();
  } catch (const std::exception &e) {
    record_error(name, e.what());
  } catch (...) {
    record_error(name, "unknown C++ exception");
  }
  return decltype(body())();
}

// ---------------------------------------------------------------- device info
struct DeviceInfo {
  int num_cus = 0;
  int num_xcds = 8;
  bool is_gfx950 = false;
};
// Cached per current device.  Throws when no GPU is usable: the product path never
// falls back to the CPU.
const DeviceInfo &device_info();

// ---------------------------------------------------------------- kernel launchers
// (each lives in the .hip file of the same topic; all only ENQUEUE on `stream`)
void cast_transpose_launch(const Tensor &in, const float *noop, Tensor &out, hipStream_t stream);
void cast_rowwise_launch(const Tensor &in, const float *noop, Tensor &out, hipStream_t stream);
void cast_dbias_launch(const Tensor &in, Tensor &out, Tensor &dbias, Tensor &workspace,
                       bool want_transpose, hipStream_t stream);
void mxfp8_quantize_launch(const Tensor &in, const float *noop, Tensor &out, hipStream_t stream);
void mxfp8_quantize_dbias_launch(const Tensor &in, Tensor &out, Tensor &dbias, Tensor &workspace,
                                 hipStream_t stream);
void dequantize_launch(const Tensor &in, Tensor &out, hipStream_t stream);
void transpose_launch(const Tensor &in, const float *noop, Tensor &out, hipStream_t stream);
void fp8_transpose_dbias_launch(const Tensor &in, Tensor &out, Tensor &dbias, Tensor &workspace,
                                hipStream_t stream);  // dbias.hip
void amax_launch(const Tensor &in, float *amax, const float *noop, hipStream_t stream);
void scale_from_amax_launch(const float *amax, float *scale, float max_fp8, bool pow2, float eps,
                            const float *noop, hipStream_t stream);
void quantize_dispatch(const Tensor &in, const float *noop, Tensor &out, hipStream_t stream);

// Fan a list of independent launches over the compute-stream pool and join them back on the caller's
// stream (ref common/util/multi_stream.cpp:20-60; callers cublaslt_gemm.cu:994-1045, cast.cu:65-88).
// Inactive unless NVTE_AMD_MULTI_STREAM is set (quantize_api.cpp): stream_of() is then the caller's stream.
class MultiStreamScope {
 public:
  MultiStreamScope(hipStream_t caller, size_t num_items);
  hipStream_t stream_of(size_t item) const;
  void join();  // record every used pool stream's own event, make the caller's stream wait for it

 private:
  hipStream_t caller_;
  int mode_ = 0, used_ = 0;
  bool joined_ = false;
};

}  // namespace te
