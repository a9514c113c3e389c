// Grouped-tensor handles (NVTEGroupedTensor), their validators and the grouped matmul config.
// Host only.  Behaviour follows ref common/transformer_engine.cpp:261-395 (validators),
// :500-580 (allocator), :1147-1365 (C API) and common/gemm/config.cpp (grouped config).
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

#include "common.h"
#include "transformer_engine/amd_ext.h"
#include "transformer_engine/gemm.h"

namespace te {
namespace {

// Handles are 1-based indices into a deque (stable addresses), recycled through a free list.
class GroupedArena {
 public:
  static GroupedArena &get() {
    static GroupedArena a;
    return a;
  }
  NVTEGroupedTensor create(NVTEScalingMode mode, size_t num_tensors, NVTEShape logical_shape) {
    std::lock_guard<std::mutex> g(mu_);
    size_t slot;
    if (!free_.empty()) {
      slot = free_.back();
      free_.pop_back();
    } else {
      NVTE_CHECK(slots_.size() < kMax, "Cannot allocate a new NVTEGroupedTensor. Maximum number of "
                 "grouped tensors reached: ", kMax, ". There is probably a memory leak in your application.");
      slot = slots_.size();
      slots_.emplace_back();
    }
    GroupedTensor &t = slots_[slot];
    t = GroupedTensor();
    t.scaling_mode = mode;
    t.num_tensors = num_tensors;
    t.logical_shape = logical_shape;
    t.live = true;
    return reinterpret_cast<NVTEGroupedTensor>(slot + 1);
  }
  void destroy(NVTEGroupedTensor h) {
    std::lock_guard<std::mutex> g(mu_);
    const uintptr_t i = reinterpret_cast<uintptr_t>(h);
    if (i == 0) return;
    NVTE_CHECK(i <= slots_.size(), "Invalid grouped tensor.");
    GroupedTensor &t = slots_[i - 1];
    if (!t.live) return;
    t = GroupedTensor();
    free_.push_back(i - 1);
  }
  GroupedTensor *find(NVTEGroupedTensor h) {
    std::lock_guard<std::mutex> g(mu_);
    const uintptr_t i = reinterpret_cast<uintptr_t>(h);
    if (i == 0 || i > slots_.size()) return nullptr;
    GroupedTensor *t = &slots_[i - 1];
    return t->live ? t : nullptr;
  }

 private:
  static constexpr size_t kMax = 1u << 20;
  std::mutex mu_;
  std::deque<GroupedTensor> slots_;
  std::vector<size_t> free_;
};

Buf *grouped_buf(GroupedTensor &t, NVTEGroupedTensorParam p) {
  switch (p) {
    case kNVTEGroupedRowwiseData: return &t.data;
    case kNVTEGroupedColumnwiseData: return &t.columnwise_data;
    case kNVTEGroupedScale: return &t.scale;
    case kNVTEGroupedAmax: return &t.amax;
    case kNVTEGroupedRowwiseScaleInv: return &t.scale_inv;
    case kNVTEGroupedColumnwiseScaleInv: return &t.columnwise_scale_inv;
    case kNVTEGroupedColumnwiseAmax: return &t.columnwise_amax;
    case kNVTEGroupedFirstDims: return &t.first_dims;
    case kNVTEGroupedLastDims: return &t.last_dims;
    case kNVTEGroupedTensorOffsets: return &t.tensor_offsets;
    default: return nullptr;
  }
}

size_t grouped_param_size(NVTEGroupedTensorParam p) {
  return p == kNVTEGroupedWithGEMMSwizzledScales ? sizeof(uint8_t) : sizeof(NVTEBasicTensor);
}

}  // namespace

GroupedTensor *lookup_grouped(NVTEGroupedTensor h) { return GroupedArena::get().find(h); }
GroupedTensor *lookup_grouped_checked(NVTEGroupedTensor h) {
  GroupedTensor *t = GroupedArena::get().find(h);
  NVTE_CHECK(t != nullptr, "Invalid grouped tensor.");
  return t;
}

size_t GroupedTensor::common_first_dim() const {
  NVTE_CHECK(all_same_first_dim(), "First dim varies across tensors");
  NVTE_CHECK(logical_shape.ndim == 2, "Logical shape must be 2D");
  return all_same_shape() ? logical_shape.data[0] / num_tensors : logical_shape.data[0];
}
size_t GroupedTensor::common_last_dim() const {
  NVTE_CHECK(all_same_last_dim(), "Last dim varies across tensors");
  NVTE_CHECK(logical_shape.ndim == 2, "Logical shape must be 2D");
  return logical_shape.data[1];
}

GroupShapes group_shapes_of(const GroupedTensor &t) {
  GroupShapes g;
  g.first_dims = static_cast<const int64_t *>(t.first_dims.ptr);
  g.last_dims = static_cast<const int64_t *>(t.last_dims.ptr);
  g.offsets = static_cast<const int64_t *>(t.tensor_offsets.ptr);
  g.uniform_first = t.all_same_first_dim() ? static_cast<int64_t>(t.common_first_dim()) : 0;
  g.uniform_last = t.all_same_last_dim() ? static_cast<int64_t>(t.common_last_dim()) : 0;
  g.num_tensors = static_cast<unsigned>(t.num_tensors);
  return g;
}

static void check_group_shape_arrays(const GroupedTensor &t, const char *name) {
  NVTE_CHECK(t.num_tensors > 0, "Grouped tensor ", name, " has no tensors!");
  auto check_arr = [&](const Buf &arr, const char *arr_name, size_t want) {
    if (!arr.present()) return;
    NVTE_CHECK(arr.shape.ndim == 1, "Grouped tensor ", name, " ", arr_name, " must be 1D");
    NVTE_CHECK(arr.dtype == kNVTEInt64, "Grouped tensor ", name, " ", arr_name, " must have dtype Int64");
    NVTE_CHECK(arr.shape.data[0] == want, "Grouped tensor ", name, " ", arr_name, " size (",
               arr.shape.data[0], ") must equal ", want);
  };
  check_arr(t.first_dims, "first_dims", t.num_tensors);
  check_arr(t.last_dims, "last_dims", t.num_tensors);
  check_arr(t.tensor_offsets, "tensor_offsets", t.num_tensors + 1);
  if (!t.all_same_shape()) {
    NVTE_CHECK(t.tensor_offsets.ptr != nullptr, "Grouped tensor ", name,
               " must have tensor_offsets when any dimension varies (first_dims or last_dims is set)");
  }
  NVTE_CHECK(t.logical_shape.ndim == 2, "Grouped tensor ", name, " logical_shape must be 2D");
  NVTE_CHECK(t.logical_shape.data[0] > 0 && t.logical_shape.data[1] > 0, "Grouped tensor ", name,
             " logical_shape must have positive dimensions");
  const size_t expected = t.logical_shape.data[0] * t.logical_shape.data[1];
  if (t.has_data()) {
    NVTE_CHECK(t.data.shape.ndim == 1, "Grouped tensor ", name, " data must be 1D");
    NVTE_CHECK(t.data.numel() == expected, "Grouped tensor ", name, " data size (", t.data.numel(),
               ") must match logical_shape size (", expected, ")");
  }
  if (t.has_columnwise_data()) {
    NVTE_CHECK(t.columnwise_data.shape.ndim == 1, "Grouped tensor ", name, " columnwise_data must be 1D");
    NVTE_CHECK(t.columnwise_data.numel() == expected, "Grouped tensor ", name, " columnwise_data size (",
               t.columnwise_data.numel(), ") must match logical_shape size (", expected, ")");
  }
}

static void check_group_scale_inv(const GroupedTensor &t, const char *name, bool is_output) {
  const char *kind = is_output ? "output" : "input";
  auto check_scales = [&](DType want) {
    if (t.has_data()) {
      NVTE_CHECK(t.scale_inv.present(), kind, " ", name, " rowwise scale_inv must be allocated");
      NVTE_CHECK(t.scale_inv.dtype == want, kind, " ", name, " rowwise scale_inv has invalid dtype (expected ",
                 dtype_name(want), ", got ", dtype_name(t.scale_inv.dtype), ")");
    }
    if (t.has_columnwise_data()) {
      NVTE_CHECK(t.columnwise_scale_inv.present(), kind, " ", name, " columnwise scale_inv must be allocated");
      NVTE_CHECK(t.columnwise_scale_inv.dtype == want, kind, " ", name,
                 " columnwise scale_inv has invalid dtype (expected ", dtype_name(want), ", got ",
                 dtype_name(t.columnwise_scale_inv.dtype), ")");
    }
  };
  if (is_fp8(t.dtype()) && is_tensor_scaling(t.scaling_mode)) {
    check_scales(kNVTEFloat32);
  } else if (is_mxfp8(t.scaling_mode)) {
    check_scales(kNVTEFloat8E8M0);
  } else if (is_tensor_scaling(t.scaling_mode)) {
    NVTE_CHECK(!t.scale_inv.present(), "Scale_inv not supported for non-quantized ", kind, " ", name);
    NVTE_CHECK(!t.columnwise_scale_inv.present(), "Scale_inv not supported for non-quantized ", kind, " ", name);
  }
}

void check_grouped_input(const GroupedTensor &t, const char *name) {
  NVTE_CHECK(t.has_data() || t.has_columnwise_data(), "Input grouped tensor ", name, " not allocated");
  check_group_scale_inv(t, name, false);
  check_group_shape_arrays(t, name);
}

void check_grouped_output(const GroupedTensor &t, const char *name, bool allow_empty) {
  if (!allow_empty) {
    NVTE_CHECK(t.has_data() || t.has_columnwise_data(), "Output grouped tensor ", name, " not allocated");
  }
  if (t.has_data() || t.has_columnwise_data()) {
    if (is_fp8(t.dtype()) && t.scaling_mode == NVTE_DELAYED_TENSOR_SCALING) {
      NVTE_CHECK(t.amax.present(), "Output ", name, " amax must be allocated");
      NVTE_CHECK(t.amax.dtype == kNVTEFloat32, "Output ", name, " amax must be Float32");
    }
    check_group_scale_inv(t, name, true);
  }
  check_group_shape_arrays(t, name);
}

}  // namespace te

using namespace te;

extern "C" {

NVTEGroupedTensor nvte_create_grouped_tensor(NVTEScalingMode scaling_mode, size_t num_tensors,
                                             NVTEShape logical_shape) {
  return api_guard("nvte_create_grouped_tensor", [&]() -> NVTEGroupedTensor {
    NVTE_CHECK(num_tensors > 0, "Number of tensors must be greater than 0");
    NVTE_CHECK(logical_shape.ndim == 2, "Logical shape must be 2D");
    return GroupedArena::get().create(scaling_mode, num_tensors, logical_shape);
  });
}

void nvte_destroy_grouped_tensor(NVTEGroupedTensor tensor) {
  api_guard("nvte_destroy_grouped_tensor", [&] { GroupedArena::get().destroy(tensor); });
}

void nvte_set_grouped_tensor_param(NVTEGroupedTensor tensor, NVTEGroupedTensorParam param, const void *buf,
                                   size_t size_in_bytes) {
  api_guard("nvte_set_grouped_tensor_param", [&] {
    NVTE_CHECK(param < kNVTENumGroupedTensorParams, "Invalid NVTEGroupedTensorParam (got ",
               static_cast<int>(param), ")");
    NVTE_CHECK(tensor != nullptr, "Grouped tensor pointer can't be NULL.");
    GroupedTensor &t = *lookup_grouped_checked(tensor);
    const size_t need = grouped_param_size(param);
    NVTE_CHECK(size_in_bytes >= need, "Buffer is too small for grouped tensor parameter (parameter ",
               static_cast<int>(param), " needs ", need, " bytes, but buffer has ", size_in_bytes, " bytes)");
    NVTE_CHECK(buf != nullptr, "Invalid buffer (got NULL)");
    if (param == kNVTEGroupedWithGEMMSwizzledScales) {
      t.with_gemm_swizzled_scales = *static_cast<const uint8_t *>(buf) != 0;
    } else {
      *grouped_buf(t, param) = Buf(*static_cast<const NVTEBasicTensor *>(buf));
    }
  });
}

void nvte_get_grouped_tensor_param(const NVTEGroupedTensor tensor, NVTEGroupedTensorParam param, void *buf,
                                   size_t size_in_bytes, size_t *size_written) {
  api_guard("nvte_get_grouped_tensor_param", [&] {
    NVTE_CHECK(param < kNVTENumGroupedTensorParams, "Invalid NVTEGroupedTensorParam (got ",
               static_cast<int>(param), ")");
    const size_t need = grouped_param_size(param);
    if (size_written != nullptr) *size_written = need;
    if (buf == nullptr) return;
    NVTE_CHECK(size_in_bytes >= need, "Buffer is too small for grouped tensor parameter (parameter ",
               static_cast<int>(param), " needs ", need, " bytes, but buffer has ", size_in_bytes, " bytes)");
    GroupedTensor dummy;  // an invalid handle reads as an empty grouped tensor
    dummy.first_dims.dtype = dummy.last_dims.dtype = dummy.tensor_offsets.dtype = kNVTEInt64;
    GroupedTensor *t = lookup_grouped(tensor);
    if (t == nullptr) t = &dummy;
    if (param == kNVTEGroupedWithGEMMSwizzledScales) {
      *static_cast<uint8_t *>(buf) = t->with_gemm_swizzled_scales ? 1 : 0;
    } else {
      *static_cast<NVTEBasicTensor *>(buf) = grouped_buf(*t, param)->as_basic();
    }
  });
}

void nvte_amd_set_grouped_columnwise_transposed(NVTEGroupedTensor tensor, int transposed) {
  api_guard("nvte_amd_set_grouped_columnwise_transposed", [&] {
    GroupedTensor &t = *lookup_grouped_checked(tensor);
    NVTE_CHECK(is_mxfp8(t.scaling_mode) || !transposed,
               "columnwise_transposed is an MXFP8 extension (per-tensor scaling always stores the transposes)");
    t.columnwise_transposed = transposed != 0;
  });
}

int nvte_amd_get_grouped_columnwise_transposed(const NVTEGroupedTensor tensor) {
  return api_guard("nvte_amd_get_grouped_columnwise_transposed", [&]() -> int {
    if (tensor == nullptr) return 0;
    return lookup_grouped_checked(tensor)->columnwise_transposed ? 1 : 0;
  });
}

size_t nvte_grouped_tensor_num_tensors(const NVTEGroupedTensor tensor) {
  GroupedTensor *t = lookup_grouped(tensor);
  return t ? t->num_tensors : 0;
}

NVTEDType nvte_grouped_tensor_type(const NVTEGroupedTensor tensor) {
  GroupedTensor *t = lookup_grouped(tensor);
  return t ? t->dtype() : kNVTEFloat32;
}

NVTEScalingMode nvte_grouped_tensor_scaling_mode(const NVTEGroupedTensor tensor) {
  return api_guard("nvte_grouped_tensor_scaling_mode", [&]() -> NVTEScalingMode {
    if (tensor == nullptr) return NVTE_DELAYED_TENSOR_SCALING;
    return lookup_grouped_checked(tensor)->scaling_mode;
  });
}

NVTEShape nvte_get_grouped_tensor_logical_shape(const NVTEGroupedTensor tensor) {
  return api_guard("nvte_get_grouped_tensor_logical_shape", [&]() -> NVTEShape {
    if (tensor == nullptr) return nvte_make_shape(nullptr, 1);
    return lookup_grouped_checked(tensor)->logical_shape;
  });
}

// ---------------------------------------------------------------- grouped matmul config
NVTEGroupedMatmulConfig nvte_create_grouped_matmul_config(void) { return new GroupedMatmulConfig; }

void nvte_destroy_grouped_matmul_config(NVTEGroupedMatmulConfig config) {
  delete static_cast<GroupedMatmulConfig *>(config);
}

static size_t gm_attr_size(NVTEGroupedMatmulConfigAttribute a) {
  switch (a) {
    case kNVTEGroupedMatmulConfigAvgM:
    case kNVTEGroupedMatmulConfigAvgN:
    case kNVTEGroupedMatmulConfigAvgK: return sizeof(int64_t);
    case kNVTEGroupedMatmulConfigSMCount: return sizeof(int);
    default: return sizeof(uint8_t);
  }
}

void nvte_get_grouped_matmul_config_attribute(NVTEGroupedMatmulConfig config, NVTEGroupedMatmulConfigAttribute attr,
                                              void *buf, size_t size_in_bytes, size_t *size_written) {
  api_guard("nvte_get_grouped_matmul_config_attribute", [&] {
    NVTE_CHECK(attr < kNVTEGroupedMatmulConfigNumAttributes, "Invalid NVTEGroupedMatmulConfigAttribute (got ",
               static_cast<int>(attr), ")");
    const size_t need = gm_attr_size(attr);
    if (size_written != nullptr) *size_written = need;
    if (buf == nullptr) return;
    NVTE_CHECK(size_in_bytes >= need, "Buffer is too small for grouped matmul config attribute (attribute ",
               static_cast<int>(attr), " needs ", need, " bytes, but buffer has ", size_in_bytes, " bytes)");
    NVTE_CHECK(config != nullptr, "Invalid NVTEGroupedMatmulConfig (got NULL)");
    const GroupedMatmulConfig &c = *static_cast<const GroupedMatmulConfig *>(config);
    switch (attr) {
      case kNVTEGroupedMatmulConfigAvgM: std::memcpy(buf, &c.avg_m, need); break;
      case kNVTEGroupedMatmulConfigAvgN: std::memcpy(buf, &c.avg_n, need); break;
      case kNVTEGroupedMatmulConfigAvgK: std::memcpy(buf, &c.avg_k, need); break;
      case kNVTEGroupedMatmulConfigSMCount: std::memcpy(buf, &c.sm_count, need); break;
      default: *static_cast<uint8_t *>(buf) = c.use_split_accumulator ? 1 : 0; break;
    }
  });
}

void nvte_set_grouped_matmul_config_attribute(NVTEGroupedMatmulConfig config, NVTEGroupedMatmulConfigAttribute attr,
                                              const void *buf, size_t size_in_bytes) {
  api_guard("nvte_set_grouped_matmul_config_attribute", [&] {
    NVTE_CHECK(attr < kNVTEGroupedMatmulConfigNumAttributes, "Invalid NVTEGroupedMatmulConfigAttribute (got ",
               static_cast<int>(attr), ")");
    const size_t need = gm_attr_size(attr);
    NVTE_CHECK(size_in_bytes >= need, "Buffer is too small for grouped matmul config attribute (attribute ",
               static_cast<int>(attr), " needs ", need, " bytes, but buffer has ", size_in_bytes, " bytes)");
    NVTE_CHECK(buf != nullptr, "Invalid buffer (got NULL)");
    NVTE_CHECK(config != nullptr, "Invalid NVTEGroupedMatmulConfig (got NULL)");
    GroupedMatmulConfig &c = *static_cast<GroupedMatmulConfig *>(config);
    switch (attr) {
      case kNVTEGroupedMatmulConfigAvgM: std::memcpy(&c.avg_m, buf, need); c.has_avg_m = true; break;
      case kNVTEGroupedMatmulConfigAvgN: std::memcpy(&c.avg_n, buf, need); c.has_avg_n = true; break;
      case kNVTEGroupedMatmulConfigAvgK: std::memcpy(&c.avg_k, buf, need); c.has_avg_k = true; break;
      case kNVTEGroupedMatmulConfigSMCount: std::memcpy(&c.sm_count, buf, need); break;
      default: c.use_split_accumulator = *static_cast<const uint8_t *>(buf) != 0; break;
    }
  });
}

}  // extern "C"
