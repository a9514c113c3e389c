// Recipe step kernels: amax scan, scale-from-amax (current scaling) and the
// amax-history roll + scale update (delayed scaling).
//
// Replaces: ref common/recipe/current_scaling.cu:24-115,247-336,
//           common/recipe/recipe_common.cuh:14-49,
//           common/recipe/delayed_scaling.cu:84-154 (single), :162-253 (bulk), :259-420.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <string>

#include "common.h"
#include "device_utils.h"
#include "transformer_engine/amd_ext.h"
#include "transformer_engine/recipe.h"

namespace te {
namespace {

// ---------------------------------------------------------------- amax scan (HBM read-only, 2 B/elem)
__global__ void zero_amax_kernel(float *amax, const float *noop) {
  if (noop_set(noop)) return;
  *amax = 0.f;
}

// The reference takes the element max in the INPUT dtype for bf16/fp16
// (current_scaling.cu:50-58); every bf16/fp16 value is exact in fp32 and both maxima
// drop NaN operands, so the fp32 max of the widened values is the same number.
template <typename VecLoader, int ELEMS, typename IT>
__global__ __launch_bounds__(512) void amax_vec_kernel(const IT *__restrict__ in,
                                                       float *__restrict__ amax_out,
                                                       const size_t nvec,
                                                       const float *__restrict__ noop) {
  if (noop_set(noop)) return;
  __shared__ float red[8];
  float m0 = 0.f, m1 = 0.f;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  // two independent vectors in flight per iteration
  for (; i + stride < nvec; i += 2 * stride) {
    float a[ELEMS], b[ELEMS];
    VecLoader::run(in + i * ELEMS, a);
    VecLoader::run(in + (i + stride) * ELEMS, b);
#pragma unroll
    for (int k = 0; k < ELEMS; ++k) {
      m0 = fmaxf(m0, __builtin_fabsf(a[k]));
      m1 = fmaxf(m1, __builtin_fabsf(b[k]));
    }
  }
  if (i < nvec) {
    float a[ELEMS];
    VecLoader::run(in + i * ELEMS, a);
#pragma unroll
    for (int k = 0; k < ELEMS; ++k) m0 = fmaxf(m0, __builtin_fabsf(a[k]));
  }
  float m = block_max<8>(fmaxf(m0, m1), red);
  if (threadIdx.x == 0) atomic_max_nonneg(amax_out, m);
}

struct LoadBf16x8 {
  static __device__ __forceinline__ void run(const bf16_t *p, float (&f)[8]) {
    const u32x4 v = *reinterpret_cast<const u32x4 *>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = __uint_as_float(v[k] << 16);
      f[2 * k + 1] = __uint_as_float(v[k] & 0xFFFF0000u);
    }
  }
};
struct LoadF16x8 {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ void run(const f16_t *p, float (&f)[8]) {
    const h8 v = *reinterpret_cast<const h8 *>(p);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = static_cast<float>(v[k]);
  }
};
struct LoadF32x4 {
  static __device__ __forceinline__ void run(const float *p, float (&f)[4]) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k) f[k] = v[k];
  }
};

template <typename IT>
__global__ __launch_bounds__(512) void amax_scalar_kernel(const IT *__restrict__ in,
                                                          float *__restrict__ amax_out,
                                                          const size_t n,
                                                          const float *__restrict__ noop) {
  if (noop_set(noop)) return;
  __shared__ float red[8];
  float m = 0.f;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    m = fmaxf(m, __builtin_fabsf(widen(in[i])));
  m = block_max<8>(m, red);
  if (threadIdx.x == 0) atomic_max_nonneg(amax_out, m);
}

// ---------------------------------------------------------------- scale from amax (device_utils.h)
__global__ void scale_from_amax_kernel(const float *amax, float *scale, float max_fp8, bool pow2,
                                       float eps, const float *noop) {
  if (noop_set(noop)) return;
  *scale = scale_from_amax(*amax, max_fp8, pow2, eps, std::numeric_limits<float>::max());
}

// ---------------------------------------------------------------- delayed scaling
enum class AmaxAlgo : int { MostRecent = 0, Max = 1 };

// Roll one history column by -1, zero its head, return the reduction the algorithm asks
// for (valid in every thread).  `last_fn()` gives the value that moves to the tail, or 0 for "leave the
// column alone" (bulk update: no amax was recorded); it is only evaluated AFTER the column's loads have been
// issued, so the loads it depends on (the head of the column, the reduced amax) and the column itself are one
// round trip to memory, not two - these kernels are a chain of dependent latencies, nothing else.
template <int BS, typename LastFn>
__device__ __forceinline__ float roll_column(const float *hist_in, float *hist_out, size_t length,
                                             size_t stride, LastFn last_fn, bool skip_if_zero, AmaxAlgo algo,
                                             float *red /* BS/64 floats */, float &last_out) {
  float amax = 0.f;
  constexpr int kInRegs = 8;   // histories up to 8 * BS entries (the default length is 1024) roll through registers:
  if (length <= static_cast<size_t>(kInRegs) * BS) {   // every load in flight at once, ONE barrier, then the stores
    float a[kInRegs];
#pragma unroll
    for (int v = 0; v < kInRegs; ++v) {
      const size_t i = static_cast<size_t>(v) * BS + threadIdx.x;
      a[v] = (i + 1 < length) ? hist_in[(i + 1) * stride] : 0.f;
    }
    const float last = last_fn();
    last_out = last;
    if (skip_if_zero && last == 0.0f) return 0.f;   // uniform over the workgroup
#pragma unroll
    for (int v = 0; v < kInRegs; ++v) {
      const size_t i = static_cast<size_t>(v) * BS + threadIdx.x;
      if (i + 1 == length) a[v] = last;
      if (i < length) amax = fmaxf(amax, a[v]);
    }
    __syncthreads();  // the roll may be in place
#pragma unroll
    for (int v = 0; v < kInRegs; ++v) {
      const size_t i = static_cast<size_t>(v) * BS + threadIdx.x;
      if (i < length) hist_out[i * stride] = (i > 0) ? a[v] : 0.f;
    }
  } else {
    const float last = last_fn();
    last_out = last;
    if (skip_if_zero && last == 0.0f) return 0.f;
    for (size_t off = 0; off < length; off += BS) {
      const size_t i = off + threadIdx.x;
      float a = 0.f;
      if (i < length) {
        a = (i < length - 1) ? hist_in[(i + 1) * stride] : last;
        amax = fmaxf(amax, a);
      }
      __syncthreads();  // the roll may be in place
      if (i < length) hist_out[i * stride] = (i > 0) ? a : 0.f;
    }
  }
  if (algo == AmaxAlgo::MostRecent) return last_out;
  amax = wave_max(amax);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < BS / 64; ++w) r = fmaxf(r, red[w]);
  return r;
}

__device__ __forceinline__ float updated_scale(float amax, float old_scale, float scaled_max) {
  float s = (__builtin_isfinite(amax) && amax > 0.f) ? scaled_max / amax : old_scale;
  if (__builtin_isinf(s)) s = std::numeric_limits<float>::max();
  return s;
}

constexpr int kDsThreads = 256;

// grid.x = num_scales
__global__ __launch_bounds__(kDsThreads) void delayed_scaling_kernel(
    const float *hist_in, const float *scale_in, float *hist_out, float *scale_out, size_t length,
    size_t stride, AmaxAlgo algo, float scaled_max) {
  __shared__ float red[kDsThreads / 64];
  const size_t col = blockIdx.x;
  const float head = hist_in[col], old_scale = scale_in[col];
  float last;
  const float amax = roll_column<kDsThreads>(hist_in + col, hist_out + col, length, stride, [&] { return head; },
                                             false, algo, red, last);
  if (threadIdx.x == 0) scale_out[col] = updated_scale(amax, old_scale, scaled_max);
}

// One launch covers up to kBulk (history, scale) pairs; grid = (pairs, max num_scale).
constexpr int kBulk = 128;
struct BulkEntry {
  float *hist;
  float *scale;
  int num_scale;
  int buf_offset;  // first entry of this tensor in the reduced-amax buffer
  float scaled_max;  // fp8_max of THIS tensor's format * 2^-margin (forward and backward states differ)
};
struct BulkParams {
  BulkEntry e[kBulk];
};

__global__ __launch_bounds__(kDsThreads) void delayed_scaling_bulk_kernel(
    const float *reduced, BulkParams p, size_t length, AmaxAlgo algo) {
  __shared__ float red[kDsThreads / 64];
  const BulkEntry ent = p.e[blockIdx.x];
  const int col = blockIdx.y;
  if (col >= ent.num_scale) return;  // uniform per block
  float *hist = ent.hist + col;
  // three independent loads up front (head of the column, the reduced amax, the scale to fall back on)
  const float head = hist[0];
  const float r = reduced != nullptr ? reduced[ent.buf_offset + col] : 0.0f;
  const float old_scale = ent.scale[col];
  float last;
  // a column with no recorded amax (last == 0) is left alone, its reduction is 0 (-> the old scale stays)
  const float amax = roll_column<kDsThreads>(hist, hist, length, ent.num_scale,
                                             [&] { return r != 0.0f ? r : head; }, true, algo, red, last);
  if (threadIdx.x == 0) ent.scale[col] = updated_scale(amax, old_scale, ent.scaled_max);
}

AmaxAlgo parse_algo(const char *s) {
  NVTE_CHECK(s != nullptr, "amax_compute_algo is NULL");
  if (std::strcmp(s, "max") == 0) return AmaxAlgo::Max;
  if (std::strcmp(s, "most_recent") == 0) return AmaxAlgo::MostRecent;
  NVTE_ERROR("Unsupported amax compute algorithm (", s, ")");
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// `dtypes` == nullptr: every tensor has format `fp8_dtype` (the reference's call); else one format per tensor
void after_reduction_impl(const NVTETensor reduced_h, const NVTETensor *hists,
                          const NVTETensor *scales, size_t n, const char *algo_s, NVTEDType fp8_dtype,
                          float margin, hipStream_t stream, const NVTEDType *dtypes = nullptr) {
  const AmaxAlgo algo = parse_algo(algo_s);
  const float margin_mult = std::pow(2.f, -margin);
  const Tensor &reduced = *lookup_checked(reduced_h);
  const float *buf = static_cast<const float *>(reduced.data.ptr);
  size_t length = 0;
  if (n > 0) {
    Tensor *h0 = lookup(hists[0]);
    NVTE_CHECK(h0 != nullptr, "Invalid tensor.");
    NVTE_CHECK(h0->data.shape.ndim == 2, "Found ", h0->data.shape.ndim, " dims");
    length = h0->data.shape.data[0];
  }
  size_t done = 0;
  int offset = 0;
  while (done < n) {
    const size_t cnt = std::min<size_t>(kBulk, n - done);
    BulkParams p;
    std::memset(&p, 0, sizeof(p));
    int max_scales = 0;
    for (size_t k = 0; k < cnt; ++k) {
      Tensor *h = lookup(hists[done + k]);
      Tensor *s = lookup(scales[done + k]);
      NVTE_CHECK(h != nullptr && s != nullptr, "Invalid tensor.");
      NVTE_CHECK(h->data.dtype == kNVTEFloat32, "Found ", dtype_name(h->data.dtype), ".");
      NVTE_CHECK(h->data.shape.ndim == 2, "Found ", h->data.shape.ndim, " dims");
      const int ns = static_cast<int>(h->data.shape.data[1]);
      NVTE_CHECK(h->data.numel() == length * ns, "Expected ", length * ns, " elements, but found ",
                 h->data.numel(), ".");
      NVTE_CHECK(s->data.dtype == kNVTEFloat32, "Found ", dtype_name(s->data.dtype), ".");
      NVTE_CHECK(s->data.shape.ndim == 1, "Found ", s->data.shape.ndim, " dims");
      NVTE_CHECK(s->data.numel() == static_cast<size_t>(ns), "Expected ", ns, " elements, Found ",
                 s->data.numel(), ".");
      p.e[k] = BulkEntry{static_cast<float *>(h->data.ptr), static_cast<float *>(s->data.ptr), ns, offset,
                         fp8_max_of(dtypes ? dtypes[done + k] : fp8_dtype) * margin_mult};
      offset += ns;
      max_scales = std::max(max_scales, ns);
    }
    if (max_scales > 0 && length > 0) {
      hipLaunchKernelGGL(delayed_scaling_bulk_kernel, dim3(cnt, max_scales), dim3(kDsThreads), 0,
                         stream, buf, p, length, algo);
      NVTE_CHECK_HIP(hipGetLastError());
    }
    done += cnt;
  }
}

}  // namespace

void amax_launch(const Tensor &in, float *amax, const float *noop, hipStream_t stream) {
  hipLaunchKernelGGL(zero_amax_kernel, dim3(1), dim3(1), 0, stream, amax, noop);
  NVTE_CHECK_HIP(hipGetLastError());
  const size_t n = in.data.numel();
  if (n == 0) return;
  const DeviceInfo &di = device_info();
  const size_t max_grid = static_cast<size_t>(di.num_cus) * 8;
  const void *p = in.data.ptr;
  auto grid_for = [&](size_t items) {
    return dim3(static_cast<unsigned>(std::max<size_t>(1, std::min(ceil_div(items, 1024), max_grid))));
  };
  switch (in.data.dtype) {
    case kNVTEBFloat16:
      if (n % 8 == 0 && aligned16(p)) {
        hipLaunchKernelGGL((amax_vec_kernel<LoadBf16x8, 8, bf16_t>), grid_for(n / 8), dim3(512), 0,
                           stream, static_cast<const bf16_t *>(p), amax, n / 8, noop);
      } else {
        hipLaunchKernelGGL((amax_scalar_kernel<bf16_t>), grid_for(n), dim3(512), 0, stream,
                           static_cast<const bf16_t *>(p), amax, n, noop);
      }
      break;
    case kNVTEFloat16:
      if (n % 8 == 0 && aligned16(p)) {
        hipLaunchKernelGGL((amax_vec_kernel<LoadF16x8, 8, f16_t>), grid_for(n / 8), dim3(512), 0,
                           stream, static_cast<const f16_t *>(p), amax, n / 8, noop);
      } else {
        hipLaunchKernelGGL((amax_scalar_kernel<f16_t>), grid_for(n), dim3(512), 0, stream,
                           static_cast<const f16_t *>(p), amax, n, noop);
      }
      break;
    case kNVTEFloat32:
      if (n % 4 == 0 && aligned16(p)) {
        hipLaunchKernelGGL((amax_vec_kernel<LoadF32x4, 4, float>), grid_for(n / 4), dim3(512), 0,
                           stream, static_cast<const float *>(p), amax, n / 4, noop);
      } else {
        hipLaunchKernelGGL((amax_scalar_kernel<float>), grid_for(n), dim3(512), 0, stream,
                           static_cast<const float *>(p), amax, n, noop);
      }
      break;
    default:
      NVTE_ERROR("Unsupported input dtype ", dtype_name(in.data.dtype));
  }
  NVTE_CHECK_HIP(hipGetLastError());
}

void scale_from_amax_launch(const float *amax, float *scale, float max_fp8, bool pow2, float eps,
                            const float *noop, hipStream_t stream) {
  hipLaunchKernelGGL(scale_from_amax_kernel, dim3(1), dim3(1), 0, stream, amax, scale, max_fp8, pow2,
                     eps, noop);
  NVTE_CHECK_HIP(hipGetLastError());
}

}  // namespace te

using namespace te;

static void compute_amax_impl(const NVTETensor input_, NVTETensor output_, hipStream_t stream,
                              const NVTEQuantizationConfig config_) {
  NVTE_CHECK(input_ != nullptr, "Invalid input tensor (got NULL)");
  const Tensor &input = *lookup_checked(input_);
  NVTE_CHECK(input.scaling_mode == NVTE_DELAYED_TENSOR_SCALING,
             "Input tensor for amax computation must unquantized, but got scaling_mode=",
             scaling_mode_name(input.scaling_mode));
  NVTE_CHECK(!is_fp8(input.data.dtype),
             "Input tensor for amax computation must be unquantized, but got dtype=",
             dtype_name(input.data.dtype));
  NVTE_CHECK(input.data.ptr != nullptr, "Input tensor for amax computation has no data");
  check_input(input, "input_compute_amax");

  NVTE_CHECK(output_ != nullptr, "Invalid output tensor (got NULL)");
  Tensor &output = *lookup_checked(output_);
  NVTE_CHECK(output.scaling_mode == NVTE_DELAYED_TENSOR_SCALING,
             "Output tensor for amax computation must be FP8 tensor with per-tensor scaling, but got "
             "scaling_mode=",
             scaling_mode_name(output.scaling_mode));
  NVTE_CHECK(output.amax.numel() == 1,
             "Output tensor for amax computation has invalid amax tensor (expected 1 entry, got "
             "shape=",
             output.amax.shape, ")");
  NVTE_CHECK(output.amax.ptr != nullptr || output.columnwise_amax.ptr != nullptr,
             "Output tensor for amax computation has amax tensor without data");
  NVTE_CHECK(output.amax.dtype == kNVTEFloat32,
             "Output tensor for amax computation has invalid amax tensor (expected FP32, got dtype=",
             dtype_name(output.amax.dtype), ")");
  check_output(output, "output_compute_amax", true);

  const float *noop = noop_ptr_of(static_cast<const QuantConfig *>(config_));
  float *amax_ptr =
      static_cast<float *>(output.amax.ptr ? output.amax.ptr : output.columnwise_amax.ptr);
  amax_launch(input, amax_ptr, noop, stream);
}

extern "C" {

void nvte_compute_amax(const NVTETensor input, NVTETensor output, hipStream_t stream) {
  api_guard("nvte_compute_amax", [&] { compute_amax_impl(input, output, stream, nullptr); });
}

void nvte_compute_amax_with_config(const NVTETensor input, NVTETensor output,
                                   const NVTEQuantizationConfig config, hipStream_t stream) {
  api_guard("nvte_compute_amax_with_config",
            [&] { compute_amax_impl(input, output, stream, config); });
}

void nvte_compute_scale_from_amax(NVTETensor output_, const NVTEQuantizationConfig config_,
                                  hipStream_t stream) {
  api_guard("nvte_compute_scale_from_amax", [&] {
    NVTE_CHECK(output_ != nullptr, "Invalid output tensor (got NULL)");
    Tensor &output = *lookup_checked(output_);
    NVTE_CHECK(output.scaling_mode == NVTE_DELAYED_TENSOR_SCALING,
               "Tensor must be FP8 tensor with per-tensor scaling, but got scaling_mode=",
               scaling_mode_name(output.scaling_mode));
    NVTE_CHECK(is_fp8(output.data.dtype), "Tensor must be FP8, but got dtype=",
               dtype_name(output.data.dtype));
    NVTE_CHECK(output.amax.numel() == 1, "Tensor has invalid amax tensor (expected 1 entry, got shape=",
               output.amax.shape, ")");
    NVTE_CHECK(output.amax.ptr != nullptr, "Tensor has amax tensor without data");
    NVTE_CHECK(output.amax.dtype == kNVTEFloat32,
               "Tensor has invalid amax tensor (expected FP32, got dtype=",
               dtype_name(output.amax.dtype), ")");
    NVTE_CHECK(output.scale.numel() == 1,
               "Tensor has invalid scale tensor (expected 1 entry, got shape=", output.scale.shape, ")");
    NVTE_CHECK(output.scale.ptr != nullptr, "Tensor has scale tensor without data");
    NVTE_CHECK(output.scale.dtype == kNVTEFloat32,
               "Tensor has invalid scale tensor (expected FP32, got dtype=",
               dtype_name(output.scale.dtype), ")");
    NVTE_CHECK(config_ != nullptr, "Invalid config (got NULL)");
    const QuantConfig &cfg = *static_cast<const QuantConfig *>(config_);
    scale_from_amax_launch(static_cast<const float *>(output.amax.ptr),
                           static_cast<float *>(output.scale.ptr), fp8_max_of(output.data.dtype),
                           cfg.force_pow_2_scales, cfg.amax_epsilon, noop_ptr_of(&cfg), stream);
  });
}

void nvte_delayed_scaling_recipe_amax_and_scale_update(
    const NVTETensor amax_history_, const NVTETensor scale_, NVTETensor updated_amax_history_,
    NVTETensor updated_scale_, const char *amax_compute_algo, NVTEDType fp8_dtype, float margin,
    hipStream_t stream) {
  api_guard("nvte_delayed_scaling_recipe_amax_and_scale_update", [&] {
    const Tensor &hist = *lookup_checked(amax_history_);
    const Tensor &scale = *lookup_checked(scale_);
    Tensor &new_hist = *lookup_checked(updated_amax_history_);
    Tensor &new_scale = *lookup_checked(updated_scale_);
    NVTE_CHECK(hist.data.shape.ndim == 2, "Found ", hist.data.shape.ndim, " dims");
    const size_t length = hist.data.shape.data[0];
    const size_t num_scales = hist.data.shape.data[1];
    NVTE_CHECK(hist.data.dtype == kNVTEFloat32, "Found ", dtype_name(hist.data.dtype), ".");
    NVTE_CHECK(scale.numel() == num_scales, "Expected ", num_scales, " elements, but found ",
               scale.numel(), ".");
    NVTE_CHECK(scale.data.dtype == kNVTEFloat32, "Found ", dtype_name(scale.data.dtype), ".");
    NVTE_CHECK(new_hist.data.shape.ndim == 2, "Found ", new_hist.data.shape.ndim, " dims.");
    NVTE_CHECK(new_hist.data.shape.data[0] == length, "Expected ", length, ", but found ",
               new_hist.data.shape.data[0]);
    NVTE_CHECK(new_hist.data.shape.data[1] == num_scales, "Expected ", num_scales, ", but found ",
               new_hist.data.shape.data[1]);
    NVTE_CHECK(new_hist.data.dtype == kNVTEFloat32, "Got ", dtype_name(new_hist.data.dtype), ".");
    NVTE_CHECK(new_scale.numel() == num_scales, "Expected ", num_scales, " elements, but found ",
               new_scale.numel(), ".");
    NVTE_CHECK(new_scale.data.dtype == kNVTEFloat32, "Got ", dtype_name(new_scale.data.dtype), ".");
    const AmaxAlgo algo = parse_algo(amax_compute_algo);
    const float scaled_max = fp8_max_of(fp8_dtype) * std::pow(2.f, -margin);
    if (num_scales == 0 || length == 0) return;
    hipLaunchKernelGGL(delayed_scaling_kernel, dim3(num_scales), dim3(kDsThreads), 0, stream,
                       static_cast<const float *>(hist.data.ptr),
                       static_cast<const float *>(scale.data.ptr),
                       static_cast<float *>(new_hist.data.ptr),
                       static_cast<float *>(new_scale.data.ptr), length, num_scales, algo, scaled_max);
    NVTE_CHECK_HIP(hipGetLastError());
  });
}

void nvte_amd_delayed_scaling_after_reduction(const NVTETensor amax_reduction_buffer,
                                              const NVTETensor *amax_histories,
                                              const NVTETensor *scales, size_t num_tensors,
                                              const char *amax_compute_algo, NVTEDType fp8_dtype,
                                              float margin, hipStream_t stream) {
  api_guard("nvte_amd_delayed_scaling_after_reduction", [&] {
    after_reduction_impl(amax_reduction_buffer, amax_histories, scales, num_tensors,
                         amax_compute_algo, fp8_dtype, margin, stream);
  });
}

void nvte_amd_delayed_scaling_after_reduction_multi(const NVTETensor amax_reduction_buffer,
                                                    const NVTETensor *amax_histories,
                                                    const NVTETensor *scales,
                                                    const NVTEDType *fp8_dtypes, size_t num_tensors,
                                                    const char *amax_compute_algo, float margin,
                                                    hipStream_t stream) {
  api_guard("nvte_amd_delayed_scaling_after_reduction_multi", [&] {
    NVTE_CHECK(num_tensors == 0 || fp8_dtypes != nullptr, "fp8_dtypes must hold one entry per tensor");
    after_reduction_impl(amax_reduction_buffer, amax_histories, scales, num_tensors, amax_compute_algo,
                         kNVTEFloat8E4M3, margin, stream, fp8_dtypes);
  });
}

}  // extern "C"

// C++-mangled like the reference's (std::vector arguments).
void nvte_delayed_scaling_recipe_amax_and_scale_update_after_reduction(
    const NVTETensor amax_reduction_buffer, std::vector<NVTETensor> amax_histories,
    std::vector<NVTETensor> scales, const char *amax_compute_algo, NVTEDType fp8_dtype, float margin,
    hipStream_t stream) {
  api_guard("nvte_delayed_scaling_recipe_amax_and_scale_update_after_reduction", [&] {
    NVTE_CHECK(amax_histories.size() == scales.size(), "history / scale list length mismatch");
    after_reduction_impl(amax_reduction_buffer, amax_histories.data(), scales.data(),
                         amax_histories.size(), amax_compute_algo, fp8_dtype, margin, stream);
  });
}
