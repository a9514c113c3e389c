// Activations multiplied by a per-row scale (MoE router probabilities), high-precision in and out:
//   nvte_scaled_swiglu / nvte_scaled_clamped_swiglu / nvte_scaled_srelu and their backward forms.
//
// Replaces: ref common/activation/scaled_activation.cu:56-278 (kernels), :284-365 (checks), :375-612
//           (launchers); common/activation/scaled_swiglu.cu, scaled_srelu.cu (C entry points);
//           declarations include/transformer_engine/activation.h:372-406,513-552,590-615.
//
//   forward : out[r][c] = act(a[r][c]) (* gate[r][c]) * scales[r]                  one cast at the store
//   backward: dL/da = dy * scales[r] * act'(a) * gate,  dL/dgate = dy * scales[r] * act(a) (0 where the
//             clamp of the gate was active),  dL/dscales[r] = sum_c dy[r][c] * act(a) * gate  (optional)
// Gated rows are [act[0:H] | gate[0:H]] or, with glu_interleave_size = S (a multiple of 32 dividing H),
// [act[0:S] | gate[0:S] | act[S:2S] | gate[S:2S] | ...]; nothing is de-interleaved in memory.
//
// MI355X shape: HBM-bound streams (bf16 forward 6 B per output element, backward 10 B).  One 256-thread
// workgroup per row, persistent over rows (the row scale is read once, the row sum for dL/dscales stays inside
// the workgroup: DPP wave sums + 4 floats of LDS behind a bare barrier), two 8-element vectors per thread in
// flight.  Vector path: all payload tensors of one dtype, 16-byte aligned, segment
// length % 8 == 0; anything else (mixed dtypes, H = 100, odd pointers) runs the same kernels one element
// at a time through the run-time dtype accessors.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <type_traits>

#include "act_math.h"
#include "common.h"
#include "device_utils.h"
#include "transformer_engine/activation.h"

namespace te {
namespace {

enum : int { kScSwiglu = 0, kScClampedSwiglu = 1, kScSrelu = 2 };

struct ScaledArgs {
  const void *in;      // [rows, hidden * (gated ? 2 : 1)]
  const void *grad;    // backward: [rows, hidden]
  const void *scales;  // [rows]
  void *out;           // forward: [rows, hidden]; backward: the shape of `in`
  void *dscales;       // backward, optional: [rows]
  size_t rows, hidden;
  size_t segment;      // gated: interleave block (== hidden when not interleaved)
  int io_dtype;        // dtype of in / grad / out on the vector path
  int in_dtype, grad_dtype, out_dtype, scale_dtype, dscale_dtype;
  float limit, alpha, offset;
};

// act value, gate value as used, and their derivatives' factors (ref scaled_activation.cu:56-98)
template <int ACT>
__device__ __forceinline__ float fwd_value(float a, float g, const ScaledArgs &p) {
  if constexpr (ACT == kScSwiglu) {
    return a * sigmoid_f(a) * g;
  } else if constexpr (ACT == kScClampedSwiglu) {
    const float gate = fminf(fmaxf(-p.limit, g), p.limit) + p.offset;
    return clamped_act(a, p.limit, p.alpha) * gate;
  } else {
    return a > 0.f ? a * a : 0.f;
  }
}
// dact = d(unscaled)/da, dgate = d(unscaled)/dgate, unscaled = the forward value before the row scale
template <int ACT>
__device__ __forceinline__ void bwd_values(float a, float g, const ScaledArgs &p, float &dact, float &dgate,
                                           float &unscaled) {
  if constexpr (ACT == kScSwiglu) {
    const float s = sigmoid_f(a);
    const float act = a * s;
    unscaled = act * g;
    dact = (s + a * s * (1.f - s)) * g;
    dgate = act;
  } else if constexpr (ACT == kScClampedSwiglu) {
    const bool gate_open = g <= p.limit && g >= -p.limit;
    const float gate = fminf(fmaxf(-p.limit, g), p.limit) + p.offset;
    const float c = fminf(a, p.limit);
    const float s = sigmoid_f(p.alpha * c);
    const float act = c * s;
    const float d = a <= p.limit ? s + p.alpha * c * s * (1.f - s) : 0.f;
    unscaled = act * gate;
    dact = d * gate;
    dgate = gate_open ? act : 0.f;
  } else {
    unscaled = a > 0.f ? a * a : 0.f;
    dact = fmaxf(2.f * a, 0.f);
    dgate = 0.f;
  }
}

template <typename T>
__device__ __forceinline__ void store8_as(T *p, const float (&v)[8]) {
  if constexpr (sizeof(T) == 4) {
    f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4 *>(p) = lo;
    *reinterpret_cast<f32x4 *>(p + 4) = hi;
  } else {
    T o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) narrow(v[k], o[k]);
    *reinterpret_cast<u32x4 *>(p) = *reinterpret_cast<const u32x4 *>(o);
  }
}

// where the activation / gate halves of output column c of a row start (element offsets inside the row);
// 32-bit arithmetic: a row is shorter than 2^31 elements (checked by the launcher)
__device__ __forceinline__ void glu_offsets(unsigned c, unsigned hidden, unsigned segment, unsigned &act_off,
                                            unsigned &gate_off) {
  if (segment == hidden) {   // not interleaved (kernel argument: uniform)
    act_off = c;
    gate_off = hidden + c;
  } else {
    const unsigned seg = c / segment;
    act_off = c + seg * segment;
    gate_off = act_off + segment;
  }
}

// Both kernels: one 256-thread workgroup per row, persistent over rows; a thread handles the 8-element
// vectors t, t + 256, ... of the row - the forward TWO at a time (the loads of both are issued before either
// is used: the compiler cannot move a load above the previous vector's store by itself).
// VEC: 8 elements per step (IT = the one payload dtype); else one element, any dtypes.

// ------------------------------------------------------------------ forward
template <int ACT, bool VEC, typename IT>
__global__ __launch_bounds__(256) void scaled_fwd_kernel(const ScaledArgs p) {
  constexpr bool GATED = ACT != kScSrelu;
  constexpr unsigned W = VEC ? 8 : 1;
  const unsigned hidden = static_cast<unsigned>(p.hidden), segment = static_cast<unsigned>(p.segment);
  const unsigned per_row = hidden / W;
  const size_t in_cols = GATED ? p.hidden * 2 : p.hidden;
  for (size_t row = blockIdx.x; row < p.rows; row += gridDim.x) {
    const float scale = load1(p.scales, row, p.scale_dtype);
    if constexpr (VEC) {
      const IT *in = static_cast<const IT *>(p.in) + row * in_cols;
      IT *out = static_cast<IT *>(p.out) + row * p.hidden;
      for (unsigned v = threadIdx.x; v < per_row; v += 512) {
        const bool two = v + 256 < per_row;
        const unsigned c0 = v * 8, c1 = two ? (v + 256) * 8 : c0;
        unsigned a0 = c0, g0 = 0, a1 = c1, g1 = 0;
        if constexpr (GATED) {
          glu_offsets(c0, hidden, segment, a0, g0);
          glu_offsets(c1, hidden, segment, a1, g1);
        }
        const Raw8<IT> ra0 = Raw8<IT>::load(in + a0), ra1 = Raw8<IT>::load(in + a1);
        Raw8<IT> rg0 = ra0, rg1 = ra1;
        if constexpr (GATED) {
          rg0 = Raw8<IT>::load(in + g0);
          rg1 = Raw8<IT>::load(in + g1);
        }
        float a[8], g[8], o[8];
        ra0.widen8(a);
        rg0.widen8(g);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fwd_value<ACT>(a[k], g[k], p) * scale;
        store8_as<IT>(out + c0, o);
        if (two) {
          ra1.widen8(a);
          rg1.widen8(g);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = fwd_value<ACT>(a[k], g[k], p) * scale;
          store8_as<IT>(out + c1, o);
        }
      }
    } else {
      for (unsigned c = threadIdx.x; c < hidden; c += 256) {
        unsigned a_off = c, g_off = 0;
        if constexpr (GATED) glu_offsets(c, hidden, segment, a_off, g_off);
        const float a = load1(p.in, row * in_cols + a_off, p.in_dtype);
        const float g = GATED ? load1(p.in, row * in_cols + g_off, p.in_dtype) : 0.f;
        store1(p.out, row * p.hidden + c, p.out_dtype, fwd_value<ACT>(a, g, p) * scale);
      }
    }
  }
}

// ------------------------------------------------------------------ backward
template <int ACT, bool VEC, typename IT>
__global__ __launch_bounds__(256) void scaled_bwd_kernel(const ScaledArgs p) {
  constexpr bool GATED = ACT != kScSrelu;
  constexpr unsigned W = VEC ? 8 : 1;
  __shared__ float red[2][4];
  unsigned phase = 0;
  const unsigned hidden = static_cast<unsigned>(p.hidden), segment = static_cast<unsigned>(p.segment);
  const unsigned per_row = hidden / W;
  const size_t in_cols = GATED ? p.hidden * 2 : p.hidden;
  for (size_t row = blockIdx.x; row < p.rows; row += gridDim.x) {
    const float scale = load1(p.scales, row, p.scale_dtype);
    float sgrad = 0.f;
    if constexpr (VEC) {
      const IT *in = static_cast<const IT *>(p.in) + row * in_cols;
      const IT *gr = static_cast<const IT *>(p.grad) + row * p.hidden;
      IT *out = static_cast<IT *>(p.out) + row * in_cols;
      auto one = [&](const Raw8<IT> &ra, const Raw8<IT> &rg, const Raw8<IT> &rdy, unsigned a_off, unsigned g_off) {
        float a[8], g[8], dy[8], da[8], dg[8];
        ra.widen8(a);
        rg.widen8(g);
        rdy.widen8(dy);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float dact, dgate, unscaled;
          bwd_values<ACT>(a[k], g[k], p, dact, dgate, unscaled);
          sgrad += dy[k] * unscaled;
          const float sg = dy[k] * scale;
          da[k] = sg * dact;
          dg[k] = sg * dgate;
        }
        store8_as<IT>(out + a_off, da);
        if constexpr (GATED) store8_as<IT>(out + g_off, dg);
      };
      // one vector per step here: two in flight per thread measured slower for the backward (484 vs 463 us
      // for dswiglu on [16384, 2 x 14336] bf16; the forward gains 6 % from it)
      for (unsigned v = threadIdx.x; v < per_row; v += 256) {
        const unsigned c0 = v * 8;
        unsigned a0 = c0, g0 = 0;
        if constexpr (GATED) glu_offsets(c0, hidden, segment, a0, g0);
        const Raw8<IT> ra0 = Raw8<IT>::load(in + a0);
        const Raw8<IT> rd0 = Raw8<IT>::load(gr + c0);
        Raw8<IT> rg0 = ra0;
        if constexpr (GATED) rg0 = Raw8<IT>::load(in + g0);
        one(ra0, rg0, rd0, a0, g0);
      }
    } else {
      for (unsigned c = threadIdx.x; c < hidden; c += 256) {
        unsigned a_off = c, g_off = 0;
        if constexpr (GATED) glu_offsets(c, hidden, segment, a_off, g_off);
        const float a = load1(p.in, row * in_cols + a_off, p.in_dtype);
        const float g = GATED ? load1(p.in, row * in_cols + g_off, p.in_dtype) : 0.f;
        const float dy = load1(p.grad, row * p.hidden + c, p.grad_dtype);
        float dact, dgate, unscaled;
        bwd_values<ACT>(a, g, p, dact, dgate, unscaled);
        sgrad += dy * unscaled;
        const float sg = dy * scale;
        store1(p.out, row * in_cols + a_off, p.out_dtype, sg * dact);
        if constexpr (GATED) store1(p.out, row * in_cols + g_off, p.out_dtype, sg * dgate);
      }
    }
    if (p.dscales != nullptr) {   // kernel argument: the whole workgroup takes the same way
      sgrad = wave_sum_to_lane63(sgrad);
      if ((threadIdx.x & 63u) == 63u) red[phase][threadIdx.x >> 6] = sgrad;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (threadIdx.x == 0)
        store1(p.dscales, row, p.dscale_dtype, red[phase][0] + red[phase][1] + red[phase][2] + red[phase][3]);
      phase ^= 1u;
    }
  }
}

bool aligned16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

template <int ACT, bool BWD>
void launch(const ScaledArgs &a, bool vec, hipStream_t stream) {
  const unsigned cus = static_cast<unsigned>(device_info().num_cus);
  unsigned grid;
  grid = static_cast<unsigned>(std::min<size_t>(a.rows, static_cast<size_t>(cus) * 8));
#define TE_SC_LAUNCH(V, T)                                                                            \
  do {                                                                                                \
    if constexpr (BWD) hipLaunchKernelGGL((scaled_bwd_kernel<ACT, V, T>), dim3(grid), dim3(256), 0, stream, a); \
    else hipLaunchKernelGGL((scaled_fwd_kernel<ACT, V, T>), dim3(grid), dim3(256), 0, stream, a);    \
  } while (0)
  if (!vec) {
    TE_SC_LAUNCH(false, float);
  } else if (a.io_dtype == kNVTEBFloat16) {
    TE_SC_LAUNCH(true, bf16_t);
  } else if (a.io_dtype == kNVTEFloat16) {
    TE_SC_LAUNCH(true, f16_t);
  } else {
    TE_SC_LAUNCH(true, float);
  }
#undef TE_SC_LAUNCH
  NVTE_CHECK_HIP(hipGetLastError());
}

void check_hp(const Tensor &t, const char *api, const char *what) {
  NVTE_CHECK(is_high_precision(t.data.dtype), api, ": ", what, " must be FP32/BF16/FP16, got ", dtype_name(t.data.dtype));
  NVTE_REQUIRE_DATA(t.data, what);
}

// ref scaled_activation.cu:284-365 (the four check_* functions, same messages)
void run(const char *api, int act, const Tensor *grad, const Tensor &input, const Tensor &scales, Tensor &out,
         Tensor *dscales, float limit, float alpha, float offset, int64_t interleave, hipStream_t stream) {
  const bool gated = act != kScSrelu, bwd = grad != nullptr;
  const auto in_dims = input.flat2d(), out_dims = out.flat2d();
  size_t rows, hidden;
  if (!bwd) {
    if (gated) {
      NVTE_CHECK(in_dims[0] == out_dims[0], api, ": input/output row mismatch.");
      NVTE_CHECK(in_dims[1] == out_dims[1] * 2, api, ": gated input last dimension must be twice output last dimension.");
    } else {
      NVTE_CHECK(in_dims[0] == out_dims[0] && in_dims[1] == out_dims[1], api, ": input/output shapes must match.");
    }
    rows = in_dims[0];
    hidden = out_dims[1];
  } else {
    const auto g_dims = grad->flat2d();
    NVTE_CHECK(g_dims[0] == in_dims[0] && in_dims[0] == out_dims[0], api, ": input/grad row mismatch.");
    if (gated) {
      NVTE_CHECK(in_dims[1] == g_dims[1] * 2 && out_dims[1] == in_dims[1], api, ": gated backward dimensions are inconsistent.");
    } else {
      NVTE_CHECK(g_dims[1] == in_dims[1] && in_dims[1] == out_dims[1], api, ": unary backward dimensions are inconsistent.");
    }
    rows = in_dims[0];
    hidden = g_dims[1];
  }
  if (gated) {
    NVTE_CHECK(interleave >= 0, api, ": glu_interleave_size must be non-negative.");
    if (interleave > 0) {
      NVTE_CHECK(interleave % 32 == 0, api, ": nonzero glu_interleave_size must be a multiple of 32.");
      NVTE_CHECK(hidden % static_cast<size_t>(interleave) == 0, api,
                 bwd ? ": grad last dimension must be divisible by glu_interleave_size."
                     : ": output last dimension must be divisible by glu_interleave_size.");
    }
  }
  NVTE_CHECK(scales.numel() == in_dims[0], api, ": act_scales must have one value per row.");
  if (dscales != nullptr) NVTE_CHECK(dscales->numel() == in_dims[0], api, ": grad_act_scales must have one value per row.");
  if (rows == 0 || hidden == 0) return;
  check_hp(input, api, "input");
  check_hp(scales, api, "act_scales");
  check_hp(out, api, bwd ? "grad_input" : "output");
  if (bwd) check_hp(*grad, api, "grad");
  if (dscales != nullptr) check_hp(*dscales, api, "grad_act_scales");
  NVTE_CHECK(hidden < (1ull << 30), api, ": rows of more than 2^30 elements are not supported");

  ScaledArgs a;
  std::memset(&a, 0, sizeof(a));
  a.in = input.data.ptr;
  a.grad = bwd ? grad->data.ptr : nullptr;
  a.scales = scales.data.ptr;
  a.out = out.data.ptr;
  a.dscales = dscales ? dscales->data.ptr : nullptr;
  a.rows = rows;
  a.hidden = hidden;
  a.segment = (gated && interleave > 0) ? static_cast<size_t>(interleave) : hidden;
  a.in_dtype = input.data.dtype;
  a.grad_dtype = bwd ? grad->data.dtype : input.data.dtype;
  a.out_dtype = out.data.dtype;
  a.scale_dtype = scales.data.dtype;
  a.dscale_dtype = dscales ? dscales->data.dtype : kNVTEFloat32;
  a.io_dtype = a.in_dtype;
  a.limit = limit;
  a.alpha = alpha;
  a.offset = offset;
  // vector path: one payload dtype, every 8-element vector inside one segment and 16-byte aligned
  // (row starts: hidden % 8 == 0 covers rows of `hidden` and `2 hidden` elements of 2 or 4 bytes)
  const bool vec = a.in_dtype == a.out_dtype && a.in_dtype == a.grad_dtype && a.segment % 8 == 0 && hidden % 8 == 0 &&
                   aligned16(a.in) && aligned16(a.out) && (!bwd || aligned16(a.grad));
  switch (act) {
    case kScSwiglu: bwd ? launch<kScSwiglu, true>(a, vec, stream) : launch<kScSwiglu, false>(a, vec, stream); break;
    case kScClampedSwiglu:
      bwd ? launch<kScClampedSwiglu, true>(a, vec, stream) : launch<kScClampedSwiglu, false>(a, vec, stream);
      break;
    default: bwd ? launch<kScSrelu, true>(a, vec, stream) : launch<kScSrelu, false>(a, vec, stream); break;
  }
}

}  // namespace
}  // namespace te

using namespace te;

extern "C" {

void nvte_scaled_swiglu(const NVTETensor input, const NVTETensor act_scales, NVTETensor output,
                        int64_t glu_interleave_size, hipStream_t stream) {
  api_guard("nvte_scaled_swiglu", [&] {
    run("nvte_scaled_swiglu", kScSwiglu, nullptr, *lookup_checked(input), *lookup_checked(act_scales),
        *lookup_checked(output), nullptr, 0.f, 0.f, 0.f, glu_interleave_size, stream);
  });
}

void nvte_scaled_clamped_swiglu(const NVTETensor input, const NVTETensor act_scales, NVTETensor output, float limit,
                                float alpha, float glu_linear_offset, int64_t glu_interleave_size,
                                hipStream_t stream) {
  api_guard("nvte_scaled_clamped_swiglu", [&] {
    run("nvte_scaled_clamped_swiglu", kScClampedSwiglu, nullptr, *lookup_checked(input), *lookup_checked(act_scales),
        *lookup_checked(output), nullptr, limit, alpha, glu_linear_offset, glu_interleave_size, stream);
  });
}

void nvte_scaled_dswiglu(const NVTETensor grad, const NVTETensor input, const NVTETensor act_scales,
                         NVTETensor grad_input, NVTETensor grad_act_scales, int64_t glu_interleave_size,
                         hipStream_t stream) {
  api_guard("nvte_scaled_dswiglu", [&] {
    run("nvte_scaled_dswiglu", kScSwiglu, lookup_checked(grad), *lookup_checked(input), *lookup_checked(act_scales),
        *lookup_checked(grad_input), grad_act_scales == nullptr ? nullptr : lookup_checked(grad_act_scales), 0.f, 0.f,
        0.f, glu_interleave_size, stream);
  });
}

void nvte_scaled_clamped_dswiglu(const NVTETensor grad, const NVTETensor input, const NVTETensor act_scales,
                                 NVTETensor grad_input, NVTETensor grad_act_scales, float limit, float alpha,
                                 float glu_linear_offset, int64_t glu_interleave_size, hipStream_t stream) {
  api_guard("nvte_scaled_clamped_dswiglu", [&] {
    run("nvte_scaled_clamped_dswiglu", kScClampedSwiglu, lookup_checked(grad), *lookup_checked(input),
        *lookup_checked(act_scales), *lookup_checked(grad_input),
        grad_act_scales == nullptr ? nullptr : lookup_checked(grad_act_scales), limit, alpha, glu_linear_offset,
        glu_interleave_size, stream);
  });
}

void nvte_scaled_srelu(const NVTETensor input, const NVTETensor act_scales, NVTETensor output, hipStream_t stream) {
  api_guard("nvte_scaled_srelu", [&] {
    run("nvte_scaled_srelu", kScSrelu, nullptr, *lookup_checked(input), *lookup_checked(act_scales),
        *lookup_checked(output), nullptr, 0.f, 0.f, 0.f, 0, stream);
  });
}

void nvte_scaled_dsrelu(const NVTETensor grad, const NVTETensor input, const NVTETensor act_scales,
                        NVTETensor grad_input, NVTETensor grad_act_scales, hipStream_t stream) {
  api_guard("nvte_scaled_dsrelu", [&] {
    run("nvte_scaled_dsrelu", kScSrelu, lookup_checked(grad), *lookup_checked(input), *lookup_checked(act_scales),
        *lookup_checked(grad_input), grad_act_scales == nullptr ? nullptr : lookup_checked(grad_act_scales), 0.f, 0.f,
        0.f, 0, stream);
  });
}

}  // extern "C"
