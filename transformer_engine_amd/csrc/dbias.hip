// Bias-gradient reduction that accompanies a quantize of dY:  dbias[c] = sum_r dY[r, c].
//
// Replaces: ref common/transpose/cast_transpose_fusion.cu:208 (reduce_dbias_kernel),
// :1260 (nvte_cast_transpose_dbias), common/cast/cast_dbias.cu:13 (nvte_quantize_dbias).
// Two-call protocol (ref transpose.h:48-49, quantize_fp8.cuh:411-415): a call whose
// workspace has no data pointer only reports the workspace shape/dtype.
//
// Round-1 structure: the cast itself is the ordinary quantize kernel; the column sums
// are fp32 partials over 128-row slabs ([ceil(R/128), C] workspace) followed by a slab
// reduce that rounds once to the input dtype.
#include <hip/hip_runtime.h>

#include "common.h"
#include "device_utils.h"

namespace te {
bool dbias_workspace_query(const Tensor &in, Tensor &workspace);
void dbias_finalize_launch(const float *ws, Tensor &dbias, size_t slabs, size_t cols, hipStream_t stream);
namespace {

constexpr int kSlab = 128;

template <typename IT>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const IT *__restrict__ in,
                                                             float *__restrict__ ws,
                                                             const size_t rows, const size_t cols) {
  const size_t c = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  const size_t r0 = static_cast<size_t>(blockIdx.y) * kSlab;
  if (c >= cols) return;
  const size_t r1 = r0 + kSlab < rows ? r0 + kSlab : rows;
  float acc = 0.f;
  for (size_t r = r0; r < r1; ++r) acc += widen(in[r * cols + c]);
  ws[static_cast<size_t>(blockIdx.y) * cols + c] = acc;
}

template <typename OT>
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ ws,
                                                           OT *__restrict__ dbias,
                                                           const size_t slabs, const size_t cols) {
  const size_t c = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (c >= cols) return;
  float acc = 0.f;
  for (size_t s = 0; s < slabs; ++s) acc += ws[s * cols + c];
  narrow(acc, dbias[c]);
}

// FP8 input (nvte_fp8_transpose_dbias): ws[slab][c] = sum over the slab's rows of float(in[r][c]) * scale_inv
// (per-element multiply, then add: the order of ref transpose/transpose_fusion.cu:23-35).  A thread owns
// 4 consecutive columns (one dword per row, 256 contiguous bytes per wave); cols % 4 != 0 or an unaligned
// pointer takes the byte-wise instantiation.
template <Fp8 F, bool VEC4>
__global__ __launch_bounds__(256) void colsum_partial_fp8_kernel(const uint8_t *__restrict__ in,
                                                                 const float *__restrict__ scale_inv,
                                                                 float *__restrict__ ws, const size_t rows,
                                                                 const size_t cols) {
  constexpr int W = VEC4 ? 4 : 1;
  const size_t c = (static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x) * W;
  const size_t r0 = static_cast<size_t>(blockIdx.y) * kSlab;
  if (c >= cols) return;
  const size_t r1 = r0 + kSlab < rows ? r0 + kSlab : rows;
  const float si = scale_inv != nullptr ? *scale_inv : 1.f;
  float acc[W];
#pragma unroll
  for (int x = 0; x < W; ++x) acc[x] = 0.f;
  for (size_t r = r0; r < r1; ++r) {
    if constexpr (VEC4) {
      const uint32_t v = *reinterpret_cast<const uint32_t *>(in + r * cols + c);
#pragma unroll
      for (int x = 0; x < 4; ++x) acc[x] += fp8_to_float<F>(v, x) * si;
    } else {
      acc[0] += fp8_to_float<F>(in[r * cols + c], 0) * si;
    }
  }
#pragma unroll
  for (int x = 0; x < W; ++x) ws[static_cast<size_t>(blockIdx.y) * cols + c + x] = acc[x];
}

}  // namespace

// nvte_fp8_transpose_dbias (ref transpose/transpose_fusion.cu:420-496): byte transpose of an FP8 [N, H]
// tensor + dbias[H] = sum over N of the dequantized input, in the dbias dtype.  The scale_inv is read
// from the OUTPUT tensor as the reference does (:469-470), else from the input, else 1.
void fp8_transpose_dbias_launch(const Tensor &in, Tensor &out, Tensor &dbias, Tensor &workspace,
                                hipStream_t stream) {
  NVTE_CHECK(in.data.shape.ndim == 2, "Input must have 2 dimensions.");
  NVTE_CHECK(out.data.shape.ndim == 2, "T output must have 2 dimensions.");
  const size_t rows = in.data.shape.data[0], cols = in.data.shape.data[1];
  NVTE_CHECK(out.data.shape.data[0] == cols && out.data.shape.data[1] == rows, "Wrong dimension of T output.");
  NVTE_CHECK(out.data.dtype == in.data.dtype, "T output must have the same type as input.");
  NVTE_CHECK(is_fp8(in.data.dtype), "nvte_fp8_transpose_dbias expects an FP8 input, got ", dtype_name(in.data.dtype));
  NVTE_CHECK(dbias.data.shape.ndim == 1 && dbias.data.shape.data[0] == cols, "Wrong shape of DBias.");
  NVTE_CHECK(is_high_precision(dbias.data.dtype), "DBias must be FP32/BF16/FP16");
  if (dbias_workspace_query(in, workspace)) return;
  NVTE_REQUIRE_DATA(in.data, "fp8_transpose_dbias input");
  NVTE_CHECK(dbias.data.ptr != nullptr, "DBias is not allocated.");
  const size_t slabs = ceil_div(rows, kSlab);
  NVTE_CHECK(workspace.data.numel() * dtype_bits(workspace.data.dtype) / 8 >= slabs * cols * sizeof(float),
             "Invalid workspace (expected dims=(", slabs, ",", cols, "), dtype=Float32; found dims=",
             workspace.data.shape, ")");
  transpose_launch(in, nullptr, out, stream);
  if (cols == 0) return;
  const float *si = static_cast<const float *>(out.scale_inv.ptr ? out.scale_inv.ptr : in.scale_inv.ptr);
  float *ws = static_cast<float *>(workspace.data.ptr);
  const uint8_t *ip = static_cast<const uint8_t *>(in.data.ptr);
  const bool vec4 = cols % 4 == 0 && (reinterpret_cast<uintptr_t>(ip) & 3u) == 0;
  const dim3 g1(static_cast<unsigned>(ceil_div(cols, vec4 ? 1024 : 256)), static_cast<unsigned>(std::max<size_t>(slabs, 1)));
  if (slabs) {
    const bool e4 = in.data.dtype == kNVTEFloat8E4M3;
#define TE_LAUNCH(F, V) hipLaunchKernelGGL((colsum_partial_fp8_kernel<F, V>), g1, dim3(256), 0, stream, ip, si, ws, rows, cols)
    if (e4) { if (vec4) TE_LAUNCH(Fp8::E4M3, true); else TE_LAUNCH(Fp8::E4M3, false); }
    else    { if (vec4) TE_LAUNCH(Fp8::E5M2, true); else TE_LAUNCH(Fp8::E5M2, false); }
#undef TE_LAUNCH
    NVTE_CHECK_HIP(hipGetLastError());
  }
  dbias_finalize_launch(ws, dbias, slabs, cols, stream);
}

// Returns true when the call was only a workspace query.
bool dbias_workspace_query(const Tensor &in, Tensor &workspace) {
  if (workspace.data.ptr != nullptr) return false;
  const size_t rows = in.rows(), cols = in.cols();
  workspace.data.shape = NVTEShape{{ceil_div(rows, kSlab), cols}, 2};
  workspace.data.dtype = kNVTEFloat32;
  return true;
}

void dbias_launch(const Tensor &in, Tensor &dbias, Tensor &workspace, hipStream_t stream) {
  const size_t rows = in.rows(), cols = in.cols();
  NVTE_CHECK(dbias.data.dtype == in.data.dtype, "DBias must have the same type as input.");
  NVTE_CHECK(dbias.data.shape.ndim == 1 && dbias.data.shape.data[0] == cols, "Wrong shape of DBias.");
  NVTE_CHECK(dbias.data.ptr != nullptr, "DBias is not allocated.");
  NVTE_REQUIRE_DATA(in.data, "dbias input");
  const size_t slabs = ceil_div(rows, kSlab);
  NVTE_CHECK(workspace.data.dtype == kNVTEFloat32 && workspace.data.numel() >= slabs * cols,
             "Workspace is too small (need ", slabs * cols, " fp32 entries)");
  if (cols == 0) return;
  float *ws = static_cast<float *>(workspace.data.ptr);
  const dim3 g1(static_cast<unsigned>(ceil_div(cols, 256)), static_cast<unsigned>(std::max<size_t>(slabs, 1)));
  const dim3 g2(static_cast<unsigned>(ceil_div(cols, 256)));
  switch (in.data.dtype) {
    case kNVTEBFloat16:
      if (slabs) hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, g1, dim3(256), 0, stream,
                                    static_cast<const bf16_t *>(in.data.ptr), ws, rows, cols);
      hipLaunchKernelGGL(colsum_final_kernel<bf16_t>, g2, dim3(256), 0, stream, ws,
                         static_cast<bf16_t *>(dbias.data.ptr), slabs, cols);
      break;
    case kNVTEFloat16:
      if (slabs) hipLaunchKernelGGL(colsum_partial_kernel<f16_t>, g1, dim3(256), 0, stream,
                                    static_cast<const f16_t *>(in.data.ptr), ws, rows, cols);
      hipLaunchKernelGGL(colsum_final_kernel<f16_t>, g2, dim3(256), 0, stream, ws,
                         static_cast<f16_t *>(dbias.data.ptr), slabs, cols);
      break;
    case kNVTEFloat32:
      if (slabs) hipLaunchKernelGGL(colsum_partial_kernel<float>, g1, dim3(256), 0, stream,
                                    static_cast<const float *>(in.data.ptr), ws, rows, cols);
      hipLaunchKernelGGL(colsum_final_kernel<float>, g2, dim3(256), 0, stream, ws,
                         static_cast<float *>(dbias.data.ptr), slabs, cols);
      break;
    default:
      NVTE_ERROR("Unsupported input dtype ", dtype_name(in.data.dtype));
  }
  NVTE_CHECK_HIP(hipGetLastError());
}

// Slab reduce alone: for kernels that produce the [slabs, cols] fp32 partials themselves (activation.hip).
void dbias_finalize_launch(const float *ws, Tensor &dbias, size_t slabs, size_t cols, hipStream_t stream) {
  if (cols == 0) return;
  const dim3 g2(static_cast<unsigned>(ceil_div(cols, 256)));
  switch (dbias.data.dtype) {
    case kNVTEBFloat16:
      hipLaunchKernelGGL(colsum_final_kernel<bf16_t>, g2, dim3(256), 0, stream, ws,
                         static_cast<bf16_t *>(dbias.data.ptr), slabs, cols);
      break;
    case kNVTEFloat16:
      hipLaunchKernelGGL(colsum_final_kernel<f16_t>, g2, dim3(256), 0, stream, ws,
                         static_cast<f16_t *>(dbias.data.ptr), slabs, cols);
      break;
    case kNVTEFloat32:
      hipLaunchKernelGGL(colsum_final_kernel<float>, g2, dim3(256), 0, stream, ws,
                         static_cast<float *>(dbias.data.ptr), slabs, cols);
      break;
    default:
      NVTE_ERROR("Unsupported dbias dtype ", dtype_name(dbias.data.dtype));
  }
  NVTE_CHECK_HIP(hipGetLastError());
}

}  // namespace te
