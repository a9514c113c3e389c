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

}  // namespace

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
