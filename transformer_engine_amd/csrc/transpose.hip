// 2-D transpose (FP8 bytes after an all-gather, or any 1/2/4-byte dtype) and
// per-tensor dequantize.
//
// Replaces: ref common/transpose/transpose.cu:98-190 (kernels), :195-297 (launcher + entry
// points), common/cast/fp8/dequantize_fp8.cuh:31-49.
//
// The aligned byte path reuses the cast_transpose tile scheme: 128x128 byte tile per
// 256-thread workgroup, 8x8 byte blocks transposed in registers with v_perm_b32, one
// XOR-swizzled LDS round trip, 16-byte coalesced stores.  2 B/element of HBM traffic.
#include <hip/hip_runtime.h>

#include "common.h"
#include "device_utils.h"
#include "transformer_engine/cast_transpose_noop.h"
#include "transformer_engine/transpose.h"

namespace te {
namespace {

constexpr int kTile = 128;

__global__ __launch_bounds__(256) void transpose_bytes_tile_kernel(
    const uint8_t *__restrict__ in, const float *__restrict__ noop, uint8_t *__restrict__ out,
    const size_t rows, const size_t cols, const unsigned tiles_c) {
  if (noop_set(noop)) return;
  __shared__ __attribute__((aligned(16))) uint8_t tile_t[kTile * kTile];
  const unsigned t = threadIdx.x;
  const unsigned cg = t & 15u, rg = t >> 4;
  const size_t tile_r = blockIdx.x / tiles_c, tile_col = blockIdx.x % tiles_c;
  const size_t row0 = tile_r * kTile + rg * 8, col0 = tile_col * kTile + cg * 8;

  uint32_t q[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32x2 v = *reinterpret_cast<const u32x2 *>(in + (row0 + i) * cols + col0);
    q[i][0] = v[0];
    q[i][1] = v[1];
  }
  transpose4x4_bytes(q[0][0], q[1][0], q[2][0], q[3][0]);
  transpose4x4_bytes(q[0][1], q[1][1], q[2][1], q[3][1]);
  transpose4x4_bytes(q[4][0], q[5][0], q[6][0], q[7][0]);
  transpose4x4_bytes(q[4][1], q[5][1], q[6][1], q[7][1]);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned c = cg * 8 + h * 4 + j;
      u32x2 v = {q[j][h], q[4 + j][h]};
      *reinterpret_cast<u32x2 *>(&tile_t[c * kTile + ((rg ^ cg) << 3)]) = v;
    }
  }
  __syncthreads();
  const unsigned k = t & 7u;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned c = (t >> 3) + 32u * p;
    const unsigned cgc = c >> 3;
    u32x4 v = *reinterpret_cast<const u32x4 *>(&tile_t[c * kTile + ((k ^ (cgc >> 1)) << 4)]);
    if (cgc & 1u) {
      u32x4 w = {v[2], v[3], v[0], v[1]};
      v = w;
    }
    *reinterpret_cast<u32x4 *>(out + (tile_col * kTile + c) * rows + tile_r * kTile + k * 16) = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_general_kernel(const T *__restrict__ in,
                                                                const float *__restrict__ noop,
                                                                T *__restrict__ out,
                                                                const size_t rows, const size_t cols,
                                                                const unsigned tiles_c) {
  if (noop_set(noop)) return;
  __shared__ T tile[64][65];
  const unsigned lx = threadIdx.x & 63u, ly = threadIdx.x >> 6;
  const size_t r_base = static_cast<size_t>(blockIdx.x / tiles_c) * 64;
  const size_t c_base = static_cast<size_t>(blockIdx.x % tiles_c) * 64;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const size_t r = r_base + ly + 4 * i, c = c_base + lx;
    if (r < rows && c < cols) tile[lx][ly + 4 * i] = in[r * cols + c];
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const size_t c = c_base + ly + 4 * i, r = r_base + lx;
    if (r < rows && c < cols) out[c * rows + r] = tile[ly + 4 * i][lx];
  }
}

// ---------------------------------------------------------------- per-tensor dequantize
template <Fp8 F, typename OT>
__global__ __launch_bounds__(256) void dequantize_kernel(const uint8_t *__restrict__ in,
                                                         const float *__restrict__ scale_inv,
                                                         OT *__restrict__ out, const size_t n) {
  const float s = *scale_inv;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  const size_t n4 = n / 4;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const uint32_t w = (reinterpret_cast<uintptr_t>(in) & 3u) == 0
                           ? reinterpret_cast<const uint32_t *>(in)[i]
                           : (static_cast<uint32_t>(in[4 * i]) | (static_cast<uint32_t>(in[4 * i + 1]) << 8) |
                              (static_cast<uint32_t>(in[4 * i + 2]) << 16) |
                              (static_cast<uint32_t>(in[4 * i + 3]) << 24));
#pragma unroll
    for (int k = 0; k < 4; ++k) narrow(fp8_to_float<F>(w, k) * s, out[4 * i + k]);
  }
  // tail
  const size_t gid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid < n - n4 * 4) {
    const size_t i = n4 * 4 + gid;
    narrow(fp8_to_float<F>(in[i], 0) * s, out[i]);
  }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

void transpose_launch(const Tensor &in, const float *noop, Tensor &out, hipStream_t stream) {
  NVTE_CHECK(in.data.shape.ndim == 2, "Input must have 2 dimensions.");
  NVTE_CHECK(out.data.shape.ndim == 2, "Output must have 2 dimensions.");
  const size_t cols = in.data.shape.data[1], rows = in.data.shape.data[0];
  NVTE_CHECK(out.data.shape.data[0] == cols, "Wrong dimension of output.");
  NVTE_CHECK(out.data.shape.data[1] == rows, "Wrong dimension of output.");
  NVTE_CHECK(in.data.ptr != nullptr, "Input is not allocated.");
  NVTE_CHECK(out.data.ptr != nullptr, "Output is not allocated.");
  NVTE_CHECK(in.data.dtype == out.data.dtype, "Input (dtype=", dtype_name(in.data.dtype),
             ") and output (dtype=", dtype_name(out.data.dtype), ") do not match.");
  if (rows == 0 || cols == 0) return;
  const size_t esz = dtype_bits(in.data.dtype) / 8;
  if (esz == 1 && rows % kTile == 0 && cols % kTile == 0 && aligned16(in.data.ptr) &&
      aligned16(out.data.ptr)) {
    const unsigned tiles_c = static_cast<unsigned>(cols / kTile);
    const size_t nblk = (rows / kTile) * tiles_c;
    NVTE_CHECK(nblk < (1ull << 31), "tensor too large");
    hipLaunchKernelGGL(transpose_bytes_tile_kernel, dim3(nblk), dim3(256), 0, stream,
                       static_cast<const uint8_t *>(in.data.ptr), noop,
                       static_cast<uint8_t *>(out.data.ptr), rows, cols, tiles_c);
  } else {
    const unsigned tiles_c = static_cast<unsigned>(ceil_div(cols, 64));
    const size_t nblk = ceil_div(rows, 64) * tiles_c;
    NVTE_CHECK(nblk < (1ull << 31), "tensor too large");
    switch (esz) {
      case 1:
        hipLaunchKernelGGL(transpose_general_kernel<uint8_t>, dim3(nblk), dim3(256), 0, stream,
                           static_cast<const uint8_t *>(in.data.ptr), noop,
                           static_cast<uint8_t *>(out.data.ptr), rows, cols, tiles_c);
        break;
      case 2:
        hipLaunchKernelGGL(transpose_general_kernel<uint16_t>, dim3(nblk), dim3(256), 0, stream,
                           static_cast<const uint16_t *>(in.data.ptr), noop,
                           static_cast<uint16_t *>(out.data.ptr), rows, cols, tiles_c);
        break;
      case 4:
        hipLaunchKernelGGL(transpose_general_kernel<uint32_t>, dim3(nblk), dim3(256), 0, stream,
                           static_cast<const uint32_t *>(in.data.ptr), noop,
                           static_cast<uint32_t *>(out.data.ptr), rows, cols, tiles_c);
        break;
      default:
        NVTE_ERROR("Unsupported element size for transpose: ", esz);
    }
  }
  NVTE_CHECK_HIP(hipGetLastError());
}

template <Fp8 F>
static void dequant_by_out(const Tensor &in, Tensor &out, size_t n, hipStream_t stream) {
  const DeviceInfo &di = device_info();
  const unsigned grid = static_cast<unsigned>(
      std::max<size_t>(1, std::min<size_t>(ceil_div(ceil_div(n, 4), 256), static_cast<size_t>(di.num_cus) * 16)));
  const uint8_t *ip = static_cast<const uint8_t *>(in.data.ptr);
  const float *sp = static_cast<const float *>(in.scale_inv.ptr);
  switch (out.data.dtype) {
    case kNVTEFloat32:
      hipLaunchKernelGGL((dequantize_kernel<F, float>), dim3(grid), dim3(256), 0, stream, ip, sp,
                         static_cast<float *>(out.data.ptr), n);
      break;
    case kNVTEFloat16:
      hipLaunchKernelGGL((dequantize_kernel<F, f16_t>), dim3(grid), dim3(256), 0, stream, ip, sp,
                         static_cast<f16_t *>(out.data.ptr), n);
      break;
    case kNVTEBFloat16:
      hipLaunchKernelGGL((dequantize_kernel<F, bf16_t>), dim3(grid), dim3(256), 0, stream, ip, sp,
                         static_cast<bf16_t *>(out.data.ptr), n);
      break;
    default:
      NVTE_ERROR("Output must be in higher precision, got ", dtype_name(out.data.dtype));
  }
  NVTE_CHECK_HIP(hipGetLastError());
}

void mxfp8_dequantize_launch(const Tensor &in, Tensor &out, hipStream_t stream);  // mxfp8.hip

void dequantize_launch(const Tensor &in, Tensor &out, hipStream_t stream) {
  check_input(in, "cast_input");
  check_output(out, "cast_output");
  if (is_mxfp8(in.scaling_mode)) {
    mxfp8_dequantize_launch(in, out, stream);
    return;
  }
  NVTE_CHECK(is_tensor_scaling(in.scaling_mode), "Not implemented scaling mode: ",
             scaling_mode_name(in.scaling_mode));
  NVTE_CHECK(is_fp8(in.data.dtype), "Input must have FP8 type.");
  NVTE_CHECK(!is_fp8(out.data.dtype), "Output must be in higher precision.");
  NVTE_CHECK(same_shape(out.data.shape, in.data.shape), "Input and output shapes need to match.");
  const size_t n = in.data.numel();
  if (n == 0) return;
  NVTE_REQUIRE_DATA(in.data, "dequantize input");
  NVTE_REQUIRE_DATA(out.data, "dequantize output");
  NVTE_CHECK(in.scale_inv.ptr != nullptr, "Input scale_inv is not allocated.");
  if (in.data.dtype == kNVTEFloat8E4M3)
    dequant_by_out<Fp8::E4M3>(in, out, n, stream);
  else
    dequant_by_out<Fp8::E5M2>(in, out, n, stream);
}

}  // namespace te

using namespace te;

extern "C" {

void nvte_transpose(const NVTETensor input, NVTETensor output, hipStream_t stream) {
  api_guard("nvte_transpose", [&] {
    transpose_launch(*lookup_checked(input), nullptr, *lookup_checked(output), stream);
  });
}

void nvte_transpose_with_noop(const NVTETensor input, const NVTETensor noop, NVTETensor output,
                              hipStream_t stream) {
  api_guard("nvte_transpose_with_noop", [&] {
    const Tensor &n = *lookup_checked(noop);
    check_noop(n, "transpose");
    transpose_launch(*lookup_checked(input), static_cast<const float *>(n.data.ptr),
                     *lookup_checked(output), stream);
  });
}

void nvte_fp8_transpose_dbias(const NVTETensor input, NVTETensor transposed_output, NVTETensor dbias,
                              NVTETensor workspace, hipStream_t stream) {
  api_guard("nvte_fp8_transpose_dbias", [&] {
    te::fp8_transpose_dbias_launch(*lookup_checked(input), *lookup_checked(transposed_output),
                                   *lookup_checked(dbias), *lookup_checked(workspace), stream);
  });
}

}  // extern "C"
