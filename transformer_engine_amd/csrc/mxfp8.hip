// MXFP8 (one e8m0 scale per 32 consecutive elements) quantize / dequantize and the
// scale-layout "swizzle" for the GEMM.
//
// Replaces: ref common/cast/mxfp8/quantize_mxfp8.cuh:49-555 (kernel), :573-873 (launcher),
//           common/cast/mxfp8/dequantize_mxfp8.cuh, common/cast/mxfp8/swizzle.cuh:30-41,
//           common/swizzle/swizzle.cu:1957 (entry point).
// Math (host twin: ref tests/cpp/operator/test_cast_mxfp8.cu:39-166):
//   block amax m -> e = float_to_e8m0(m * (1/fp8_max)) (value rounded UP to a power of
//   two) -> q = fp8_satfinite(x * 2^(127-e)).  Rowwise blocks are 1x32, columnwise
//   blocks are 32x1; the columnwise payload keeps the [rows, cols] layout.
//
// Aligned path: a 256-thread workgroup owns a 128x128 tile, thread (cg, rg) an 8x8
// block held in registers for BOTH directions, so the input is read once.  A 1x32 row
// block is 4 neighbouring lanes (2 shuffle steps); a 32x1 column block is the 4 lanes
// {cg, cg+16, cg+32, cg+48} of one wave (2 shuffle steps).  No LDS is needed: neither
// payload is transposed.  HBM: 2 + 1 + 1 + 2/32 B per bf16 element.
#include <hip/hip_runtime.h>

#include "common.h"
#include "device_utils.h"
#include "transformer_engine/swizzle.h"

namespace te {
namespace {

constexpr int kTile = 128;

template <typename IT>
struct Ld8;
template <>
struct Ld8<bf16_t> {
  static __device__ __forceinline__ void run(const bf16_t *p, float (&f)[8]) {
    const u32x4 v = ld_stream<u32x4>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = __uint_as_float(v[k] << 16);
      f[2 * k + 1] = __uint_as_float(v[k] & 0xFFFF0000u);
    }
  }
};
template <>
struct Ld8<f16_t> {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ void run(const f16_t *p, float (&f)[8]) {
    const h8 v = ld_stream<h8>(p);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = static_cast<float>(v[k]);
  }
};
template <>
struct Ld8<float> {
  static __device__ __forceinline__ void run(const float *p, float (&f)[8]) {
    const f32x4 a = ld_stream<f32x4>(p);
    const f32x4 b = ld_stream<f32x4>(p + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[k] = a[k];
      f[4 + k] = b[k];
    }
  }
};

// compact -> GEMM layout index (ref common/cast/mxfp8/swizzle.cuh:30-41)
__host__ __device__ __forceinline__ size_t swizzled_idx(size_t i, size_t j, size_t tiles_x) {
  const size_t tile = (i / 128) * tiles_x + (j / 4);
  const size_t y = i % 128, x = j % 4;
  return tile * 512 + (y % 32) * 16 + (y / 32) * 4 + x;
}

// TCOL: store the columnwise payload transposed ([cols, rows]) and its compact scales as
// [cols, rows/32] (amd_ext.h, nvte_amd_set_columnwise_transposed)
template <typename IT, Fp8 F, bool ROW, bool COL, bool SWZ, bool TCOL = false>
__global__ __launch_bounds__(256) void mxfp8_tile_kernel(
    const IT *__restrict__ in, const float *__restrict__ noop, uint8_t *__restrict__ out_row,
    uint8_t *__restrict__ out_col, uint8_t *__restrict__ sc_row, uint8_t *__restrict__ sc_col,
    float *__restrict__ amax_ptr, const size_t rows, const size_t cols, const size_t stride_r,
    const size_t stride_c, const unsigned tiles_c) {
  if (noop_set(noop)) return;
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) uint8_t tile_t[TCOL ? kTile * kTile : 16];
  constexpr float rcp_max = 1.0f / Fp8Traits<F>::max_finite;  // correctly rounded constant

  const unsigned t = threadIdx.x;
  const unsigned cg = t & 15u, rg = t >> 4;
  const size_t tile_r = blockIdx.x / tiles_c, tile_col = blockIdx.x % tiles_c;
  const size_t row0 = tile_r * kTile + rg * 8, col0 = tile_col * kTile + cg * 8;

  float f[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) Ld8<IT>::run(in + (row0 + i) * cols + col0, f[i]);

  float tmax = 0.f;

  if constexpr (ROW) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, __builtin_fabsf(f[i][j]));
      m = fmaxf(m, __shfl_xor(m, 1, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      tmax = fmaxf(tmax, m);
      const uint8_t e = float_to_e8m0(m * rcp_max);
      const float r = e8m0_rcp(e);
      // the block's largest magnitude times its own reciprocal scale is inside the range unless the block
      // holds an infinity: no per-value clamp then (cvt8_fp8)
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f[i][j] * r;
      const u32x2 q = cvt8_fp8<F>(o, m * r <= Fp8Traits<F>::max_finite);
      *reinterpret_cast<u32x2 *>(out_row + (row0 + i) * cols + col0) = q;
      // gather the 4 block scales of this row's 128-column segment into lane cg == 0
      uint32_t e32 = e;
      const uint32_t e1 = __shfl_down(e32, 4, 64);
      const uint32_t e2 = __shfl_down(e32, 8, 64);
      const uint32_t e3 = __shfl_down(e32, 12, 64);
      if (cg == 0) {
        const uint32_t packed = e32 | (e1 << 8) | (e2 << 16) | (e3 << 24);
        const size_t r_idx = row0 + i, c_idx = tile_col * 4;
        const size_t idx = SWZ ? swizzled_idx(r_idx, c_idx, stride_r / 4) : r_idx * stride_r + c_idx;
        *reinterpret_cast<uint32_t *>(sc_row + idx) = packed;
      }
    }
  }

  if constexpr (COL) {
    uint32_t qc[8][2];
    uint32_t ebytes[2] = {0u, 0u};
    bool in_range = true;   // every one of this thread's 8 column blocks scales into the finite range
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) m = fmaxf(m, __builtin_fabsf(f[i][j]));
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      if constexpr (!ROW) tmax = fmaxf(tmax, m);
      const uint8_t e = float_to_e8m0(m * rcp_max);
      ebytes[j >> 2] |= static_cast<uint32_t>(e) << (8 * (j & 3));
      const float r = e8m0_rcp(e);
      in_range = in_range && (m * r <= Fp8Traits<F>::max_finite);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i][j] *= r;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32x2 w = cvt8_fp8<F>(f[i], in_range);
      qc[i][0] = w[0];
      qc[i][1] = w[1];
      if constexpr (!TCOL) {
        u32x2 q = {qc[i][0], qc[i][1]};
        *reinterpret_cast<u32x2 *>(out_col + (row0 + i) * cols + col0) = q;
      }
    }
    if constexpr (TCOL)
      store_tile_transposed(tile_t, qc, t, out_col + (tile_col * kTile) * rows + tile_r * kTile, rows);
    if ((t & 63u) < 16u) {  // rg_local == 0 holds the wave's 32-row block scale for 8 columns
      const size_t blk = tile_r * 4 + (t >> 6);
      if constexpr (TCOL && !SWZ) {
        // compact scales of the transposed operand: [column][32-row block], stride_c = pad4(rows/32)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          sc_col[(col0 + j) * stride_c + blk] = static_cast<uint8_t>(ebytes[j >> 2] >> (8 * (j & 3)));
      } else if constexpr (SWZ) {
        const size_t tiles_x = ceil_div_dev(rows, 128);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          sc_col[swizzled_idx(col0 + j, blk, tiles_x)] =
              static_cast<uint8_t>(ebytes[j >> 2] >> (8 * (j & 3)));
      } else {
        u32x2 v = {ebytes[0], ebytes[1]};
        *reinterpret_cast<u32x2 *>(sc_col + blk * stride_c + col0) = v;
      }
    }
  }

  if (amax_ptr != nullptr) {
    tmax = block_max<4>(tmax, red);
    if (t == 0) atomic_max_nonneg(amax_ptr, tmax);
  }
}

// ---------------------------------------------------------------- general (any shape) kernel: one 32x32 tile per workgroup
template <typename IT, Fp8 F, bool SWZ>
__global__ __launch_bounds__(256) void mxfp8_general_kernel(
    const IT *__restrict__ in, const float *__restrict__ noop, uint8_t *__restrict__ out_row,
    uint8_t *__restrict__ out_col, uint8_t *__restrict__ sc_row, uint8_t *__restrict__ sc_col,
    float *__restrict__ amax_ptr, const size_t rows, const size_t cols, const size_t stride_r,
    const size_t stride_c, const unsigned tiles_c) {
  if (noop_set(noop)) return;
  __shared__ float tile[32][33];
  __shared__ float col_rcp[32];
  __shared__ float red[4];
  constexpr float rcp_max = 1.0f / Fp8Traits<F>::max_finite;
  const unsigned t = threadIdx.x, tx = t & 31u, ty = t >> 5;
  const size_t tile_y = blockIdx.x / tiles_c, tile_x = blockIdx.x % tiles_c;
  const size_t r0 = tile_y * 32, c0 = tile_x * 32;
  float tmax = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const size_t r = r0 + ty + 8 * i, c = c0 + tx;
    const float x = (r < rows && c < cols) ? widen(in[r * cols + c]) : 0.f;
    tile[ty + 8 * i][tx] = x;
    tmax = fmaxf(tmax, __builtin_fabsf(x));
  }
  __syncthreads();
  if (out_row != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lr = ty + 8 * i;
      const size_t r = r0 + lr, c = c0 + tx;
      const float x = tile[lr][tx];
      float m = __builtin_fabsf(x);
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 32));
      const uint8_t e = float_to_e8m0(m * rcp_max);
      if (r < rows && c < cols) out_row[r * cols + c] = cvt1_fp8<F>(x * e8m0_rcp(e));
      if (tx == 0 && r < rows) {
        const size_t idx = SWZ ? swizzled_idx(r, tile_x, ceil_div_dev(cols, 128)) : r * stride_r + tile_x;
        sc_row[idx] = e;
      }
    }
  }
  if (out_col != nullptr) {
    if (t < 32) {
      float m = 0.f;
      for (int i = 0; i < 32; ++i) m = fmaxf(m, __builtin_fabsf(tile[i][t]));
      const uint8_t e = float_to_e8m0(m * rcp_max);
      col_rcp[t] = e8m0_rcp(e);
      const size_t c = c0 + t;
      if (c < cols) {
        const size_t idx = SWZ ? swizzled_idx(c, tile_y, ceil_div_dev(rows, 128)) : tile_y * stride_c + c;
        sc_col[idx] = e;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lr = ty + 8 * i;
      const size_t r = r0 + lr, c = c0 + tx;
      if (r < rows && c < cols) out_col[r * cols + c] = cvt1_fp8<F>(tile[lr][tx] * col_rcp[tx]);
    }
  }
  if (amax_ptr != nullptr) {
    tmax = block_max<4>(tmax, red);
    if (t == 0) atomic_max_nonneg(amax_ptr, tmax);
  }
}

// ---------------------------------------------------------------- dequantize
template <Fp8 F, typename OT, bool ROWWISE, bool SWZ>
__global__ __launch_bounds__(256) void mxfp8_dequant_kernel(const uint8_t *__restrict__ in,
                                                            const uint8_t *__restrict__ sc,
                                                            OT *__restrict__ out, const size_t rows,
                                                            const size_t cols, const size_t stride) {
  const size_t n = rows * cols;
  const size_t step = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += step) {
    const size_t r = i / cols, c = i % cols;
    size_t idx;
    if constexpr (ROWWISE) {
      idx = SWZ ? swizzled_idx(r, c / 32, ceil_div_dev(cols, 128)) : r * stride + c / 32;
    } else {
      idx = SWZ ? swizzled_idx(c, r / 32, ceil_div_dev(rows, 128)) : (r / 32) * stride + c;
    }
    narrow(fp8_to_float<F>(in[i], 0) * e8m0_to_float(sc[idx]), out[i]);
  }
}

// ---------------------------------------------------------------- scale swizzle / unswizzle
// in: compact [M, K] byte matrix (already padded to 128 x 4 multiples); out: tiled layout.
template <bool FORWARD>
__global__ __launch_bounds__(256) void swizzle_scales_kernel(const uint8_t *__restrict__ in,
                                                             uint8_t *__restrict__ out,
                                                             const size_t m, const size_t k) {
  // one thread per dword of the SWIZZLED image: (tile, y%32, y/32) -> 4 consecutive x
  const size_t n4 = m * k / 4;
  const size_t step = static_cast<size_t>(gridDim.x) * blockDim.x;
  const size_t tiles_x = k / 4;
  for (size_t d = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; d < n4; d += step) {
    const size_t tile = d / 128, in_tile = d % 128;
    const size_t y = (in_tile / 4) + 32 * (in_tile % 4);
    const size_t i = (tile / tiles_x) * 128 + y, j = (tile % tiles_x) * 4;
    if constexpr (FORWARD) {
      reinterpret_cast<uint32_t *>(out)[d] = *reinterpret_cast<const uint32_t *>(in + i * k + j);
    } else {
      *reinterpret_cast<uint32_t *>(out + i * k + j) = reinterpret_cast<const uint32_t *>(in)[d];
    }
  }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

template <typename IT, Fp8 F>
void launch_quant(const Tensor &in, const float *noop, Tensor &out, hipStream_t stream) {
  const size_t rows = in.rows(), cols = in.cols();
  const bool row = out.has_data(), col = out.has_columnwise_data();
  const bool swz = out.with_gemm_swizzled_scales;
  const bool tcol = out.has_columnwise_data() && out.columnwise_transposed;
  const IT *ip = static_cast<const IT *>(in.data.ptr);
  uint8_t *orow = static_cast<uint8_t *>(out.data.ptr);
  uint8_t *ocol = static_cast<uint8_t *>(out.columnwise_data.ptr);
  uint8_t *srow = static_cast<uint8_t *>(out.scale_inv.ptr);
  uint8_t *scol = static_cast<uint8_t *>(out.columnwise_scale_inv.ptr);
  float *amax = static_cast<float *>(out.amax.ptr);
  const size_t stride_r = row ? out.scale_inv.shape.data[1] : 1;
  const size_t stride_c = col ? out.columnwise_scale_inv.shape.data[1] : 1;

  const bool fast = rows % kTile == 0 && cols % kTile == 0 && aligned16(ip) &&
                    (!row || (aligned8(orow) && aligned8(srow))) &&
                    (!col || (aligned8(ocol) && aligned8(scol)));
  NVTE_CHECK(!tcol || (fast && aligned16(ocol)),
             "MXFP8 transposed-columnwise output needs 128-multiple dims and 16-byte aligned buffers (got ",
             rows, "x", cols, ")");
  if (fast) {
    const unsigned tiles_c = static_cast<unsigned>(cols / kTile);
    const size_t nblk = (rows / kTile) * tiles_c;
    NVTE_CHECK(nblk < (1ull << 31), "tensor too large");
#define TE_MX_LAUNCH(R, C, S)                                                                      \
  hipLaunchKernelGGL((mxfp8_tile_kernel<IT, F, R, C, S>), dim3(nblk), dim3(256), 0, stream, ip,   \
                     noop, orow, ocol, srow, scol, amax, rows, cols, stride_r, stride_c, tiles_c)
#define TE_MX_LAUNCH_T(R, S)                                                                        \
  hipLaunchKernelGGL((mxfp8_tile_kernel<IT, F, R, true, S, true>), dim3(nblk), dim3(256), 0, stream, ip, \
                     noop, orow, ocol, srow, scol, amax, rows, cols, stride_r, stride_c, tiles_c)
    if (tcol) {
      if (row) {
        if (swz) TE_MX_LAUNCH_T(true, true); else TE_MX_LAUNCH_T(true, false);
      } else {
        if (swz) TE_MX_LAUNCH_T(false, true); else TE_MX_LAUNCH_T(false, false);
      }
    } else if (row && col) {
      if (swz) TE_MX_LAUNCH(true, true, true); else TE_MX_LAUNCH(true, true, false);
    } else if (row) {
      if (swz) TE_MX_LAUNCH(true, false, true); else TE_MX_LAUNCH(true, false, false);
    } else {
      if (swz) TE_MX_LAUNCH(false, true, true); else TE_MX_LAUNCH(false, true, false);
    }
#undef TE_MX_LAUNCH
#undef TE_MX_LAUNCH_T
  } else {
    const unsigned tiles_c = static_cast<unsigned>(ceil_div(cols, 32));
    const size_t nblk = ceil_div(rows, 32) * tiles_c;
    NVTE_CHECK(nblk < (1ull << 31), "tensor too large");
    if (swz) {
      hipLaunchKernelGGL((mxfp8_general_kernel<IT, F, true>), dim3(nblk), dim3(256), 0, stream, ip,
                         noop, orow, ocol, srow, scol, amax, rows, cols, stride_r, stride_c, tiles_c);
    } else {
      hipLaunchKernelGGL((mxfp8_general_kernel<IT, F, false>), dim3(nblk), dim3(256), 0, stream, ip,
                         noop, orow, ocol, srow, scol, amax, rows, cols, stride_r, stride_c, tiles_c);
    }
  }
  NVTE_CHECK_HIP(hipGetLastError());
}

template <Fp8 F, typename OT>
void launch_dequant(const Tensor &in, Tensor &out, hipStream_t stream) {
  const size_t rows = in.rows(), cols = in.cols();
  const bool rowwise = in.has_data();
  const uint8_t *ip = static_cast<const uint8_t *>(rowwise ? in.data.ptr : in.columnwise_data.ptr);
  const Buf &sb = rowwise ? in.scale_inv : in.columnwise_scale_inv;
  const uint8_t *sp = static_cast<const uint8_t *>(sb.ptr);
  NVTE_CHECK(sp != nullptr, "Scaling tensor must be allocated");
  const size_t stride = sb.shape.data[1];
  const DeviceInfo &di = device_info();
  const unsigned grid = static_cast<unsigned>(std::max<size_t>(
      1, std::min<size_t>(ceil_div(rows * cols, 256), static_cast<size_t>(di.num_cus) * 16)));
  OT *op = static_cast<OT *>(out.data.ptr);
  const bool swz = in.with_gemm_swizzled_scales;
#define TE_DQ(RW, S)                                                                              \
  hipLaunchKernelGGL((mxfp8_dequant_kernel<F, OT, RW, S>), dim3(grid), dim3(256), 0, stream, ip,  \
                     sp, op, rows, cols, stride)
  if (rowwise) {
    if (swz) TE_DQ(true, true); else TE_DQ(true, false);
  } else {
    if (swz) TE_DQ(false, true); else TE_DQ(false, false);
  }
#undef TE_DQ
  NVTE_CHECK_HIP(hipGetLastError());
}

void swizzle_one(const Buf &in, Buf &out, bool forward, const char *what, hipStream_t stream) {
  NVTE_CHECK(in.ptr != nullptr && out.ptr != nullptr, what, " scale buffers must be allocated");
  NVTE_CHECK(in.shape.ndim == 2, what, " scale_inv must be 2-D, got ", in.shape);
  const size_t m = in.shape.data[0], k = in.shape.data[1];
  NVTE_CHECK(m % 128 == 0 && k % 4 == 0, what, " scale_inv shape must be padded to (128, 4), got ",
             in.shape);
  NVTE_CHECK(out.numel() == m * k, what, " output scale_inv has wrong size (expected ", m * k,
             ", got ", out.numel(), ")");
  if (m * k == 0) return;
  const unsigned grid = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(ceil_div(m * k / 4, 256), 4096)));
  if (forward) {
    hipLaunchKernelGGL(swizzle_scales_kernel<true>, dim3(grid), dim3(256), 0, stream,
                       static_cast<const uint8_t *>(in.ptr), static_cast<uint8_t *>(out.ptr), m, k);
  } else {
    hipLaunchKernelGGL(swizzle_scales_kernel<false>, dim3(grid), dim3(256), 0, stream,
                       static_cast<const uint8_t *>(in.ptr), static_cast<uint8_t *>(out.ptr), m, k);
  }
  NVTE_CHECK_HIP(hipGetLastError());
}

// Columnwise scales are [ceil(R/32) pad4, C pad128]; the GEMM layout treats them as the
// transposed matrix [C, R/32] (ref quantize_mxfp8.cuh:273-278).  A transposing variant of
// the copy handles that case.
template <bool FORWARD>
__global__ __launch_bounds__(256) void swizzle_col_scales_kernel(const uint8_t *__restrict__ in,
                                                                 uint8_t *__restrict__ out,
                                                                 const size_t kblocks /*rows of compact*/,
                                                                 const size_t n /*cols of compact*/) {
  const size_t total = kblocks * n;
  const size_t step = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total; idx += step) {
    const size_t j = idx / n, i = idx % n;  // compact element (j, i): k-block j of column i
    const size_t s = swizzled_idx(i, j, kblocks / 4);
    if constexpr (FORWARD) out[s] = in[idx]; else out[idx] = in[s];
  }
}

void swizzle_col(const Buf &in, Buf &out, bool forward, hipStream_t stream) {
  NVTE_CHECK(in.ptr != nullptr && out.ptr != nullptr, "columnwise scale buffers must be allocated");
  NVTE_CHECK(in.shape.ndim == 2, "columnwise scale_inv must be 2-D, got ", in.shape);
  const size_t kb = in.shape.data[0], n = in.shape.data[1];
  NVTE_CHECK(kb % 4 == 0 && n % 128 == 0, "columnwise scale_inv shape must be padded to (4, 128), got ",
             in.shape);
  NVTE_CHECK(out.numel() == kb * n, "output columnwise scale_inv has wrong size");
  if (kb * n == 0) return;
  const unsigned grid = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(ceil_div(kb * n, 256), 4096)));
  if (forward) {
    hipLaunchKernelGGL(swizzle_col_scales_kernel<true>, dim3(grid), dim3(256), 0, stream,
                       static_cast<const uint8_t *>(in.ptr), static_cast<uint8_t *>(out.ptr), kb, n);
  } else {
    hipLaunchKernelGGL(swizzle_col_scales_kernel<false>, dim3(grid), dim3(256), 0, stream,
                       static_cast<const uint8_t *>(in.ptr), static_cast<uint8_t *>(out.ptr), kb, n);
  }
  NVTE_CHECK_HIP(hipGetLastError());
}

void swizzle_tensor(const Tensor &in, Tensor &out, bool forward, hipStream_t stream) {
  NVTE_CHECK(is_mxfp8(in.scaling_mode) && is_mxfp8(out.scaling_mode),
             "Scale swizzle is implemented for NVTE_MXFP8_1D_SCALING tensors only, got ",
             scaling_mode_name(in.scaling_mode));
  NVTE_CHECK(in.with_gemm_swizzled_scales != forward,
             forward ? "Input scales are already in GEMM layout" : "Input scales are not in GEMM layout");
  NVTE_CHECK(out.with_gemm_swizzled_scales == forward,
             "Output tensor's with_gemm_swizzled_scales flag does not match the requested layout");
  const bool has_r = in.scale_inv.present(), has_c = in.columnwise_scale_inv.present();
  NVTE_CHECK(has_r || has_c, "Input tensor has no scaling factors");
  if (has_r) swizzle_one(in.scale_inv, out.scale_inv, forward, "rowwise", stream);
  if (has_c) {
    NVTE_CHECK(in.columnwise_transposed == out.columnwise_transposed, "input and output column-wise layouts differ");
    // transposed columnwise layout (amd_ext.h): the scale matrix [pad128(cols), pad4(rows / 32)] is already
    // row-per-GEMM-row, i.e. swizzles like a row-wise one
    if (in.columnwise_transposed)
      swizzle_one(in.columnwise_scale_inv, out.columnwise_scale_inv, forward, "columnwise (transposed)", stream);
    else
      swizzle_col(in.columnwise_scale_inv, out.columnwise_scale_inv, forward, stream);
  }
}

// ref swizzle/swizzle.cu:2467-2481 (swizzle_grouped_scaling_factors): every member of a SAME-SHAPE MXFP8
// grouped tensor, scale matrices padded to (128, 4) / (4, 128) and stored back to back.  128-row scale tiles
// never cross members, so the row-wise matrices of all members swizzle as one stacked matrix; the
// column-wise ones member by member (reference layout: transposing copy; transposed layout of amd_ext.h:
// a row-type matrix per member).
Buf scale_view(const Buf &b, size_t byte_off, size_t rows, size_t cols) {
  Buf v = b;
  v.ptr = static_cast<uint8_t *>(b.ptr) + byte_off;
  v.shape = NVTEShape{{rows, cols}, 2};
  return v;
}

void swizzle_grouped(const GroupedTensor &in, GroupedTensor &out, bool forward, hipStream_t stream) {
  NVTE_CHECK(is_mxfp8(in.scaling_mode) && is_mxfp8(out.scaling_mode),
             "Grouped scale swizzle is implemented for NVTE_MXFP8_1D_SCALING tensors only, got ",
             scaling_mode_name(in.scaling_mode));
  NVTE_CHECK(in.all_same_shape() && out.all_same_shape(), "all tensors in the grouped tensor must have the same shape");
  NVTE_CHECK(in.num_tensors == out.num_tensors && in.logical_shape.data[0] == out.logical_shape.data[0] &&
                 in.logical_shape.data[1] == out.logical_shape.data[1],
             "input and output grouped tensors differ in shape");
  NVTE_CHECK(in.with_gemm_swizzled_scales != forward,
             forward ? "Input scales are already in GEMM layout" : "Input scales are not in GEMM layout");
  NVTE_CHECK(out.with_gemm_swizzled_scales == forward,
             "Output grouped tensor's with_gemm_swizzled_scales flag does not match the requested layout");
  const size_t T = in.num_tensors, R = in.logical_shape.data[0], C = in.logical_shape.data[1];
  NVTE_CHECK(R % T == 0, "First logical dimension must be divisible by num_tensors.");
  const size_t first = R / T;
  const bool has_r = in.scale_inv.present(), has_c = in.columnwise_scale_inv.present();
  NVTE_CHECK(has_r || has_c, "Input grouped tensor has no scaling factors");
  if (has_r) {
    const size_t rows = round_up(first, 128), cols = round_up(ceil_div(C, 32), 4);
    NVTE_CHECK(in.scale_inv.numel() >= T * rows * cols && out.scale_inv.numel() >= T * rows * cols,
               "rowwise scale_inv must hold ", T, " matrices padded to (128, 4): ", T * rows * cols, " bytes");
    Buf o = scale_view(out.scale_inv, 0, T * rows, cols);
    swizzle_one(scale_view(in.scale_inv, 0, T * rows, cols), o, forward, "rowwise", stream);
  }
  if (has_c) {
    NVTE_CHECK(in.columnwise_transposed == out.columnwise_transposed, "input and output column-wise layouts differ");
    const size_t kb = round_up(ceil_div(first, 32), 4), n = round_up(C, 128), per = kb * n;
    NVTE_CHECK(in.columnwise_scale_inv.numel() >= T * per && out.columnwise_scale_inv.numel() >= T * per,
               "columnwise scale_inv must hold ", T, " matrices padded to (4, 128): ", T * per, " bytes");
    for (size_t i = 0; i < T; ++i) {
      if (in.columnwise_transposed) {
        Buf o = scale_view(out.columnwise_scale_inv, i * per, n, kb);
        swizzle_one(scale_view(in.columnwise_scale_inv, i * per, n, kb), o, forward, "columnwise (transposed)", stream);
      } else {
        Buf o = scale_view(out.columnwise_scale_inv, i * per, kb, n);
        swizzle_col(scale_view(in.columnwise_scale_inv, i * per, kb, n), o, forward, stream);
      }
    }
  }
}

}  // namespace

void mxfp8_quantize_launch(const Tensor &in, const float *noop, Tensor &out, hipStream_t stream) {
  NVTE_CHECK(in.has_data(), "Cannot quantize tensor without rowwise data.");
  NVTE_CHECK(is_fp8(out.dtype()), "Output must have FP8 type.");
  NVTE_CHECK(!is_fp8(in.dtype()), "Input must be in higher precision.");
  check_input(in, "cast_input");
  check_output(out, "cast_output");
  const bool row = out.has_data(), col = out.has_columnwise_data();
  if (row) {
    NVTE_CHECK(out.scale_inv.ptr != nullptr, "Scaling tensor must be allocated");
    NVTE_CHECK(same_shape(out.data.shape, in.data.shape), "Input and output shapes need to match.");
  }
  if (col) {
    NVTE_CHECK(out.columnwise_scale_inv.ptr != nullptr, "Columnwise scaling tensor must be allocated");
    if (out.columnwise_transposed) {
      const auto [r, c] = in.flat2d();
      NVTE_CHECK(out.columnwise_data.numel() == r * c && out.columnwise_data.shape.ndim >= 1 &&
                     out.columnwise_data.shape.data[0] == c,
                 "Transposed columnwise output must have shape [", c, ", ", r, "], got ",
                 out.columnwise_data.shape);
    } else {
      NVTE_CHECK(same_shape(out.columnwise_data.shape, in.data.shape),
                 "Input and columnwise output shapes need to match.");
    }
  }
  if (in.data.numel() == 0) return;
  NVTE_REQUIRE_DATA(in.data, "MXFP8 quantize input");
  if (row) NVTE_REQUIRE_DATA(out.data, "MXFP8 row-wise output");
  if (col) NVTE_REQUIRE_DATA(out.columnwise_data, "MXFP8 column-wise output");
  const bool e4 = out.dtype() == kNVTEFloat8E4M3;
  switch (in.data.dtype) {
    case kNVTEBFloat16:
      e4 ? launch_quant<bf16_t, Fp8::E4M3>(in, noop, out, stream)
         : launch_quant<bf16_t, Fp8::E5M2>(in, noop, out, stream);
      break;
    case kNVTEFloat16:
      e4 ? launch_quant<f16_t, Fp8::E4M3>(in, noop, out, stream)
         : launch_quant<f16_t, Fp8::E5M2>(in, noop, out, stream);
      break;
    case kNVTEFloat32:
      e4 ? launch_quant<float, Fp8::E4M3>(in, noop, out, stream)
         : launch_quant<float, Fp8::E5M2>(in, noop, out, stream);
      break;
    default:
      NVTE_ERROR("Unsupported input dtype ", dtype_name(in.data.dtype));
  }
}

void mxfp8_dequantize_launch(const Tensor &in, Tensor &out, hipStream_t stream) {
  NVTE_CHECK(in.shape().ndim >= 2, "Input must have at least 2 dimensions.");
  NVTE_CHECK(in.has_data() || in.has_columnwise_data(), "Cannot dequantize tensor without data.");
  NVTE_CHECK(is_fp8(in.dtype()), "Input must have FP8 type.");
  NVTE_CHECK(!is_fp8(out.data.dtype), "Output must be in higher precision.");
  NVTE_CHECK(same_shape(out.shape(), in.shape()), "Input and output shapes need to match.");
  if (in.numel() == 0) return;
  const bool e4 = in.dtype() == kNVTEFloat8E4M3;
  switch (out.data.dtype) {
    case kNVTEFloat32:
      e4 ? launch_dequant<Fp8::E4M3, float>(in, out, stream) : launch_dequant<Fp8::E5M2, float>(in, out, stream);
      break;
    case kNVTEFloat16:
      e4 ? launch_dequant<Fp8::E4M3, f16_t>(in, out, stream) : launch_dequant<Fp8::E5M2, f16_t>(in, out, stream);
      break;
    case kNVTEBFloat16:
      e4 ? launch_dequant<Fp8::E4M3, bf16_t>(in, out, stream) : launch_dequant<Fp8::E5M2, bf16_t>(in, out, stream);
      break;
    default:
      NVTE_ERROR("Unsupported output dtype ", dtype_name(out.data.dtype));
  }
}

}  // namespace te

using namespace te;

extern "C" {

void nvte_swizzle_scaling_factors(const NVTETensor input, NVTETensor output, hipStream_t stream) {
  api_guard("nvte_swizzle_scaling_factors", [&] {
    swizzle_tensor(*lookup_checked(input), *lookup_checked(output), true, stream);
  });
}

void nvte_multi_tensor_swizzle_scaling_factors(const NVTETensor *inputs, NVTETensor *outputs,
                                               const size_t num_tensors, hipStream_t stream) {
  api_guard("nvte_multi_tensor_swizzle_scaling_factors", [&] {
    for (size_t i = 0; i < num_tensors; ++i)
      swizzle_tensor(*lookup_checked(inputs[i]), *lookup_checked(outputs[i]), true, stream);
  });
}

void nvte_unswizzle_scaling_factors(const NVTETensor input, NVTETensor output, hipStream_t stream) {
  api_guard("nvte_unswizzle_scaling_factors", [&] {
    swizzle_tensor(*lookup_checked(input), *lookup_checked(output), false, stream);
  });
}

void nvte_multi_tensor_unswizzle_scaling_factors(const NVTETensor *inputs, NVTETensor *outputs,
                                                 const size_t num_tensors, hipStream_t stream) {
  api_guard("nvte_multi_tensor_unswizzle_scaling_factors", [&] {
    for (size_t i = 0; i < num_tensors; ++i)
      swizzle_tensor(*lookup_checked(inputs[i]), *lookup_checked(outputs[i]), false, stream);
  });
}

void nvte_swizzle_grouped_scaling_factors(const NVTEGroupedTensor input, NVTEGroupedTensor output, hipStream_t stream) {
  api_guard("nvte_swizzle_grouped_scaling_factors", [&] {
    swizzle_grouped(*lookup_grouped_checked(input), *lookup_grouped_checked(output), true, stream);
  });
}

void nvte_unswizzle_grouped_scaling_factors(const NVTEGroupedTensor input, NVTEGroupedTensor output, hipStream_t stream) {
  api_guard("nvte_unswizzle_grouped_scaling_factors", [&] {
    swizzle_grouped(*lookup_grouped_checked(input), *lookup_grouped_checked(output), false, stream);
  });
}

}  // extern "C"
