// Per-tensor-scaled quantize kernels: cast + transpose (+amax, +scale_inv) and the
// flat rowwise cast.  HBM-bound: 2 B read + 1 B (+1 B) written per bf16 element.
//
// Replaces (behaviour, not structure):
//   ref common/transpose/cast_transpose.cu:104-216 (general kernel), :220-344 (launcher),
//       common/transpose/rtc/cast_transpose.cu:24-129 (tuned aligned kernel),
//       common/util/vectorized_pointwise.h:176-227 (rowwise cast),
//       common/cast/fp8/quantize_fp8.cuh:510-573 (rowwise dispatch).
//
// MI355X design of the aligned path: one 256-thread workgroup owns a 128x128 element
// tile.  Thread (cg = t%16, rg = t/16) loads an 8-row x 8-column block with eight
// 16-byte loads (a wave instruction covers four 256-byte row segments), converts with
// v_cvt_pk_{fp8,bf8}_f32, stores the rowwise bytes straight from registers (8 B per
// lane, 128 B contiguous per 16 lanes), transposes its 8x8 byte block in registers
// with v_perm_b32 and writes eight 8-byte column pieces into a 16 KiB XOR-swizzled LDS
// image of the transposed tile.  After one barrier every thread reads 16 bytes per
// pass and stores them with 16-byte stores: a wave instruction writes eight full
// 128-byte lines of the transposed output.  Both LDS phases are bank-conflict free
// (see DESIGN.md for the bank arithmetic).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "common.h"
#include "device_utils.h"

namespace te {
namespace {

constexpr int kTile = 128;      // tile edge, elements
constexpr int kCtThreads = 256;

// ---------------------------------------------------------------- 128x128 tile kernel, persistent
// Requires cols % 8 == 0, rows % 16 == 0 and 16-byte aligned base pointers (every 16-byte load, 8-byte
// rowwise store and 16-byte transposed store is then aligned); GUARD = false additionally requires whole
// tiles (rows % 128 == 0, cols % 128 == 0) and drops every bounds check.
// Persistent: the grid is a few workgroups per CU and workgroup w walks tiles w, w + grid, ...  The eight
// 16-byte loads of the NEXT tile are issued before the current tile is converted, stored and pushed
// through LDS, so a workgroup always has a tile's worth of loads in flight (a one-tile workgroup spends
// its life waiting: load -> convert -> barrier -> store), the LDS image is double-buffered (one barrier
// per tile) and amax costs one atomic per workgroup instead of one per tile.
// DBIAS: additionally leaves the tile's column sums (over its 128 rows, of the UNSCALED input) in
// dbias_part[tile_r][col] - the [ceil(R/128), C] fp32 workspace of nvte_quantize_dbias, reduced by
// dbias.hip afterwards: cast, transpose and bias gradient from ONE read of dY.
template <typename IT, Fp8 F, bool DBIAS = false, bool GUARD = false>
__global__ __launch_bounds__(kCtThreads) void cast_transpose_tile_kernel(
    const IT *__restrict__ in, const float *__restrict__ noop, uint8_t *__restrict__ out_c,
    uint8_t *__restrict__ out_t, const float *__restrict__ scale_ptr, float *__restrict__ amax_ptr,
    float *__restrict__ scale_inv_ptr, const size_t rows, const size_t cols, const unsigned tiles_c,
    const unsigned ntiles, float *__restrict__ dbias_part = nullptr) {
  if (noop_set(noop)) return;

  __shared__ __attribute__((aligned(16))) uint8_t tile_t[2][kTile * kTile];
  __shared__ float red[kCtThreads / kWave];
  __shared__ float colpart[DBIAS ? 2 : 1][DBIAS ? 16 : 1][kTile];

  const unsigned t = threadIdx.x;
  const unsigned cg = t & 15u;   // 8-column group
  const unsigned rg = t >> 4;    // 8-row group
  const float scale = scale_ptr ? *scale_ptr : 1.0f;
  float amax = 0.f;

  auto load_tile = [&](unsigned tile, Raw8<IT> (&raw)[8]) {
    const size_t row0 = static_cast<size_t>(tile / tiles_c) * kTile + rg * 8;
    const size_t col0 = static_cast<size_t>(tile % tiles_c) * kTile + cg * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (GUARD) {
        // out-of-range lanes load a valid (clamped) address and the value is dropped at use: a bounds branch
        // in front of the loads keeps them from being issued together
        const size_t r = row0 + i < rows ? row0 + i : rows - 1, c = col0 < cols ? col0 : 0;
        raw[i] = Raw8<IT>::load(in + r * cols + c);
      } else {
        raw[i] = Raw8<IT>::load(in + (row0 + i) * cols + col0);
      }
    }
  };

  Raw8<IT> cur[8], nxt[8];
  unsigned tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile, cur);
  for (unsigned it = 0; tile < ntiles; tile += gridDim.x, ++it) {
    const unsigned next = tile + gridDim.x;
    if (next < ntiles) load_tile(next, nxt);   // in flight while this tile is processed

    const size_t tile_r = tile / tiles_c, tile_col = tile % tiles_c;
    const size_t row0 = tile_r * kTile + rg * 8;
    const size_t col0 = tile_col * kTile + cg * 8;
    const unsigned buf = it & 1u;

    // q[i][h]: row i of this thread's block, columns 4h..4h+3
    uint32_t q[8][2];
    float csum[DBIAS ? 8 : 1];
    if constexpr (DBIAS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) csum[j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float f[8];
      cur[i].widen8(f);
      if constexpr (GUARD) {
        if (!(col0 < cols && row0 + i < rows)) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = 0.f;
        }
      }
      const u32x2 w = quantize8<F>(f, scale, amax);   // amax of the unscaled values, sat(f * scale)
      q[i][0] = w[0];
      q[i][1] = w[1];
      if constexpr (DBIAS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) csum[j] += f[j];
      }
      // rowwise output straight from registers
      if (!GUARD || (col0 < cols && row0 + i < rows)) {
        u32x2 v = {q[i][0], q[i][1]};
        *reinterpret_cast<u32x2 *>(out_c + (row0 + i) * cols + col0) = v;
      }
    }
    if constexpr (DBIAS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) colpart[buf][rg][cg * 8 + j] = csum[j];
    }

    // transposed output through the LDS image (device_utils.h); holds this tile's only barrier
    if constexpr (GUARD) {
      const size_t rows_left = rows - tile_r * kTile, cols_left = cols - tile_col * kTile;
      store_tile_transposed<true>(tile_t[buf], q, t, out_t + (tile_col * kTile) * rows + tile_r * kTile, rows,
                                  static_cast<unsigned>(rows_left >= kTile ? 8 : rows_left / 16),
                                  static_cast<unsigned>(cols_left >= kTile ? kTile : cols_left));
    } else {
      store_tile_transposed(tile_t[buf], q, t, out_t + (tile_col * kTile) * rows + tile_r * kTile, rows);
    }

    if constexpr (DBIAS) {
      // colpart[buf] was written before the barrier inside store_tile_transposed
      if (t < kTile && (!GUARD || tile_col * kTile + t < cols)) {
        float c = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) c += colpart[buf][r][t];
        dbias_part[tile_r * cols + tile_col * kTile + t] = c;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
  }

  if (amax_ptr != nullptr) {
    amax = block_max<kCtThreads / kWave>(amax, red);
    if (t == 0) atomic_max_nonneg(amax_ptr, amax);
  }
  if (blockIdx.x == 0 && t == 0 && scale_inv_ptr != nullptr) *scale_inv_ptr = 1.0f / scale;
}

// ---------------------------------------------------------------- general kernel (any shape, any dtype pair)
template <typename OT>
struct OutCvt;
template <>
struct OutCvt<float> {
  static __device__ __forceinline__ float run(float x) { return x; }
};
template <>
struct OutCvt<bf16_t> {
  static __device__ __forceinline__ bf16_t run(float x) {
    bf16_t o;
    narrow(x, o);
    return o;
  }
};
template <>
struct OutCvt<f16_t> {
  static __device__ __forceinline__ f16_t run(float x) {
    f16_t o;
    narrow(x, o);
    return o;
  }
};
struct e4m3_t {
  uint8_t b;
};
struct e5m2_t {
  uint8_t b;
};
template <>
struct OutCvt<e4m3_t> {
  static __device__ __forceinline__ e4m3_t run(float x) { return e4m3_t{cvt1_fp8<Fp8::E4M3>(x)}; }
};
template <>
struct OutCvt<e5m2_t> {
  static __device__ __forceinline__ e5m2_t run(float x) { return e5m2_t{cvt1_fp8<Fp8::E5M2>(x)}; }
};

constexpr int kGTile = 64;

template <typename IT, typename OT, bool WITH_T>
__global__ __launch_bounds__(256) void cast_transpose_general_kernel(
    const IT *__restrict__ in, const float *__restrict__ noop, OT *__restrict__ out_c,
    OT *__restrict__ out_t, const float *__restrict__ scale_ptr, float *__restrict__ amax_ptr,
    float *__restrict__ scale_inv_ptr, const size_t rows, const size_t cols, const unsigned tiles_c) {
  if (noop_set(noop)) return;
  __shared__ OT tile[WITH_T ? kGTile : 1][kGTile + 1];
  __shared__ float red[4];

  const unsigned t = threadIdx.x;
  const unsigned lx = t & 63u, ly = t >> 6;
  const size_t r_base = static_cast<size_t>(blockIdx.x / tiles_c) * kGTile;
  const size_t c_base = static_cast<size_t>(blockIdx.x % tiles_c) * kGTile;
  const float scale = scale_ptr ? *scale_ptr : 1.0f;
  float amax = 0.f;

#pragma unroll 4
  for (int i = 0; i < kGTile / 4; ++i) {
    const size_t r = r_base + ly + 4 * i, c = c_base + lx;
    if (r < rows && c < cols) {
      const float x = widen(in[r * cols + c]);
      amax = fmaxf(amax, __builtin_fabsf(x));
      const OT o = OutCvt<OT>::run(x * scale);
      if (out_c != nullptr) out_c[r * cols + c] = o;
      if constexpr (WITH_T) tile[lx][ly + 4 * i] = o;
    }
  }
  if constexpr (WITH_T) {
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < kGTile / 4; ++i) {
      const size_t c = c_base + ly + 4 * i, r = r_base + lx;
      if (r < rows && c < cols) out_t[c * rows + r] = tile[ly + 4 * i][lx];
    }
  }
  if (amax_ptr != nullptr) {
    amax = block_max<4>(amax, red);
    if (t == 0) atomic_max_nonneg(amax_ptr, amax);
  }
  if (blockIdx.x == 0 && t == 0 && scale_inv_ptr != nullptr) *scale_inv_ptr = 1.0f / scale;
}

// ---------------------------------------------------------------- flat rowwise cast (aligned, FP8 out)
// 3 B/element (bf16 in).  The tensor is a list of 8-element chunks (16 bytes in, 8 bytes out for bf16); a workgroup
// takes 4 x 256 consecutive chunks per step, thread t chunks t, t + 256, t + 512, t + 768 of them: every load
// instruction of a wave covers 1 KiB of consecutive input, every store 512 bytes of consecutive output.  As in the
// tile kernel the loads of the NEXT step are issued before this step is converted and stored, so a workgroup keeps
// 32 KiB of loads in flight.  N % 8 == 0, 16-byte aligned input, 8-byte aligned output.
constexpr int kFlatUnroll = 4;
template <typename IT, Fp8 F>
__global__ __launch_bounds__(256) void cast_flat_kernel(
    const IT *__restrict__ in, const float *__restrict__ noop, uint8_t *__restrict__ out,
    const float *__restrict__ scale_ptr, float *__restrict__ amax_ptr,
    float *__restrict__ scale_inv_ptr, const size_t n8) {
  if (noop_set(noop)) return;
  __shared__ float red[4];
  const float scale = scale_ptr ? *scale_ptr : 1.0f;
  float amax = 0.f;
  constexpr size_t kStep = 256 * kFlatUnroll;   // chunks per workgroup step
  const size_t stride = static_cast<size_t>(gridDim.x) * kStep;
  auto load_step = [&](size_t base, Raw8<IT> (&raw)[kFlatUnroll]) {
#pragma unroll
    for (int u = 0; u < kFlatUnroll; ++u) {
      const size_t c = base + u * 256 + threadIdx.x;
      raw[u] = Raw8<IT>::load(in + (c < n8 ? c : n8 - 1) * 8);   // clamped, the value is dropped at use
    }
  };
  Raw8<IT> cur[kFlatUnroll], nxt[kFlatUnroll];
  size_t base = static_cast<size_t>(blockIdx.x) * kStep;
  if (base < n8) load_step(base, cur);
  for (; base < n8; base += stride) {
    if (base + stride < n8) load_step(base + stride, nxt);
#pragma unroll
    for (int u = 0; u < kFlatUnroll; ++u) {
      const size_t c = base + u * 256 + threadIdx.x;
      float f[8];
      cur[u].widen8(f);
      if (c >= n8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = 0.f;
      }
      const u32x2 w = quantize8<F>(f, scale, amax);   // amax of the unscaled values, sat(f * scale)
      if (c < n8) *reinterpret_cast<u32x2 *>(out + c * 8) = w;
    }
#pragma unroll
    for (int u = 0; u < kFlatUnroll; ++u) cur[u] = nxt[u];
  }
  if (amax_ptr != nullptr) {
    amax = block_max<4>(amax, red);
    if (threadIdx.x == 0) atomic_max_nonneg(amax_ptr, amax);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && scale_inv_ptr != nullptr)
    *scale_inv_ptr = 1.0f / scale;
}

// Scalar flat cast for shapes / alignments / dtype pairs the vector kernel does not take.
template <typename IT, typename OT>
__global__ __launch_bounds__(256) void cast_flat_general_kernel(
    const IT *__restrict__ in, const float *__restrict__ noop, OT *__restrict__ out,
    const float *__restrict__ scale_ptr, float *__restrict__ amax_ptr,
    float *__restrict__ scale_inv_ptr, const size_t n) {
  if (noop_set(noop)) return;
  __shared__ float red[4];
  const float scale = scale_ptr ? *scale_ptr : 1.0f;
  float amax = 0.f;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = widen(in[i]);
    amax = fmaxf(amax, __builtin_fabsf(x));
    out[i] = OutCvt<OT>::run(x * scale);
  }
  if (amax_ptr != nullptr) {
    amax = block_max<4>(amax, red);
    if (threadIdx.x == 0) atomic_max_nonneg(amax_ptr, amax);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && scale_inv_ptr != nullptr) *scale_inv_ptr = 1.0f / scale;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename IT, typename OT>
void launch_general(const Tensor &in, const float *noop, Tensor &out, bool with_t, size_t rows,
                    size_t cols, hipStream_t stream) {
  const unsigned tiles_c = static_cast<unsigned>(ceil_div(cols, kGTile));
  const size_t nblk = ceil_div(rows, kGTile) * tiles_c;
  NVTE_CHECK(nblk < (1ull << 31), "tensor too large for the general cast kernel");
  const IT *ip = static_cast<const IT *>(in.data.ptr);
  OT *oc = static_cast<OT *>(out.data.ptr);
  OT *ot = static_cast<OT *>(out.columnwise_data.ptr);
  const float *sp = static_cast<const float *>(out.scale.ptr);
  float *ap = static_cast<float *>(out.amax.ptr);
  float *sip = static_cast<float *>(out.scale_inv.ptr);
  if (with_t) {
    hipLaunchKernelGGL((cast_transpose_general_kernel<IT, OT, true>), dim3(nblk), dim3(256), 0,
                       stream, ip, noop, oc, ot, sp, ap, sip, rows, cols, tiles_c);
  } else {
    hipLaunchKernelGGL((cast_transpose_general_kernel<IT, OT, false>), dim3(nblk), dim3(256), 0,
                       stream, ip, noop, oc, ot, sp, ap, sip, rows, cols, tiles_c);
  }
  NVTE_CHECK_HIP(hipGetLastError());
}

template <typename IT>
void launch_general_by_out(const Tensor &in, const float *noop, Tensor &out, bool with_t,
                           size_t rows, size_t cols, hipStream_t stream) {
  switch (out.data.dtype) {
    case kNVTEFloat8E4M3: launch_general<IT, e4m3_t>(in, noop, out, with_t, rows, cols, stream); break;
    case kNVTEFloat8E5M2: launch_general<IT, e5m2_t>(in, noop, out, with_t, rows, cols, stream); break;
    case kNVTEFloat32: launch_general<IT, float>(in, noop, out, with_t, rows, cols, stream); break;
    case kNVTEFloat16: launch_general<IT, f16_t>(in, noop, out, with_t, rows, cols, stream); break;
    case kNVTEBFloat16: launch_general<IT, bf16_t>(in, noop, out, with_t, rows, cols, stream); break;
    default: NVTE_ERROR("Unsupported output dtype ", dtype_name(out.data.dtype));
  }
}

// Persistent grid for `items` equal work items on at most `max_wgs` workgroups: every workgroup gets the same number of
// items (+-1).  X [8192, 4096] is 2048 tiles: on 768 workgroups a third of them would walk 3 tiles and the rest 2 - the
// kernel then takes 3 tile times at 89 % occupancy - on 683 every workgroup walks 3.
unsigned balanced_grid(size_t items, size_t max_wgs) {
  if (items <= max_wgs) return static_cast<unsigned>(items);
  const size_t rounds = ceil_div(items, max_wgs);
  return static_cast<unsigned>(ceil_div(items, rounds));
}

// Workgroups per CU of the persistent tile kernel (LDS: 32 KiB each; more resident workgroups = more
// loads in flight, fewer = longer-lived workgroups).  NVTE_AMD_CT_WGS_PER_CU overrides (tuning aid).
// 3, not the 4 that fit: inside the configs[1] step X [8192, 4096] (2048 tiles) takes 28 us at 2 or 3 and
// 35 us at 4, where every workgroup owns exactly two tiles and all of them move through load / store in step;
// dY [8192, 14336] is the same 78 us from 2 to 6 (scratch/run_ct_instep.sh).
unsigned ct_grid(size_t ntiles) {
  static const unsigned per_cu = [] {
    const char *e = std::getenv("NVTE_AMD_CT_WGS_PER_CU");
    const int v = e ? std::atoi(e) : 0;
    return static_cast<unsigned>(v >= 1 && v <= 16 ? v : 3);
  }();
  return balanced_grid(ntiles, static_cast<size_t>(device_info().num_cus) * per_cu);
}

template <typename IT, Fp8 F, bool DBIAS>
void launch_tile(const Tensor &in, const float *noop, Tensor &out, size_t rows, size_t cols, float *part,
                 hipStream_t stream) {
  const unsigned tiles_c = static_cast<unsigned>(ceil_div(cols, kTile));
  const size_t ntiles = ceil_div(rows, kTile) * tiles_c;
  NVTE_CHECK(ntiles < (1ull << 31), "tensor too large for the tile cast kernel");
  const bool whole = rows % kTile == 0 && cols % kTile == 0;
  const dim3 grid(ct_grid(ntiles)), block(kCtThreads);
#define TE_CT_ARGS                                                                                        \
  static_cast<const IT *>(in.data.ptr), noop, static_cast<uint8_t *>(out.data.ptr),                      \
      static_cast<uint8_t *>(out.columnwise_data.ptr), static_cast<const float *>(out.scale.ptr),        \
      static_cast<float *>(out.amax.ptr), static_cast<float *>(out.scale_inv.ptr), rows, cols, tiles_c,  \
      static_cast<unsigned>(ntiles), part
  if (whole)
    hipLaunchKernelGGL((cast_transpose_tile_kernel<IT, F, DBIAS, false>), grid, block, 0, stream, TE_CT_ARGS);
  else
    hipLaunchKernelGGL((cast_transpose_tile_kernel<IT, F, DBIAS, true>), grid, block, 0, stream, TE_CT_ARGS);
#undef TE_CT_ARGS
  NVTE_CHECK_HIP(hipGetLastError());
}

// Shapes the vector tile kernel takes: every 16-byte load / 8-byte rowwise store / 16-byte transposed store aligned.
bool tile_path_ok(const Tensor &in, const Tensor &out, size_t rows, size_t cols) {
  return is_fp8(out.data.dtype) && cols % 8 == 0 && rows % 16 == 0 && aligned16(in.data.ptr) &&
         aligned16(out.data.ptr) && aligned16(out.columnwise_data.ptr);
}

// Workgroups per CU of the flat cast (NVTE_AMD_CAST_WGS_PER_CU overrides, tuning aid).  Measured on the configs[1]
// tensors, isolated (scratch/perf_cast_flat.py): dY [8192, 14336] 66 / 60 / 58 / 58 us and X [8192, 4096] 22.8 / 19.8 /
// 19.7 / 19.6 us at 2 / 4 / 6 / 8 workgroups per CU (the grid-stride kernel it replaces: 63-70 and 21.7 us).
unsigned flat_grid(size_t steps) {
  static const unsigned per_cu = [] {
    const char *e = std::getenv("NVTE_AMD_CAST_WGS_PER_CU");
    const int v = e ? std::atoi(e) : 0;
    return static_cast<unsigned>(v >= 1 && v <= 16 ? v : 6);
  }();
  return balanced_grid(steps, static_cast<size_t>(device_info().num_cus) * per_cu);
}

template <typename IT, Fp8 F>
void launch_flat(const Tensor &in, const float *noop, Tensor &out, size_t n, hipStream_t stream) {
  const size_t n16 = n / 8;   // 8-element chunks
  const unsigned grid = flat_grid(ceil_div(n16, static_cast<size_t>(256 * kFlatUnroll)));
  hipLaunchKernelGGL((cast_flat_kernel<IT, F>), dim3(grid), dim3(256), 0, stream,
                     static_cast<const IT *>(in.data.ptr), noop, static_cast<uint8_t *>(out.data.ptr),
                     static_cast<const float *>(out.scale.ptr), static_cast<float *>(out.amax.ptr),
                     static_cast<float *>(out.scale_inv.ptr), n16);
  NVTE_CHECK_HIP(hipGetLastError());
}

template <typename IT, typename OT>
void launch_flat_general(const Tensor &in, const float *noop, Tensor &out, size_t n,
                         hipStream_t stream) {
  const DeviceInfo &di = device_info();
  const unsigned grid = static_cast<unsigned>(
      std::min<size_t>(ceil_div(n, 256), static_cast<size_t>(di.num_cus) * 16));
  hipLaunchKernelGGL((cast_flat_general_kernel<IT, OT>), dim3(grid), dim3(256), 0, stream,
                     static_cast<const IT *>(in.data.ptr), noop, static_cast<OT *>(out.data.ptr),
                     static_cast<const float *>(out.scale.ptr), static_cast<float *>(out.amax.ptr),
                     static_cast<float *>(out.scale_inv.ptr), n);
  NVTE_CHECK_HIP(hipGetLastError());
}

template <typename IT>
void launch_flat_general_by_out(const Tensor &in, const float *noop, Tensor &out, size_t n,
                                hipStream_t stream) {
  switch (out.data.dtype) {
    case kNVTEFloat8E4M3: launch_flat_general<IT, e4m3_t>(in, noop, out, n, stream); break;
    case kNVTEFloat8E5M2: launch_flat_general<IT, e5m2_t>(in, noop, out, n, stream); break;
    case kNVTEFloat32: launch_flat_general<IT, float>(in, noop, out, n, stream); break;
    case kNVTEFloat16: launch_flat_general<IT, f16_t>(in, noop, out, n, stream); break;
    case kNVTEBFloat16: launch_flat_general<IT, bf16_t>(in, noop, out, n, stream); break;
    default: NVTE_ERROR("Unsupported output dtype ", dtype_name(out.data.dtype));
  }
}

#define TE_SWITCH_IN(dtype, IT, ...)                                        \
  switch (dtype) {                                                          \
    case kNVTEFloat32: { using IT = float; __VA_ARGS__; } break;            \
    case kNVTEFloat16: { using IT = f16_t; __VA_ARGS__; } break;            \
    case kNVTEBFloat16: { using IT = bf16_t; __VA_ARGS__; } break;          \
    default: NVTE_ERROR("Unsupported input dtype ", dtype_name(dtype));     \
  }

}  // namespace

// cast + transpose with shared scale / amax / scale_inv  (ref cast_transpose.cu:220-344)
void cast_transpose_launch(const Tensor &in, const float *noop, Tensor &out, hipStream_t stream) {
  check_input(in, "cast_transpose_input");
  check_output(out, "cast_transpose_output");
  NVTE_CHECK(in.has_data(), "Input is not allocated");
  NVTE_CHECK(out.has_data(), "Output rowwise data is not allocated");
  NVTE_CHECK(out.has_columnwise_data(), "Output columnwise is not allocated");
  NVTE_CHECK(same_shape(in.data.shape, out.data.shape),
             "Input and output shapes do not match (input=", in.data.shape,
             ", output=", out.data.shape, ")");
  NVTE_CHECK(!is_fp8(in.dtype()), "Input must be in higher precision.");
  NVTE_CHECK(is_tensor_scaling(out.scaling_mode), "Not implemented scaling mode: ",
             scaling_mode_name(out.scaling_mode));
  const size_t rows = in.rows(), cols = in.cols();
  NVTE_CHECK(out.columnwise_data.numel() == rows * cols, "Invalid columnwise output size (expected ",
             rows * cols, " elements, got ", out.columnwise_data.numel(), ")");
  NVTE_CHECK(out.data.dtype == out.columnwise_data.dtype,
             "Cast and transposed output types must match.");
  NVTE_CHECK(out.scale_inv.ptr == out.columnwise_scale_inv.ptr,
             "Cast and transposed outputs need to share scale-inverse tensor.");
  if (rows == 0 || cols == 0) return;
  NVTE_REQUIRE_DATA(in.data, "cast_transpose input");
  NVTE_REQUIRE_DATA(out.data, "cast_transpose output");
  NVTE_REQUIRE_DATA(out.columnwise_data, "cast_transpose transposed output");

  const bool fast = tile_path_ok(in, out, rows, cols);
  TE_SWITCH_IN(in.data.dtype, IT, {
    if (fast) {
      if (out.data.dtype == kNVTEFloat8E4M3)
        launch_tile<IT, Fp8::E4M3, false>(in, noop, out, rows, cols, nullptr, stream);
      else
        launch_tile<IT, Fp8::E5M2, false>(in, noop, out, rows, cols, nullptr, stream);
    } else {
      launch_general_by_out<IT>(in, noop, out, true, rows, cols, stream);
    }
  });
}

// Tile-aligned cast + transpose + per-slab column sums; returns false when the shape / alignment needs
// the general fused kernel of activation.hip instead.  `part` = [rows/128, cols] fp32.
bool cast_transpose_dbias_tile_launch(const Tensor &in, Tensor &out, float *part, hipStream_t stream) {
  const size_t rows = in.rows(), cols = in.cols();
  if (!(out.has_data() && out.has_columnwise_data() && rows > 0 && cols > 0 && tile_path_ok(in, out, rows, cols)))
    return false;
  if (ceil_div(rows, kTile) * ceil_div(cols, kTile) >= (1ull << 31)) return false;
  TE_SWITCH_IN(in.data.dtype, IT, {
    if (out.data.dtype == kNVTEFloat8E4M3)
      launch_tile<IT, Fp8::E4M3, true>(in, nullptr, out, rows, cols, part, stream);
    else
      launch_tile<IT, Fp8::E5M2, true>(in, nullptr, out, rows, cols, part, stream);
  });
  return true;
}

// rowwise-only cast (ref quantize_fp8.cuh:510-573 + vectorized_pointwise.h:176-227)
void cast_rowwise_launch(const Tensor &in, const float *noop, Tensor &out, hipStream_t stream) {
  check_input(in, "cast_input");
  check_output(out, "cast_output");
  NVTE_CHECK(!is_fp8(in.dtype()), "Input must be in higher precision.");
  NVTE_CHECK(in.has_data(), "Input is not allocated");
  NVTE_CHECK(same_shape(out.data.shape, in.data.shape), "Input and output shapes need to match.");
  NVTE_CHECK(is_tensor_scaling(out.scaling_mode), "Not implemented scaling mode: ",
             scaling_mode_name(out.scaling_mode));
  const size_t n = in.data.numel();
  if (n == 0) return;
  NVTE_REQUIRE_DATA(in.data, "cast input");
  NVTE_REQUIRE_DATA(out.data, "cast output");
  const bool fp8_out = is_fp8(out.data.dtype);
  const bool fast = fp8_out && n % 8 == 0 && aligned16(in.data.ptr) && aligned16(out.data.ptr);
  TE_SWITCH_IN(in.data.dtype, IT, {
    if (fast) {
      if (out.data.dtype == kNVTEFloat8E4M3)
        launch_flat<IT, Fp8::E4M3>(in, noop, out, n, stream);
      else
        launch_flat<IT, Fp8::E5M2>(in, noop, out, n, stream);
    } else {
      launch_flat_general_by_out<IT>(in, noop, out, n, stream);
    }
  });
}

}  // namespace te
