// Grouped (multi-expert) FP8 per-tensor-scaling quantize path with DEVICE-side member shapes:
// offsets helpers, per-member amax, per-member scale / scale_inv, grouped cast (+ per-member
// transpose), grouped dequantize and the grouped bias add.  Nothing here reads member shapes on
// the host, so every entry point is HIP-graph safe (ref: the m_splits.tolist() host sync of the
// legacy GroupedLinear path is what this removes, pytorch/module/grouped_linear.py:630).
//
// Replaces (behaviour, not structure):
//   ref common/util/splits_to_offsets.cu                      (prefix sums)
//       common/gemm/cublaslt_grouped_gemm.cu:2077-2129        (int32->int64, grouped offsets)
//       common/cast/fp8/group_amax_fp8.cuh:43-280             (grouped amax, grouped scale)
//       common/recipe/current_scaling.cu:178-262              (nvte_group_compute_amax_with_config)
//       common/cast/cast_grouped.cu:18-81                     (nvte_group_quantize / _scale_from_amax)
//       common/cast/fp8/group_quantize_fp8.cuh:984-1226       (FP8 grouped cast + transpose)
//       common/gemm/cublaslt_grouped_gemm.cu:1200-1316,1902-2015 (grouped bias add)
//
// HBM-bound like the ungrouped kernels (2 B read + 1 B per FP8 layout written per bf16 element).
// The grouped cast reuses the 128x128 register-transpose tile of cast_transpose.hip; a block first
// finds "which member, which tile" with a 256-thread prefix scan over the members' tile counts.
#include <hip/hip_runtime.h>

#include <limits>
#include <type_traits>

#include "common.h"
#include "device_utils.h"
#include "transformer_engine/cast.h"
#include "transformer_engine/gemm.h"
#include "transformer_engine/recipe.h"

#include "group_lookup.h"

namespace te {
namespace {

// ---------------------------------------------------------------- prefix sums (one wave)
// out[lead + i] = stride * sum_{j <= i} a[j] * (b ? b[j] : 1), out[0] = 0 when lead.
__device__ __forceinline__ int64_t ld_i(const void *p, int is64, size_t i) {
  return is64 ? static_cast<const int64_t *>(p)[i] : static_cast<int64_t>(static_cast<const int32_t *>(p)[i]);
}
__device__ __forceinline__ void st_i(void *p, int is64, size_t i, int64_t v) {
  if (is64) static_cast<int64_t *>(p)[i] = v; else static_cast<int32_t *>(p)[i] = static_cast<int32_t>(v);
}
__global__ __launch_bounds__(64) void scan_kernel(const void *a, int a64, const void *b, int b64, void *out,
                                                  int out64, size_t n, int64_t stride, int lead) {
  const unsigned lane = threadIdx.x;
  const size_t per = (n + 63) / 64, lo = lane * per, hi = lo + per < n ? lo + per : n;
  int64_t local = 0;
  for (size_t i = lo; i < hi; ++i) local += ld_i(a, a64, i) * (b ? ld_i(b, b64, i) : 1);
  int64_t incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t up = __shfl_up(incl, off, 64);
    if (lane >= static_cast<unsigned>(off)) incl += up;
  }
  int64_t run = incl - local;
  if (lane == 0 && lead) st_i(out, out64, 0, 0);
  for (size_t i = lo; i < hi; ++i) {
    run += ld_i(a, a64, i) * (b ? ld_i(b, b64, i) : 1);
    st_i(out, out64, i + (lead ? 1 : 0), run * stride);
  }
}

__global__ void convert_i32_i64_kernel(const int32_t *src, int64_t *dst, size_t n, int64_t mult) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = static_cast<int64_t>(src[i]) * mult;
}

int is64_of(const Tensor &t, const char *what) {
  NVTE_CHECK(t.data.dtype == kNVTEInt32 || t.data.dtype == kNVTEInt64, what, " must be int32 or int64, got ",
             dtype_name(t.data.dtype));
  return t.data.dtype == kNVTEInt64;
}

// ---------------------------------------------------------------- grouped amax
constexpr int kAmaxChunk = 256 * 8 * 32;  // elements per workgroup: the two member lookups (dependent
                                           // global loads) are paid once per 128 KiB of bf16 input

__global__ void group_zero_amax_kernel(float *amax, unsigned n, const float *noop) {
  if (noop_set(noop)) return;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) amax[i] = 0.f;
}

template <typename IT>
__global__ __launch_bounds__(256) void group_amax_kernel(const IT *__restrict__ in, float *__restrict__ amax,
                                                         const GroupShapes g, const float *__restrict__ noop,
                                                         const int vec_ok) {
  if (noop_set(noop)) return;
  __shared__ float red[4];
  const int64_t active = active_numel(g);
  const int64_t start = static_cast<int64_t>(blockIdx.x) * kAmaxChunk;
  if (start >= active) return;
  const int64_t end = start + kAmaxChunk < active ? start + kAmaxChunk : active;
  const unsigned t_lo = member_holding(g, start), t_hi = member_holding(g, end - 1);
  // one pass per member that owns a piece of this chunk (nearly always exactly one)
  for (unsigned t = t_lo; t <= t_hi; ++t) {
    const Member mem = member_of(g, t);
    const int64_t s0 = mem.off > start ? mem.off : start;
    const int64_t mem_end = mem.off + mem.rows * mem.cols;
    const int64_t s1 = mem_end < end ? mem_end : end;
    if (s0 >= s1) continue;  // empty member (block-uniform)
    float m = 0.f, m1 = 0.f;
    // scalar head up to the next 8-element boundary, vector body, scalar tail
    const int64_t body0 = vec_ok ? ((s0 + 7) & ~static_cast<int64_t>(7)) : s1;
    const int64_t head_end = body0 < s1 ? body0 : s1;
    for (int64_t e = s0 + threadIdx.x; e < head_end; e += 256) m = fmaxf(m, __builtin_fabsf(widen(in[e])));
    int64_t i = body0 + threadIdx.x * 8;
    for (; i + 3 * 256 * 8 + 8 <= s1; i += 4 * 256 * 8) {  // four independent vectors in flight
      float f[8], h[8], f2[8], h2[8];
      Load8<IT>::run(in + i, f);
      Load8<IT>::run(in + i + 256 * 8, h);
      Load8<IT>::run(in + i + 2 * 256 * 8, f2);
      Load8<IT>::run(in + i + 3 * 256 * 8, h2);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        m = fmaxf(m, fmaxf(__builtin_fabsf(f[k]), __builtin_fabsf(f2[k])));
        m1 = fmaxf(m1, fmaxf(__builtin_fabsf(h[k]), __builtin_fabsf(h2[k])));
      }
    }
    for (; i + 256 * 8 + 8 <= s1; i += 2 * 256 * 8) {  // two
      float f[8], h[8];
      Load8<IT>::run(in + i, f);
      Load8<IT>::run(in + i + 256 * 8, h);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        m = fmaxf(m, __builtin_fabsf(f[k]));
        m1 = fmaxf(m1, __builtin_fabsf(h[k]));
      }
    }
    for (; i + 8 <= s1; i += 256 * 8) {
      float f[8];
      Load8<IT>::run(in + i, f);
#pragma unroll
      for (int k = 0; k < 8; ++k) m = fmaxf(m, __builtin_fabsf(f[k]));
    }
    if (i < s1 && i >= body0)
      for (int64_t e = i; e < s1; ++e) m = fmaxf(m, __builtin_fabsf(widen(in[e])));
    m = block_max<4>(fmaxf(m, m1), red);
    if (threadIdx.x == 0) atomic_max_nonneg(amax + t, m);
    __syncthreads();  // `red` is reused by the next member
  }
}

// ref cast/fp8/group_amax_fp8.cuh:204-221
__global__ void group_scale_kernel(const float *amax, float *scale, float *scale_inv, unsigned n, float max_fp8,
                                   bool pow2, float eps, const float *noop) {
  if (noop_set(noop)) return;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float s = scale_from_amax(amax[i], max_fp8, pow2, eps, std::numeric_limits<float>::max());
    scale[i] = s;
    scale_inv[i] = 1.0f / s;
  }
}

// ---------------------------------------------------------------- grouped cast (+ per-member transpose)
constexpr int kTile = 128;

template <typename IT, Fp8 F>
__global__ __launch_bounds__(256) void group_cast_transpose_kernel(
    const IT *__restrict__ in, uint8_t *__restrict__ out_c, uint8_t *__restrict__ out_t,
    const float *__restrict__ scales, const float *__restrict__ noop, const GroupShapes g, const int base_aligned) {
  if (noop_set(noop)) return;
  __shared__ __attribute__((aligned(16))) uint8_t tile_t[kTile * kTile];
  unsigned mem, lt;
  Member m;
  if (!locate_tile(g, blockIdx.x, mem, lt, m)) return;
  const size_t rows = static_cast<size_t>(m.rows), cols = static_cast<size_t>(m.cols);
  const unsigned tiles_c = static_cast<unsigned>((cols + kTile - 1) / kTile);
  const size_t tile_r = lt / tiles_c, tile_col = lt % tiles_c;
  const float scale = scales[mem];
  const IT *src = in + m.off;
  uint8_t *dst_c = out_c ? out_c + m.off : nullptr;
  uint8_t *dst_t = out_t ? out_t + m.off : nullptr;

  const unsigned t = threadIdx.x, cg = t & 15u, rg = t >> 4;
  const size_t row0 = tile_r * kTile + rg * 8, col0 = tile_col * kTile + cg * 8;
  // vector path: 16-byte loads of 8 elements, 8-byte rowwise stores, 16-byte transposed stores
  const bool vec = base_aligned && (m.off % 16 == 0) && (cols % 8 == 0) && (dst_t == nullptr || rows % 16 == 0);
  if (vec) {
    uint32_t q[8][2];
    const bool col_ok = col0 < cols;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const bool ok = col_ok && row0 + i < rows;
      if (ok) Load8<IT>::run(src + (row0 + i) * cols + col0, f);
      float unused_amax = 0.f;   // the group's amaxes come from their own pass (current scaling)
      const u32x2 w8 = quantize8<F>(f, scale, unused_amax);
      q[i][0] = w8[0];
      q[i][1] = w8[1];
      if (ok && dst_c) {
        u32x2 v = {q[i][0], q[i][1]};
        *reinterpret_cast<u32x2 *>(dst_c + (row0 + i) * cols + col0) = v;
      }
    }
    if (dst_t) {
      const size_t rows_left = rows - tile_r * kTile, cols_left = cols - tile_col * kTile;
      store_tile_transposed<true>(tile_t, q, t, dst_t + (tile_col * kTile) * rows + tile_r * kTile, rows,
                                  static_cast<unsigned>(rows_left >= kTile ? 8 : rows_left / 16),
                                  static_cast<unsigned>(cols_left >= kTile ? kTile : cols_left));
    }
  } else {
    // any shape / alignment: one element at a time
    for (int i = 0; i < 8; ++i) {
      const size_t r = row0 + i;
      if (r >= rows) break;
      for (int j = 0; j < 8; ++j) {
        const size_t c = col0 + j;
        if (c >= cols) break;
        const uint8_t o = cvt1_fp8<F>(widen(src[r * cols + c]) * scale);
        if (dst_c) dst_c[r * cols + c] = o;
        if (dst_t) dst_t[c * rows + r] = o;
      }
    }
  }
}


// ---------------------------------------------------------------- grouped MXFP8 quantize
// ref common/cast/mxfp8/group_quantize_mxfp8.cuh (same-shape / varying-first-dim grouped tensors are
// quantized as ONE [sum first_i, last] tensor there, :478, :795; scales of member i therefore start at
// global row offsets[i] / last).  One workgroup per 128x128 tile of a member, the thread geometry of
// mxfp8_tile_kernel (mxfp8.hip): thread t owns rows 8*(t>>4).., columns 8*(t&15).. of the tile, a wave
// 32 rows = one column block.  Member rows are multiples of 128 by contract (amd_ext.h); rows of a
// member's last, partial 32-row block are skipped (nothing outside the member is read or written).
//   ROW  : payload [first_i, last] at the member offset, scales at the member's GLOBAL rows of the
//          [pad128(sum first), pad4(last/32)] matrix (compact or GEMM-swizzled)
//   COL  : reference layout - payload un-transposed, scales compact [sum first / 32, pad128(last)] (= the members'
//          [first_i / 32, pad128(last)] matrices back to back) or GEMM-swizzled per member like TCOL's
//   TCOL : per-member transposed payload [last, first_i], scales [pad128(last), first_i/32] per member at
//          pad128(last) * (member's first row / 32) (grouped columnwise_transposed, amd_ext.h)
__host__ __device__ __forceinline__ size_t mx_swizzled_idx(size_t i, size_t j, size_t tiles_x) {
  const size_t tile = (i / 128) * tiles_x + (j / 4);
  const size_t y = i % 128, x = j % 4;
  return tile * 512 + (y % 32) * 16 + (y / 32) * 4 + x;   // ref common/cast/mxfp8/swizzle.cuh:30-41
}

template <typename IT, Fp8 F, bool ROW, bool COL, bool TCOL, bool SWZ>
__global__ __launch_bounds__(256) void group_mxfp8_kernel(
    const IT *__restrict__ in, uint8_t *__restrict__ out_row, uint8_t *__restrict__ out_col,
    uint8_t *__restrict__ sc_row, uint8_t *__restrict__ sc_col, const float *__restrict__ noop,
    const GroupShapes g, const size_t total_rows, const size_t stride_r, const size_t pad_cols,
    const size_t sc_col_cap) {
  if (noop_set(noop)) return;
  __shared__ __attribute__((aligned(16))) uint8_t tile_t[TCOL ? kTile * kTile : 16];
  constexpr float rcp_max = 1.0f / Fp8Traits<F>::max_finite;
  unsigned mem, lt;
  Member m;
  if (!locate_tile(g, blockIdx.x, mem, lt, m)) return;
  const size_t rows = static_cast<size_t>(m.rows), cols = static_cast<size_t>(m.cols);
  const unsigned tiles_c = static_cast<unsigned>(cols / kTile);
  const size_t tile_r = lt / tiles_c, tile_col = lt % tiles_c;
  const size_t row_start = static_cast<size_t>(m.off) / cols;  // the member's first row in the group

  const unsigned t = threadIdx.x, cg = t & 15u, rg = t >> 4, w = t >> 6;
  const size_t row0 = tile_r * kTile + rg * 8, col0 = tile_col * kTile + cg * 8;
  const IT *src = in + m.off;

  // Two copies of the tile body: members that keep the 128-row contract (every valid call) run without the
  // per-wave row guards - with them the eight loads sit behind a branch and the kernel loses 9-17 % (245 vs
  // 204-216 us on [16384, 14336]; the single-tensor kernel: 203) - the guarded copy only keeps a member
  // that breaks the contract inside its own rows.  (The same split of group_cast_transpose_kernel into
  // whole and partial tiles was measured and gained nothing: 131 vs 127 us.)
  auto body = [&](auto guard_tag) {
  constexpr bool GUARD = decltype(guard_tag)::value;
  const bool live = !GUARD || tile_r * kTile + (w + 1) * 32 <= rows;  // wave-uniform: this wave's 32-row block exists

  float f[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (live) {
      Load8<IT>::run(src + (row0 + i) * cols + col0, f[i]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[i][j] = 0.f;
    }
  }

  if constexpr (ROW) {
    uint8_t *dst = out_row + m.off;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float mx = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) mx = fmaxf(mx, __builtin_fabsf(f[i][j]));
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      const uint8_t e = float_to_e8m0(mx * rcp_max);
      const float r = e8m0_rcp(e);
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f[i][j] * r;
      const u32x2 q = cvt8_fp8<F>(o, mx * r <= Fp8Traits<F>::max_finite);   // as mxfp8_tile_kernel
      if (live) *reinterpret_cast<u32x2 *>(dst + (row0 + i) * cols + col0) = q;
      uint32_t e32 = e;
      const uint32_t e1 = __shfl_down(e32, 4, 64);
      const uint32_t e2 = __shfl_down(e32, 8, 64);
      const uint32_t e3 = __shfl_down(e32, 12, 64);
      if (cg == 0 && live) {
        const uint32_t packed = e32 | (e1 << 8) | (e2 << 16) | (e3 << 24);
        const size_t r_idx = row_start + row0 + i, c_idx = tile_col * 4;
        const size_t idx = SWZ ? mx_swizzled_idx(r_idx, c_idx, stride_r / 4) : r_idx * stride_r + c_idx;
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
      float mx = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) mx = fmaxf(mx, __builtin_fabsf(f[i][j]));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const uint8_t e = float_to_e8m0(mx * rcp_max);
      ebytes[j >> 2] |= static_cast<uint32_t>(e) << (8 * (j & 3));
      const float r = e8m0_rcp(e);
      in_range = in_range && (mx * r <= Fp8Traits<F>::max_finite);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i][j] *= r;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32x2 w8 = cvt8_fp8<F>(f[i], in_range);
      qc[i][0] = w8[0];
      qc[i][1] = w8[1];
      if constexpr (!TCOL) {
        u32x2 q = {qc[i][0], qc[i][1]};
        if (live) *reinterpret_cast<u32x2 *>(out_col + m.off + (row0 + i) * cols + col0) = q;
      }
    }
    if constexpr (TCOL) {
      if constexpr (GUARD) {
        const size_t rows_left = rows - tile_r * kTile;
        store_tile_transposed<true>(tile_t, qc, t, out_col + m.off + (tile_col * kTile) * rows + tile_r * kTile, rows,
                                    static_cast<unsigned>(rows_left >= kTile ? 8 : rows_left / 32 * 2), kTile);
      } else {
        store_tile_transposed(tile_t, qc, t, out_col + m.off + (tile_col * kTile) * rows + tile_r * kTile, rows);
      }
    }
    if (live && (t & 63u) < 16u) {  // rg_local == 0 holds the wave's 32-row block scale for 8 columns
      if constexpr (TCOL) {
        const size_t sb = pad_cols * (row_start / 32);
        const size_t blk = tile_r * 4 + w;                      // block inside the member
        const size_t stride_c = (rows / 32 + 3) / 4 * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const size_t idx = sb + (SWZ ? mx_swizzled_idx(col0 + j, blk, stride_c / 4) : (col0 + j) * stride_c + blk);
          // (members that break the 128-row contract could otherwise run past the buffer)
          if (idx < sc_col_cap) sc_col[idx] = static_cast<uint8_t>(ebytes[j >> 2] >> (8 * (j & 3)));
        }
      } else if constexpr (SWZ) {
        // GEMM-swizzled: every member has a scale matrix of its own, [pad128(last), pad4(first_i / 32)] swizzled,
        // behind those of the members before it (ref group_quantize_mxfp8.cuh:114-126) - the bytes of TCOL
        const size_t sb = pad_cols * (row_start / 32);
        const size_t blk = tile_r * 4 + w;
        const size_t stride_c = (rows / 32 + 3) / 4 * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const size_t idx = sb + mx_swizzled_idx(col0 + j, blk, stride_c / 4);
          if (idx < sc_col_cap) sc_col[idx] = static_cast<uint8_t>(ebytes[j >> 2] >> (8 * (j & 3)));
        }
      } else {
        const size_t blk = (row_start + tile_r * kTile) / 32 + w;  // block of the whole group
        u32x2 v = {ebytes[0], ebytes[1]};
        *reinterpret_cast<u32x2 *>(sc_col + blk * pad_cols + col0) = v;
      }
    }
  }
  };  // body
  if (rows % kTile == 0) body(std::false_type{}); else body(std::true_type{});
}

// ---------------------------------------------------------------- grouped dequantize / bias add
template <Fp8 F, typename OT>
__global__ __launch_bounds__(256) void group_dequant_kernel(const uint8_t *__restrict__ in, OT *__restrict__ out,
                                                            const float *__restrict__ scale_inv, const GroupShapes g) {
  const int64_t active = active_numel(g);
  const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (i0 >= active) return;
  const unsigned t0 = member_holding(g, i0);
  for (int k = 0; k < 4 && i0 + k < active; ++k) {
    const unsigned t = (k == 0) ? t0 : member_holding(g, i0 + k);
    OT o;
    narrow(fp8_to_float<F>(in[i0 + k], 0) * scale_inv[t], o);
    out[i0 + k] = o;
  }
}

// out[row, col] += bias[member(row)][col] (* scale[row]); members share the last dim n.
template <typename T>
__global__ __launch_bounds__(256) void group_bias_add_kernel(T *__restrict__ out, const T *__restrict__ bias,
                                                             const float *__restrict__ row_scale,
                                                             const GroupShapes g, const int64_t n) {
  const int64_t active = active_numel(g);
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= active) return;
  const unsigned t = member_holding(g, i);
  const int64_t col = i % n;
  float b = widen(bias[static_cast<int64_t>(t) * n + col]);
  if (row_scale) b *= row_scale[i / n];
  T o;
  narrow(widen(out[i]) + b, o);
  out[i] = o;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

const float *noop_of(const NVTEQuantizationConfig config) { return noop_ptr_of(static_cast<const QuantConfig *>(config)); }

// MXFP8 branch of nvte_group_quantize (ref cast/dispatch/quantize.cuh:470-471 -> mxfp8::group_quantize)
void group_quantize_mxfp8(const GroupedTensor &in, GroupedTensor &out, const NVTEQuantizationConfig quant_config,
                                 hipStream_t stream) {
  NVTE_CHECK(in.num_tensors == out.num_tensors, "Number of input and output tensors must be same.");
  NVTE_CHECK(in.has_data(), "Cannot quantize tensor without rowwise data.");
  NVTE_CHECK(is_high_precision(in.data.dtype), "Grouped quantize input must be FP32/BF16/FP16");
  NVTE_CHECK(is_fp8(out.dtype()), "Output must have FP8 type.");
  const bool rowwise = out.has_data(), colwise = out.has_columnwise_data();
  NVTE_CHECK(rowwise || colwise, "Either rowwise or columnwise output data need to be allocated.");
  NVTE_CHECK(out.all_same_last_dim(),
             "Grouped MXFP8 quantization is implemented for same-shape and varying-first-dimension grouped tensors");
  NVTE_CHECK(in.logical_shape.data[1] == out.logical_shape.data[1] && in.logical_shape.data[0] == out.logical_shape.data[0],
             "Grouped MXFP8 input and output must have the same logical shape.");
  NVTE_CHECK(out.num_tensors <= kMaxMembers, "at most ", kMaxMembers, " member tensors are supported");
  const size_t R = out.logical_shape.data[0], C = out.logical_shape.data[1], T = out.num_tensors;
  // ref group_quantize_mxfp8.cuh:842 (first logical dim) and :846 (last dim)
  NVTE_CHECK(R % 128 == 0, "First logical dimension of a grouped MXFP8 tensor must be divisible by 128, got ", R);
  NVTE_CHECK(C % 128 == 0, "Last dimension of a grouped MXFP8 tensor must be divisible by 128, got ", C);
  const GroupShapes g = group_shapes_of(out);
  size_t tiles;
  if (out.all_same_shape()) {
    NVTE_CHECK(R % T == 0, "First logical dimension must be divisible by num_tensors.");
    NVTE_CHECK((R / T) % 128 == 0, "Member first dimension of a grouped MXFP8 tensor must be divisible by 128, got ", R / T);
    tiles = (R / kTile) * (C / kTile);
  } else {
    NVTE_CHECK(g.offsets && g.first_dims,
               "Varying first-dimension grouped MXFP8 quantization requires first_dims and tensor_offsets.");
    tiles = (R / kTile + T) * (C / kTile);
  }
  NVTE_CHECK(tiles < (1ull << 31), "grouped tensor too large");
  const size_t stride_r = round_up(C / 32, 4), pad_cols = round_up(C, 128);
  if (rowwise)
    NVTE_CHECK(out.scale_inv.numel() >= round_up(R, 128) * stride_r, "Grouped MXFP8 rowwise scale_inv holds ",
               out.scale_inv.numel(), " bytes, need ", round_up(R, 128) * stride_r);
  if (colwise)
    NVTE_CHECK(out.columnwise_scale_inv.numel() >= round_up(R / 32, 4) * pad_cols,
               "Grouped MXFP8 columnwise scale_inv holds ", out.columnwise_scale_inv.numel(), " bytes, need ",
               round_up(R / 32, 4) * pad_cols);
  (void)device_info();
  const float *noop = noop_of(quant_config);
  uint8_t *orow = static_cast<uint8_t *>(out.data.ptr), *ocol = static_cast<uint8_t *>(out.columnwise_data.ptr);
  uint8_t *srow = static_cast<uint8_t *>(out.scale_inv.ptr), *scol = static_cast<uint8_t *>(out.columnwise_scale_inv.ptr);
  NVTE_REQUIRE_DATA(in.data, "grouped quantize input");
  NVTE_CHECK((!rowwise || (orow && srow)) && (!colwise || (ocol && scol)),
             "grouped quantize output has shapes but no data pointers");
  NVTE_CHECK(aligned16(in.data.ptr) && (!orow || aligned16(orow)) && (!ocol || aligned16(ocol)) &&
                 (!srow || aligned16(srow)) && (!scol || aligned16(scol)),
             "Grouped MXFP8 quantization needs 16-byte aligned buffers");
  const bool tcol = colwise && out.columnwise_transposed, swz = out.with_gemm_swizzled_scales;
  const dim3 grid(static_cast<unsigned>(tiles)), block(256);
#define TE_GMX_LAUNCH(IT_, F_, R_, C_, T_, S_)                                                              \
  hipLaunchKernelGGL((group_mxfp8_kernel<IT_, F_, R_, C_, T_, S_>), grid, block, 0, stream,                  \
                     static_cast<const IT_ *>(in.data.ptr), orow, ocol, srow, scol, noop, g, R, stride_r, pad_cols, \
                     colwise ? out.columnwise_scale_inv.numel() : 0)
#define TE_GMX_BY_LAYOUT(IT_, F_)                                                                           \
  do {                                                                                                      \
    if (rowwise && tcol) { if (swz) TE_GMX_LAUNCH(IT_, F_, true, true, true, true); else TE_GMX_LAUNCH(IT_, F_, true, true, true, false); } \
    else if (tcol) { if (swz) TE_GMX_LAUNCH(IT_, F_, false, true, true, true); else TE_GMX_LAUNCH(IT_, F_, false, true, true, false); }      \
    else if (rowwise && colwise) { if (swz) TE_GMX_LAUNCH(IT_, F_, true, true, false, true); else TE_GMX_LAUNCH(IT_, F_, true, true, false, false); } \
    else if (rowwise) { if (swz) TE_GMX_LAUNCH(IT_, F_, true, false, false, true); else TE_GMX_LAUNCH(IT_, F_, true, false, false, false); } \
    else { if (swz) TE_GMX_LAUNCH(IT_, F_, false, true, false, true); else TE_GMX_LAUNCH(IT_, F_, false, true, false, false); }             \
  } while (0)
#define TE_GMX_BY_OUT(IT_)                                                                                  \
  if (out.dtype() == kNVTEFloat8E4M3) TE_GMX_BY_LAYOUT(IT_, Fp8::E4M3); else TE_GMX_BY_LAYOUT(IT_, Fp8::E5M2)
  switch (in.data.dtype) {
    case kNVTEBFloat16: TE_GMX_BY_OUT(bf16_t); break;
    case kNVTEFloat16: TE_GMX_BY_OUT(f16_t); break;
    default: TE_GMX_BY_OUT(float); break;
  }
#undef TE_GMX_BY_OUT
#undef TE_GMX_BY_LAYOUT
#undef TE_GMX_LAUNCH
  NVTE_CHECK_HIP(hipGetLastError());
}

}  // namespace
}  // namespace te

using namespace te;

extern "C" {

// ---------------------------------------------------------------- offsets helpers
void nvte_splits_to_offsets(const int64_t *split_sizes, int64_t *output, size_t num_splits, int64_t stride,
                            hipStream_t stream) {
  api_guard("nvte_splits_to_offsets", [&] {
    NVTE_CHECK(split_sizes != nullptr && output != nullptr, "split_sizes and output must be device pointers");
    (void)device_info();
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(64), 0, stream, split_sizes, 1, nullptr, 1, output, 1,
                       num_splits, stride, 1);
    NVTE_CHECK_HIP(hipGetLastError());
  });
}

void nvte_compute_grouped_tensor_offsets(const int64_t *first_dims, int64_t *offsets, size_t n_groups,
                                         int64_t last_dim, hipStream_t stream) {
  api_guard("nvte_compute_grouped_tensor_offsets", [&] {
    NVTE_CHECK(first_dims != nullptr && offsets != nullptr, "first_dims and offsets must be device pointers");
    (void)device_info();
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(64), 0, stream, first_dims, 1, nullptr, 1, offsets, 1, n_groups,
                       last_dim, 1);
    NVTE_CHECK_HIP(hipGetLastError());
  });
}

void nvte_splits_to_offsets_2d(const NVTETensor first_dims, const NVTETensor last_dims, NVTETensor output,
                               hipStream_t stream) {
  api_guard("nvte_splits_to_offsets_2d", [&] {
    const Tensor &f = *lookup_checked(first_dims);
    const Tensor &l = *lookup_checked(last_dims);
    Tensor &o = *lookup_checked(output);
    const size_t n = f.data.numel();
    NVTE_CHECK(l.data.numel() == n, "first_dims and last_dims must have the same number of entries");
    NVTE_CHECK(o.data.numel() == n + 1, "output must have ", n + 1, " entries, got ", o.data.numel());
    (void)device_info();
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(64), 0, stream, f.data.ptr, is64_of(f, "first_dims"), l.data.ptr,
                       is64_of(l, "last_dims"), o.data.ptr, is64_of(o, "output"), n, static_cast<int64_t>(1), 1);
    NVTE_CHECK_HIP(hipGetLastError());
  });
}

void nvte_splits_to_offsets_multi(NVTETensor split_sizes, NVTETensor *outputs, const int64_t *strides,
                                  const int *include_leading_zero, size_t num_outputs, hipStream_t stream) {
  api_guard("nvte_splits_to_offsets_multi", [&] {
    const Tensor &s = *lookup_checked(split_sizes);
    const size_t n = s.data.numel();
    NVTE_CHECK(num_outputs == 0 || (outputs && strides && include_leading_zero), "null argument array");
    (void)device_info();
    for (size_t k = 0; k < num_outputs; ++k) {
      Tensor &o = *lookup_checked(outputs[k]);
      const int lead = include_leading_zero[k] != 0;
      NVTE_CHECK(o.data.numel() == n + lead, "outputs[", k, "] must have ", n + lead, " entries, got ",
                 o.data.numel());
      hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(64), 0, stream, s.data.ptr, is64_of(s, "split_sizes"), nullptr,
                         1, o.data.ptr, is64_of(o, "output"), n, strides[k], lead);
      NVTE_CHECK_HIP(hipGetLastError());
    }
  });
}

void nvte_convert_int32_to_int64(const int32_t *src, int64_t *dst, size_t n, hipStream_t stream) {
  nvte_convert_int32_to_int64_with_multiplier(src, dst, n, 1, stream);
}

void nvte_convert_int32_to_int64_with_multiplier(const int32_t *src, int64_t *dst, size_t n, int64_t multiplier,
                                                 hipStream_t stream) {
  api_guard("nvte_convert_int32_to_int64_with_multiplier", [&] {
    if (n == 0) return;
    NVTE_CHECK(src != nullptr && dst != nullptr, "src and dst must be device pointers");
    (void)device_info();
    hipLaunchKernelGGL(convert_i32_i64_kernel, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256), 0, stream,
                       src, dst, n, multiplier);
    NVTE_CHECK_HIP(hipGetLastError());
  });
}

// ---------------------------------------------------------------- grouped amax / scale
void nvte_group_compute_amax_with_config(const NVTEGroupedTensor input_, NVTEGroupedTensor output_,
                                         const NVTEQuantizationConfig config, hipStream_t stream) {
  api_guard("nvte_group_compute_amax_with_config", [&] {
    NVTE_CHECK(input_ != nullptr, "Invalid grouped input tensor (got NULL)");
    NVTE_CHECK(output_ != nullptr, "Invalid grouped output tensor (got NULL)");
    const GroupedTensor &in = *lookup_grouped_checked(input_);
    GroupedTensor &out = *lookup_grouped_checked(output_);
    NVTE_CHECK(in.num_tensors == out.num_tensors, "Number of grouped input and output tensors must match.");
    NVTE_CHECK(in.has_data(), "Grouped amax input must have rowwise data.");
    NVTE_CHECK(!is_fp8(in.data.dtype), "Grouped amax input must be unquantized, but got dtype=",
               dtype_name(in.data.dtype));
    NVTE_CHECK(out.amax.present() || out.columnwise_amax.present(), "Grouped amax output must have an amax buffer.");
    check_grouped_input(in, "group_compute_amax_input");
    check_grouped_output(out, "group_compute_amax_output", true);
    NVTE_CHECK(out.num_tensors <= kMaxMembers, "at most ", kMaxMembers, " member tensors are supported");
    const GroupShapes g = group_shapes_of(out);
    if (!out.all_same_first_dim()) NVTE_CHECK(g.first_dims, "Grouped amax requires first_dims for varying first dimensions.");
    if (!out.all_same_last_dim()) NVTE_CHECK(g.last_dims, "Grouped amax requires last_dims for varying last dimensions.");
    const float *noop = noop_of(config);
    float *amax = static_cast<float *>(out.amax.ptr ? out.amax.ptr : out.columnwise_amax.ptr);
    NVTE_CHECK(amax != nullptr, "Grouped amax buffer has no data pointer");
    NVTE_REQUIRE_DATA(in.data, "grouped amax input");
    (void)device_info();
    const unsigned T = static_cast<unsigned>(out.num_tensors);
    hipLaunchKernelGGL(group_zero_amax_kernel, dim3(ceil_div(T, 256)), dim3(256), 0, stream, amax, T, noop);
    NVTE_CHECK_HIP(hipGetLastError());
    const size_t total = out.logical_shape.data[0] * out.logical_shape.data[1];
    const unsigned grid = static_cast<unsigned>(ceil_div(total, kAmaxChunk));
    const int vec_ok = aligned16(in.data.ptr);
    switch (in.data.dtype) {
      case kNVTEBFloat16:
        hipLaunchKernelGGL(group_amax_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream,
                           static_cast<const bf16_t *>(in.data.ptr), amax, g, noop, vec_ok);
        break;
      case kNVTEFloat16:
        hipLaunchKernelGGL(group_amax_kernel<f16_t>, dim3(grid), dim3(256), 0, stream,
                           static_cast<const f16_t *>(in.data.ptr), amax, g, noop, vec_ok);
        break;
      case kNVTEFloat32:
        hipLaunchKernelGGL(group_amax_kernel<float>, dim3(grid), dim3(256), 0, stream,
                           static_cast<const float *>(in.data.ptr), amax, g, noop, vec_ok);
        break;
      default: NVTE_ERROR("Unsupported grouped amax input dtype ", dtype_name(in.data.dtype));
    }
    NVTE_CHECK_HIP(hipGetLastError());
  });
}

void nvte_group_compute_scale_from_amax(NVTEGroupedTensor output, const NVTEQuantizationConfig config,
                                        hipStream_t stream) {
  api_guard("nvte_group_compute_scale_from_amax", [&] {
    NVTE_CHECK(output != nullptr, "Invalid grouped output tensor (got NULL)");
    GroupedTensor &out = *lookup_grouped_checked(output);
    if (out.num_tensors == 0) return;
    NVTE_CHECK(is_fp8(out.dtype()), "Grouped scale update requires an FP8 output tensor, but got dtype=",
               dtype_name(out.dtype()));
    float *amax = static_cast<float *>(out.amax.ptr ? out.amax.ptr : out.columnwise_amax.ptr);
    NVTE_CHECK(amax != nullptr, "Grouped scale update requires an amax buffer.");
    float *scale = static_cast<float *>(out.scale.ptr);
    NVTE_CHECK(scale != nullptr, "Grouped scale update requires a scale buffer.");
    float *scale_inv = static_cast<float *>(out.scale_inv.ptr ? out.scale_inv.ptr : out.columnwise_scale_inv.ptr);
    NVTE_CHECK(scale_inv != nullptr, "Grouped scale update requires a scale_inv buffer.");
    const QuantConfig *cfg = static_cast<const QuantConfig *>(config);
    (void)device_info();
    const unsigned T = static_cast<unsigned>(out.num_tensors);
    hipLaunchKernelGGL(group_scale_kernel, dim3(ceil_div(T, 256)), dim3(256), 0, stream, amax, scale, scale_inv, T,
                       fp8_max_of(out.dtype()), cfg ? cfg->force_pow_2_scales : false,
                       cfg ? cfg->amax_epsilon : 0.f, noop_ptr_of(cfg));
    NVTE_CHECK_HIP(hipGetLastError());
  });
}

// ---------------------------------------------------------------- grouped quantize
void nvte_group_quantize(const NVTEGroupedTensor input_, NVTEGroupedTensor output_,
                         const NVTEQuantizationConfig quant_config, hipStream_t stream) {
  api_guard("nvte_group_quantize", [&] {
    const GroupedTensor &in = *lookup_grouped_checked(input_);
    GroupedTensor &out = *lookup_grouped_checked(output_);
    check_grouped_input(in, "input");
    check_grouped_output(out, "output", false);
    if (is_mxfp8(out.scaling_mode)) {
      group_quantize_mxfp8(in, out, quant_config, stream);
      return;
    }
    NVTE_CHECK(out.scaling_mode == NVTE_DELAYED_TENSOR_SCALING,
               "nvte_group_quantize: FP8 per-tensor scaling and MXFP8 are implemented in this library, got ",
               scaling_mode_name(out.scaling_mode));
    NVTE_CHECK(in.num_tensors == out.num_tensors, "Number of input and output tensors must be same.");
    NVTE_CHECK(in.has_data(), "Cannot quantize tensor without rowwise data.");
    NVTE_CHECK(is_high_precision(in.data.dtype), "Grouped quantize input must be FP32/BF16/FP16");
    NVTE_CHECK(is_fp8(out.dtype()), "Output must have FP8 type.");
    NVTE_CHECK(out.scale.present(), "Grouped FP8 tensor-scaling output scale must be set.");
    NVTE_CHECK(out.scale.dtype == kNVTEFloat32, "Grouped FP8 tensor-scaling scale must be FP32.");
    NVTE_CHECK(out.scale.numel() >= out.num_tensors,
               "Grouped FP8 tensor-scaling scale must have at least one entry per tensor.");
    const bool rowwise = out.has_data(), colwise = out.has_columnwise_data();
    NVTE_CHECK(rowwise || colwise, "Either rowwise or columnwise output data need to be allocated.");
    NVTE_CHECK(!out.varying_both_dims(),
               "Grouped FP8 tensor-scaling quantization only supports same-shape, varying first-dimension, "
               "or varying last-dimension grouped tensors.");
    NVTE_CHECK(in.logical_shape.data[1] == out.logical_shape.data[1],
               "Grouped FP8 tensor-scaling input and output must have the same last dimension.");
    NVTE_CHECK(out.logical_shape.data[0] <= in.logical_shape.data[0],
               "Grouped FP8 tensor-scaling output first dimension must not exceed input first dimension.");
    NVTE_CHECK(out.num_tensors <= kMaxMembers, "at most ", kMaxMembers, " member tensors are supported");
    const size_t R = out.logical_shape.data[0], C = out.logical_shape.data[1], T = out.num_tensors;
    if (R == 0 || C == 0) return;
    const GroupShapes g = group_shapes_of(out);
    // upper bound of the number of 128x128 tiles (member shapes are only known on the device)
    size_t tiles;
    if (out.all_same_shape()) {
      NVTE_CHECK(R % T == 0, "First logical dimension must be divisible by num_tensors.");
      tiles = T * ceil_div(R / T, kTile) * ceil_div(C, kTile);
    } else if (out.all_same_last_dim()) {
      NVTE_CHECK(g.offsets && g.first_dims,
                 "Varying first-dimension grouped FP8 quantization requires first_dims and tensor_offsets.");
      tiles = (ceil_div(R, kTile) + T) * ceil_div(C, kTile);
    } else {
      NVTE_CHECK(g.offsets && g.last_dims,
                 "Varying last-dimension grouped FP8 quantization requires last_dims and tensor_offsets.");
      tiles = ceil_div(R, kTile) * (ceil_div(C, kTile) + T);
    }
    NVTE_CHECK(tiles < (1ull << 31), "grouped tensor too large");
    (void)device_info();
    const float *noop = noop_of(quant_config);
    uint8_t *oc = static_cast<uint8_t *>(out.data.ptr);
    uint8_t *ot = static_cast<uint8_t *>(out.columnwise_data.ptr);
    const float *sc = static_cast<const float *>(out.scale.ptr);
    NVTE_REQUIRE_DATA(in.data, "grouped quantize input");
    NVTE_CHECK(oc != nullptr || ot != nullptr, "grouped quantize output has shapes but no data pointers");
    NVTE_CHECK(sc != nullptr, "grouped quantize output scale has no data pointer");
    const int al = aligned16(in.data.ptr) && (!oc || aligned16(oc)) && (!ot || aligned16(ot));
    const dim3 grid(static_cast<unsigned>(tiles)), block(256);
#define TE_GQ_LAUNCH(IT_, F_)                                                                         \
  hipLaunchKernelGGL((group_cast_transpose_kernel<IT_, F_>), grid, block, 0, stream,                  \
                     static_cast<const IT_ *>(in.data.ptr), oc, ot, sc, noop, g, al)
#define TE_GQ_BY_OUT(IT_)                                                                              \
  if (out.dtype() == kNVTEFloat8E4M3) TE_GQ_LAUNCH(IT_, Fp8::E4M3); else TE_GQ_LAUNCH(IT_, Fp8::E5M2)
    switch (in.data.dtype) {
      case kNVTEBFloat16: TE_GQ_BY_OUT(bf16_t); break;
      case kNVTEFloat16: TE_GQ_BY_OUT(f16_t); break;
      default: TE_GQ_BY_OUT(float); break;
    }
#undef TE_GQ_BY_OUT
#undef TE_GQ_LAUNCH
    NVTE_CHECK_HIP(hipGetLastError());
  });
}

void nvte_group_dequantize(const NVTEGroupedTensor input_, NVTEGroupedTensor output_, hipStream_t stream) {
  api_guard("nvte_group_dequantize", [&] {
    const GroupedTensor &in = *lookup_grouped_checked(input_);
    GroupedTensor &out = *lookup_grouped_checked(output_);
    check_grouped_input(in, "input");
    NVTE_CHECK(in.scaling_mode == NVTE_DELAYED_TENSOR_SCALING && is_fp8(in.dtype()) && in.has_data(),
               "nvte_group_dequantize: rowwise FP8 per-tensor-scaled input expected");
    NVTE_CHECK(out.has_data() && is_high_precision(out.data.dtype), "output must be FP32/BF16/FP16 rowwise data");
    NVTE_CHECK(out.data.numel() == in.data.numel(), "input and output sizes differ");
    NVTE_CHECK(in.num_tensors <= kMaxMembers, "at most ", kMaxMembers, " member tensors are supported");
    const GroupShapes g = group_shapes_of(in);
    const size_t total = in.data.numel();
    if (total == 0) return;
    (void)device_info();
    const dim3 grid(static_cast<unsigned>(ceil_div(total, 1024))), block(256);
    const uint8_t *ip = static_cast<const uint8_t *>(in.data.ptr);
    const float *si = static_cast<const float *>(in.scale_inv.ptr);
    NVTE_CHECK(ip != nullptr && si != nullptr && out.data.ptr != nullptr,
               "grouped dequantize: input data, scale_inv and output need data pointers");
#define TE_GD(F_, OT_) hipLaunchKernelGGL((group_dequant_kernel<F_, OT_>), grid, block, 0, stream, ip, \
                                          static_cast<OT_ *>(out.data.ptr), si, g)
#define TE_GD_BY_OUT(F_)                                         \
  switch (out.data.dtype) {                                      \
    case kNVTEBFloat16: TE_GD(F_, bf16_t); break;                \
    case kNVTEFloat16: TE_GD(F_, f16_t); break;                  \
    default: TE_GD(F_, float); break;                            \
  }
    if (in.dtype() == kNVTEFloat8E4M3) { TE_GD_BY_OUT(Fp8::E4M3) } else { TE_GD_BY_OUT(Fp8::E5M2) }
#undef TE_GD_BY_OUT
#undef TE_GD
    NVTE_CHECK_HIP(hipGetLastError());
  });
}

// ---------------------------------------------------------------- grouped bias add
static void grouped_bias_add_impl(const char *api, const NVTEGroupedTensor output, const NVTEGroupedTensor bias,
                                  const NVTETensor scale, hipStream_t stream) {
  const GroupedTensor &out = *lookup_grouped_checked(output);
  const GroupedTensor &b = *lookup_grouped_checked(bias);
  NVTE_CHECK(out.num_tensors >= 1, api, ": number of tensors must be at least 1");
  NVTE_CHECK(out.num_tensors == b.num_tensors, api, ": output and bias must have the same number of tensors");
  NVTE_CHECK(out.has_data(), api, ": output is missing row-wise data");
  NVTE_CHECK(b.has_data(), api, ": bias is missing row-wise data");
  NVTE_CHECK(out.dtype() == b.dtype(), api, ": output and bias must have the same dtype");
  NVTE_CHECK(b.all_same_first_dim(), api, ": bias first dim must be uniform");
  NVTE_CHECK(b.common_first_dim() == 1, api, ": bias first dim must be 1");
  NVTE_CHECK(out.all_same_last_dim() && b.all_same_last_dim(), api, ": requires a uniform last dimension");
  NVTE_CHECK(out.common_last_dim() == b.common_last_dim(), api, ": output and bias last dims differ");
  NVTE_CHECK(out.num_tensors <= kMaxMembers, api, " supports at most ", kMaxMembers, " tensors");
  const float *rs = nullptr;
  if (scale != nullptr) {
    const Tensor &s = *lookup_checked(scale);
    NVTE_CHECK(s.data.ptr != nullptr, "Grouped scaled bias add: scale tensor has no data");
    NVTE_CHECK(s.data.dtype == kNVTEFloat32, "Grouped scaled bias add: scale must be Float32");
    NVTE_CHECK(s.data.shape.ndim == 1, "Grouped scaled bias add: scale must be 1D");
    NVTE_CHECK(s.data.shape.data[0] == out.logical_shape.data[0], "Grouped scaled bias add: scale size (",
               s.data.shape.data[0], ") must equal the number of output rows (", out.logical_shape.data[0], ")");
    rs = static_cast<const float *>(s.data.ptr);
  }
  const GroupShapes g = group_shapes_of(out);
  const size_t total = out.data.numel();
  if (total == 0) return;
  (void)device_info();
  const int64_t n = static_cast<int64_t>(out.common_last_dim());
  const dim3 grid(static_cast<unsigned>(ceil_div(total, 256))), block(256);
  switch (out.dtype()) {
    case kNVTEBFloat16:
      hipLaunchKernelGGL(group_bias_add_kernel<bf16_t>, grid, block, 0, stream, static_cast<bf16_t *>(out.data.ptr),
                         static_cast<const bf16_t *>(b.data.ptr), rs, g, n);
      break;
    case kNVTEFloat16:
      hipLaunchKernelGGL(group_bias_add_kernel<f16_t>, grid, block, 0, stream, static_cast<f16_t *>(out.data.ptr),
                         static_cast<const f16_t *>(b.data.ptr), rs, g, n);
      break;
    case kNVTEFloat32:
      hipLaunchKernelGGL(group_bias_add_kernel<float>, grid, block, 0, stream, static_cast<float *>(out.data.ptr),
                         static_cast<const float *>(b.data.ptr), rs, g, n);
      break;
    default: NVTE_ERROR(api, ": output must be BF16, FP16, or FP32");
  }
  NVTE_CHECK_HIP(hipGetLastError());
}

void nvte_grouped_bias_add(const NVTEGroupedTensor output, const NVTEGroupedTensor bias, hipStream_t stream) {
  api_guard("nvte_grouped_bias_add", [&] { grouped_bias_add_impl("nvte_grouped_bias_add", output, bias, nullptr, stream); });
}

void nvte_grouped_scaled_bias_add(const NVTEGroupedTensor output, const NVTEGroupedTensor bias,
                                  const NVTETensor scale, hipStream_t stream) {
  api_guard("nvte_grouped_scaled_bias_add", [&] {
    NVTE_CHECK(scale != nullptr, "Grouped scaled bias add: scale tensor is required");
    grouped_bias_add_impl("nvte_grouped_scaled_bias_add", output, bias, scale, stream);
  });
}

}  // extern "C"
