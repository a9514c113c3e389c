// Activations (plain and gated, forward and backward) fused with the FP8 cast, the transpose and
// the bias-gradient reduction: one pass over the data for what the unfused sequence does in three.
//
// Replaces (behaviour, not structure):
//   ref common/activation/{gelu,relu,swiglu}.cu + activation_template.h (entry points),
//       common/util/math.h:21-134 (the functions), util/vectorized_pointwise.h:176-227 (unary),
//       :410-560 (gated forward / backward: input [N, 2H], activation half first),
//       common/cast/cast_dbias*.cu, transpose/cast_transpose_fusion.cu:1260-1420
//       (nvte_quantize_dbias_d<act>, nvte_cast_transpose_dbias_d<act>, nvte_d<gated>_cast_transpose).
//
// HBM-bound.  Algorithmic bytes per output element (bf16 in, FP8 out with transpose): unary 2+2,
// d-act 4+2, gated forward 4+2, gated backward 3+2 per OUTPUT element.
// Kernel: the 128x128 tile of cast_transpose.hip (8x8 values per thread, 16-byte loads, row-wise
// bytes straight from registers, transposes through the XOR-swizzled LDS image), with the value
// produced by a functor of up to three inputs; the dbias partial of a tile (sum over its 128 rows)
// goes through 8 KiB of LDS into the [ceil(R/128), C] fp32 workspace of the existing two-call
// protocol.  The activation is a wave-uniform runtime switch (free next to the memory traffic),
// the dataflow (MODE) and the output kind are template parameters.
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>

#include "common.h"
#include "device_utils.h"
#include "act_math.h"
#include "group_lookup.h"
#include "transformer_engine/activation.h"
#include "transformer_engine/cast.h"
#include "transformer_engine/transpose.h"

namespace te {

void dbias_finalize_launch(const float *ws, Tensor &dbias, size_t slabs, size_t cols, hipStream_t stream);  // dbias.hip
bool cast_transpose_dbias_tile_launch(const Tensor &in, Tensor &out, float *part, hipStream_t stream);     // cast_transpose.hip

namespace {

enum : int { kActNone = 0, kActGelu, kActSilu, kActRelu, kActQGelu, kActSRelu, kActSigmoid, kActClampedSilu };
enum : int { kUnary = 0, kDAct = 1, kGatedFwd = 2, kGatedBwd = 3 };
enum : int { kOutE4M3 = 0, kOutE5M2 = 1, kOutHP = 2, kOutMXE4M3 = 3, kOutMXE5M2 = 4 };

// ref common/util/math.h:21-134
__device__ __forceinline__ float act_f(int id, float x) {
  switch (id) {
    case kActGelu: return x * sigmoid_f(2.f * x * (0.79788456f + 0.03567741f * x * x));
    case kActSilu: return x * sigmoid_f(x);
    case kActRelu: return fmaxf(x, 0.f);
    case kActQGelu: return x * sigmoid_f(1.702f * x);
    case kActSRelu: return x > 0.f ? x * x : 0.f;
    case kActSigmoid: return sigmoid_f(x);
    default: return x;
  }
}
__device__ __forceinline__ float dact_f(int id, float x) {
  switch (id) {
    case kActGelu: {
      const float p = sigmoid_f(2.f * 0.79788456f * x * (1.f + 0.044715f * x * x));
      return 2.f * x * (p * (1.f - p) * (0.79788456f + 0.1070322243f * x * x)) + p;
    }
    case kActSilu: {
      const float s = sigmoid_f(x);
      return x * s * (1.f - s) + s;
    }
    case kActRelu: return x > 0.f ? 1.f : 0.f;
    case kActQGelu: {
      const float s = sigmoid_f(1.702f * x);
      return 1.702f * x * s * (1.f - s) + s;
    }
    case kActSRelu: return fmaxf(2.f * x, 0.f);
    case kActSigmoid: {
      const float s = sigmoid_f(x);
      return s * (1.f - s);
    }
    default: return 1.f;
  }
}

// compile-time id: the tile kernel dispatches on the activation ONCE per workgroup (a per-element switch
// costs a taken scalar branch per value)
template <int ID>
__device__ __forceinline__ float act_c(float x) { return act_f(ID, x); }
template <int ID>
__device__ __forceinline__ float dact_c(float x) { return dact_f(ID, x); }

struct FusedArgs {
  const void *in;    // unary: input; d-act / gated: the activation input ([R, H] or [R, 2H])
  const void *grad;  // d-act / gated backward: incoming gradient [R, H]
  void *out_c;       // row-wise output ([R, H], gated backward [R, 2H]) or null
  uint8_t *out_t;    // transposed FP8 output or null
  const float *scale;
  float *amax, *scale_inv;
  float *dbias_part;  // [ceil(R/128)][H] fp32 or null
  unsigned rows, cols;  // cols = H, the width of the activation
  unsigned ld_in, ld_out;
  int in_dtype, out_dtype;
  int act;
  // MXFP8 output (out_c = row-wise payload): column-wise payload, e8m0 scales and their layouts
  uint8_t *mx_col, *sc_row, *sc_col;
  unsigned stride_r, stride_c;  // second dims of the two scale arrays
  int swz, tcol;                // GEMM-swizzled scales; column-wise payload stored transposed (amd_ext.h)
  // clamped SwiGLU (ref math.h:80-98, vectorized_pointwise.h:436-438,543-546)
  float limit, alpha, glu_offset;
  // grouped launch (nvte_group_*; grp.num_tensors != 0): the buffers above are the GROUP's flat buffers, `rows`
  // the logical first dim of the group, `scale` one entry per member; every workgroup looks its member up
  // on the device and re-bases the pointers (unary / d-act modes, uniform last dim)
  GroupShapes grp;
  unsigned pad_cols;  // grouped MXFP8, transposed column-wise scales: pad128(cols), rows of a member's scale matrix
};

// compact -> GEMM-swizzled scale index (ref common/cast/mxfp8/swizzle.cuh:30-41; twin of mxfp8.hip)
__device__ __forceinline__ size_t mx_swizzled_idx(size_t i, size_t j, size_t tiles_x) {
  const size_t tile = (i / 128) * tiles_x + (j / 4);
  const size_t y = i % 128, x = j % 4;
  return tile * 512 + (y % 32) * 16 + (y / 32) * 4 + x;
}

// activation / derivative / linear-half transform with the launch's parameters (compile-time id)
template <int ACT>
__device__ __forceinline__ float act_p(float x, const FusedArgs &a) {
  if constexpr (ACT == kActClampedSilu) return clamped_act(x, a.limit, a.alpha);
  else return act_c<ACT>(x);
}
template <int ACT>
__device__ __forceinline__ float dact_p(float x, const FusedArgs &a) {
  if constexpr (ACT == kActClampedSilu) return clamped_dact(x, a.limit, a.alpha);
  else return dact_c<ACT>(x);
}
// returns the value the linear half contributes; `pass` = its gradient flows (clamp inactive)
template <int ACT>
__device__ __forceinline__ float gate_p(float g, const FusedArgs &a, bool &pass) {
  if constexpr (ACT == kActClampedSilu) {
    pass = g <= a.limit && g >= -a.limit;
    return fminf(fmaxf(-a.limit, g), a.limit) + a.glu_offset;
  } else {
    pass = true;
    return g;
  }
}
// the same with a run-time id (element-wise kernels)
__device__ __forceinline__ float act_r(float x, const FusedArgs &a) {
  return a.act == kActClampedSilu ? clamped_act(x, a.limit, a.alpha) : act_f(a.act, x);
}
__device__ __forceinline__ float dact_r(float x, const FusedArgs &a) {
  return a.act == kActClampedSilu ? clamped_dact(x, a.limit, a.alpha) : dact_f(a.act, x);
}
__device__ __forceinline__ float gate_r(float g, const FusedArgs &a, bool &pass) {
  pass = true;
  if (a.act != kActClampedSilu) return g;
  pass = g <= a.limit && g >= -a.limit;
  return fminf(fmaxf(-a.limit, g), a.limit) + a.glu_offset;
}

constexpr int kTile = 128;

template <int OK>
__device__ __forceinline__ uint32_t pack4(float a, float b, float c, float d) {
  if constexpr (OK == kOutE4M3) return cvt4_fp8<Fp8::E4M3>(a, b, c, d);
  else return cvt4_fp8<Fp8::E5M2>(a, b, c, d);
}

// raw 8-element vectors: issued back to back for all 8 rows of a thread BEFORE any arithmetic, so that
// every load of a tile is in flight at once (a guarded load-compute-store per row serialises them:
// measured 143 us instead of 75 us for the plain cast + dbias of dY)
template <typename IT>
struct Raw8 {
  u32x4 v;
  __device__ __forceinline__ void load(const IT *p) { v = ld_stream<u32x4>(p); }
  __device__ __forceinline__ void unpack(float (&f)[8]) const {
    if constexpr (sizeof(IT) == 2 && std::is_same<IT, bf16_t>::value) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f[2 * k] = __uint_as_float(v[k] << 16);
        f[2 * k + 1] = __uint_as_float(v[k] & 0xFFFF0000u);
      }
    } else {
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      const h8 h = __builtin_bit_cast(h8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = static_cast<float>(h[k]);
    }
  }
};
template <>
struct Raw8<float> {
  f32x4 lo, hi;
  __device__ __forceinline__ void load(const float *p) {
    lo = ld_stream<f32x4>(p);
    hi = ld_stream<f32x4>(p + 4);
  }
  __device__ __forceinline__ void unpack(float (&f)[8]) const {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[k] = lo[k];
      f[4 + k] = hi[k];
    }
  }
};

// MXFP8 block quantize of a thread's 8x8 values of output half `o` (ref cast/mxfp8/gated_mxfp8.cuh,
// quantize_mxfp8.cuh; same thread map and arithmetic as mxfp8_tile_kernel in mxfp8.hip): a 1x32 row
// block is 4 neighbouring lanes, a 32x1 column block the lanes {cg, cg+16, cg+32, cg+48} of a wave.
// Whole 128x128 tiles only.  Which of the two directions, the scale layout and the transposed
// column-wise store are block-uniform run-time choices around whole sections, not per element.
template <Fp8 F>
__device__ __forceinline__ void mx_store_block(float (&f)[8][8], const FusedArgs &a, unsigned t, size_t tile_r,
                                               size_t tile_col, unsigned o, uint8_t *tile_t) {
  constexpr float rcp_max = 1.0f / Fp8Traits<F>::max_finite;
  const unsigned cg = t & 15u, rg = t >> 4;
  const size_t row0 = tile_r * kTile + rg * 8;
  const size_t ocol0 = static_cast<size_t>(o) * a.cols + tile_col * kTile + cg * 8;  // column of the output matrix
  if (a.out_c) {
    uint8_t *out_row = static_cast<uint8_t *>(a.out_c);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, __builtin_fabsf(f[i][j]));
      m = fmaxf(m, __shfl_xor(m, 1, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      const uint8_t e = float_to_e8m0(m * rcp_max);
      const float r = e8m0_rcp(e);
      const u32x2 q = {cvt4_fp8<F>(f[i][0] * r, f[i][1] * r, f[i][2] * r, f[i][3] * r),
                       cvt4_fp8<F>(f[i][4] * r, f[i][5] * r, f[i][6] * r, f[i][7] * r)};
      *reinterpret_cast<u32x2 *>(out_row + (row0 + i) * a.ld_out + ocol0) = q;
      const uint32_t e32 = e;
      const uint32_t e1 = __shfl_down(e32, 4, 64), e2 = __shfl_down(e32, 8, 64), e3 = __shfl_down(e32, 12, 64);
      if (cg == 0) {
        const size_t r_idx = row0 + i, c_idx = (static_cast<size_t>(o) * a.cols + tile_col * kTile) / 32;
        const size_t idx = a.swz ? mx_swizzled_idx(r_idx, c_idx, a.stride_r / 4) : r_idx * a.stride_r + c_idx;
        *reinterpret_cast<uint32_t *>(a.sc_row + idx) = e32 | (e1 << 8) | (e2 << 16) | (e3 << 24);
      }
    }
  }
  if (a.mx_col) {
    uint32_t qc[8][2];
    uint32_t ebytes[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) m = fmaxf(m, __builtin_fabsf(f[i][j]));
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const uint8_t e = float_to_e8m0(m * rcp_max);
      ebytes[j >> 2] |= static_cast<uint32_t>(e) << (8 * (j & 3));
      const float r = e8m0_rcp(e);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i][j] *= r;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      qc[i][0] = cvt4_fp8<F>(f[i][0], f[i][1], f[i][2], f[i][3]);
      qc[i][1] = cvt4_fp8<F>(f[i][4], f[i][5], f[i][6], f[i][7]);
      if (!a.tcol) {
        const u32x2 q = {qc[i][0], qc[i][1]};
        *reinterpret_cast<u32x2 *>(a.mx_col + (row0 + i) * a.ld_out + ocol0) = q;
      }
    }
    if (a.tcol)
      store_tile_transposed(tile_t, qc, t,
                            a.mx_col + (static_cast<size_t>(o) * a.cols + tile_col * kTile) * a.rows + tile_r * kTile, a.rows);
    if ((t & 63u) < 16u) {  // the wave's first row group holds the 32-row block scales of its 8 columns
      const size_t blk = tile_r * 4 + (t >> 6);
      if (a.swz) {
        const size_t tiles_x = ceil_div_dev(a.rows, 128);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          a.sc_col[mx_swizzled_idx(ocol0 + j, blk, tiles_x)] = static_cast<uint8_t>(ebytes[j >> 2] >> (8 * (j & 3)));
      } else if (a.tcol) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          a.sc_col[(ocol0 + j) * a.stride_c + blk] = static_cast<uint8_t>(ebytes[j >> 2] >> (8 * (j & 3)));
      } else {
        const u32x2 v = {ebytes[0], ebytes[1]};
        *reinterpret_cast<u32x2 *>(a.sc_col + blk * a.stride_c + ocol0) = v;
      }
    }
  }
}

template <typename IT>
__device__ __forceinline__ float round_to_input_type(float v) {
  if constexpr (std::is_same<IT, float>::value) {
    return v;
  } else {
    IT o;
    narrow(v, o);
    return widen(o);
  }
}

// ACT is a template parameter on purpose: dispatching on a run-time id inside the kernel (even once per
// workgroup around seven inlined bodies) ends up as a branch tree PER ELEMENT after LLVM merges the
// bodies back together - measured 103 us instead of 76 us for relu -> FP8 + transpose.
template <typename IT, int OK, int MODE, int ACT, bool DBIAS>
__global__ __launch_bounds__(256) void fused_act_tile_kernel(const FusedArgs a_in) {
  constexpr bool FP8 = OK == kOutE4M3 || OK == kOutE5M2;       // per-tensor scaling
  constexpr bool MXO = OK == kOutMXE4M3 || OK == kOutMXE5M2;   // MXFP8 block scaling (whole tiles only)
  constexpr int NOUT = MODE == kGatedBwd ? 2 : 1;
  constexpr bool GATED = MODE == kGatedFwd || MODE == kGatedBwd;
  constexpr bool GRAD = MODE == kDAct || MODE == kGatedBwd;
  __shared__ __attribute__((aligned(16))) uint8_t tile_t[(FP8 || MXO) ? kTile * kTile : 16];
  __shared__ float colpart[DBIAS ? 16 : 1][kTile];
  __shared__ float red[4];
  const unsigned t = threadIdx.x, cg = t & 15u, rg = t >> 4;
  FusedArgs a = a_in;
  const unsigned tiles_c = (a.cols + kTile - 1) / kTile;
  size_t tile_r = blockIdx.x / tiles_c, tile_col = blockIdx.x % tiles_c;
  if constexpr (!GATED) {
    if (a.grp.num_tensors != 0) {
      // grouped: which member, which of its tiles; then everything below sees ONE member
      unsigned mem, lt;
      Member m;
      if (!locate_tile(a.grp, blockIdx.x, mem, lt, m)) return;
      const size_t off = static_cast<size_t>(m.off), row_base = off / a.cols;
      const size_t out_esz = (FP8 || MXO) ? 1 : (a.out_dtype == kNVTEFloat32 ? 4 : 2);
      tile_r = lt / tiles_c;
      tile_col = lt % tiles_c;
      a.rows = static_cast<unsigned>(m.rows);
      a.in = static_cast<const IT *>(a.in) + off;
      if (a.grad) a.grad = static_cast<const IT *>(a.grad) + off;
      if (a.out_c) a.out_c = static_cast<char *>(a.out_c) + off * out_esz;
      if (a.out_t) a.out_t += off;
      if (a.scale) a.scale += mem;
      a.scale_inv = nullptr;  // per-member scale_inv belongs to nvte_group_compute_scale_from_amax
      if (a.mx_col) a.mx_col += off;
      if (a.sc_row) a.sc_row += row_base * a.stride_r;  // the member's GLOBAL rows (compact and swizzled alike)
      if (a.sc_col) {
        if (a.tcol) {
          a.sc_col += static_cast<size_t>(a.pad_cols) * (row_base / 32);
          a.stride_c = static_cast<unsigned>((m.rows / 32 + 3) / 4 * 4);
        } else {
          a.sc_col += (row_base / 32) * a.stride_c;
        }
      }
      // dbias slabs are numbered by the global tile row (unique for any member sizes)
      if (a.dbias_part) a.dbias_part += (blockIdx.x / tiles_c - tile_r) * static_cast<size_t>(a.cols);
    }
  }
  const size_t row0 = tile_r * kTile + rg * 8, col0 = tile_col * kTile + cg * 8;
  const bool col_ok = col0 < a.cols;
  const size_t col_ld = col_ok ? col0 : 0;  // out-of-tile threads load a valid address and drop the value
  const float scale = (FP8 && a.scale) ? *a.scale : 1.f;
  const IT *in = static_cast<const IT *>(a.in), *grad = static_cast<const IT *>(a.grad);
  Raw8<IT> r0[8], r1[GATED ? 8 : 1], rg8[GRAD ? 8 : 1];
  auto load_tile = [&] {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t r = min(row0 + i, static_cast<size_t>(a.rows - 1));
      r0[i].load(in + r * a.ld_in + col_ld);
      if constexpr (GATED) r1[i].load(in + r * a.ld_in + a.cols + col_ld);
      if constexpr (GRAD) rg8[i].load(grad + r * a.cols + col_ld);
    }
  };
  if constexpr (!MXO) load_tile();
  float amax = 0.f;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint32_t q[NOUT][FP8 ? 8 : 1][2];
  if constexpr (MXO) {
    // One output half at a time (the gated backward has two): the thread's 8x8 block of fp32 values must
    // be complete before its block scales exist, and with both halves unrolled into one body the
    // compiler keeps 270-370 VGPRs alive (one wave per SIMD: measured 485 us for dswiglu against 427 us
    // unfused).  So the halves are iterations of a REAL loop: the second one reloads the tile (L2-hot)
    // and picks its value with a select - both candidates share the sigmoid.
#pragma unroll 1
    for (int o = 0; o < NOUT; ++o) {
      load_tile();
      float f[8][8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float x0[8], x1[8], g[8];
        r0[i].unpack(x0);
        if constexpr (GATED) r1[i].unpack(x1);
        if constexpr (GRAD) rg8[i].unpack(g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float v;
          if constexpr (MODE == kUnary) v = act_p<ACT>(x0[k], a);
          if constexpr (MODE == kDAct) v = g[k] * dact_p<ACT>(x0[k], a);
          if constexpr (GATED) {
            bool pass;
            const float lin = gate_p<ACT>(x1[k], a, pass);
            if constexpr (MODE == kGatedFwd) v = act_p<ACT>(x0[k], a) * lin;
            if constexpr (MODE == kGatedBwd) {
              const float va = dact_p<ACT>(x0[k], a) * g[k] * lin;
              const float vg = pass ? g[k] * act_p<ACT>(x0[k], a) : 0.f;
              v = o == 0 ? va : vg;
            }
          }
          if constexpr (DBIAS) cs[k] += v;  // NOUT == 1 whenever DBIAS
          // the block quantize sees the value rounded to the input dtype, i.e. what the unfused
          // activation -> quantize sequence would read back (ref gated_mxfp8.cuh:277-283,
          // quantize_mxfp8.cuh:244-247; the dbias sum takes the unrounded value, as there)
          f[i][k] = round_to_input_type<IT>(v);
          amax = fmaxf(amax, __builtin_fabsf(f[i][k]));
        }
      }
      if (o && a.tcol) __syncthreads();  // the LDS image is reused
      mx_store_block<OK == kOutMXE4M3 ? Fp8::E4M3 : Fp8::E5M2>(f, a, t, tile_r, tile_col, o, tile_t);
    }
  } else {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const size_t r = row0 + i;
    const bool ok = col_ok && r < a.rows;
    float x0[8], x1[8], g[8];
    r0[i].unpack(x0);
    if constexpr (GATED) r1[i].unpack(x1);
    if constexpr (GRAD) rg8[i].unpack(g);
    float v[NOUT][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (MODE == kUnary) v[0][k] = act_p<ACT>(x0[k], a);
      if constexpr (MODE == kDAct) v[0][k] = g[k] * dact_p<ACT>(x0[k], a);
      if constexpr (GATED) {
        bool pass;
        const float lin = gate_p<ACT>(x1[k], a, pass);
        if constexpr (MODE == kGatedFwd) v[0][k] = act_p<ACT>(x0[k], a) * lin;
        if constexpr (MODE == kGatedBwd) {
          v[0][k] = dact_p<ACT>(x0[k], a) * g[k] * lin;
          v[1][k] = pass ? g[k] * act_p<ACT>(x0[k], a) : 0.f;
        }
      }
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        v[o][k] = ok ? v[o][k] : 0.f;
        if constexpr (!FP8) amax = fmaxf(amax, __builtin_fabsf(v[o][k]));   // FP8: folded into quantize8 below
      }
      if constexpr (DBIAS) cs[k] += v[0][k];
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      if constexpr (FP8) {
        // amax of the unscaled values and sat(v * scale); the clamp is decided once for the 8 values
        const u32x2 w8 = quantize8<OK == kOutE4M3 ? Fp8::E4M3 : Fp8::E5M2>(v[o], scale, amax);
        q[o][i][0] = w8[0];
        q[o][i][1] = w8[1];
        if (ok && a.out_c) {
          u32x2 w = {q[o][i][0], q[o][i][1]};
          *reinterpret_cast<u32x2 *>(static_cast<uint8_t *>(a.out_c) + r * a.ld_out + o * a.cols + col0) = w;
        }
      } else {
        if (ok) store8(a.out_c, r * a.ld_out + o * a.cols + col0, a.out_dtype, v[o]);
      }
    }
  }
  }
  if constexpr (FP8) {
    if (a.out_t) {
      const size_t rows_left = a.rows - tile_r * kTile, cols_left = a.cols - tile_col * kTile;
      const unsigned vch = static_cast<unsigned>(rows_left >= kTile ? 8 : rows_left / 16);
      const unsigned vco = static_cast<unsigned>(cols_left >= kTile ? kTile : cols_left);
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        if (o) __syncthreads();  // the LDS image is reused
        uint8_t *dst_t = a.out_t + (o * a.cols + tile_col * kTile) * static_cast<size_t>(a.rows) + tile_r * kTile;
        // whole tiles (block-uniform test) take the unguarded store: its four LDS reads issue back to back
        if (vch == 8 && vco == kTile) store_tile_transposed<false>(tile_t, q[o], t, dst_t, a.rows);
        else store_tile_transposed<true>(tile_t, q[o], t, dst_t, a.rows, vch, vco);
      }
    }
  }
  if constexpr (DBIAS) {
#pragma unroll
    for (int k = 0; k < 8; ++k) colpart[rg][cg * 8 + k] = cs[k];
    __syncthreads();
    if (t < kTile) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += colpart[r][t];
      const size_t c = tile_col * kTile + t;
      if (c < a.cols) a.dbias_part[tile_r * a.cols + c] = s;
    }
  }
  if (a.amax) {
    __syncthreads();
    amax = block_max<4>(amax, red);
    if (t == 0) atomic_max_nonneg(a.amax, amax);
  }
  if (FP8 && blockIdx.x == 0 && t == 0 && a.scale_inv) *a.scale_inv = 1.f / scale;
}

// Any shape / alignment: one element per thread step, every choice at run time.
__global__ __launch_bounds__(256) void fused_act_general_kernel(const FusedArgs a, const int mode) {
  __shared__ float red[4];
  const bool fp8 = a.out_dtype == kNVTEFloat8E4M3 || a.out_dtype == kNVTEFloat8E5M2;
  const float scale = (fp8 && a.scale) ? *a.scale : 1.f;
  const int nout = mode == kGatedBwd ? 2 : 1;
  float amax = 0.f;
  const size_t total = static_cast<size_t>(a.rows) * a.cols;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
    const size_t r = i / a.cols, c = i % a.cols;
    const float x0 = load1(a.in, r * a.ld_in + c, a.in_dtype);
    const float x1 = (mode == kGatedFwd || mode == kGatedBwd) ? load1(a.in, r * a.ld_in + a.cols + c, a.in_dtype) : 0.f;
    const float g = (mode == kDAct || mode == kGatedBwd) ? load1(a.grad, i, a.in_dtype) : 0.f;
    float v[2] = {0.f, 0.f};
    bool pass;
    const float lin = gate_r(x1, a, pass);
    switch (mode) {
      case kUnary: v[0] = act_r(x0, a); break;
      case kDAct: v[0] = g * dact_r(x0, a); break;
      case kGatedFwd: v[0] = act_r(x0, a) * lin; break;
      default: v[0] = dact_r(x0, a) * g * lin; v[1] = pass ? g * act_r(x0, a) : 0.f; break;
    }
    if (a.dbias_part) atomicAdd(a.dbias_part + c, v[0]);
    for (int o = 0; o < nout; ++o) {
      amax = fmaxf(amax, __builtin_fabsf(v[o]));
      const float y = v[o] * scale;
      if (a.out_c) store1(a.out_c, r * a.ld_out + o * a.cols + c, a.out_dtype, y);
      if (a.out_t) store1(a.out_t, (o * a.cols + c) * static_cast<size_t>(a.rows) + r, a.out_dtype, y);
    }
  }
  if (a.amax) {
    amax = block_max<4>(amax, red);
    if (threadIdx.x == 0) atomic_max_nonneg(a.amax, amax);
  }
  if (fp8 && blockIdx.x == 0 && threadIdx.x == 0 && a.scale_inv) *a.scale_inv = 1.f / scale;
}

// MXFP8 output for any shape: one 32x32 tile of the OUTPUT matrix per workgroup (the structure of
// mxfp8_general_kernel in mxfp8.hip), value by value with every choice at run time.
template <Fp8 F>
__global__ __launch_bounds__(256) void fused_act_mx_general_kernel(const FusedArgs a, const int mode) {
  __shared__ float tile[32][33];
  __shared__ float col_rcp[32];
  __shared__ float red[4];
  constexpr float rcp_max = 1.0f / Fp8Traits<F>::max_finite;
  const unsigned t = threadIdx.x, tx = t & 31u, ty = t >> 5;
  const size_t out_cols = a.ld_out;
  const size_t tiles_c = ceil_div_dev(out_cols, 32);
  const size_t tile_y = blockIdx.x / tiles_c, tile_x = blockIdx.x % tiles_c;
  const size_t r0 = tile_y * 32, c0 = tile_x * 32;
  float tmax = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const size_t r = r0 + ty + 8 * i, co = c0 + tx;
    float v = 0.f;
    if (r < a.rows && co < out_cols) {
      const size_t o = co / a.cols, c = co % a.cols;  // o = 1: the gate half of a gated backward
      const float x0 = load1(a.in, r * a.ld_in + c, a.in_dtype);
      const float x1 = (mode == kGatedFwd || mode == kGatedBwd) ? load1(a.in, r * a.ld_in + a.cols + c, a.in_dtype) : 0.f;
      const float g = (mode == kDAct || mode == kGatedBwd) ? load1(a.grad, r * a.cols + c, a.in_dtype) : 0.f;
      bool pass;
      const float lin = gate_r(x1, a, pass);
      switch (mode) {
        case kUnary: v = act_r(x0, a); break;
        case kDAct: v = g * dact_r(x0, a); break;
        case kGatedFwd: v = act_r(x0, a) * lin; break;
        default: v = o == 0 ? dact_r(x0, a) * g * lin : (pass ? g * act_r(x0, a) : 0.f); break;
      }
      if (a.dbias_part) atomicAdd(a.dbias_part + c, v);
      v = round_to(v, a.in_dtype);
    }
    tile[ty + 8 * i][tx] = v;
    tmax = fmaxf(tmax, __builtin_fabsf(v));
  }
  __syncthreads();
  if (a.out_c != nullptr) {
    uint8_t *out_row = static_cast<uint8_t *>(a.out_c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lr = ty + 8 * i;
      const size_t r = r0 + lr, c = c0 + tx;
      const float x = tile[lr][tx];
      float m = __builtin_fabsf(x);
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 32));
      const uint8_t e = float_to_e8m0(m * rcp_max);
      if (r < a.rows && c < out_cols) out_row[r * out_cols + c] = cvt1_fp8<F>(x * e8m0_rcp(e));
      if (tx == 0 && r < a.rows) {
        const size_t idx = a.swz ? mx_swizzled_idx(r, tile_x, ceil_div_dev(out_cols, 128)) : r * a.stride_r + tile_x;
        a.sc_row[idx] = e;
      }
    }
  }
  if (a.mx_col != nullptr) {
    if (t < 32) {
      float m = 0.f;
      for (int i = 0; i < 32; ++i) m = fmaxf(m, __builtin_fabsf(tile[i][t]));
      const uint8_t e = float_to_e8m0(m * rcp_max);
      col_rcp[t] = e8m0_rcp(e);
      const size_t c = c0 + t;
      if (c < out_cols) {
        const size_t idx = a.swz ? mx_swizzled_idx(c, tile_y, ceil_div_dev(a.rows, 128))
                                 : (a.tcol ? c * a.stride_c + tile_y : tile_y * a.stride_c + c);
        a.sc_col[idx] = e;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lr = ty + 8 * i;
      const size_t r = r0 + lr, c = c0 + tx;
      if (r < a.rows && c < out_cols) {
        const uint8_t q = cvt1_fp8<F>(tile[lr][tx] * col_rcp[tx]);
        if (a.tcol) a.mx_col[c * a.rows + r] = q;
        else a.mx_col[r * out_cols + c] = q;
      }
    }
  }
  if (a.amax) {
    tmax = block_max<4>(tmax, red);
    if (t == 0) atomic_max_nonneg(a.amax, tmax);
  }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename IT, int OK, int MODE, int ACT>
void launch_tile(const FusedArgs &a, bool dbias, unsigned grid, hipStream_t stream) {
  // combinations that exist: the identity only as a plain cast, the sigmoid only as the GLU gate
  constexpr bool valid = (ACT != kActNone || MODE == kUnary) &&
                         ((ACT != kActSigmoid && ACT != kActClampedSilu) || MODE == kGatedFwd || MODE == kGatedBwd);
  if constexpr (!valid) {
    NVTE_ERROR("invalid activation / mode combination");
  } else {
    if constexpr ((MODE == kUnary || MODE == kDAct) && OK != kOutHP) {
      if (dbias) {
        hipLaunchKernelGGL((fused_act_tile_kernel<IT, OK, MODE, ACT, true>), dim3(grid), dim3(256), 0, stream, a);
        return;
      }
    }
    NVTE_CHECK(!dbias, "dbias is fused for FP8 outputs of the plain / d-act modes only");
    hipLaunchKernelGGL((fused_act_tile_kernel<IT, OK, MODE, ACT, false>), dim3(grid), dim3(256), 0, stream, a);
  }
}

template <typename IT, int OK, int MODE>
void launch_tile_by_act(const FusedArgs &a, bool dbias, unsigned grid, hipStream_t stream) {
  switch (a.act) {
    case kActGelu: launch_tile<IT, OK, MODE, kActGelu>(a, dbias, grid, stream); break;
    case kActSilu: launch_tile<IT, OK, MODE, kActSilu>(a, dbias, grid, stream); break;
    case kActRelu: launch_tile<IT, OK, MODE, kActRelu>(a, dbias, grid, stream); break;
    case kActQGelu: launch_tile<IT, OK, MODE, kActQGelu>(a, dbias, grid, stream); break;
    case kActSRelu: launch_tile<IT, OK, MODE, kActSRelu>(a, dbias, grid, stream); break;
    case kActSigmoid: launch_tile<IT, OK, MODE, kActSigmoid>(a, dbias, grid, stream); break;
    case kActClampedSilu: launch_tile<IT, OK, MODE, kActClampedSilu>(a, dbias, grid, stream); break;
    default: launch_tile<IT, OK, MODE, kActNone>(a, dbias, grid, stream); break;
  }
}

template <typename IT, int OK>
void launch_tile_by_mode(int mode, const FusedArgs &a, bool dbias, unsigned grid, hipStream_t stream) {
  switch (mode) {
    case kUnary: launch_tile_by_act<IT, OK, kUnary>(a, dbias, grid, stream); break;
    case kDAct: launch_tile_by_act<IT, OK, kDAct>(a, dbias, grid, stream); break;
    case kGatedFwd: launch_tile_by_act<IT, OK, kGatedFwd>(a, dbias, grid, stream); break;
    default: launch_tile_by_act<IT, OK, kGatedBwd>(a, dbias, grid, stream); break;
  }
}

template <int OK>
void launch_tile_by_input(int mode, const FusedArgs &a, bool dbias, unsigned grid, hipStream_t stream) {
  switch (a.in_dtype) {
    case kNVTEBFloat16: launch_tile_by_mode<bf16_t, OK>(mode, a, dbias, grid, stream); break;
    case kNVTEFloat16: launch_tile_by_mode<f16_t, OK>(mode, a, dbias, grid, stream); break;
    default: launch_tile_by_mode<float, OK>(mode, a, dbias, grid, stream); break;
  }
}

// ---------------------------------------------------------------- grouped forms (nvte_group_*)
// dbias[i, c] = sum over member i's slabs of part[slab, c]; slabs are numbered by the global tile row, member
// i's first slab is the number of tile rows of the members before it.  One workgroup per (member, 256 columns).
__global__ __launch_bounds__(256) void group_dbias_reduce_kernel(const float *__restrict__ part, void *__restrict__ dbias,
                                                                 int out_dtype, const GroupShapes g, unsigned cols) {
  const unsigned mem = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  size_t first = 0;
  for (unsigned j = 0; j < mem; ++j) first += static_cast<size_t>((member_of(g, j).rows + kTile - 1) / kTile);
  const size_t n = static_cast<size_t>((member_of(g, mem).rows + kTile - 1) / kTile);
  if (c >= cols) return;
  float s = 0.f;
  for (size_t k = 0; k < n; ++k) s += part[(first + k) * cols + c];
  store1(dbias, static_cast<size_t>(mem) * cols + c, out_dtype, s);
}

}  // namespace

// ref cast/dispatch/quantize.cuh:430-540 (group_quantize_fwd_helper / group_quantize_bwd_helper): unary
// activation or d-activation of every member, then the cast the output's scaling mode asks for - FP8
// per-tensor scaling with output.scale[i] per member (amax / scale_inv untouched, as nvte_group_quantize)
// or MXFP8 - and optionally dbias[i, :] = column sums of member i.  Same-shape and varying-first-dim groups.
void fused_act_group_launch(const char *api, int mode, int act, const GroupedTensor *grad, const GroupedTensor &in,
                            GroupedTensor &out, GroupedTensor *dbias, Tensor *workspace,
                            const NVTEQuantizationConfig quant_config, hipStream_t stream) {
  (void)quant_config;
  check_grouped_input(in, "input");
  NVTE_CHECK(in.has_data() && is_high_precision(in.data.dtype), api, ": input must be FP32/BF16/FP16 row-wise data");
  NVTE_CHECK(in.num_tensors == out.num_tensors, api, ": number of input and output tensors must be same");
  NVTE_CHECK(out.all_same_last_dim() && in.all_same_last_dim(), api,
             ": implemented for same-shape and varying-first-dimension grouped tensors");
  const size_t R = out.logical_shape.data[0], C = out.logical_shape.data[1], T = out.num_tensors;
  NVTE_CHECK(in.logical_shape.data[0] == R && in.logical_shape.data[1] == C, api, ": input and output shapes differ");
  NVTE_CHECK(T <= kMaxMembers, api, ": at most ", kMaxMembers, " member tensors are supported");
  if (grad) {
    check_grouped_input(*grad, "grad");
    NVTE_CHECK(grad->has_data() && grad->data.dtype == in.data.dtype, api, ": grad and input must have the same dtype");
    NVTE_CHECK(grad->logical_shape.data[0] == R && grad->logical_shape.data[1] == C && grad->num_tensors == T, api,
               ": grad and input shapes differ");
  }
  const bool mx = is_mxfp8(out.scaling_mode);
  const size_t tile_rows = out.all_same_shape() ? T * ceil_div(R / T, kTile) : ceil_div(R, kTile) + T;
  if (dbias) {
    NVTE_CHECK(workspace != nullptr, api, ": dbias needs a workspace tensor");
    if (workspace->data.ptr == nullptr) {  // ref two-call protocol (cast.h:157-158)
      workspace->data.shape = NVTEShape{{tile_rows, C}, 2};
      workspace->data.dtype = kNVTEFloat32;
      return;
    }
    NVTE_CHECK(dbias->has_data() && dbias->data.dtype == in.data.dtype, api, ": DBias must have the same type as input.");
    NVTE_CHECK(dbias->data.numel() == T * C, api, ": Wrong shape of DBias (expected ", T, "x", C, ")");
    NVTE_CHECK(workspace->data.dtype == kNVTEFloat32 && workspace->data.numel() >= tile_rows * C, api,
               ": Workspace is too small (need ", tile_rows * C, " fp32 entries)");
  }
  check_grouped_output(out, "output", false);
  NVTE_CHECK(mx || (is_tensor_scaling(out.scaling_mode) && is_fp8(out.dtype())), api,
             ": FP8 per-tensor scaling and MXFP8 grouped outputs are implemented, got ", scaling_mode_name(out.scaling_mode));
  NVTE_CHECK(R < (1ull << 31) && C < (1ull << 31), "tensor too large");
  if (R == 0 || C == 0) return;
  const GroupShapes g = group_shapes_of(out);
  if (!out.all_same_shape()) NVTE_CHECK(g.offsets && g.first_dims, api, ": varying first dims need first_dims and tensor_offsets");
  else NVTE_CHECK(R % T == 0, api, ": first logical dimension must be divisible by num_tensors");

  FusedArgs a;
  std::memset(&a, 0, sizeof(a));
  a.in = in.data.ptr;
  a.grad = grad ? grad->data.ptr : nullptr;
  a.dbias_part = dbias ? static_cast<float *>(workspace->data.ptr) : nullptr;
  a.rows = static_cast<unsigned>(R);
  a.cols = static_cast<unsigned>(C);
  a.ld_in = a.ld_out = static_cast<unsigned>(C);
  a.in_dtype = in.data.dtype;
  a.out_dtype = out.dtype();
  a.act = act;
  a.grp = g;
  NVTE_REQUIRE_DATA(in.data, "grouped activation input");
  const size_t tiles = tile_rows * ceil_div(C, kTile);
  NVTE_CHECK(tiles < (1ull << 31), "grouped tensor too large");
  (void)device_info();
  if (mx) {
    NVTE_CHECK(R % 128 == 0 && C % 128 == 0, api, ": grouped MXFP8 tensors need 128-multiple dims, got ", R, "x", C);
    if (out.all_same_shape()) NVTE_CHECK((R / T) % 128 == 0, api, ": member first dim must be a multiple of 128");
    a.swz = out.with_gemm_swizzled_scales ? 1 : 0;
    a.tcol = (out.has_columnwise_data() && out.columnwise_transposed) ? 1 : 0;
    a.pad_cols = static_cast<unsigned>(round_up(C, 128));
    if (out.has_data()) {
      a.out_c = out.data.ptr;
      a.sc_row = static_cast<uint8_t *>(out.scale_inv.ptr);
      a.stride_r = static_cast<unsigned>(round_up(C / 32, 4));
      NVTE_CHECK(a.out_c && a.sc_row && out.scale_inv.numel() >= round_up(R, 128) * a.stride_r, api,
                 ": row-wise MXFP8 output or its scale_inv is missing / too small");
    }
    if (out.has_columnwise_data()) {
      NVTE_CHECK(a.tcol || !a.swz, api, ": swizzled column-wise scales of a grouped output need the transposed layout");
      a.mx_col = static_cast<uint8_t *>(out.columnwise_data.ptr);
      a.sc_col = static_cast<uint8_t *>(out.columnwise_scale_inv.ptr);
      a.stride_c = a.pad_cols;  // reference layout: [sum first / 32, pad128(cols)]
      NVTE_CHECK(a.mx_col && a.sc_col && out.columnwise_scale_inv.numel() >= round_up(R / 32, 4) * a.pad_cols, api,
                 ": column-wise MXFP8 output or its scale_inv is missing / too small");
    }
    NVTE_CHECK(aligned16(a.in) && (!a.grad || aligned16(a.grad)) && (!a.out_c || aligned16(a.out_c)) &&
                   (!a.mx_col || aligned16(a.mx_col)) && (!a.sc_row || aligned16(a.sc_row)) && (!a.sc_col || aligned16(a.sc_col)),
               api, ": grouped MXFP8 needs 16-byte aligned buffers");
    if (a.out_dtype == kNVTEFloat8E4M3) launch_tile_by_input<kOutMXE4M3>(mode, a, dbias != nullptr, static_cast<unsigned>(tiles), stream);
    else launch_tile_by_input<kOutMXE5M2>(mode, a, dbias != nullptr, static_cast<unsigned>(tiles), stream);
  } else {
    a.out_c = out.data.ptr;
    a.out_t = static_cast<uint8_t *>(out.columnwise_data.ptr);
    a.scale = static_cast<const float *>(out.scale.ptr);
    NVTE_CHECK(a.scale != nullptr && out.scale.numel() >= T, api, ": grouped FP8 output needs one scale per member");
    NVTE_CHECK(a.out_c || a.out_t, api, ": output has shapes but no data pointers");
    // vector path only (members padded to 16 rows for the transposes, as nvte_group_quantize's fast path)
    NVTE_CHECK(C % 8 == 0 && aligned16(a.in) && (!a.grad || aligned16(a.grad)) && (!a.out_c || aligned16(a.out_c)) &&
                   (!a.out_t || aligned16(a.out_t)),
               api, ": grouped FP8 activation needs a last dim multiple of 8 and 16-byte aligned buffers");
    if (a.out_dtype == kNVTEFloat8E4M3) launch_tile_by_input<kOutE4M3>(mode, a, dbias != nullptr, static_cast<unsigned>(tiles), stream);
    else launch_tile_by_input<kOutE5M2>(mode, a, dbias != nullptr, static_cast<unsigned>(tiles), stream);
  }
  NVTE_CHECK_HIP(hipGetLastError());
  if (dbias) {
    hipLaunchKernelGGL(group_dbias_reduce_kernel, dim3(static_cast<unsigned>(T), static_cast<unsigned>(ceil_div(C, 256))),
                       dim3(256), 0, stream, a.dbias_part, dbias->data.ptr, static_cast<int>(dbias->data.dtype), g,
                       static_cast<unsigned>(C));
    NVTE_CHECK_HIP(hipGetLastError());
  }
}

namespace {
}  // namespace

// mode / act: the enums above.  grad may be null (unary, gated forward); dbias + workspace are both
// null or both given (two-call protocol: an empty workspace only reports its shape).
struct ClampParams {
  float limit = 0.f, alpha = 0.f, glu_offset = 0.f;
};

void fused_act_launch(int mode, int act, const Tensor *grad, const Tensor &in, Tensor &out, Tensor *dbias,
                      Tensor *workspace, hipStream_t stream, const ClampParams &cp = ClampParams()) {
  const bool gated = mode == kGatedFwd || mode == kGatedBwd;
  const size_t rows = in.rows();
  NVTE_CHECK(!gated || in.cols() % 2 == 0, "Gated activation input must have an even last dimension, got ", in.cols());
  const size_t H = gated ? in.cols() / 2 : in.cols();
  const size_t out_cols = mode == kGatedBwd ? 2 * H : H;
  check_input(in, "input");
  NVTE_CHECK(is_high_precision(in.data.dtype), "Activation input must be FP32/BF16/FP16");
  if (grad) {
    check_input(*grad, "grad");
    NVTE_CHECK(grad->data.dtype == in.data.dtype, "grad and input must have the same dtype");
    NVTE_CHECK(grad->rows() == rows && grad->cols() == H, "grad has wrong shape (expected ", rows, "x", H, ")");
  }
  constexpr size_t kSlab = 128;
  if (dbias) {
    NVTE_CHECK(workspace != nullptr, "dbias needs a workspace tensor");
    if (workspace->data.ptr == nullptr) {
      workspace->data.shape = NVTEShape{{ceil_div(rows, kSlab), H}, 2};
      workspace->data.dtype = kNVTEFloat32;
      return;
    }
    NVTE_CHECK(dbias->data.dtype == in.data.dtype, "DBias must have the same type as input.");
    NVTE_CHECK(dbias->data.shape.ndim == 1 && dbias->data.shape.data[0] == H, "Wrong shape of DBias.");
    NVTE_CHECK(dbias->data.ptr != nullptr, "DBias is not allocated.");
    NVTE_CHECK(workspace->data.dtype == kNVTEFloat32 && workspace->data.numel() >= ceil_div(rows, kSlab) * H,
               "Workspace is too small (need ", ceil_div(rows, kSlab) * H, " fp32 entries)");
  }
  check_output(out, "output");
  NVTE_CHECK(is_tensor_scaling(out.scaling_mode) || is_mxfp8(out.scaling_mode),
             "fused activation + cast: high-precision, FP8 per-tensor-scaling and MXFP8 outputs are implemented, got ",
             scaling_mode_name(out.scaling_mode));
  const DType odt = out.dtype();
  if (is_mxfp8(out.scaling_mode)) {
    NVTE_CHECK(is_fp8(odt), "MXFP8 output needs an FP8 dtype, got ", dtype_name(odt));
    NVTE_CHECK(out.has_data() || out.has_columnwise_data(), "Output is not allocated");
    NVTE_CHECK(rows < (1ull << 31) && out_cols < (1ull << 31), "tensor too large");
    FusedArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = in.data.ptr;
    a.grad = grad ? grad->data.ptr : nullptr;
    a.dbias_part = dbias ? static_cast<float *>(workspace->data.ptr) : nullptr;
    a.amax = static_cast<float *>(out.amax.ptr);
    a.rows = static_cast<unsigned>(rows);
    a.cols = static_cast<unsigned>(H);
    a.ld_in = static_cast<unsigned>(in.cols());
    a.ld_out = static_cast<unsigned>(out_cols);
    a.in_dtype = in.data.dtype;
    a.out_dtype = odt;
    a.act = act;
    a.limit = cp.limit;
    a.alpha = cp.alpha;
    a.glu_offset = cp.glu_offset;
    a.swz = out.with_gemm_swizzled_scales ? 1 : 0;
    a.tcol = out.columnwise_transposed ? 1 : 0;
    if (out.has_data()) {
      NVTE_CHECK(out.data.numel() == rows * out_cols, "Output has wrong size (expected ", rows, "x", out_cols, ")");
      NVTE_CHECK(out.scale_inv.ptr != nullptr && out.scale_inv.shape.ndim == 2 &&
                     out.scale_inv.shape.data[0] >= rows && out.scale_inv.shape.data[1] >= ceil_div(out_cols, 32),
                 "Row-wise scale_inv of the MXFP8 output is missing or too small");
      a.out_c = out.data.ptr;
      a.sc_row = static_cast<uint8_t *>(out.scale_inv.ptr);
      a.stride_r = static_cast<unsigned>(out.scale_inv.shape.data[1]);
    }
    if (out.has_columnwise_data()) {
      NVTE_CHECK(out.columnwise_data.numel() == rows * out_cols, "Column-wise output has wrong size");
      const NVTEShape &cs = out.columnwise_scale_inv.shape;
      NVTE_CHECK(out.columnwise_scale_inv.ptr != nullptr && cs.ndim == 2, "Column-wise scale_inv of the MXFP8 output is missing");
      if (a.tcol)
        NVTE_CHECK(cs.data[0] >= out_cols && cs.data[1] >= ceil_div(rows, 32), "Column-wise scale_inv is too small");
      else
        NVTE_CHECK(cs.data[0] >= ceil_div(rows, 32) && cs.data[1] >= out_cols, "Column-wise scale_inv is too small");
      a.mx_col = static_cast<uint8_t *>(out.columnwise_data.ptr);
      a.sc_col = static_cast<uint8_t *>(out.columnwise_scale_inv.ptr);
      a.stride_c = static_cast<unsigned>(cs.data[1]);
    }
    if (a.swz) {
      // the swizzled layout addresses whole 128 x 4 scale tiles: the arrays must be padded as the reference pads them
      if (a.sc_row) NVTE_CHECK(out.scale_inv.shape.data[0] % 128 == 0 && a.stride_r % 4 == 0, "swizzled row-wise scales need (128, 4) padding");
      if (a.sc_col) NVTE_CHECK(out.columnwise_scale_inv.numel() >= round_up(out_cols, 128) * round_up(ceil_div(rows, 32), 4),
                               "swizzled column-wise scales need (128, 4) padding");
    }
    if (rows == 0 || H == 0) return;
    NVTE_REQUIRE_DATA(in.data, "activation input");
    if (grad) NVTE_REQUIRE_DATA(grad->data, "grad");
    if (out.has_data()) NVTE_REQUIRE_DATA(out.data, "output");
    if (out.has_columnwise_data()) NVTE_REQUIRE_DATA(out.columnwise_data, "column-wise output");
    (void)device_info();
    if (a.amax) zero_async(a.amax, sizeof(float), stream);
    const size_t slabs = ceil_div(rows, kSlab);
    const bool whole = rows % kTile == 0 && H % kTile == 0 && aligned16(a.in) && (!a.grad || aligned16(a.grad)) &&
                       (!a.out_c || aligned16(a.out_c)) && (!a.mx_col || aligned16(a.mx_col)) &&
                       (!a.sc_row || (reinterpret_cast<uintptr_t>(a.sc_row) & 3u) == 0) &&
                       (!a.sc_col || (reinterpret_cast<uintptr_t>(a.sc_col) & 7u) == 0) &&
                       (a.stride_r % 4 == 0) && (a.tcol || a.swz || a.stride_c % 8 == 0);
    if (whole) {
      const size_t tiles = slabs * (H / kTile);
      NVTE_CHECK(tiles < (1ull << 31), "tensor too large");
      if (odt == kNVTEFloat8E4M3) launch_tile_by_input<kOutMXE4M3>(mode, a, dbias != nullptr, static_cast<unsigned>(tiles), stream);
      else launch_tile_by_input<kOutMXE5M2>(mode, a, dbias != nullptr, static_cast<unsigned>(tiles), stream);
    } else {
      if (dbias) zero_async(a.dbias_part, slabs * H * sizeof(float), stream);
      const size_t tiles = ceil_div(rows, 32) * ceil_div(out_cols, 32);
      NVTE_CHECK(tiles < (1ull << 31), "tensor too large");
      if (odt == kNVTEFloat8E4M3)
        hipLaunchKernelGGL(fused_act_mx_general_kernel<Fp8::E4M3>, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, stream, a, mode);
      else
        hipLaunchKernelGGL(fused_act_mx_general_kernel<Fp8::E5M2>, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, stream, a, mode);
    }
    NVTE_CHECK_HIP(hipGetLastError());
    if (dbias) dbias_finalize_launch(a.dbias_part, *dbias, slabs, H, stream);
    return;
  }
  const bool fp8 = is_fp8(odt);
  NVTE_CHECK(fp8 || is_high_precision(odt), "Unsupported output dtype ", dtype_name(odt));
  NVTE_CHECK(fp8 || out.has_data(), "High-precision output needs row-wise data");
  NVTE_CHECK(out.has_data() || out.has_columnwise_data(), "Output is not allocated");
  if (out.has_data()) NVTE_CHECK(out.data.numel() == rows * out_cols, "Output has wrong size (expected ", rows, "x", out_cols, ")");
  if (out.has_columnwise_data()) {
    NVTE_CHECK(fp8, "A transposed output needs an FP8 dtype");
    NVTE_CHECK(out.columnwise_data.numel() == rows * out_cols, "Transposed output has wrong size");
  }
  NVTE_CHECK(rows < (1ull << 31) && out_cols < (1ull << 31), "tensor too large");
  if (rows == 0 || H == 0) return;
  NVTE_REQUIRE_DATA(in.data, "activation input");
  if (grad) NVTE_REQUIRE_DATA(grad->data, "grad");
  if (out.has_data()) NVTE_REQUIRE_DATA(out.data, "output");
  if (out.has_columnwise_data()) NVTE_REQUIRE_DATA(out.columnwise_data, "column-wise output");
  (void)device_info();

  FusedArgs a;
  std::memset(&a, 0, sizeof(a));
  a.in = in.data.ptr;
  a.grad = grad ? grad->data.ptr : nullptr;
  a.out_c = out.data.ptr;
  a.out_t = static_cast<uint8_t *>(out.columnwise_data.ptr);
  a.scale = fp8 ? static_cast<const float *>(out.scale.ptr) : nullptr;
  a.amax = static_cast<float *>(out.amax.ptr);
  a.scale_inv = fp8 ? static_cast<float *>(out.scale_inv.ptr ? out.scale_inv.ptr : out.columnwise_scale_inv.ptr) : nullptr;
  a.dbias_part = dbias ? static_cast<float *>(workspace->data.ptr) : nullptr;
  a.rows = static_cast<unsigned>(rows);
  a.cols = static_cast<unsigned>(H);
  a.ld_in = static_cast<unsigned>(in.cols());
  a.ld_out = static_cast<unsigned>(out_cols);
  a.in_dtype = in.data.dtype;
  a.out_dtype = odt;
  a.act = act;
  a.limit = cp.limit;
  a.alpha = cp.alpha;
  a.glu_offset = cp.glu_offset;

  const size_t slabs = ceil_div(rows, kSlab);
  // plain cast + transpose + dbias on whole tiles: the tuned cast_transpose tile kernel itself
  if (mode == kUnary && act == kActNone && dbias && fp8 &&
      cast_transpose_dbias_tile_launch(in, out, a.dbias_part, stream)) {
    dbias_finalize_launch(a.dbias_part, *dbias, slabs, H, stream);
    return;
  }
  const bool vec = !(dbias && !fp8) && H % 8 == 0 && aligned16(a.in) && (!a.grad || aligned16(a.grad)) && (!a.out_c || aligned16(a.out_c)) &&
                   (!a.out_t || (aligned16(a.out_t) && rows % 16 == 0)) &&
                   (fp8 || (H * dtype_bits(odt) / 8) % 16 == 0);
  if (vec) {
    const size_t tiles = slabs * ceil_div(H, kTile);
    NVTE_CHECK(tiles < (1ull << 31), "tensor too large");
    const unsigned grid = static_cast<unsigned>(tiles);
    if (odt == kNVTEFloat8E4M3) launch_tile_by_input<kOutE4M3>(mode, a, dbias != nullptr, grid, stream);
    else if (odt == kNVTEFloat8E5M2) launch_tile_by_input<kOutE5M2>(mode, a, dbias != nullptr, grid, stream);
    else launch_tile_by_input<kOutHP>(mode, a, dbias != nullptr, grid, stream);
  } else {
    if (dbias) zero_async(a.dbias_part, slabs * H * sizeof(float), stream);
    const unsigned grid = static_cast<unsigned>(std::min<size_t>(ceil_div(rows * H, 256), static_cast<size_t>(device_info().num_cus) * 16));
    hipLaunchKernelGGL(fused_act_general_kernel, dim3(grid), dim3(256), 0, stream, a, mode);
  }
  NVTE_CHECK_HIP(hipGetLastError());
  if (dbias) dbias_finalize_launch(a.dbias_part, *dbias, slabs, H, stream);
}

// nvte_quantize_dbias / nvte_cast_transpose_dbias for FP8 per-tensor-scaling outputs: same kernel with
// the identity "activation" (the cast and the column sums share ONE read of the input)
void fused_cast_dbias_launch(const Tensor &in, Tensor &out, Tensor &dbias, Tensor &workspace, hipStream_t stream) {
  fused_act_launch(kUnary, kActNone, nullptr, in, out, &dbias, &workspace, stream);
}

}  // namespace te

using namespace te;

extern "C" {

#define TE_UNARY(name, act_id)                                                                          \
  void name(const NVTETensor input, NVTETensor output, hipStream_t stream) {                            \
    api_guard(#name, [&] {                                                                              \
      fused_act_launch(kUnary, act_id, nullptr, *lookup_checked(input), *lookup_checked(output), nullptr, \
                       nullptr, stream);                                                                \
    });                                                                                                 \
  }
#define TE_DACT(name, act_id)                                                                           \
  void name(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream) {     \
    api_guard(#name, [&] {                                                                              \
      fused_act_launch(kDAct, act_id, lookup_checked(grad), *lookup_checked(input), *lookup_checked(output), \
                       nullptr, nullptr, stream);                                                       \
    });                                                                                                 \
  }
#define TE_GATED(name, act_id)                                                                          \
  void name(const NVTETensor input, NVTETensor output, hipStream_t stream) {                            \
    api_guard(#name, [&] {                                                                              \
      fused_act_launch(kGatedFwd, act_id, nullptr, *lookup_checked(input), *lookup_checked(output), nullptr, \
                       nullptr, stream);                                                                \
    });                                                                                                 \
  }
#define TE_DGATED(name, act_id)                                                                         \
  void name(const NVTETensor grad, const NVTETensor input, NVTETensor output, hipStream_t stream) {     \
    api_guard(#name, [&] {                                                                              \
      fused_act_launch(kGatedBwd, act_id, lookup_checked(grad), *lookup_checked(input),                 \
                       *lookup_checked(output), nullptr, nullptr, stream);                              \
    });                                                                                                 \
  }
#define TE_DBIAS_DACT(name, act_id)                                                                     \
  void name(const NVTETensor input, const NVTETensor act_input, NVTETensor output, NVTETensor dbias,    \
            NVTETensor workspace, hipStream_t stream) {                                                 \
    api_guard(#name, [&] {                                                                              \
      fused_act_launch(kDAct, act_id, lookup_checked(input), *lookup_checked(act_input),                \
                       *lookup_checked(output), lookup_checked(dbias), lookup_checked(workspace), stream); \
    });                                                                                                 \
  }

TE_UNARY(nvte_gelu, kActGelu)
TE_UNARY(nvte_silu, kActSilu)
TE_UNARY(nvte_relu, kActRelu)
TE_UNARY(nvte_qgelu, kActQGelu)
TE_UNARY(nvte_srelu, kActSRelu)
TE_DACT(nvte_dgelu, kActGelu)
TE_DACT(nvte_dsilu, kActSilu)
TE_DACT(nvte_drelu, kActRelu)
TE_DACT(nvte_dqgelu, kActQGelu)
TE_DACT(nvte_dsrelu, kActSRelu)
TE_GATED(nvte_glu, kActSigmoid)
TE_GATED(nvte_geglu, kActGelu)
TE_GATED(nvte_swiglu, kActSilu)
TE_GATED(nvte_reglu, kActRelu)
TE_GATED(nvte_qgeglu, kActQGelu)
TE_GATED(nvte_sreglu, kActSRelu)
TE_DGATED(nvte_dglu, kActSigmoid)
TE_DGATED(nvte_dgeglu, kActGelu)
TE_DGATED(nvte_dswiglu, kActSilu)
TE_DGATED(nvte_dreglu, kActRelu)
TE_DGATED(nvte_dqgeglu, kActQGelu)
TE_DGATED(nvte_dsreglu, kActSRelu)
TE_DBIAS_DACT(nvte_quantize_dbias_dgelu, kActGelu)
TE_DBIAS_DACT(nvte_quantize_dbias_dsilu, kActSilu)
TE_DBIAS_DACT(nvte_quantize_dbias_drelu, kActRelu)
TE_DBIAS_DACT(nvte_quantize_dbias_dqgelu, kActQGelu)
TE_DBIAS_DACT(nvte_quantize_dbias_dsrelu, kActSRelu)
TE_DBIAS_DACT(nvte_cast_transpose_dbias_dgelu, kActGelu)
TE_DBIAS_DACT(nvte_cast_transpose_dbias_dsilu, kActSilu)
TE_DBIAS_DACT(nvte_cast_transpose_dbias_drelu, kActRelu)
TE_DBIAS_DACT(nvte_cast_transpose_dbias_dqgelu, kActQGelu)
TE_DBIAS_DACT(nvte_cast_transpose_dbias_dsrelu, kActSRelu)

// ref activation.h:346-370,487-512: the GPT-OSS gated Swish (clamped pre-activation and gate, sigmoid(alpha x));
// the un-suffixed forms are the deprecated ones with glu_linear_offset = 1
void nvte_clamped_swiglu_v2(const NVTETensor input, NVTETensor output, float limit, float alpha,
                            float glu_linear_offset, hipStream_t stream) {
  api_guard("nvte_clamped_swiglu_v2", [&] {
    fused_act_launch(kGatedFwd, kActClampedSilu, nullptr, *lookup_checked(input), *lookup_checked(output), nullptr,
                     nullptr, stream, ClampParams{limit, alpha, glu_linear_offset});
  });
}
void nvte_clamped_swiglu(const NVTETensor input, NVTETensor output, float limit, float alpha, hipStream_t stream) {
  nvte_clamped_swiglu_v2(input, output, limit, alpha, 1.0f, stream);
}
void nvte_clamped_dswiglu_v2(const NVTETensor grad, const NVTETensor input, NVTETensor output, float limit,
                             float alpha, float glu_linear_offset, hipStream_t stream) {
  api_guard("nvte_clamped_dswiglu_v2", [&] {
    fused_act_launch(kGatedBwd, kActClampedSilu, lookup_checked(grad), *lookup_checked(input),
                     *lookup_checked(output), nullptr, nullptr, stream, ClampParams{limit, alpha, glu_linear_offset});
  });
}
void nvte_clamped_dswiglu(const NVTETensor grad, const NVTETensor input, NVTETensor output, float limit,
                          float alpha, hipStream_t stream) {
  nvte_clamped_dswiglu_v2(grad, input, output, limit, alpha, 1.0f, stream);
}

// ref transpose.h:242-318: (input = grad, act_input, output)
#define TE_DGATED_CT(name, act_id)                                                                      \
  void name(const NVTETensor input, const NVTETensor act_input, NVTETensor output, hipStream_t stream) { \
    api_guard(#name, [&] {                                                                              \
      fused_act_launch(kGatedBwd, act_id, lookup_checked(input), *lookup_checked(act_input),            \
                       *lookup_checked(output), nullptr, nullptr, stream);                              \
    });                                                                                                 \
  }
TE_DGATED_CT(nvte_dgeglu_cast_transpose, kActGelu)
TE_DGATED_CT(nvte_dswiglu_cast_transpose, kActSilu)
TE_DGATED_CT(nvte_dreglu_cast_transpose, kActRelu)
TE_DGATED_CT(nvte_dqgeglu_cast_transpose, kActQGelu)
TE_DGATED_CT(nvte_dsreglu_cast_transpose, kActSRelu)

// ---- grouped forms: ref activation.h:65-149 (forward), :173-275 (backward), cast.h:167-410 (dbias [+ d-act])
#define TE_GROUP_UNARY(name, act_id)                                                                       \
  void name(const NVTEGroupedTensor input, NVTEGroupedTensor output, hipStream_t stream) {                 \
    api_guard(#name, [&] {                                                                                 \
      fused_act_group_launch(#name, kUnary, act_id, nullptr, *lookup_grouped_checked(input),               \
                             *lookup_grouped_checked(output), nullptr, nullptr, nullptr, stream);          \
    });                                                                                                    \
  }
#define TE_GROUP_DACT(name, act_id)                                                                        \
  void name(const NVTEGroupedTensor grad, const NVTEGroupedTensor input, NVTETensor output, hipStream_t stream) { \
    api_guard(#name, [&] {                                                                                 \
      fused_act_group_launch(#name, kDAct, act_id, lookup_grouped_checked(grad), *lookup_grouped_checked(input), \
                             *lookup_grouped_checked(output), nullptr, nullptr, nullptr, stream);          \
    });                                                                                                    \
  }
#define TE_GROUP_DBIAS_DACT(name, act_id)                                                                  \
  void name(const NVTEGroupedTensor input, const NVTEGroupedTensor act_input, NVTEGroupedTensor output,    \
            NVTEGroupedTensor dbias, NVTETensor workspace, hipStream_t stream) {                           \
    api_guard(#name, [&] {                                                                                 \
      fused_act_group_launch(#name, kDAct, act_id, lookup_grouped_checked(input), *lookup_grouped_checked(act_input), \
                             *lookup_grouped_checked(output), lookup_grouped_checked(dbias),               \
                             lookup_checked(workspace), nullptr, stream);                                  \
    });                                                                                                    \
  }
TE_GROUP_UNARY(nvte_group_gelu, kActGelu)
TE_GROUP_UNARY(nvte_group_silu, kActSilu)
TE_GROUP_UNARY(nvte_group_relu, kActRelu)
TE_GROUP_UNARY(nvte_group_qgelu, kActQGelu)
TE_GROUP_UNARY(nvte_group_srelu, kActSRelu)
TE_GROUP_DACT(nvte_group_dgelu, kActGelu)
TE_GROUP_DACT(nvte_group_dsilu, kActSilu)
TE_GROUP_DACT(nvte_group_drelu, kActRelu)
TE_GROUP_DACT(nvte_group_dqgelu, kActQGelu)
TE_GROUP_DACT(nvte_group_dsrelu, kActSRelu)
TE_GROUP_DBIAS_DACT(nvte_group_quantize_dbias_dgelu, kActGelu)
TE_GROUP_DBIAS_DACT(nvte_group_quantize_dbias_dsilu, kActSilu)
TE_GROUP_DBIAS_DACT(nvte_group_quantize_dbias_drelu, kActRelu)
TE_GROUP_DBIAS_DACT(nvte_group_quantize_dbias_dqgelu, kActQGelu)
TE_GROUP_DBIAS_DACT(nvte_group_quantize_dbias_dsrelu, kActSRelu)

void nvte_group_quantize_dbias(const NVTEGroupedTensor input, NVTEGroupedTensor output, NVTEGroupedTensor dbias,
                               NVTETensor workspace, const NVTEQuantizationConfig quant_config, hipStream_t stream) {
  api_guard("nvte_group_quantize_dbias", [&] {
    fused_act_group_launch("nvte_group_quantize_dbias", kUnary, kActNone, nullptr, *lookup_grouped_checked(input),
                           *lookup_grouped_checked(output), lookup_grouped_checked(dbias), lookup_checked(workspace),
                           quant_config, stream);
  });
}

}  // extern "C"
