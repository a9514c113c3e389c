// FP8 weight-update path: what an "FP8 primary weights" training loop calls
// each optimizer step to turn sharded fp32/bf16 master weights into the FP8 bytes + scales the
// GEMMs read.
//
// Replaces: ref common/multi_tensor/multi_tensor_apply.cuh:18-118 (the chunk scheduler),
//           common/multi_tensor/scale.cu:40-170, compute_scale.cu:26-132,
//           common/multi_tensor/adam.cu:52-176,769-857,1005-1019 (Adam with FP8 parameters),
//           common/recipe/mxfp8_scaling.cu:26-253 (partial amax / partial cast of a shard).
//
// MI355X design notes
//  * multi-tensor launches: the reference packs <= 320 (tensor, chunk) pairs per launch into
//    byte tables.  Here one launch takes up to mt_capacity(D) tensors with ANY number of chunks
//    (a 58 M-element weight is 896 workgroups, >> 256 CUs): the kernel argument holds a prefix
//    sum of chunks per tensor and each workgroup finds its tensor by a binary search on wave-
//    uniform data (scalar loads from the kernarg segment).
//  * all kernels are HBM-bound streams: 16-byte loads, every load of an iteration issued before
//    the arithmetic, FP8 bytes packed by the hardware converter.
//  * the shard kernels give each wave a 32-row x 512-column tile: a lane owns 8 consecutive
//    columns (one 16 B load per row for bf16), the 1x32 row-block maximum is a 4-lane exchange,
//    the 32x1 column-block maximum stays in registers.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "common.h"
#include "device_utils.h"
#include "transformer_engine/multi_tensor.h"
#include "transformer_engine/recipe.h"

namespace te {
namespace {

// ====================================================================================
// multi-tensor table
// ====================================================================================
constexpr int mt_capacity(int depth) { return 3400 / (8 * depth + 12); }

template <int D>
struct MtTable {
  static constexpr int N = mt_capacity(D);
  void *addr[D][N];
  int64_t size[N];
  uint32_t first_chunk[N + 1];  // prefix sum of chunks; first_chunk[count] = grid size
  int count;
};

struct MtWork {
  int tensor;
  int64_t begin;  // first element of this workgroup's chunk
  int64_t len;    // elements in it
};

template <int D>
__device__ __forceinline__ MtWork mt_locate(const MtTable<D> &t, int64_t chunk_size) {
  const uint32_t wg = blockIdx.x;
  int lo = 0, hi = t.count;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.first_chunk[mid] <= wg) lo = mid; else hi = mid;
  }
  MtWork w;
  w.tensor = lo;
  w.begin = static_cast<int64_t>(wg - t.first_chunk[lo]) * chunk_size;
  const int64_t left = t.size[lo] - w.begin;
  w.len = left < chunk_size ? left : chunk_size;
  return w;
}

using TensorLists = std::vector<std::vector<Tensor *>>;

// ref common/multi_tensor (convert_tensor_array): [num_lists][num_tensors] handles -> descriptors
TensorLists lists_of(NVTETensor **tensor_lists, size_t num_lists, size_t per_list) {
  NVTE_CHECK(tensor_lists != nullptr || num_lists == 0, "tensor_lists is NULL");
  TensorLists out(num_lists);
  for (size_t i = 0; i < num_lists; ++i) {
    NVTE_CHECK(tensor_lists[i] != nullptr || per_list == 0, "tensor list ", i, " is NULL");
    out[i].reserve(per_list);
    for (size_t j = 0; j < per_list; ++j) out[i].push_back(lookup_checked(tensor_lists[i][j]));
  }
  return out;
}

// Walks the lists in batches of MtTable<D>::N tensors.  `depth_checked` leading lists must hold
// tensors of the same element count as list 0 (the kernels index them with one offset).
template <int D, typename Launch>
void mt_run(const TensorLists &lists, int depth_checked, int64_t chunk_size, Launch &&launch) {
  NVTE_CHECK(static_cast<int>(lists.size()) == D, "Expected ", D, " tensor lists, but found ",
             lists.size());
  NVTE_CHECK(chunk_size > 0, "chunk_size must be positive, got ", chunk_size);
  const size_t n = lists[0].size();
  for (int d = 1; d < D; ++d)
    NVTE_CHECK(lists[d].size() == n, "Tensor list ", d, " has size=", lists[d].size(),
               ", but expected size=", n);
  MtTable<D> t;
  std::memset(&t, 0, sizeof(t));
  uint64_t chunks = 0;
  auto flush = [&] {
    if (t.count == 0) return;
    t.first_chunk[t.count] = static_cast<uint32_t>(chunks);
    launch(t, static_cast<unsigned>(chunks));
    std::memset(&t, 0, sizeof(t));
    chunks = 0;
  };
  for (size_t j = 0; j < n; ++j) {
    const size_t numel = lists[0][j]->data.numel();
    // ref multi_tensor_apply.cuh:78-80: zero-sized tensors must be filtered by the caller
    NVTE_CHECK(numel > 0, "multi_tensor_apply expects tensors with at least one chunk; zero-sized "
               "tensors must be filtered before launch");
    for (int d = 1; d < depth_checked; ++d)
      NVTE_CHECK(lists[d][j]->data.numel() == numel, "Tensor ", j, " of list ", d, " has ",
                 lists[d][j]->data.numel(), " elements, but list 0 has ", numel);
    // the kernels dereference these unconditionally
    for (int d = 0; d < depth_checked; ++d)
      NVTE_CHECK(lists[d][j]->data.ptr != nullptr, "Tensor ", j, " of list ", d, " has no data");
    const uint64_t c = (numel + chunk_size - 1) / chunk_size;
    if (chunks + c > 0x7FFFFFFFull) flush();
    NVTE_CHECK(c <= 0x7FFFFFFFull, "tensor ", j, " needs too many chunks; raise chunk_size");
    for (int d = 0; d < D; ++d) t.addr[d][t.count] = lists[d][j]->data.ptr;
    t.size[t.count] = static_cast<int64_t>(numel);
    t.first_chunk[t.count] = static_cast<uint32_t>(chunks);
    chunks += c;
    if (++t.count == MtTable<D>::N) flush();
  }
  flush();
}

constexpr int kMtThreads = 512;

// ====================================================================================
// out = in * scale, flag non-finite inputs (ref multi_tensor/scale.cu:40-103)
// ====================================================================================
template <typename IT, typename OT, bool SCALE_PTR>
__global__ __launch_bounds__(kMtThreads) void mt_scale_kernel(MtTable<2> t, int64_t chunk_size,
                                                              int *is_infinite, float scale_v,
                                                              const float *scale_p) {
  const MtWork w = mt_locate(t, chunk_size);
  const IT *in = static_cast<const IT *>(t.addr[0][w.tensor]) + w.begin;
  OT *out = static_cast<OT *>(t.addr[1][w.tensor]) + w.begin;
  const float scale = SCALE_PTR ? *scale_p : scale_v;
  bool finite = true;
  int64_t i = threadIdx.x;
  // four independent elements per thread and iteration
  for (; i + 3 * kMtThreads < w.len; i += 4 * kMtThreads) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = widen(in[i + k * kMtThreads]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      finite = finite && __builtin_isfinite(v[k]);
      narrow(v[k] * scale, out[i + k * kMtThreads]);
    }
  }
  for (; i < w.len; i += kMtThreads) {
    const float v = widen(in[i]);
    finite = finite && __builtin_isfinite(v);
    narrow(v * scale, out[i]);
  }
  if (!finite && is_infinite != nullptr) *is_infinite = 1;  // racing writers store the same value
}

template <typename IT, typename OT>
void mt_scale_typed(const TensorLists &lists, int64_t chunk, int *flag, float scale_v,
                    const float *scale_p, hipStream_t stream) {
  mt_run<2>(lists, 2, chunk, [&](const MtTable<2> &t, unsigned grid) {
    if (scale_p != nullptr) {
      hipLaunchKernelGGL((mt_scale_kernel<IT, OT, true>), dim3(grid), dim3(kMtThreads), 0, stream, t,
                         chunk, flag, scale_v, scale_p);
    } else {
      hipLaunchKernelGGL((mt_scale_kernel<IT, OT, false>), dim3(grid), dim3(kMtThreads), 0, stream, t,
                         chunk, flag, scale_v, scale_p);
    }
    NVTE_CHECK_HIP(hipGetLastError());
  });
}

template <typename IT>
void mt_scale_by_out(DType ot, const TensorLists &lists, int64_t chunk, int *flag, float scale_v,
                     const float *scale_p, hipStream_t stream) {
  switch (ot) {
    case kNVTEFloat32: mt_scale_typed<IT, float>(lists, chunk, flag, scale_v, scale_p, stream); break;
    case kNVTEFloat16: mt_scale_typed<IT, f16_t>(lists, chunk, flag, scale_v, scale_p, stream); break;
    case kNVTEBFloat16: mt_scale_typed<IT, bf16_t>(lists, chunk, flag, scale_v, scale_p, stream); break;
    default: NVTE_ERROR("Unsupported output dtype ", dtype_name(ot));
  }
}

void mt_scale(int chunk_size, const Tensor &is_infinite, const TensorLists &lists, float scale_v,
              const float *scale_p, hipStream_t stream) {
  NVTE_CHECK(lists.size() == 2, "Expected 2 tensor lists, but found ", lists.size());
  if (lists[0].empty()) return;
  const DType it = lists[0][0]->data.dtype, ot = lists[1][0]->data.dtype;
  for (size_t j = 0; j < lists[0].size(); ++j) {
    NVTE_CHECK(lists[0][j]->data.dtype == it, "Input tensor ", j, " has dtype=",
               dtype_name(lists[0][j]->data.dtype), ", but expected ", dtype_name(it));
    NVTE_CHECK(lists[1][j]->data.dtype == ot, "Output tensor ", j, " has dtype=",
               dtype_name(lists[1][j]->data.dtype), ", but expected ", dtype_name(ot));
  }
  int *flag = static_cast<int *>(is_infinite.data.ptr);
  switch (it) {
    case kNVTEFloat32: mt_scale_by_out<float>(ot, lists, chunk_size, flag, scale_v, scale_p, stream); break;
    case kNVTEFloat16: mt_scale_by_out<f16_t>(ot, lists, chunk_size, flag, scale_v, scale_p, stream); break;
    case kNVTEBFloat16: mt_scale_by_out<bf16_t>(ot, lists, chunk_size, flag, scale_v, scale_p, stream); break;
    default: NVTE_ERROR("Unsupported input dtype ", dtype_name(it));
  }
}

// ====================================================================================
// amax -> (scale, 1/scale) per element (ref multi_tensor/compute_scale.cu:26-57)
// ====================================================================================
__global__ __launch_bounds__(kMtThreads) void mt_scale_and_inv_kernel(MtTable<3> t, int64_t chunk_size,
                                                                      float max_fp8, bool pow2,
                                                                      float eps) {
  const MtWork w = mt_locate(t, chunk_size);
  const float *amax = static_cast<const float *>(t.addr[0][w.tensor]) + w.begin;
  float *scale = static_cast<float *>(t.addr[1][w.tensor]) + w.begin;
  float *scale_inv = static_cast<float *>(t.addr[2][w.tensor]) + w.begin;
  for (int64_t i = threadIdx.x; i < w.len; i += kMtThreads) {
    const float s = scale_from_amax(amax[i], max_fp8, pow2, eps, std::numeric_limits<float>::max());
    scale[i] = s;
    scale_inv[i] = 1.0f / s;  // IEEE division = the reference's __frcp_rn
  }
}

// bf16 amax -> e8m0 decode scale of an E4M3 block (ref compute_scale.cu:59-79)
__global__ __launch_bounds__(kMtThreads) void mt_e8m0_kernel(MtTable<2> t, int64_t chunk_size) {
  const MtWork w = mt_locate(t, chunk_size);
  const bf16_t *amax = static_cast<const bf16_t *>(t.addr[0][w.tensor]) + w.begin;
  uint8_t *scale_inv = static_cast<uint8_t *>(t.addr[1][w.tensor]) + w.begin;
  constexpr float max_norm_rcp = 1.0f / 448.0f;
  int64_t done = 0;
  if (((reinterpret_cast<uintptr_t>(amax) & 15) | (reinterpret_cast<uintptr_t>(scale_inv) & 7)) == 0) {
    const int64_t nvec = w.len >> 3;  // 16 B of amaxes -> 8 B of exponents per lane
    for (int64_t i = threadIdx.x; i < nvec; i += kMtThreads) {
      const u32x4 raw = *reinterpret_cast<const u32x4 *>(amax + 8 * i);
      uint32_t e[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        e[2 * k] = float_to_e8m0(__uint_as_float(raw[k] << 16) * max_norm_rcp);
        e[2 * k + 1] = float_to_e8m0(__uint_as_float(raw[k] & 0xFFFF0000u) * max_norm_rcp);
      }
      const u32x2 o = {e[0] | (e[1] << 8) | (e[2] << 16) | (e[3] << 24),
                       e[4] | (e[5] << 8) | (e[6] << 16) | (e[7] << 24)};
      *reinterpret_cast<u32x2 *>(scale_inv + 8 * i) = o;
    }
    done = nvec << 3;
  }
  for (int64_t i = done + threadIdx.x; i < w.len; i += kMtThreads)
    scale_inv[i] = float_to_e8m0(widen(amax[i]) * max_norm_rcp);
}

// ====================================================================================
// Adam on fp32 master weights with an FP8 model copy (ref multi_tensor/adam.cu:52-176)
// lists: g, p_fp8, m, v, p_master, scale, amax, scale_inv
// ====================================================================================
struct AdamArgs {
  float beta1, beta2, corr1, corr2, eps, lr, decay;
  int mode;  // 0 = L2 regularisation folded into the gradient, 1 = decoupled weight decay
};

// One element, written without fused multiply-adds: every operation rounds on its own, so the
// numpy oracle (tests) reproduces it bit for bit.
__device__ __forceinline__ void adam_step(const AdamArgs &a, float g, float &p, float &m, float &v) {
#pragma clang fp contract(off)
  if (a.mode == 0) {
    g = g + (a.decay * p);
    m = a.beta1 * m + (1 - a.beta1) * g;
    v = a.beta2 * v + (1 - a.beta2) * g * g;
    const float mh = m / a.corr1;
    const float vh = v / a.corr2;
    const float denom = __builtin_sqrtf(vh) + a.eps;
    const float upd = mh / denom;
    p = p - (a.lr * upd);
  } else {
    m = a.beta1 * m + (1 - a.beta1) * g;
    v = a.beta2 * v + (1 - a.beta2) * g * g;
    const float mh = m / a.corr1;
    const float vh = v / a.corr2;
    const float denom = __builtin_sqrtf(vh) + a.eps;
    const float upd = (mh / denom) + (a.decay * p);
    p = p - (a.lr * upd);
  }
}

template <typename GT>
__device__ __forceinline__ void load4(const GT *p, float (&f)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) f[k] = widen(p[k]);
}
template <>
__device__ __forceinline__ void load4<float>(const float *p, float (&f)[4]) {
  const f32x4 v = *reinterpret_cast<const f32x4 *>(p);
#pragma unroll
  for (int k = 0; k < 4; ++k) f[k] = v[k];
}
template <>
__device__ __forceinline__ void load4<bf16_t>(const bf16_t *p, float (&f)[4]) {
  const u32x2 v = *reinterpret_cast<const u32x2 *>(p);
  f[0] = __uint_as_float(v[0] << 16);
  f[1] = __uint_as_float(v[0] & 0xFFFF0000u);
  f[2] = __uint_as_float(v[1] << 16);
  f[3] = __uint_as_float(v[1] & 0xFFFF0000u);
}
template <>
__device__ __forceinline__ void load4<f16_t>(const f16_t *p, float (&f)[4]) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 v = *reinterpret_cast<const h4 *>(p);
#pragma unroll
  for (int k = 0; k < 4; ++k) f[k] = static_cast<float>(v[k]);
}

template <Fp8 F, typename GT>
__global__ __launch_bounds__(kMtThreads) void mt_adam_fp8_kernel(MtTable<8> t, int64_t chunk_size,
                                                                 AdamArgs a) {
  __shared__ float red[kMtThreads / 64];
  const MtWork w = mt_locate(t, chunk_size);
  const GT *g = static_cast<const GT *>(t.addr[0][w.tensor]) + w.begin;
  uint8_t *p8 = static_cast<uint8_t *>(t.addr[1][w.tensor]) + w.begin;
  float *m = static_cast<float *>(t.addr[2][w.tensor]) + w.begin;
  float *v = static_cast<float *>(t.addr[3][w.tensor]) + w.begin;
  float *pm = static_cast<float *>(t.addr[4][w.tensor]) + w.begin;
  const float *scale_ptr = static_cast<const float *>(t.addr[5][w.tensor]);
  float *amax_ptr = static_cast<float *>(t.addr[6][w.tensor]);
  float *scale_inv_ptr = static_cast<float *>(t.addr[7][w.tensor]);
  const float scale = scale_ptr != nullptr ? *scale_ptr : 1.f;
  float amax = 0.f;

  const bool vec = ((reinterpret_cast<uintptr_t>(g) % (4 * sizeof(GT))) | (reinterpret_cast<uintptr_t>(p8) & 3) |
                    (reinterpret_cast<uintptr_t>(m) & 15) | (reinterpret_cast<uintptr_t>(v) & 15) |
                    (reinterpret_cast<uintptr_t>(pm) & 15)) == 0;
  int64_t done = 0;
  if (vec) {
    const int64_t nvec = w.len >> 2;
    for (int64_t i = threadIdx.x; i < nvec; i += kMtThreads) {
      float rg[4], rp[4], rm[4], rv[4];
      load4<GT>(g + 4 * i, rg);
      load4<float>(pm + 4 * i, rp);
      load4<float>(m + 4 * i, rm);
      load4<float>(v + 4 * i, rv);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        adam_step(a, rg[k], rp[k], rm[k], rv[k]);
        amax = fmaxf(amax, __builtin_fabsf(rp[k]));
      }
      *reinterpret_cast<f32x4 *>(pm + 4 * i) = f32x4{rp[0], rp[1], rp[2], rp[3]};
      *reinterpret_cast<f32x4 *>(m + 4 * i) = f32x4{rm[0], rm[1], rm[2], rm[3]};
      *reinterpret_cast<f32x4 *>(v + 4 * i) = f32x4{rv[0], rv[1], rv[2], rv[3]};
      *reinterpret_cast<uint32_t *>(p8 + 4 * i) =
          cvt4_fp8<F>(rp[0] * scale, rp[1] * scale, rp[2] * scale, rp[3] * scale);
    }
    done = nvec << 2;
  }
  for (int64_t i = done + threadIdx.x; i < w.len; i += kMtThreads) {
    float rp = pm[i], rm = m[i], rv = v[i];
    adam_step(a, widen(g[i]), rp, rm, rv);
    amax = fmaxf(amax, __builtin_fabsf(rp));
    pm[i] = rp;
    m[i] = rm;
    v[i] = rv;
    p8[i] = cvt1_fp8<F>(rp * scale);
  }
  amax = block_max<kMtThreads / 64>(amax, red);
  if (threadIdx.x == 0) {
    // fmaxf drops NaN operands, as the reference's reduce_max does
    if (amax_ptr != nullptr) atomic_max_nonneg(amax_ptr, amax);
    if (scale_inv_ptr != nullptr) *scale_inv_ptr = 1.0f / scale;
  }
}

template <Fp8 F, typename GT>
void mt_adam_fp8_typed(const TensorLists &lists, int64_t chunk, const AdamArgs &a, hipStream_t stream) {
  mt_run<8>(lists, 5, chunk, [&](const MtTable<8> &t, unsigned grid) {
    hipLaunchKernelGGL((mt_adam_fp8_kernel<F, GT>), dim3(grid), dim3(kMtThreads), 0, stream, t, chunk, a);
    NVTE_CHECK_HIP(hipGetLastError());
  });
}

template <Fp8 F>
void mt_adam_fp8_by_grad(DType gt, const TensorLists &lists, int64_t chunk, const AdamArgs &a,
                         hipStream_t stream) {
  switch (gt) {
    case kNVTEFloat32: mt_adam_fp8_typed<F, float>(lists, chunk, a, stream); break;
    case kNVTEFloat16: mt_adam_fp8_typed<F, f16_t>(lists, chunk, a, stream); break;
    case kNVTEBFloat16: mt_adam_fp8_typed<F, bf16_t>(lists, chunk, a, stream); break;
    default: NVTE_ERROR("Unsupported gradient dtype ", dtype_name(gt));
  }
}

// ====================================================================================
// MXFP8 shard kernels (ref recipe/mxfp8_scaling.cu)
// ====================================================================================
constexpr int kShardRows = 32;          // one columnwise block
constexpr int kShardCols = 64 * 8;      // a wave's tile width
constexpr int kShardWaves = 4;          // tiles per workgroup, side by side

struct ShardGeom {
  int rows, cols;
  int64_t start, end;  // the shard covers flattened elements [start, end)
  int row_stride;      // second dim of the rowwise amax / scale array
  int col_stride;      // second dim of the columnwise amax / scale array
};

// 8 consecutive elements at flattened index idx (a multiple of 8) of the full matrix; elements
// outside the shard read as 0.  ALIGNED: start % 8 == 0 and the shard pointer is 16 B aligned.
template <typename IT, bool ALIGNED>
__device__ __forceinline__ void shard_load8(const IT *shard, const ShardGeom &g, int64_t idx, float (&f)[8]) {
  if (ALIGNED && idx >= g.start && idx + 8 <= g.end) {
    Load8<IT>::run(shard + (idx - g.start), f);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t e = idx + k;
      f[k] = (e >= g.start && e < g.end) ? widen(shard[e - g.start]) : 0.f;
    }
  }
}

// grid = (ceil(cols / (4*512)), rows / 32), 256 threads
template <typename IT, bool ALIGNED>
__global__ __launch_bounds__(kShardWaves * 64) void shard_amax_kernel(const IT *__restrict__ shard,
                                                                      IT *__restrict__ amax_row,
                                                                      IT *__restrict__ amax_col,
                                                                      ShardGeom g) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * kShardWaves + wave) * kShardCols + lane * 8;
  const int r0 = blockIdx.y * kShardRows;
  // written region = the reference's grid of 32 x 128 tiles: row blocks padded to 4 per tile,
  // columns padded to 128
  const int cb_written = static_cast<int>(ceil_div_dev(g.cols, 128)) * 4;
  const int c_written = cb_written * 32;
  if (c >= c_written) return;  // whole 4-lane groups leave together (c_written % 32 == 0)
  const bool live = c < g.cols;
  const int64_t tile_lo = static_cast<int64_t>(r0) * g.cols + c;
  const int64_t tile_hi = static_cast<int64_t>(r0 + kShardRows - 1) * g.cols + c + 8;
  const bool touches = live && tile_hi > g.start && tile_lo < g.end;

  float cmax[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int rb = 0; rb < kShardRows; rb += 8) {
    float x[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (touches) {
        shard_load8<IT, ALIGNED>(shard, g, static_cast<int64_t>(r0 + rb + i) * g.cols + c, x[i]);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[i][k] = 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float rmax = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float a = __builtin_fabsf(x[i][k]);
        rmax = fmaxf(rmax, a);
        cmax[k] = fmaxf(cmax[k], a);
      }
      // 1x32 block = 4 neighbouring lanes
      rmax = fmaxf(rmax, __shfl_xor(rmax, 1, 64));
      rmax = fmaxf(rmax, __shfl_xor(rmax, 2, 64));
      if ((lane & 3) == 0) {
        IT o;
        narrow(rmax, o);
        amax_row[static_cast<int64_t>(r0 + rb + i) * g.row_stride + (c >> 5)] = o;
      }
    }
  }
  IT o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) narrow(cmax[k], o[k]);
  IT *dst = amax_col + static_cast<int64_t>(blockIdx.y) * g.col_stride + c;
#pragma unroll
  for (int k = 0; k < 8; ++k) dst[k] = o[k];
}

template <typename IT, bool ALIGNED>
__global__ __launch_bounds__(kShardWaves * 64) void shard_cast_kernel(const IT *__restrict__ shard,
                                                                      uint8_t *__restrict__ out_row,
                                                                      uint8_t *__restrict__ out_col,
                                                                      const uint8_t *__restrict__ sc_row,
                                                                      const uint8_t *__restrict__ sc_col,
                                                                      ShardGeom g) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * kShardWaves + wave) * kShardCols + lane * 8;
  const int r0 = blockIdx.y * kShardRows;
  if (c >= g.cols) return;
  const int64_t tile_lo = static_cast<int64_t>(r0) * g.cols + c;
  const int64_t tile_hi = static_cast<int64_t>(r0 + kShardRows - 1) * g.cols + c + 8;
  if (tile_hi <= g.start || tile_lo >= g.end) return;

  float cs[8];
  {
    const uint8_t *p = sc_col + static_cast<int64_t>(blockIdx.y) * g.col_stride + c;
#pragma unroll
    for (int k = 0; k < 8; ++k) cs[k] = e8m0_rcp(p[k]);  // once per tile; no alignment assumed
  }
  for (int rb = 0; rb < kShardRows; rb += 8) {
    float x[8][8];
    float rs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + rb + i;
      shard_load8<IT, ALIGNED>(shard, g, static_cast<int64_t>(r) * g.cols + c, x[i]);
      rs[i] = e8m0_rcp(sc_row[static_cast<int64_t>(r) * g.row_stride + (c >> 5)]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t idx = static_cast<int64_t>(r0 + rb + i) * g.cols + c;
      const u32x2 qr = {cvt4_fp8<Fp8::E4M3>(x[i][0] * rs[i], x[i][1] * rs[i], x[i][2] * rs[i], x[i][3] * rs[i]),
                        cvt4_fp8<Fp8::E4M3>(x[i][4] * rs[i], x[i][5] * rs[i], x[i][6] * rs[i], x[i][7] * rs[i])};
      const u32x2 qc = {cvt4_fp8<Fp8::E4M3>(x[i][0] * cs[0], x[i][1] * cs[1], x[i][2] * cs[2], x[i][3] * cs[3]),
                        cvt4_fp8<Fp8::E4M3>(x[i][4] * cs[4], x[i][5] * cs[5], x[i][6] * cs[6], x[i][7] * cs[7])};
      if (ALIGNED && idx >= g.start && idx + 8 <= g.end) {
        *reinterpret_cast<u32x2 *>(out_row + (idx - g.start)) = qr;
        *reinterpret_cast<u32x2 *>(out_col + (idx - g.start)) = qc;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int64_t e = idx + k;
          if (e >= g.start && e < g.end) {
            out_row[e - g.start] = static_cast<uint8_t>((qr[k >> 2] >> (8 * (k & 3))) & 0xFFu);
            out_col[e - g.start] = static_cast<uint8_t>((qc[k >> 2] >> (8 * (k & 3))) & 0xFFu);
          }
        }
      }
    }
  }
}

// ref mxfp8_scaling.cu:133-159,174-212: shapes of the padded amax / scale arrays
void check_shard_scale_array(const Buf &b, const char *name, size_t min_rows, size_t min_cols,
                             size_t row_pad, size_t col_pad) {
  NVTE_CHECK(b.shape.ndim == 2, name, " must be a 2D tensor");
  NVTE_CHECK(b.shape.data[0] % row_pad == 0, "Wrong padding of ", name, "'s rows");
  NVTE_CHECK(b.shape.data[0] >= min_rows, "Invalid rows");
  NVTE_CHECK(b.shape.data[1] % col_pad == 0, "Wrong padding of ", name, "'s cols");
  NVTE_CHECK(b.shape.data[1] >= min_cols, "Invalid cols");
}

ShardGeom shard_geom(const Tensor &input, int rows, int cols, size_t start_offset, const Buf &rowwise,
                     const Buf &colwise) {
  NVTE_CHECK(rows > 0 && cols > 0, "rows and cols must be positive");
  NVTE_CHECK(rows % 32 == 0, "rows must be divisible by 32");
  NVTE_CHECK(cols % 32 == 0, "cols must be divisible by 32");
  NVTE_CHECK(input.data.shape.ndim == 1, "input must be a 1D tensor");
  NVTE_CHECK(start_offset + input.data.shape.data[0] <= static_cast<size_t>(rows) * cols,
             "Invalid start_offset");
  ShardGeom g;
  g.rows = rows;
  g.cols = cols;
  g.start = static_cast<int64_t>(start_offset);
  g.end = g.start + static_cast<int64_t>(input.data.shape.data[0]);
  g.row_stride = static_cast<int>(rowwise.shape.data[1]);
  g.col_stride = static_cast<int>(colwise.shape.data[1]);
  return g;
}

dim3 shard_grid(const ShardGeom &g) {
  return dim3(static_cast<unsigned>(ceil_div(g.cols, kShardWaves * kShardCols)), static_cast<unsigned>(g.rows / 32));
}

bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

template <typename IT>
void shard_amax_typed(const Tensor &in, Tensor &ar, Tensor &ac, const ShardGeom &g, hipStream_t stream) {
  const bool aligned = (g.start % 8 == 0) && aligned_to(in.data.ptr, 16);
  auto *k = aligned ? shard_amax_kernel<IT, true> : shard_amax_kernel<IT, false>;
  hipLaunchKernelGGL(k, shard_grid(g), dim3(kShardWaves * 64), 0, stream, static_cast<const IT *>(in.data.ptr),
                     static_cast<IT *>(ar.data.ptr), static_cast<IT *>(ac.data.ptr), g);
  NVTE_CHECK_HIP(hipGetLastError());
}

template <typename IT>
void shard_cast_typed(const Tensor &in, Tensor &orow, Tensor &ocol, const Tensor &sr, const Tensor &sc,
                      const ShardGeom &g, hipStream_t stream) {
  const bool aligned = (g.start % 8 == 0) && aligned_to(in.data.ptr, 16) &&
                       aligned_to(orow.data.ptr, 8) && aligned_to(ocol.data.ptr, 8);
  auto *k = aligned ? shard_cast_kernel<IT, true> : shard_cast_kernel<IT, false>;
  hipLaunchKernelGGL(k, shard_grid(g), dim3(kShardWaves * 64), 0, stream, static_cast<const IT *>(in.data.ptr),
                     static_cast<uint8_t *>(orow.data.ptr), static_cast<uint8_t *>(ocol.data.ptr),
                     static_cast<const uint8_t *>(sr.data.ptr), static_cast<const uint8_t *>(sc.data.ptr), g);
  NVTE_CHECK_HIP(hipGetLastError());
}

}  // namespace
}  // namespace te

using namespace te;

extern "C" {

void nvte_multi_tensor_scale_cuda(int chunk_size, NVTETensor is_infinite, NVTETensor **tensor_lists,
                                  const size_t num_tensor_lists, const size_t num_tensors_per_list,
                                  float scale, hipStream_t stream) {
  api_guard("nvte_multi_tensor_scale_cuda", [&] {
    mt_scale(chunk_size, *lookup_checked(is_infinite),
             lists_of(tensor_lists, num_tensor_lists, num_tensors_per_list), scale, nullptr, stream);
  });
}

void nvte_multi_tensor_scale_tensor_cuda(int chunk_size, NVTETensor is_infinite,
                                         NVTETensor **tensor_lists, const size_t num_tensor_lists,
                                         const size_t num_tensors_per_list, NVTETensor scale,
                                         hipStream_t stream) {
  api_guard("nvte_multi_tensor_scale_tensor_cuda", [&] {
    const Tensor &s = *lookup_checked(scale);
    NVTE_CHECK(s.data.dtype == kNVTEFloat32 && s.data.ptr != nullptr, "scale must be an FP32 device tensor");
    mt_scale(chunk_size, *lookup_checked(is_infinite),
             lists_of(tensor_lists, num_tensor_lists, num_tensors_per_list), 0.f,
             static_cast<const float *>(s.data.ptr), stream);
  });
}

void nvte_multi_tensor_compute_scale_and_scale_inv_cuda(int chunk_size, NVTETensor noop_flag,
                                                        NVTETensor **tensor_lists,
                                                        const size_t num_tensor_lists,
                                                        const size_t num_tensors_per_list,
                                                        float max_fp8, int force_pow_2_scales,
                                                        float epsilon, hipStream_t stream) {
  api_guard("nvte_multi_tensor_compute_scale_and_scale_inv_cuda", [&] {
    lookup_checked(noop_flag);  // the reference takes it and never reads it (compute_scale.cu:31-33)
    const TensorLists lists = lists_of(tensor_lists, num_tensor_lists, num_tensors_per_list);
    NVTE_CHECK(lists.size() == 3, "Expected 3 tensor lists (amax, scale, scale_inv), but found ", lists.size());
    for (size_t d = 0; d < 3; ++d)
      for (const Tensor *t : lists[d])
        NVTE_CHECK(t->data.dtype == kNVTEFloat32, "amax / scale / scale_inv must be FP32, got ",
                   dtype_name(t->data.dtype));
    const int64_t chunk = chunk_size;
    mt_run<3>(lists, 3, chunk, [&](const MtTable<3> &t, unsigned grid) {
      hipLaunchKernelGGL(mt_scale_and_inv_kernel, dim3(grid), dim3(kMtThreads), 0, stream, t, chunk, max_fp8,
                         force_pow_2_scales != 0, epsilon);
      NVTE_CHECK_HIP(hipGetLastError());
    });
  });
}

void nvte_multi_tensor_compute_scale_inv_e8m0_cuda(int chunk_size, NVTETensor **tensor_lists,
                                                   const size_t num_tensor_lists,
                                                   const size_t num_tensors_per_list,
                                                   hipStream_t stream) {
  api_guard("nvte_multi_tensor_compute_scale_inv_e8m0_cuda", [&] {
    const TensorLists lists = lists_of(tensor_lists, num_tensor_lists, num_tensors_per_list);
    NVTE_CHECK(lists.size() == 2, "Expected 2 tensor lists (amax, scale_inv), but found ", lists.size());
    for (const Tensor *t : lists[0])
      NVTE_CHECK(t->data.dtype == kNVTEBFloat16, "amax should be bf16");
    for (const Tensor *t : lists[1])
      NVTE_CHECK(t->data.dtype == kNVTEByte || t->data.dtype == kNVTEFloat8E8M0,
                 "scale_inv should be e8m0/uint8");
    const int64_t chunk = chunk_size;
    mt_run<2>(lists, 2, chunk, [&](const MtTable<2> &t, unsigned grid) {
      hipLaunchKernelGGL(mt_e8m0_kernel, dim3(grid), dim3(kMtThreads), 0, stream, t, chunk);
      NVTE_CHECK_HIP(hipGetLastError());
    });
  });
}

void nvte_multi_tensor_adam_fp8_cuda(int chunk_size, NVTETensor noop_flag, NVTETensor **tensor_lists,
                                     const size_t num_tensor_lists, const size_t num_tensors_per_list,
                                     const float lr, const float beta1, const float beta2,
                                     const float epsilon, const int step, const int mode,
                                     const int bias_correction, const float weight_decay,
                                     const NVTEDType fp8_dtype, hipStream_t stream) {
  api_guard("nvte_multi_tensor_adam_fp8_cuda", [&] {
    lookup_checked(noop_flag);  // taken and never read (adam.cu:62-64)
    const TensorLists lists = lists_of(tensor_lists, num_tensor_lists, num_tensors_per_list);
    // 8 tensor lists: g, p_fp8, m, v, p_master, scale, amax, scale_inv
    NVTE_CHECK(lists.size() == 8, "Expected 8 tensor lists, but found ", lists.size());
    NVTE_CHECK(is_fp8(fp8_dtype), "Expected FP8 dtype, but got ", dtype_name(fp8_dtype));
    if (num_tensors_per_list == 0) return;
    const DType gt = lists[0][0]->data.dtype;
    static const char *names[8] = {"Grad", "Param", "First moment", "Second moment", "Master param",
                                   "Scale", "Absmax", "Scale-inverse"};
    for (size_t j = 0; j < num_tensors_per_list; ++j) {
      NVTE_CHECK(lists[0][j]->data.dtype == gt, "Grad tensor ", j, " has dtype=",
                 dtype_name(lists[0][j]->data.dtype), ", but expected dtype=", dtype_name(gt));
      const DType pt = lists[1][j]->data.dtype;
      NVTE_CHECK(pt == fp8_dtype || pt == kNVTEByte, "Param tensor ", j, " has dtype=", dtype_name(pt),
                 ", but expected dtype=", dtype_name(fp8_dtype));
      for (int d = 2; d < 8; ++d)
        NVTE_CHECK(lists[d][j]->data.dtype == kNVTEFloat32, names[d], " tensor ", j, " has dtype=",
                   dtype_name(lists[d][j]->data.dtype), ", but expected dtype=Float32");
    }
    AdamArgs a;
    a.beta1 = beta1;
    a.beta2 = beta2;
    a.corr1 = 1.f;
    a.corr2 = 1.f;
    if (bias_correction == 1) {
      a.corr1 = 1 - std::pow(beta1, step);
      a.corr2 = 1 - std::pow(beta2, step);
    }
    a.eps = epsilon;
    a.lr = lr;
    a.decay = weight_decay;
    a.mode = mode;
    const int64_t chunk = chunk_size;
    if (fp8_dtype == kNVTEFloat8E4M3) {
      mt_adam_fp8_by_grad<Fp8::E4M3>(gt, lists, chunk, a, stream);
    } else {
      mt_adam_fp8_by_grad<Fp8::E5M2>(gt, lists, chunk, a, stream);
    }
  });
}

void nvte_mxfp8_scaling_compute_partial_amax(const NVTETensor input_, NVTETensor amax_rowwise_,
                                             NVTETensor amax_colwise_, int rows, int cols,
                                             size_t start_offset, hipStream_t stream) {
  api_guard("nvte_mxfp8_scaling_compute_partial_amax", [&] {
    const Tensor &input = *lookup_checked(input_);
    Tensor &ar = *lookup_checked(amax_rowwise_);
    Tensor &ac = *lookup_checked(amax_colwise_);
    check_shard_scale_array(ar.data, "amax_rowwise", rows, cols / 32, 128, 4);
    NVTE_CHECK(ar.data.dtype == input.data.dtype, "Wrong dtype of amax_rowwise");
    check_shard_scale_array(ac.data, "amax_colwise", rows / 32, cols, 4, 128);
    NVTE_CHECK(ac.data.dtype == input.data.dtype, "Wrong dtype of amax_colwise");
    const ShardGeom g = shard_geom(input, rows, cols, start_offset, ar.data, ac.data);
    NVTE_CHECK(input.data.ptr != nullptr || g.end == g.start, "input has no data");
    NVTE_CHECK(ar.data.ptr != nullptr && ac.data.ptr != nullptr, "amax_rowwise / amax_colwise have no data");
    switch (input.data.dtype) {
      case kNVTEFloat32: shard_amax_typed<float>(input, ar, ac, g, stream); break;
      case kNVTEFloat16: shard_amax_typed<f16_t>(input, ar, ac, g, stream); break;
      case kNVTEBFloat16: shard_amax_typed<bf16_t>(input, ar, ac, g, stream); break;
      default: NVTE_ERROR("Unsupported input dtype ", dtype_name(input.data.dtype));
    }
  });
}

void nvte_mxfp8_scaling_partial_cast(const NVTETensor input_, NVTETensor output_rowwise_,
                                     NVTETensor output_colwise_, const NVTETensor scale_inv_rowwise_,
                                     const NVTETensor scale_inv_colwise_, int rows, int cols,
                                     size_t start_offset, hipStream_t stream) {
  api_guard("nvte_mxfp8_scaling_partial_cast", [&] {
    const Tensor &input = *lookup_checked(input_);
    Tensor &orow = *lookup_checked(output_rowwise_);
    Tensor &ocol = *lookup_checked(output_colwise_);
    const Tensor &sr = *lookup_checked(scale_inv_rowwise_);
    const Tensor &sc = *lookup_checked(scale_inv_colwise_);
    NVTE_CHECK(orow.data.shape.ndim == 1, "output_rowwise must be a 1D tensor");
    NVTE_CHECK(ocol.data.shape.ndim == 1, "output_colwise must be a 1D tensor");
    NVTE_CHECK(input.data.shape.ndim == 1, "input must be a 1D tensor");
    NVTE_CHECK(orow.data.shape.data[0] == input.data.shape.data[0], "Size of input and output_rowwise mismatch");
    NVTE_CHECK(ocol.data.shape.data[0] == input.data.shape.data[0], "Size of input and output_colwise mismatch");
    NVTE_CHECK(orow.data.dtype == kNVTEFloat8E4M3 || orow.data.dtype == kNVTEByte,
               "output_rowwise should be e4m3 or uint8");
    NVTE_CHECK(ocol.data.dtype == kNVTEFloat8E4M3 || ocol.data.dtype == kNVTEByte,
               "output_colwise should be e4m3 or uint8");
    check_shard_scale_array(sr.data, "scale_inv_rowwise", rows, cols / 32, 128, 4);
    NVTE_CHECK(sr.data.dtype == kNVTEByte || sr.data.dtype == kNVTEFloat8E8M0, "Wrong dtype of scale_inv_rowwise");
    check_shard_scale_array(sc.data, "scale_inv_colwise", rows / 32, cols, 4, 128);
    NVTE_CHECK(sc.data.dtype == kNVTEByte || sc.data.dtype == kNVTEFloat8E8M0, "Wrong dtype of scale_inv_colwise");
    const ShardGeom g = shard_geom(input, rows, cols, start_offset, sr.data, sc.data);
    if (g.end == g.start) return;
    NVTE_CHECK(input.data.ptr != nullptr && orow.data.ptr != nullptr && ocol.data.ptr != nullptr &&
                   sr.data.ptr != nullptr && sc.data.ptr != nullptr,
               "input, outputs and scale_inv tensors must all have data");
    switch (input.data.dtype) {
      case kNVTEFloat32: shard_cast_typed<float>(input, orow, ocol, sr, sc, g, stream); break;
      case kNVTEFloat16: shard_cast_typed<f16_t>(input, orow, ocol, sr, sc, g, stream); break;
      case kNVTEBFloat16: shard_cast_typed<bf16_t>(input, orow, ocol, sr, sc, g, stream); break;
      default: NVTE_ERROR("Unsupported input dtype ", dtype_name(input.data.dtype));
    }
  });
}

}  // extern "C"
