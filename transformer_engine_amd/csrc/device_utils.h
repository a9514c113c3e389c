// Device-side helpers shared by the quantize / transpose / recipe kernels (gfx950).
//
// Numerics contract (what the CPU oracle in oracle/te_oracle.c restates):
//   * a value is widened to fp32, multiplied by the fp32 scale (ONE rounding), then
//     converted to FP8 with round-to-nearest-even and SATURATION to the largest
//     finite magnitude - the reference's `OType(in * scale)` through CUDA's
//     satfinite conversion (ref common/transpose/cast_transpose.cu:162-165,
//     common/util/ptx.cuh:973,993);
//   * amax = max |float(in)| with maxNum semantics (a NaN operand is ignored),
//     ref cast_transpose.cu:164-165;
//   * scale_inv = correctly rounded 1/scale (ref common/utils.cuh:950).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "transformer_engine/transformer_engine.h"

namespace te {

constexpr int kWave = 64;

__host__ __device__ __forceinline__ size_t ceil_div_dev(size_t a, size_t b) { return (a + b - 1) / b; }

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ input widening
struct bf16_t {
  uint16_t bits;
};
struct f16_t {
  _Float16 v;
};

__device__ __forceinline__ float widen(float x) { return x; }
__device__ __forceinline__ float widen(bf16_t x) { return __uint_as_float(static_cast<uint32_t>(x.bits) << 16); }
__device__ __forceinline__ float widen(f16_t x) { return static_cast<float>(x.v); }

// fp32 -> bf16 / fp16, round-to-nearest-even, NaN stays NaN (plain casts lower to
// v_cvt_pk_bf16_f32 / v_cvt_f16_f32 on gfx950).
__device__ __forceinline__ void narrow(float x, float &out) { out = x; }
__device__ __forceinline__ void narrow(float x, bf16_t &out) {
  __bf16 b = static_cast<__bf16>(x);
  out.bits = __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ void narrow(float x, f16_t &out) { out.v = static_cast<_Float16>(x); }

// ------------------------------------------------------------------ FP8 conversion
enum class Fp8 { E4M3, E5M2 };

template <Fp8 F>
struct Fp8Traits;
template <>
struct Fp8Traits<Fp8::E4M3> {
  static constexpr float max_finite = 448.f;
};
template <>
struct Fp8Traits<Fp8::E5M2> {
  static constexpr float max_finite = 57344.f;
};

// Clamp to the finite range but let NaN through (v_med3 / v_max would turn a NaN into a
// finite bound, so the NaN test is explicit: the compare is false for NaN).
template <Fp8 F>
__device__ __forceinline__ float sat_finite(float x) {
  constexpr float m = Fp8Traits<F>::max_finite;
  return (__builtin_fabsf(x) > m) ? __builtin_copysignf(m, x) : x;
}

// Two floats -> two FP8 bytes placed in the low (hi=false) or high (hi=true) half of
// `old`; the other half of `old` is preserved.  Hardware RNE conversion.
template <Fp8 F, bool HI>
__device__ __forceinline__ uint32_t cvt_pk_fp8(float a, float b, uint32_t old) {
  a = sat_finite<F>(a);
  b = sat_finite<F>(b);
  if constexpr (F == Fp8::E4M3) {
    return static_cast<uint32_t>(__builtin_amdgcn_cvt_pk_fp8_f32(a, b, static_cast<int>(old), HI));
  } else {
    return static_cast<uint32_t>(__builtin_amdgcn_cvt_pk_bf8_f32(a, b, static_cast<int>(old), HI));
  }
}

// Four floats -> one dword of four FP8 bytes (byte i = value i).
template <Fp8 F>
__device__ __forceinline__ uint32_t cvt4_fp8(float a, float b, float c, float d) {
  uint32_t r = cvt_pk_fp8<F, false>(a, b, 0u);
  return cvt_pk_fp8<F, true>(c, d, r);
}

// Eight values -> eight FP8 bytes.  The finite-range clamp of cvt_pk_fp8 costs three instructions per value
// (compare, copysign, select, plus VCC wait states); a caller that already knows the largest magnitude of
// the values (it computed an amax over them anyway) passes in_range = "that magnitude is inside the format's
// finite range" and the plain hardware conversions run.  NaN needs no clamp (the conversion keeps it) and
// is ignored by fmaxf, +-inf is larger than every range and takes the clamping branch: same bytes as the
// per-value form in every case.
template <Fp8 F>
__device__ __forceinline__ u32x2 cvt8_fp8(const float (&o)[8], bool in_range) {
  u32x2 w;
  if (in_range) {
    if constexpr (F == Fp8::E4M3) {
      w[0] = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], 0, false), true);
      w[1] = __builtin_amdgcn_cvt_pk_fp8_f32(o[6], o[7], __builtin_amdgcn_cvt_pk_fp8_f32(o[4], o[5], 0, false), true);
    } else {
      w[0] = __builtin_amdgcn_cvt_pk_bf8_f32(o[2], o[3], __builtin_amdgcn_cvt_pk_bf8_f32(o[0], o[1], 0, false), true);
      w[1] = __builtin_amdgcn_cvt_pk_bf8_f32(o[6], o[7], __builtin_amdgcn_cvt_pk_bf8_f32(o[4], o[5], 0, false), true);
    }
  } else {
    w[0] = cvt4_fp8<F>(o[0], o[1], o[2], o[3]);
    w[1] = cvt4_fp8<F>(o[4], o[5], o[6], o[7]);
  }
  return w;
}
// largest magnitude of eight values (NaN ignored, as fmaxf)
__device__ __forceinline__ float absmax8(const float (&t)[8]) {
  float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(t[0]), __builtin_fabsf(t[1])), __builtin_fabsf(t[2]));
  m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(t[3])), __builtin_fabsf(t[4]));
  m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(t[5])), __builtin_fabsf(t[6]));
  return __builtin_fmaxf(m, __builtin_fabsf(t[7]));
}
// Eight values t -> eight FP8 bytes of sat(t * scale), and amax = max(amax, |t|).  fl(|t| * scale) is
// monotonic in |t|, so the scaled largest magnitude is the largest scaled magnitude.
template <Fp8 F>
__device__ __forceinline__ u32x2 quantize8(const float (&t)[8], float scale, float &amax) {
  const float m = absmax8(t);
  amax = __builtin_fmaxf(amax, m);
  float o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = t[k] * scale;
  return cvt8_fp8<F>(o, __builtin_fabsf(m * scale) <= Fp8Traits<F>::max_finite);
}

template <Fp8 F>
__device__ __forceinline__ uint8_t cvt1_fp8(float a) {
  return static_cast<uint8_t>(cvt_pk_fp8<F, false>(a, a, 0u) & 0xFFu);
}

// FP8 byte -> float (hardware, exact).
template <Fp8 F>
__device__ __forceinline__ float fp8_to_float(uint32_t packed, int byte_idx) {
  // The byte selector of the builtin must be a constant: switch on it.
  if constexpr (F == Fp8::E4M3) {
    switch (byte_idx) {
      case 0: return __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(packed), 0);
      case 1: return __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(packed), 1);
      case 2: return __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(packed), 2);
      default: return __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(packed), 3);
    }
  } else {
    switch (byte_idx) {
      case 0: return __builtin_amdgcn_cvt_f32_bf8(static_cast<int>(packed), 0);
      case 1: return __builtin_amdgcn_cvt_f32_bf8(static_cast<int>(packed), 1);
      case 2: return __builtin_amdgcn_cvt_f32_bf8(static_cast<int>(packed), 2);
      default: return __builtin_amdgcn_cvt_f32_bf8(static_cast<int>(packed), 3);
    }
  }
}

// ------------------------------------------------------------------ e8m0 block scales
// float -> e8m0 exponent, rounding the VALUE up to the next power of two
// (ref common/util/ptx.cuh:386-417, host twin tests/cpp/test_common.h:480-500).
__device__ __forceinline__ uint8_t float_to_e8m0(float v) {
  const uint32_t u = __float_as_uint(v);
  if (v != v) return 0xFF;
  if ((u & 0x7FFFFFFFu) == 0x7F800000u) return 0xFE;
  if (v == 0.0f) return 0x00;
  uint32_t e = (u >> 23) & 0xFFu;
  const uint32_t m = u & 0x7FFFFFu;
  if ((m > 0 && e != 0xFE) && !(e == 0 && m <= 0x400000u)) ++e;
  return static_cast<uint8_t>(e);
}

// 2^(127 - e) as a float (ref ptx.cuh:356-364, test_common.h:502-513).
__device__ __forceinline__ float e8m0_rcp(uint8_t e) {
  uint32_t bits;
  if (e == 255) {
    bits = 0x7FFFFFFFu;
  } else if (e == 254) {
    bits = 0x00400000u;
  } else {
    bits = static_cast<uint32_t>(254 - e) << 23;
  }
  return __uint_as_float(bits);
}

// 2^(e - 127) for dequantisation, built the way the reference does (exponent field only,
// ref common/util/ptx.cuh:382-384): e = 0 reads as 0.0, e = 255 as +inf.
__device__ __forceinline__ float e8m0_to_float(uint8_t e) {
  return __uint_as_float(static_cast<uint32_t>(e) << 23);
}

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Max over a workgroup of NWAVES waves; result valid in thread 0.  `scratch` holds NWAVES floats.
template <int NWAVES>
__device__ __forceinline__ float block_max(float v, float *scratch) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) v = fmaxf(v, scratch[w]);
  }
  return v;
}

// Wave sum on the DPP path (cross-lane operands of ordinary vector adds) instead of six dependent
// ds_bpermute round trips through the LDS crossbar: quad swaps, two row rotations, then the row totals
// travel down the rows (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  Lane 63 ends
// up with the total of all 64 lanes.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0xb1, 0xf>(v);   // quad_perm:[1,0,3,2]
  v = dpp_add<0x4e, 0xf>(v);   // quad_perm:[2,3,0,1]
  v = dpp_add<0x124, 0xf>(v);  // row_ror:4
  v = dpp_add<0x128, 0xf>(v);  // row_ror:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3
  return v;
}

// amax values are >= 0, so their fp32 bit patterns order like unsigned ints
// (ref common/utils.cuh:935: int-punned atomicMax).  The target only ever grows during a
// launch, so a relaxed device-scope read that already shows a value >= v makes the atomic
// redundant even if the read is stale; thousands of same-address atomics per launch
// otherwise serialise at the memory side (measured: 7168 of them cost ~13 us).
__device__ __forceinline__ void atomic_max_nonneg(float *addr, float v) {
  unsigned int *p = reinterpret_cast<unsigned int *>(addr);
  const unsigned int bits = __float_as_uint(v);
  const unsigned int seen = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (bits > seen) atomicMax(p, bits);
}

// ------------------------------------------------------------------ byte shuffles
// v_perm_b32: result byte i = byte sel[i] of the 64-bit value {hi:lo} (0-3 -> lo, 4-7 -> hi).
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {
  return __builtin_amdgcn_perm(hi, lo, sel);
}

// Streaming (non-temporal) load for high-precision inputs that a quantize kernel touches once.
// Measured on MI355X (cast_transpose of dY, 235 MB in + 235 MB out): plain loads 5.05 TB/s, nt
// loads 6.3 TB/s; nt STORES lose (the FP8 outputs are re-read by the GEMM and by repeated
// launches out of the Infinity Cache), so only the loads stream.
template <typename V>
__device__ __forceinline__ V ld_stream(const void *p) {
  return __builtin_nontemporal_load(reinterpret_cast<const V *>(p));
}

// In-register transpose of a 4x4 byte block: r[i] holds row i (byte j = column j);
// on return r[j] holds column j (byte i = row i).
__device__ __forceinline__ void transpose4x4_bytes(uint32_t &r0, uint32_t &r1, uint32_t &r2, uint32_t &r3) {
  const uint32_t t0 = perm(r1, r0, 0x05010400u);  // r0.b0 r1.b0 r0.b1 r1.b1
  const uint32_t t1 = perm(r1, r0, 0x07030602u);  // r0.b2 r1.b2 r0.b3 r1.b3
  const uint32_t t2 = perm(r3, r2, 0x05010400u);
  const uint32_t t3 = perm(r3, r2, 0x07030602u);
  r0 = perm(t2, t0, 0x05040100u);
  r1 = perm(t2, t0, 0x07060302u);
  r2 = perm(t3, t1, 0x05040100u);
  r3 = perm(t3, t1, 0x07060302u);
}

// Transposed store of a 128x128 FP8 tile held as 8x8 byte blocks by 256 threads: thread t owns
// rows 8*(t>>4).., columns 8*(t&15).. of the tile; q[i][h] = row i, columns 4h..4h+3.
// The block is transposed in registers (four 4x4 byte transposes), written as eight
// ds_write_b64 into a 16 KiB LDS image of the transposed tile (row c = 128 B = 16 eight-row
// chunks, chunk rg at slot rg ^ (c/8): conflict-free for the 16-lane b64 groups), and after one
// barrier written out 16 B per thread per pass: one wave instruction = eight full 128-B lines
// of out_t, whose row c starts at out_t + c * ld.  `q` is clobbered.
// GUARD: only 16-row chunks k < valid_chunks of transposed rows c < valid_cols are written
// (partial tiles of the grouped quantize; rows per member are a multiple of 16 there).
template <bool GUARD = false>
__device__ __forceinline__ void store_tile_transposed(uint8_t *tile_t, uint32_t (&q)[8][2], unsigned t,
                                                      uint8_t *out_t, size_t ld, unsigned valid_chunks = 8,
                                                      unsigned valid_cols = 128) {
  const unsigned cg = t & 15u, rg = t >> 4;
  transpose4x4_bytes(q[0][0], q[1][0], q[2][0], q[3][0]);  // rows 0-3, cols 0-3
  transpose4x4_bytes(q[0][1], q[1][1], q[2][1], q[3][1]);  // rows 0-3, cols 4-7
  transpose4x4_bytes(q[4][0], q[5][0], q[6][0], q[7][0]);  // rows 4-7, cols 0-3
  transpose4x4_bytes(q[4][1], q[5][1], q[6][1], q[7][1]);  // rows 4-7, cols 4-7
  // now q[j][h] = column 4h+j, rows 0-3 ; q[4+j][h] = column 4h+j, rows 4-7
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned c = cg * 8 + h * 4 + j;
      u32x2 v = {q[j][h], q[4 + j][h]};
      *reinterpret_cast<u32x2 *>(&tile_t[c * 128 + ((rg ^ cg) << 3)]) = v;
    }
  }
  // the bare barrier behind a wait for this wave's LDS writes only: __syncthreads() would also wait for every
  // outstanding global access of the wave (one counter on gfx9), i.e. drain the next tile's prefetch and
  // the rowwise stores just issued
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const unsigned k = t & 7u;  // 16-row chunk of the output row
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned c = (t >> 3) + 32u * p;
    const unsigned cgc = c >> 3;
    u32x4 v = *reinterpret_cast<const u32x4 *>(&tile_t[c * 128 + ((k ^ (cgc >> 1)) << 4)]);
    if (cgc & 1u) {
      u32x4 w = {v[2], v[3], v[0], v[1]};
      v = w;
    }
    if constexpr (GUARD) {
      if (k >= valid_chunks || c >= valid_cols) continue;
    }
    *reinterpret_cast<u32x4 *>(out_t + c * ld + k * 16) = v;
  }
}

__device__ __forceinline__ bool noop_set(const float *noop) { return noop != nullptr && noop[0] == 1.0f; }

// ------------------------------------------------------------------ 8-element streaming loads -> fp32
// Raw8: the same 8 elements kept as loaded (no arithmetic), so that a software-pipelined kernel can issue
// the NEXT tile's loads before it touches the current tile's values.
template <typename IT>
struct Raw8;
template <>
struct Raw8<bf16_t> {
  u32x4 v;
  static __device__ __forceinline__ Raw8 zero() { return Raw8{u32x4{0u, 0u, 0u, 0u}}; }
  static __device__ __forceinline__ Raw8 load(const bf16_t *p) { return Raw8{ld_stream<u32x4>(p)}; }
  static __device__ __forceinline__ Raw8 load_cached(const bf16_t *p) { return Raw8{*reinterpret_cast<const u32x4 *>(p)}; }
  __device__ __forceinline__ void widen8(float (&f)[8]) const {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = __uint_as_float(v[k] << 16);
      f[2 * k + 1] = __uint_as_float(v[k] & 0xFFFF0000u);
    }
  }
};
template <>
struct Raw8<f16_t> {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  h8 v;
  static __device__ __forceinline__ Raw8 zero() { return Raw8{h8{0, 0, 0, 0, 0, 0, 0, 0}}; }
  static __device__ __forceinline__ Raw8 load(const f16_t *p) { return Raw8{ld_stream<h8>(p)}; }
  static __device__ __forceinline__ Raw8 load_cached(const f16_t *p) { return Raw8{*reinterpret_cast<const h8 *>(p)}; }
  __device__ __forceinline__ void widen8(float (&f)[8]) const {
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = static_cast<float>(v[k]);
  }
};
template <>
struct Raw8<float> {
  f32x4 a, b;
  static __device__ __forceinline__ Raw8 zero() { return Raw8{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}; }
  static __device__ __forceinline__ Raw8 load(const float *p) { return Raw8{ld_stream<f32x4>(p), ld_stream<f32x4>(p + 4)}; }
  static __device__ __forceinline__ Raw8 load_cached(const float *p) {
    return Raw8{*reinterpret_cast<const f32x4 *>(p), *reinterpret_cast<const f32x4 *>(p + 4)};
  }
  __device__ __forceinline__ void widen8(float (&f)[8]) const {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[k] = a[k];
      f[4 + k] = b[k];
    }
  }
};

template <typename IT>
struct Load8;

template <>
struct Load8<bf16_t> {
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
struct Load8<f16_t> {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ void run(const f16_t *p, float (&f)[8]) {
    const h8 v = ld_stream<h8>(p);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = static_cast<float>(v[k]);
  }
};

template <>
struct Load8<float> {
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


// ------------------------------------------------------------------ typed (runtime dtype) element / 8-element access
// dtype values are NVTEDType (transformer_engine.h); the switches are wave-uniform
__device__ __forceinline__ void load8(const void *base, size_t idx, int dtype, float (&f)[8]) {
  switch (dtype) {
    case kNVTEBFloat16: Load8<bf16_t>::run(static_cast<const bf16_t *>(base) + idx, f); break;
    case kNVTEFloat16: Load8<f16_t>::run(static_cast<const f16_t *>(base) + idx, f); break;
    default: Load8<float>::run(static_cast<const float *>(base) + idx, f); break;
  }
}
// plain (cached) loads for the small per-column vectors
__device__ __forceinline__ float load1(const void *base, size_t idx, int dtype) {
  switch (dtype) {
    case kNVTEBFloat16: return widen(static_cast<const bf16_t *>(base)[idx]);
    case kNVTEFloat16: return widen(static_cast<const f16_t *>(base)[idx]);
    default: return static_cast<const float *>(base)[idx];
  }
}
__device__ __forceinline__ void store1(void *base, size_t idx, int dtype, float v) {
  switch (dtype) {
    case kNVTEBFloat16: narrow(v, static_cast<bf16_t *>(base)[idx]); break;
    case kNVTEFloat16: narrow(v, static_cast<f16_t *>(base)[idx]); break;
    case kNVTEFloat8E4M3: static_cast<uint8_t *>(base)[idx] = cvt1_fp8<Fp8::E4M3>(v); break;
    case kNVTEFloat8E5M2: static_cast<uint8_t *>(base)[idx] = cvt1_fp8<Fp8::E5M2>(v); break;
    default: static_cast<float *>(base)[idx] = v; break;
  }
}
// value after rounding to the output dtype (amax of a non-FP8 output, ref rmsnorm_fwd_kernels.cuh:113-117)
__device__ __forceinline__ float round_to(float v, int dtype) {
  switch (dtype) {
    case kNVTEBFloat16: { bf16_t o; narrow(v, o); return widen(o); }
    case kNVTEFloat16: { f16_t o; narrow(v, o); return widen(o); }
    default: return v;
  }
}
__device__ __forceinline__ void store8(void *base, size_t idx, int dtype, const float (&v)[8]) {
  switch (dtype) {
    case kNVTEFloat8E4M3: {
      u32x2 o = {cvt4_fp8<Fp8::E4M3>(v[0], v[1], v[2], v[3]), cvt4_fp8<Fp8::E4M3>(v[4], v[5], v[6], v[7])};
      *reinterpret_cast<u32x2 *>(static_cast<uint8_t *>(base) + idx) = o;
      break;
    }
    case kNVTEFloat8E5M2: {
      u32x2 o = {cvt4_fp8<Fp8::E5M2>(v[0], v[1], v[2], v[3]), cvt4_fp8<Fp8::E5M2>(v[4], v[5], v[6], v[7])};
      *reinterpret_cast<u32x2 *>(static_cast<uint8_t *>(base) + idx) = o;
      break;
    }
    case kNVTEBFloat16: {
      bf16_t o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) narrow(v[k], o[k]);
      *reinterpret_cast<u32x4 *>(static_cast<bf16_t *>(base) + idx) = *reinterpret_cast<const u32x4 *>(o);
      break;
    }
    case kNVTEFloat16: {
      f16_t o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) narrow(v[k], o[k]);
      *reinterpret_cast<u32x4 *>(static_cast<f16_t *>(base) + idx) = *reinterpret_cast<const u32x4 *>(o);
      break;
    }
    default: {
      f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
      float *p = static_cast<float *>(base) + idx;
      *reinterpret_cast<f32x4 *>(p) = lo;
      *reinterpret_cast<f32x4 *>(p + 4) = hi;
      break;
    }
  }
}

// ------------------------------------------------------------------ scale from amax
// ref common/recipe/recipe_common.cuh:14-49
__device__ __forceinline__ float scale_from_amax(float amax, float max_fp8, bool pow2, float eps,
                                                 float value_for_inf) {
  if (amax < eps) amax = eps;  // false for NaN
  if (__builtin_isinf(amax) || amax == 0.f || amax != amax) return 1.f;
  float scale = max_fp8 / amax;  // IEEE-rounded division (no fast-math in this library)
  if (__builtin_isinf(scale)) scale = value_for_inf;
  if (pow2) scale = __uint_as_float(__float_as_uint(scale) & 0xFF800000u);
  return scale;
}


}  // namespace te
