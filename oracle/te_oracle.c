/* TEST INFRASTRUCTURE - NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's FP8 linear hot path, used ONLY as the parity
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing
 * under transformer_engine_amd/ may import, link or call this file.
 *
 * Every function cites the reference lines it follows ("ref" = /root/reference).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - float -> FP8 (RNE, saturate-to-finite): checked exhaustively against torch's CPU
 *     float8_e4m3fn / float8_e5m2 casts for every in-range fp32 bit pattern class and
 *     against ROCm's host-side __hip_fp8 conversions incl. saturation
 *     (tests/test_oracle_pinning.py).  NaN -> 0xFF (E4M3) / 0xFE (E5M2) is what the
 *     gfx950 converter returns; the reference's tests never feed NaN to a cast, so the
 *     NaN code points are "parity unpinned".
 *   - scale-from-amax: checked against the reference's own pure-torch
 *     tests/pytorch/references/quantize_scale_calc.py (imported by file path) and the
 *     golden vectors generated from it (tests/golden/).
 *   - e8m0 helpers: transcribed semantics of ref tests/cpp/test_common.h:480-513, checked
 *     against torch.float8_e8m0fnu where they overlap.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define TE_E4M3 0
#define TE_E5M2 1

/* ---------------------------------------------------------------- bit helpers */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float te_bf16_to_float(uint16_t b) { return u2f((uint32_t)b << 16); }

/* fp32 -> bf16, round-to-nearest-even, NaN kept quiet (what a plain cast does). */
uint16_t te_float_to_bf16(float f) {
  uint32_t u = f2u(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

float te_half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
  if (e == 0) {
    if (m == 0) return u2f(sign);
    /* subnormal: normalise */
    int shift = 0;
    while (!(m & 0x400u)) { m <<= 1; ++shift; }
    m &= 0x3FFu;
    return u2f(sign | ((uint32_t)(113 - shift) << 23) | (m << 13));
  }
  if (e == 31) return u2f(sign | 0x7F800000u | (m << 13));
  return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

uint16_t te_float_to_half(float f) {
  const uint32_t u = f2u(f);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const uint32_t a = u & 0x7FFFFFFFu;
  if (a > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);
  if (a >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u); /* >= 65520 rounds to inf */
  if (a < 0x33000001u) return (uint16_t)sign;               /* < 2^-25 (or tie to 0) */
  int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7FFFFFu) | 0x800000u;
  int shift;
  uint32_t base;
  if (e < -14) { shift = 13 + (-14 - e); base = 0; } else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7FFFFFu; }
  uint32_t q = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
  return (uint16_t)(sign | (base + q));
}

/* ---------------------------------------------------------------- FP8 <-> float
 * The reference casts with CUDA's __nv_fp8_e4m3 / __nv_fp8_e5m2 constructors
 * (ref tests/cpp/test_common.h:67-68, common/common.h:645-646) = cvt.rn.satfinite
 * (ref common/util/ptx.cuh:973,993): round-to-nearest-even, finite overflow AND
 * infinity saturate to +-448 / +-57344, NaN -> NaN code.
 */
uint8_t te_float_to_fp8(float x, int fmt) {
  const int man_bits = (fmt == TE_E4M3) ? 3 : 2;
  const int bias = (fmt == TE_E4M3) ? 7 : 15;
  const float maxv = (fmt == TE_E4M3) ? 448.f : 57344.f;
  const uint8_t maxcode = (fmt == TE_E4M3) ? 0x7E : 0x7B;
  const uint32_t u = f2u(x);
  const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
  /* NaN: the gfx950 converters (v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32) return ONE code for every
   * NaN input whatever its sign or payload: 0xFF (E4M3, S.1111.111) and 0xFE (E5M2,
   * S.11111.10) - measured.  CUDA's converter would give 0x7F / 0x7E-class codes; the reference's
   * tests never cast a NaN, so these code points are "parity unpinned". */
  if (x != x) return (uint8_t)((fmt == TE_E4M3) ? 0xFF : 0xFE);
  const float ax = fabsf(x);
  if (ax >= maxv) return (uint8_t)(sign | maxcode); /* includes +-inf */
  if (ax == 0.f) return sign;
  /* Represent ax = q * 2^(emin - man_bits) in the subnormal range, or 1.m * 2^e. */
  int e;
  (void)frexpf(ax, &e); /* ax = f * 2^e, f in [0.5,1) -> unbiased exponent e-1 */
  int ue = e - 1;
  const int emin = 1 - bias;
  if (ue < emin) ue = emin; /* subnormal: fixed quantum */
  /* quantum = 2^(ue - man_bits); the scaled value is exact in double */
  const double scaled = ldexp((double)ax, man_bits - ue); /* in [0, 2^(man_bits+1)) */
  double fl = floor(scaled);
  const double rem = scaled - fl;
  uint32_t q = (uint32_t)fl;
  if (rem > 0.5 || (rem == 0.5 && (q & 1u))) ++q;
  /* q in [0, 2^(man_bits+1)]; q == 2^(man_bits+1) means carry into the next binade */
  uint32_t code;
  if (ue == emin && q < (1u << man_bits)) {
    code = q; /* subnormal (exponent field 0) */
  } else {
    uint32_t ef = (uint32_t)(ue + bias);
    if (q == (2u << man_bits)) { q = 1u << man_bits; ++ef; }
    code = (ef << man_bits) | (q - (1u << man_bits));
  }
  if (code > maxcode) code = maxcode; /* rounding up past max finite saturates */
  return (uint8_t)(sign | code);
}

float te_fp8_to_float(uint8_t b, int fmt) {
  const int man_bits = (fmt == TE_E4M3) ? 3 : 2;
  const int bias = (fmt == TE_E4M3) ? 7 : 15;
  const uint32_t ef = (b & 0x7Fu) >> man_bits, m = b & ((1u << man_bits) - 1u);
  const float s = (b & 0x80u) ? -1.f : 1.f;
  if (fmt == TE_E4M3) {
    if ((b & 0x7Fu) == 0x7Fu) return s * NAN;
  } else {
    if (ef == 31) return m ? s * NAN : s * INFINITY;
  }
  if (ef == 0) return s * ldexpf((float)m, 1 - bias - man_bits);
  return s * ldexpf((float)(m | (1u << man_bits)), (int)ef - bias - man_bits);
}

/* ref tests/cpp/test_common.h:480-500 (device twin common/util/ptx.cuh:386-417) */
uint8_t te_float_to_e8m0(float val) {
  if (val != val) return 0xFF;
  if (isinf(val)) return 0xFE;
  if (val == 0.0f) return 0x00;
  const uint32_t u = f2u(val);
  uint32_t exponent = (u >> 23) & 0xFFu; /* the reference stores the shifted word in a byte */
  const uint32_t mantissa = u & 0x7FFFFFu;
  if ((mantissa > 0 && exponent != 0xFE) && !(exponent == 0 && mantissa <= 0x400000u)) ++exponent;
  return (uint8_t)exponent;
}

/* ref tests/cpp/test_common.h:502-513 */
float te_exp2f_rcp(uint8_t biased_exp) {
  uint32_t v;
  if (biased_exp == 255) v = 0x7FFFFFFFu;
  else if (biased_exp == 254) v = 0x00400000u;
  else v = (uint32_t)(254 - biased_exp) << 23;
  return u2f(v);
}

/* ref common/util/ptx.cuh:382-384 */
float te_e8m0_to_float(uint8_t e) { return u2f((uint32_t)e << 23); }

static inline float load_in(const void *p, size_t i, int in_dtype) {
  switch (in_dtype) { /* NVTEDType values: 4 fp32, 5 fp16, 6 bf16 */
    case 4: return ((const float *)p)[i];
    case 5: return te_half_to_float(((const uint16_t *)p)[i]);
    default: return te_bf16_to_float(((const uint16_t *)p)[i]);
  }
}

/* ---------------------------------------------------------------- cast + transpose
 * ref tests/cpp/operator/test_cast_transpose.cu:24-39 (compute_ref), plus the kernel-side
 * details the test leaves implicit: scale==NULL means 1 (ref common/transpose/
 * cast_transpose.cu:143), scale_inv = 1/scale (:213-215).  The test's serial double loop
 * is kept serial on purpose: it is the reference's CPU cost (bench cpu_baseline).
 */
void te_cast_transpose_ref(const void *in, int in_dtype, uint8_t *out_c, uint8_t *out_t, size_t N,
                           size_t H, float scale, int fmt, float *amax, float *scale_inv) {
  float current_max = 0.f; /* reference starts at -1e100 and takes fmaxf(.., |x|): same for N*H>0 */
  for (size_t i = 0; i < N; ++i) {
    for (size_t j = 0; j < H; ++j) {
      const float current = load_in(in, i * H + j, in_dtype);
      current_max = fmaxf(current_max, fabsf(current));
      const uint8_t q = te_float_to_fp8(scale * current, fmt);
      if (out_c) out_c[i * H + j] = q;
      if (out_t) out_t[j * N + i] = q;
    }
  }
  if (amax) *amax = current_max;
  if (scale_inv) *scale_inv = 1.0f / scale;
}

/* Same math, rows split over OpenMP threads (used as the "all host cores" baseline). */
void te_cast_transpose_ref_omp(const void *in, int in_dtype, uint8_t *out_c, uint8_t *out_t,
                               size_t N, size_t H, float scale, int fmt, float *amax,
                               float *scale_inv) {
  float gmax = 0.f;
#pragma omp parallel for reduction(max : gmax) schedule(static)
  for (long i = 0; i < (long)N; ++i) {
    float m = 0.f;
    for (size_t j = 0; j < H; ++j) {
      const float current = load_in(in, (size_t)i * H + j, in_dtype);
      m = fmaxf(m, fabsf(current));
      const uint8_t q = te_float_to_fp8(scale * current, fmt);
      if (out_c) out_c[(size_t)i * H + j] = q;
      if (out_t) out_t[j * N + (size_t)i] = q;
    }
    gmax = fmaxf(gmax, m);
  }
  if (amax) *amax = gmax;
  if (scale_inv) *scale_inv = 1.0f / scale;
}

/* ---------------------------------------------------------------- current scaling
 * ref tests/cpp/operator/test_cast_transpose_current_scaling.cu:41-88 and the device
 * function ref common/recipe/recipe_common.cuh:14-49 (epsilon floor, pow2, NaN).
 */
float te_amax_ref(const void *in, int in_dtype, size_t n) {
  float m = 0.f;
  for (size_t i = 0; i < n; ++i) m = fmaxf(m, fabsf(load_in(in, i, in_dtype)));
  return m;
}

float te_scale_from_amax_ref(float amax, float max_fp8, int force_pow_2, float epsilon) {
  if (amax < epsilon) amax = epsilon;
  if (isinf(amax) || amax == 0.f || amax != amax) return 1.f;
  float scale = max_fp8 / amax;
  if (isinf(scale)) scale = 3.402823466e+38f;
  if (force_pow_2) scale = u2f(f2u(scale) & 0xFF800000u);
  return scale;
}

/* ---------------------------------------------------------------- MXFP8
 * ref tests/cpp/operator/test_cast_mxfp8.cu:39-166 (CAST_ONLY path): 32x32 tiles; rowwise
 * blocks 1x32, columnwise blocks 32x1; e = float_to_e8m0(amax * (1/fp8_max));
 * q = fp8(x * exp2f_rcp(e)).  Scale strides are the PADDED row lengths.
 */
void te_mxfp8_ref(const void *in, int in_dtype, size_t rows, size_t cols, int fmt,
                  uint8_t *out_row, uint8_t *sc_row, size_t stride_row, uint8_t *out_col,
                  uint8_t *sc_col, size_t stride_col) {
  const float max_rcp = (float)(1.0 / (double)((fmt == TE_E4M3) ? 448.f : 57344.f));
  const size_t ty = (rows + 31) / 32, tx = (cols + 31) / 32;
#pragma omp parallel for schedule(static)
  for (long t = 0; t < (long)(ty * tx); ++t) {
    const size_t tile_Y = (size_t)t / tx, tile_X = (size_t)t % tx;
    const size_t i0 = tile_Y * 32, i1 = (i0 + 32 < rows) ? i0 + 32 : rows;
    const size_t j0 = tile_X * 32, j1 = (j0 + 32 < cols) ? j0 + 32 : cols;
    float cache[32][32];
    for (size_t i = i0; i < i1; ++i)
      for (size_t j = j0; j < j1; ++j) cache[i - i0][j - j0] = load_in(in, i * cols + j, in_dtype);
    if (out_row) {
      for (size_t i = i0; i < i1; ++i) {
        float block_amax = 0.f;
        for (size_t j = j0; j < j1; ++j) {
          const float a = fabsf(cache[i - i0][j - j0]);
          if (block_amax < a) block_amax = a; /* std::max(a,b): NaN b is ignored */
        }
        const uint8_t e = te_float_to_e8m0(block_amax * max_rcp);
        sc_row[i * stride_row + tile_X] = e;
        const float r = te_exp2f_rcp(e);
        for (size_t j = j0; j < j1; ++j) out_row[i * cols + j] = te_float_to_fp8(cache[i - i0][j - j0] * r, fmt);
      }
    }
    if (out_col) {
      for (size_t j = j0; j < j1; ++j) {
        float block_amax = 0.f;
        for (size_t i = i0; i < i1; ++i) {
          const float a = fabsf(cache[i - i0][j - j0]);
          if (block_amax < a) block_amax = a;
        }
        const uint8_t e = te_float_to_e8m0(block_amax * max_rcp);
        sc_col[tile_Y * stride_col + j] = e;
        const float r = te_exp2f_rcp(e);
        for (size_t i = i0; i < i1; ++i) out_col[i * cols + j] = te_float_to_fp8(cache[i - i0][j - j0] * r, fmt);
      }
    }
  }
}

/* ref common/cast/mxfp8/swizzle.cuh:30-41 */
size_t te_swizzled_scale_idx(size_t i, size_t j, size_t num_tiles_X) {
  const size_t tile_idx_X = j / 4, tile_idx_Y = i / 128;
  const size_t x = j % 4, y = i % 128;
  return (tile_idx_Y * num_tiles_X + tile_idx_X) * 512 + (y % 32) * 16 + (y / 32) * 4 + x;
}

/* compact [M,K] (M % 128 == 0, K % 4 == 0) -> swizzled image */
void te_swizzle_scales_ref(const uint8_t *in, uint8_t *out, size_t M, size_t K) {
  for (size_t i = 0; i < M; ++i)
    for (size_t j = 0; j < K; ++j) out[te_swizzled_scale_idx(i, j, K / 4)] = in[i * K + j];
}

/* columnwise scales: compact [KB, N] viewed as the transposed [N, KB] matrix
 * (ref common/cast/mxfp8/quantize_mxfp8.cuh:273-278) */
void te_swizzle_col_scales_ref(const uint8_t *in, uint8_t *out, size_t KB, size_t N) {
  for (size_t j = 0; j < KB; ++j)
    for (size_t i = 0; i < N; ++i) out[te_swizzled_scale_idx(i, j, KB / 4)] = in[j * N + i];
}

/* ---------------------------------------------------------------- delayed scaling
 * ref common/recipe/delayed_scaling.cu:84-154 (one launch per history tensor) and the
 * pure-torch twin ref transformer_engine/pytorch/quantization.py:1057-1106;
 * expectations ref tests/pytorch/test_recipe.py:116-189.
 * algo: 0 = most_recent, 1 = max.  hist is [H, S] row-major; may run in place.
 */
static float ds_scale(float amax, float old_scale, float scaled_max) {
  float s = (isfinite(amax) && amax > 0.f) ? scaled_max / amax : old_scale;
  if (isinf(s)) s = 3.402823466e+38f;
  return s;
}

void te_delayed_scaling_ref(const float *hist, const float *scale, float *new_hist, float *new_scale,
                            size_t H, size_t S, int algo, float fp8_max, float margin) {
  const float scaled_max = fp8_max * powf(2.f, -margin);
  for (size_t s = 0; s < S; ++s) {
    const float last = hist[s];
    float amax = 0.f;
    float prev_row0 = 0.f;
    (void)prev_row0;
    /* roll by -1: new[i] = old[i+1], new[H-1] = old[0]; then new[0] = 0 */
    float carry = last;
    for (size_t i = 0; i < H; ++i) {
      const float a = (i + 1 < H) ? hist[(i + 1) * S + s] : carry;
      amax = fmaxf(amax, a);
      new_hist[i * S + s] = (i > 0) ? a : 0.f;
    }
    if (algo == 0) amax = last;
    new_scale[s] = ds_scale(amax, scale[s], scaled_max);
  }
}

/* ref common/recipe/delayed_scaling.cu:162-253 (bulk, after the amax all-reduce): `last` comes
 * from the reduced buffer when that entry is non-zero; the roll is SKIPPED when last == 0. */
void te_delayed_scaling_bulk_ref(const float *reduced /*may be NULL*/, float *hist, float *scale,
                                 size_t H, size_t S, int algo, float fp8_max, float margin) {
  const float scaled_max = fp8_max * powf(2.f, -margin);
  for (size_t s = 0; s < S; ++s) {
    float last = hist[s];
    if (reduced && reduced[s] != 0.0f) last = reduced[s];
    float amax = 0.f;
    if (last != 0.0f) {
      for (size_t i = 0; i < H; ++i) {
        const float a = (i + 1 < H) ? hist[(i + 1) * S + s] : last;
        amax = fmaxf(amax, a);
        hist[i * S + s] = (i > 0) ? a : 0.f;
      }
    }
    if (algo == 0) amax = last;
    scale[s] = ds_scale(amax, scale[s], scaled_max);
  }
}

/* ---------------------------------------------------------------- dequantize + FP32 GEMM
 * The reference has no host GEMM (ref tests/cpp/operator/test_grouped_gemm.cu:366); its
 * contract is "within tolerance of the matmul of the dequantized operands"
 * (ref tests/pytorch/utils.py:97-104).  D[i][j] = alpha * sum_k P[i][k] Q[j][k] + beta * D.
 */
void te_dequant_ref(const uint8_t *q, int fmt, float scale_inv, float *out, size_t n) {
  for (size_t i = 0; i < n; ++i) out[i] = te_fp8_to_float(q[i], fmt) * scale_inv;
}

void te_dequant_mxfp8_ref(const uint8_t *q, int fmt, const uint8_t *sc, size_t stride, int rowwise,
                          float *out, size_t rows, size_t cols) {
  for (size_t i = 0; i < rows; ++i)
    for (size_t j = 0; j < cols; ++j) {
      const uint8_t e = rowwise ? sc[i * stride + j / 32] : sc[(i / 32) * stride + j];
      out[i * cols + j] = te_fp8_to_float(q[i * cols + j], fmt) * te_e8m0_to_float(e);
    }
}

void te_gemm_tn_f32_ref(const float *P, const float *Q, float *D, size_t n, size_t m, size_t k,
                        float alpha, float beta) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)n; ++i) {
    for (size_t j = 0; j < m; ++j) {
      const float *p = P + (size_t)i * k, *q = Q + j * k;
      double acc = 0.0; /* wide accumulator: the oracle should not add its own rounding noise */
      for (size_t kk = 0; kk < k; ++kk) acc += (double)p[kk] * (double)q[kk];
      const float prev = (beta != 0.f) ? D[(size_t)i * m + j] : 0.f;
      D[(size_t)i * m + j] = (float)(alpha * acc) + beta * prev;
    }
  }
}

/* Faster blocked float-accumulate variant, used only as the CPU baseline timing leg. */
void te_gemm_tn_f32_fast(const float *P, const float *Q, float *D, size_t n, size_t m, size_t k) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)n; ++i) {
    for (size_t j = 0; j < m; ++j) {
      const float *p = P + (size_t)i * k, *q = Q + j * k;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      size_t kk = 0;
      for (; kk + 4 <= k; kk += 4) {
        a0 += p[kk] * q[kk]; a1 += p[kk + 1] * q[kk + 1];
        a2 += p[kk + 2] * q[kk + 2]; a3 += p[kk + 3] * q[kk + 3];
      }
      for (; kk < k; ++kk) a0 += p[kk] * q[kk];
      D[(size_t)i * m + j] = (a0 + a1) + (a2 + a3);
    }
  }
}

void te_colsum_ref(const void *in, int in_dtype, size_t rows, size_t cols, float *out) {
  for (size_t j = 0; j < cols; ++j) out[j] = 0.f;
  for (size_t i = 0; i < rows; ++i)
    for (size_t j = 0; j < cols; ++j) out[j] += load_in(in, i * cols + j, in_dtype);
}

/* GEMM activation epilogues: tanh-form GELU and its derivative, ref common/util/math.h:21-24 and
 * :27-32 (the forms TE's own activation kernels use; the cuBLASLt GELU epilogues the reference
 * GEMM selects at cublaslt_gemm.cu:650-689 are the same tanh approximation). mode 0 = gelu,
 * 1 = dgelu. */
void te_gelu_ref(const float *in, float *out, size_t n, int mode) {
  for (size_t i = 0; i < n; ++i) {
    const float x = in[i];
    if (mode == 0) {
      out[i] = x * (0.5f + 0.5f * tanhf(x * (0.79788456f + 0.03567741f * x * x)));
    } else {
      const float t = tanhf(0.79788456f * x * (1.f + 0.044715f * x * x));
      out[i] = 0.5f * x * ((1.f - t * t) * (0.79788456f + 0.1070322243f * x * x)) + 0.5f * (1.f + t);
    }
  }
}

/* ------------------------------------------------------------------ LayerNorm / RMSNorm
 * Restatement of the reference's own host oracle, ref tests/cpp/operator/test_normalization.h:
 * compute_ref_stats :33-57, compute_gamma :60-92 (non-cuDNN branch), compute_ref_output :95-135,
 * compute_ref_backward :138-190.  fp32 arithmetic in the same order. */
static inline float round_out(float v, int out_dtype) {
  if (out_dtype == 6) { uint32_t u = ((uint32_t)te_float_to_bf16(v)) << 16; float f; memcpy(&f, &u, 4); return f; }
  if (out_dtype == 5) return te_half_to_float(te_float_to_half(v));
  return v;
}

void te_norm_stats_ref(int is_ln, const void *x, int dtype, float *mu, float *rsigma, size_t N, size_t H,
                       double epsilon) {
  for (size_t i = 0; i < N; ++i) {
    float sum = 0.f, m = 0.f;
    for (size_t j = 0; j < H; ++j) sum += load_in(x, i * H + j, dtype);
    if (is_ln) { mu[i] = sum / H; m = mu[i]; }
    float sum_sq = 0.f;
    for (size_t j = 0; j < H; ++j) {
      const float cur = load_in(x, i * H + j, dtype);
      sum_sq += (cur - m) * (cur - m);
    }
    rsigma[i] = (float)(1.0 / sqrt((double)((sum_sq / H) + epsilon)));
  }
}

/* out[i] = the fp32 value BEFORE the conversion to the output type (tmp * scale); amax: on tmp when
 * scale != 1 (FP8 paths), else on tmp rounded to out_dtype. */
void te_norm_output_ref(int is_ln, const void *x, const void *gamma, const void *beta, int dtype, int wdtype,
                        float *out, const float *mu, const float *rsigma, size_t N, size_t H, float *amax,
                        float scale, int zero_centered_gamma, int out_dtype) {
  float current_max = 0.f;
  for (size_t i = 0; i < N; ++i)
    for (size_t j = 0; j < H; ++j) {
      const float cur = load_in(x, i * H + j, dtype);
      float g = load_in(gamma, j, wdtype);
      if (zero_centered_gamma) g += 1.f;
      float tmp;
      if (is_ln) tmp = (cur - mu[i]) * rsigma[i] * g + load_in(beta, j, wdtype);
      else tmp = cur * rsigma[i] * g;
      out[i * H + j] = tmp * scale;
      if (scale != 1.f) current_max = fmaxf(current_max, fabsf(tmp));
      else current_max = fmaxf(current_max, fabsf(round_out(tmp, out_dtype)));
    }
  if (amax) *amax = current_max;
}

void te_norm_backward_ref(int is_ln, const void *dz, const void *add, const void *x, int dtype, int wdtype,
                          const float *mu, const float *rsigma, const void *gamma, float *dx, float *dgamma,
                          float *dbeta, size_t N, size_t H, int zero_centered_gamma) {
  for (size_t j = 0; j < H; ++j) { dgamma[j] = 0.f; if (dbeta) dbeta[j] = 0.f; }
  for (size_t i = 0; i < N; ++i) {
    const float local_mu = is_ln ? mu[i] : 0.f;
    float mdy = 0.f, mdyy = 0.f;
    for (size_t j = 0; j < H; ++j) {
      const float xv = load_in(x, i * H + j, dtype);
      const float y = (xv - local_mu) * rsigma[i];
      float g = load_in(gamma, j, wdtype);
      if (zero_centered_gamma) g += 1.f;
      const float dzv = load_in(dz, i * H + j, wdtype);
      const float dy = g * dzv;
      dgamma[j] += y * dzv;
      if (is_ln) { dbeta[j] += dzv; mdy += dy; }
      mdyy += dy * y;
    }
    mdy /= H;
    mdyy /= H;
    for (size_t j = 0; j < H; ++j) {
      const float xv = load_in(x, i * H + j, dtype);
      const float y = (xv - local_mu) * rsigma[i];
      float g = load_in(gamma, j, wdtype);
      if (zero_centered_gamma) g += 1.f;
      const float dy = g * load_in(dz, i * H + j, wdtype);
      const float a = add ? load_in(add, i * H + j, wdtype) : 0.f;
      dx[i * H + j] = a + rsigma[i] * (dy - mdyy * y - mdy);
    }
  }
}

/* ------------------------------------------------------------------ activations
 * ref common/util/math.h:21-134.  id: 1 gelu, 2 silu, 3 relu, 4 qgelu, 5 srelu, 6 sigmoid (the "act"
 * of plain GLU); deriv != 0 returns the derivative. */
static inline float sigmoid_ref(float x) { return 1.f / (1.f + expf(-x)); }
float te_act_scalar_ref(int id, int deriv, float x) {
  switch (id) {
    case 1:
      if (!deriv) return x * (0.5f + 0.5f * tanhf(x * (0.79788456f + 0.03567741f * x * x)));
      else {
        const float t = tanhf(0.79788456f * x * (1.f + 0.044715f * x * x));
        return 0.5f * x * ((1.f - t * t) * (0.79788456f + 0.1070322243f * x * x)) + 0.5f * (1.f + t);
      }
    case 2: { const float s = sigmoid_ref(x); return deriv ? x * (s * (1.f - s)) + s : x * s; }
    case 3: return deriv ? (x > 0.f ? 1.f : 0.f) : fmaxf(x, 0.f);
    case 4: { const float s = sigmoid_ref(1.702f * x);
              return deriv ? 1.702f * x * (s * (1.f - s)) + s : x * s; }
    case 5: return deriv ? fmaxf(2.f * x, 0.f) : (x > 0.f ? x * x : 0.f);
    case 6: { const float s = sigmoid_ref(x); return deriv ? s * (1.f - s) : s; }
    default: return deriv ? 1.f : x;
  }
}

/* mode 0: out[r,c] = act(in[r,c]); 1: grad*act'(in); 2 (gated fwd, in [R,2H]): act(a)*b;
 * 3 (gated bwd, out [R,2H]): [act'(a)*grad*b, act(a)*grad]   (ref util/vectorized_pointwise.h:410-560).
 * fp32 results before any output conversion. */
void te_act_ref(int mode, int id, const void *in, const void *grad, int dtype, float *out, size_t R, size_t H) {
  const size_t ld_in = (mode >= 2) ? 2 * H : H, ld_out = (mode == 3) ? 2 * H : H;
  for (size_t r = 0; r < R; ++r)
    for (size_t c = 0; c < H; ++c) {
      const float a = load_in(in, r * ld_in + c, dtype);
      const float b = (mode >= 2) ? load_in(in, r * ld_in + H + c, dtype) : 0.f;
      const float g = (mode == 1 || mode == 3) ? load_in(grad, r * H + c, dtype) : 0.f;
      if (mode == 0) out[r * ld_out + c] = te_act_scalar_ref(id, 0, a);
      else if (mode == 1) out[r * ld_out + c] = g * te_act_scalar_ref(id, 1, a);
      else if (mode == 2) out[r * ld_out + c] = te_act_scalar_ref(id, 0, a) * b;
      else {
        out[r * ld_out + c] = te_act_scalar_ref(id, 1, a) * g * b;
        out[r * ld_out + H + c] = g * te_act_scalar_ref(id, 0, a);
      }
    }
}

/* ---------------------------------------------------------------- FP8 weight-update path
 * Shard ("partial") amax / cast of an MXFP8 master weight: restates the torch references the
 * reference's own test holds, ref tests/pytorch/test_partial_cast.py:18-53
 * (compute_partial_amax_reference: scatter the shard into a zero [h, w] matrix, |x|, max over
 * 1x32 / 32x1 blocks; partial_cast_reference: x * 2^(127 - e) -> e4m3 with the rowwise and the
 * columnwise block scale), kernels ref common/recipe/mxfp8_scaling.cu:26-131.
 * amax arrays are fp32 here (the values are exact in the input dtype). */
void te_mxfp8_partial_amax_ref(const void *in, int in_dtype, size_t len, size_t rows, size_t cols,
                               size_t start, float *amax_row, size_t row_stride, float *amax_col,
                               size_t col_stride) {
  for (size_t r = 0; r < rows; ++r)
    for (size_t cb = 0; cb < cols / 32; ++cb) amax_row[r * row_stride + cb] = 0.f;
  for (size_t rb = 0; rb < rows / 32; ++rb)
    for (size_t c = 0; c < cols; ++c) amax_col[rb * col_stride + c] = 0.f;
  for (size_t i = 0; i < len; ++i) {
    const size_t idx = start + i, r = idx / cols, c = idx % cols;
    const float a = fabsf(load_in(in, i, in_dtype));
    float *pr = &amax_row[r * row_stride + c / 32], *pc = &amax_col[(r / 32) * col_stride + c];
    *pr = fmaxf(*pr, a);
    *pc = fmaxf(*pc, a);
  }
}

/* ref common/multi_tensor/compute_scale.cu:59-79: e = float_to_e8m0(amax * (1/448)); the
 * reference's test restates it with a division, tests/pytorch/test_multi_tensor.py:398-406
 * (the two agree on every bf16 amax: tests/test_oracle_weight_update.py). */
void te_e8m0_from_amax_ref(const float *amax, size_t n, uint8_t *out) {
  const float max_norm_rcp = 1.0f / 448.0f;
  for (size_t i = 0; i < n; ++i) out[i] = te_float_to_e8m0(amax[i] * max_norm_rcp);
}

void te_mxfp8_partial_cast_ref(const void *in, int in_dtype, size_t len, size_t rows, size_t cols,
                               size_t start, const uint8_t *sc_row, size_t row_stride,
                               const uint8_t *sc_col, size_t col_stride, uint8_t *out_row,
                               uint8_t *out_col) {
  (void)rows;
  for (size_t i = 0; i < len; ++i) {
    const size_t idx = start + i, r = idx / cols, c = idx % cols;
    const float x = load_in(in, i, in_dtype);
    out_row[i] = te_float_to_fp8(x * te_exp2f_rcp(sc_row[r * row_stride + c / 32]), 0);
    out_col[i] = te_float_to_fp8(x * te_exp2f_rcp(sc_col[(r / 32) * col_stride + c]), 0);
  }
}

/* ref common/multi_tensor/compute_scale.cu:26-57 (+ recipe_common.cuh:14-49); the reference's
 * test compares with scale_from_amax_tensor, rtol = atol = 0 (tests/pytorch/test_multi_tensor.py:
 * 343-374). */
void te_scale_and_scale_inv_ref(const float *amax, size_t n, float max_fp8, int force_pow_2, float eps,
                                float *scale, float *scale_inv) {
  for (size_t i = 0; i < n; ++i) {
    scale[i] = te_scale_from_amax_ref(amax[i], max_fp8, force_pow_2, eps);
    scale_inv[i] = 1.0f / scale[i];
  }
}

/* ref common/multi_tensor/adam.cu:52-176 (AdamFunctorMaster with an FP8 PARAM_T) and :775-780
 * (bias corrections).  Every operation rounds to fp32 on its own (this file is compiled with
 * -ffp-contract=off, the HIP kernel with contraction off in the same function); the reference
 * binary may fuse some multiply-adds, which moves results by at most an fp32 ulp per step.
 * Returns max |p_master| after the update. */
float te_adam_fp8_ref(const void *g, int g_dtype, size_t n, float *pm, float *m, float *v, uint8_t *p8,
                      float scale, int fmt, float lr, float beta1, float beta2, float eps, int step,
                      int mode, int bias_correction, float decay) {
  float c1 = 1.f, c2 = 1.f;
  if (bias_correction == 1) {
    c1 = (float)(1 - pow((double)beta1, step));
    c2 = (float)(1 - pow((double)beta2, step));
  }
  float amax = 0.f;
  for (size_t i = 0; i < n; ++i) {
    float rg = load_in(g, i, g_dtype), rp = pm[i], rm = m[i], rv = v[i];
    if (mode == 0) {
      rg = rg + (decay * rp);
      rm = beta1 * rm + (1 - beta1) * rg;
      rv = beta2 * rv + (1 - beta2) * rg * rg;
      const float mh = rm / c1, vh = rv / c2;
      const float denom = sqrtf(vh) + eps;
      rp = rp - (lr * (mh / denom));
    } else {
      rm = beta1 * rm + (1 - beta1) * rg;
      rv = beta2 * rv + (1 - beta2) * rg * rg;
      const float mh = rm / c1, vh = rv / c2;
      const float denom = sqrtf(vh) + eps;
      const float upd = (mh / denom) + (decay * rp);
      rp = rp - (lr * upd);
    }
    pm[i] = rp;
    m[i] = rm;
    v[i] = rv;
    amax = fmaxf(amax, fabsf(rp));
    p8[i] = te_float_to_fp8(rp * scale, fmt);
  }
  return amax;
}

/* ---------------------------------------------------------------- clamped SwiGLU (GPT-OSS)
 * ref common/util/math.h:80-98 (clamped_silu, clamped_dsilu = qgelu_with_alpha / dqgelu_with_alpha of the
 * clamped value) and util/vectorized_pointwise.h:436-438 (forward: gate clamped to [-limit, limit] + offset),
 * :543-553 (backward: the gate half's gradient is 0 where the clamp was active).
 * mode 0: out [R, H]; mode 1: out [R, 2H]. */
void te_clamped_swiglu_ref(int mode, const void *in, const void *grad, int dtype, float *out, size_t R, size_t H,
                           float limit, float alpha, float offset) {
  for (size_t r = 0; r < R; ++r)
    for (size_t c = 0; c < H; ++c) {
      const float a = load_in(in, r * 2 * H + c, dtype);
      float g = load_in(in, r * 2 * H + H + c, dtype);
      const float cv = fminf(a, limit);
      const float s = sigmoid_ref(alpha * cv);
      const float act = cv * s;
      const int pass = g <= limit && g >= -limit;
      g = fminf(fmaxf(-limit, g), limit) + offset;
      if (mode == 0) {
        out[r * H + c] = act * g;
      } else {
        const float dy = load_in(grad, r * H + c, dtype);
        const float dact = a <= limit ? alpha * cv * (s * (1.f - s)) + s : 0.f;
        out[r * 2 * H + c] = dact * dy * g;
        out[r * 2 * H + H + c] = pass ? dy * act : 0.f;
      }
    }
}

int te_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
