// Scaled FP8 / MXFP8 GEMM on CDNA4 matrix cores (gfx950).  No BLAS library.
//
// Replaces: ref common/gemm/cublaslt_gemm.cu:102-307 (operand canonicalisation),
//           :315-799 (cublas_gemm), :803-968 (C entry points), :970-1060 (multi-tensor).
//
// Problem after canonicalisation ("TN everywhere", see nvte_is_non_tn_fp8_gemm_supported):
//     D[i][j] = alpha * sP * sQ * sum_k P[i][k] * Q[j][k]  (+ beta * D[i][j]) (+ bias[j])
// with P = the reference's operand B as an [n, k] K-contiguous byte matrix,
//      Q = the reference's operand A as an [m, k] K-contiguous byte matrix,
//      D row-major [n, m] (ld = m), fp32 accumulation.
// MXFP8: sP / sQ are e8m0 bytes per (row, 32-element K block) applied by the
// block-scaled MFMA itself.
//
// Kernel: 256x256 output tile per 512-thread workgroup (8 waves as 2 x 4, each 128 x 64 =
// 4 x 2 MFMA tiles of 32x32), K tile 128 bytes, LDS double buffer 2 x (32 KiB P + 32 KiB Q)
// filled by buffer_load ... lds (LDS-DMA, 16 B per lane), v_mfma_[scale_]f32_32x32x64_f8f6f4
// (the rate-doubled FP8 path: 64 K per instruction).  The LDS image is lane-linear as LDS-DMA
// requires; bank conflicts of the ds_read_b128 fragment reads are removed by XOR-ing the
// 16-byte chunk index with ((row >> 1) & 7) on the SOURCE address and on the read address.
// Out-of-range rows and the K tail are zero-filled by the buffer descriptor's range check, so
// any n, m and any K % 16 == 0 run through the same kernel.
//
// Measured on MI355X while shaping the main loop: the MFMA pipe alone
// sustains 4.97 PFLOP/s on constant operands but 3.92 PFLOP/s on random operand bits (clock
// drops under switching power), global->LDS ingest alone tops out near 19 TB/s chip-wide, and a
// 4-wave / 128x128-per-wave variant (1/3 fewer LDS reads) was not faster than this layout.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "device_utils.h"
#include "transformer_engine/gemm.h"
#include "transformer_engine/multi_stream.h"

// Build split: the 60 kernel instantiations take minutes to compile, so the Makefile builds this file five times in
// parallel - TE_GEMM_PART 0..3 = the kernels of one (P format, Q format) pair behind gemm_launch_fmt_<part>(), 4 = everything
// else (host code, the small helper kernels).  Undefined = one translation unit with all of it (diagnostic builds).
#ifndef TE_GEMM_PART
#define TE_GEMM_PART -1
#endif
#define TE_GEMM_HAS_FMT(i) (TE_GEMM_PART == -1 || TE_GEMM_PART == (i))
#define TE_GEMM_HAS_HOST (TE_GEMM_PART == -1 || TE_GEMM_PART == 4)

namespace te {
// kernel launchers of one format pair (P format * 2 + Q format); `key` = variant_key(MX, EPI, GROUPED, KM), `args` = GemmArgs
void gemm_launch_fmt_0(int key, const void *args, hipStream_t stream);
void gemm_launch_fmt_1(int key, const void *args, hipStream_t stream);
void gemm_launch_fmt_2(int key, const void *args, hipStream_t stream);
void gemm_launch_fmt_3(int key, const void *args, hipStream_t stream);

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int BM = 256;   // tile rows   (index i, rows of P)
constexpr int BN = 256;   // tile cols   (index j, rows of Q)
constexpr int BK = 128;   // bytes of K per stage
constexpr int kThreads = 512;
constexpr int kStageBytes = (BM + BN) * BK;  // 64 KiB
constexpr int kScaleSlotBytes = 2048;        // MX: e8m0 scale dwords of one K tile, 256 P rows + 256 Q rows
constexpr unsigned kOOB = 0x80000000u;
[[maybe_unused]] constexpr size_t kNtStoreBytes = 32u << 20;  // outputs beyond this are written with streaming stores
// cache-policy bits of the operand staging loads (gfx940+: bit 0 = sc0, bit 1 = nt, bit 4 = sc1); a build-time
// knob for A/B runs (make variant), the shipped value is the measured best
#ifndef TE_GEMM_AUX
#define TE_GEMM_AUX 0
#endif
constexpr int kStageAux = TE_GEMM_AUX;

struct GemmArgs {
  const uint8_t *P;   // [n, k]
  const uint8_t *Q;   // [m, k]
  void *D;            // [n, m]
  const void *bias;   // [m] or null
  const float *sP;    // per-tensor scale_inv (device) or null
  const float *sQ;
  const uint8_t *eP;  // MX scales for P rows
  const uint8_t *eQ;
  unsigned n, m, k;
  unsigned ldd;
  unsigned ldP, ldQ;              // K-MAJOR operands (kernel variant KM): bytes between consecutive k of one row
  unsigned eP_stride, eQ_stride;  // compact layout row stride (bytes)
  unsigned eP_swz, eQ_swz;        // 1 = GEMM-swizzled 128x4 tiles
  unsigned eP_bytes, eQ_bytes;    // size of the scale buffers
  float alpha, beta;
  int d_dtype;     // NVTEDType of D
  int bias_dtype;  // NVTEDType of bias
  unsigned tiles_j;  // number of tile columns
  unsigned ntiles;
  unsigned grid;     // persistent workgroups (<= ntiles)
  unsigned nt_store; // 1 = the output is large (> kNtStoreBytes): full tiles use streaming stores
  // extended epilogues (kernel variant EPI = true only)
  int epi;                // kEpiNone / kEpiGeluAux / kEpiDgelu
  void *aux;              // [n, m] pre-GELU values: written (GELU_AUX) or read (DGELU)
  int aux_dtype;
  const float *d_scale;   // FP8 output: scale (device, null = 1) ...
  float *d_amax;          // ... and amax of the unscaled output (device, may be null)
  float *dbias_acc;       // DGELU_BGRAD: fp32 [m] column sums of D (atomics), or null
  // grouped launch (kernel variant GROUPED = true only): per-member problems + tile prefix,
  // both written on the device by grouped_setup_kernel
  const struct GroupDesc *groups;
  const struct GroupScaleDesc *group_scales;  // MX members only: where each member's e8m0 scales sit
  const unsigned *group_tile_start;  // [num_groups + 1]
  unsigned num_groups;
#ifdef TE_GEMM_STAMPS
  unsigned long long *stamps;  // diagnostic build only: [grid][kStampSlots] s_memrealtime values
  unsigned long long *wstamps; // diagnostic build only: [grid][8 waves][4] = cycles in {vmcnt wait, barrier wait, K loop, K tiles}
  int dbg;                     // diagnostic build only: TE_GEMM_DBG (tile-order experiments)
#endif
};
#ifdef TE_GEMM_STAMPS
constexpr unsigned kStampSlots = 512;
static unsigned long long *g_stamps = nullptr;
static unsigned long long *g_wstamps = nullptr;
#define TE_STAMP()                                                                                   \
  do {                                                                                               \
    if (a.stamps != nullptr && tid == 0 && sidx < kStampSlots)                                       \
      a.stamps[blockIdx.x * kStampSlots + sidx] = __builtin_amdgcn_s_memrealtime();                  \
    ++sidx;                                                                                          \
  } while (0)
#else
#define TE_STAMP() do {} while (0)
#endif

// One member GEMM of a grouped launch.
struct GroupDesc {
  const uint8_t *P;  // [n, k]
  const uint8_t *Q;  // [m, k]
  void *D;           // [n, m]
  const void *C;     // accumulation input (== D, another buffer, or null)
  float mult;        // alpha * scale_inv(P) * scale_inv(Q); NaN marks an inconsistent member
  float beta;
  unsigned n, m, k;
  unsigned tiles_j;
};

// MXFP8 members: the e8m0 scale matrices of the member's P and Q operands (device-written)
struct GroupScaleDesc {
  const uint8_t *eP, *eQ;
  unsigned eP_stride, eQ_stride;  // bytes per scale row: pad4(ceil(k / 32))
  unsigned eP_bytes, eQ_bytes;    // padded size of the member's scale matrix
};

[[maybe_unused]] constexpr int kEpiNone = 0;
constexpr int kEpiGeluAux = 1;  // aux = pre-activation, D = gelu(aux)
constexpr int kEpiDgelu = 2;    // D = dgelu(aux) * (alpha P Q^T + beta C)

// tanh-form GELU and its derivative, the forms of ref common/util/math.h:21-32
__device__ __forceinline__ float gelu_tanh(float x) {
  return x * (0.5f + 0.5f * tanhf(x * (0.79788456f + 0.03567741f * x * x)));
}
__device__ __forceinline__ float dgelu_tanh(float x) {
  const float t = tanhf(0.79788456f * x * (1.f + 0.044715f * x * x));
  return 0.5f * x * ((1.f - t * t) * (0.79788456f + 0.1070322243f * x * x)) + 0.5f * (1.f + t);
}

// XCD-aware, bijective remap of the flat workgroup id: workgroups that share an XCD
// (id % 8) get a contiguous run of tiles; inside a run tiles are walked in groups of
// GROUP_I tile rows so that co-resident workgroups share P and Q panels in L2.
__device__ __forceinline__ void tile_of(unsigned wg, unsigned ntiles, unsigned tiles_i,
                                        unsigned tiles_j, unsigned &ti, unsigned &tj, unsigned GROUP_I = 8) {
  const unsigned q = ntiles / 8, r = ntiles % 8, xcd = wg % 8, local = wg / 8;
  const unsigned lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  const unsigned per_group = GROUP_I * tiles_j;
  const unsigned g = lin / per_group;
  const unsigned first_i = g * GROUP_I;
  const unsigned gsz = min(tiles_i - first_i, GROUP_I);
  const unsigned in_g = lin % per_group;
  ti = first_i + in_g % gsz;
  tj = in_g / gsz;
}

template <typename T>
__device__ __forceinline__ float load_as_float(const void *p, size_t idx);
template <>
__device__ __forceinline__ float load_as_float<float>(const void *p, size_t idx) {
  return static_cast<const float *>(p)[idx];
}
template <>
__device__ __forceinline__ float load_as_float<bf16_t>(const void *p, size_t idx) {
  return widen(static_cast<const bf16_t *>(p)[idx]);
}
template <>
__device__ __forceinline__ float load_as_float<f16_t>(const void *p, size_t idx) {
  return widen(static_cast<const f16_t *>(p)[idx]);
}

__device__ __forceinline__ float load_any(const void *p, size_t idx, int dtype) {
  switch (dtype) {
    case kNVTEFloat32: return load_as_float<float>(p, idx);
    case kNVTEBFloat16: return load_as_float<bf16_t>(p, idx);
    default: return load_as_float<f16_t>(p, idx);
  }
}

__device__ __forceinline__ void store_any(void *p, size_t idx, int dtype, float v) {
  switch (dtype) {
    case kNVTEFloat32: static_cast<float *>(p)[idx] = v; break;
    case kNVTEBFloat16: narrow(v, static_cast<bf16_t *>(p)[idx]); break;
    case kNVTEFloat16: narrow(v, static_cast<f16_t *>(p)[idx]); break;
    case kNVTEFloat8E4M3: static_cast<uint8_t *>(p)[idx] = cvt1_fp8<Fp8::E4M3>(v); break;
    default: static_cast<uint8_t *>(p)[idx] = cvt1_fp8<Fp8::E5M2>(v); break;
  }
}

// LDS-DMA (buffer_load ... lds) issued from inline assembly.  The compiler's own form of the instruction
// (__builtin_amdgcn_raw_ptr_buffer_load_lds) makes its wait-count pass treat EVERY later LDS store - and the transposing
// LDS loads of the K-major variants, which it models as stores - as a possible read of the DMA's destination: it then puts
// an s_waitcnt vmcnt(0), a wait for the K tile staged a moment ago, in front of them (4 per K tile in the K-major kernels,
// one at the start of every epilogue).  The kernel orders DMA against LDS reads itself (vmcnt + s_barrier at the K-tile
// hand-over), so the instruction is hidden from that pass
// in the kernel variants where that pays (kAsmDma in the kernel); TE_GEMM_BUILTIN_DMA forces the builtin everywhere (A/B builds).
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i_t make_rsrc(const void *base, unsigned bytes) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  return v4i_t{static_cast<int>(static_cast<uint32_t>(b)), static_cast<int>(static_cast<uint32_t>(b >> 32) & 0xFFFFu),
               static_cast<int>(bytes), 0x00020000};
}
// 64 lanes x 16 bytes -> LDS [lds_addr, lds_addr + 1024); source byte = rsrc base + voff (per lane) + soff
// (the scalar operands are wave-uniform by construction; readfirstlane makes that known where the values come out of
// memory - the grouped kernels' member descriptors - and folds away elsewhere)
__device__ __forceinline__ v4i_t uniform4(const v4i_t r) {
  return v4i_t{__builtin_amdgcn_readfirstlane(r[0]), __builtin_amdgcn_readfirstlane(r[1]),
               __builtin_amdgcn_readfirstlane(r[2]), __builtin_amdgcn_readfirstlane(r[3])};
}
// Wait states inside the string (the compiler pads nothing there): `s_nop 3` separates the write of M0 from the DMA that
// reads it (1 state needed) and, together with the s_mov, gives the 5 states a VMEM instruction needs after a VALU
// instruction wrote one of its SGPR operands - which happens right in front of the statement whenever the compiler
// reloads a spilled SGPR (v_readlane) or makes a descriptor word uniform (v_readfirstlane, the grouped kernels).
__device__ __forceinline__ void glds_b128(v4i_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  rsrc = uniform4(rsrc);
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// Two DMAs in one statement (the P and the Q piece of one staging step): the second one's scalar operands are more than
// 5 states past anything in front of the statement, so it only needs the one state behind its own write of M0.
// BARE: no padding for a VALU-written SGPR operand - only where the generated code is KNOWN to hold no VALU instruction
// that writes an SGPR (v_readlane / v_readfirstlane / v_cmp to an SGPR pair) anywhere near: the steady-state K loop of
// the dense per-tensor kernels, checked at every build of the test suite by tests/test_gemm_codegen.py.
template <bool BARE>
__device__ __forceinline__ void glds_b128_pair(v4i_t r0, unsigned lds0, unsigned voff0, unsigned soff0,
                                               v4i_t r1, unsigned lds1, unsigned voff1, unsigned soff1) {
  r0 = uniform4(r0);
  r1 = uniform4(r1);
  lds0 = __builtin_amdgcn_readfirstlane(lds0);
  lds1 = __builtin_amdgcn_readfirstlane(lds1);
  soff0 = __builtin_amdgcn_readfirstlane(soff0);
  soff1 = __builtin_amdgcn_readfirstlane(soff1);
  if constexpr (BARE)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds"
                 :: "s"(lds0), "v"(voff0), "s"(r0), "s"(soff0), "s"(lds1), "v"(voff1), "s"(r1), "s"(soff1) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, %7 offen lds"
                 :: "s"(lds0), "v"(voff0), "s"(r0), "s"(soff0), "s"(lds1), "v"(voff1), "s"(r1), "s"(soff1) : "memory");
}
// 64 lanes x 4 bytes -> LDS [lds_addr, lds_addr + 256)
__device__ __forceinline__ void glds_b32(v4i_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  rsrc = uniform4(rsrc);
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dword %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// Main-loop pipeline:
//  * every MFMA fragment register is refilled from LDS right after its last use, 5..7 MFMAs
//    before its next one, so the matrix pipe never waits on LDS; LDS and global->LDS
//    instructions are issued in the shadow of the MFMAs (hand-made order, pinned with
//    sched_barrier);
//  * ONE barrier per 128-byte K tile, placed after MFMA 10 of 16: by then every wave has pulled
//    all of K tile t out of LDS slot t&1, so the slot is refilled with K tile t+2 immediately, a
//    full tile before it is needed (the loads of K tile t+1, issued one tile ago, are what the
//    barrier's vmcnt(0) waits for);
//  * PERSISTENT workgroups: workgroup w walks output tiles w, w + grid, w + 2 grid, ... and the
//    K-tile pipeline runs straight across output-tile boundaries: the last two K tiles of an
//    output tile already stage the first two K tiles of the next one, so neither the load
//    latency of a fresh tile nor the drain of the previous tile's stores is exposed;
//  * the accumulators leave through a wave-private LDS transpose (its own 32 KiB of LDS, the
//    two pipeline slots stay live) so that global stores are full 128/256-byte row segments.
//  * EPI = true is the same kernel with the extended epilogue (GELU / dGELU with the aux matrix,
//    bias-gradient column sums, FP8 output with amax) - element-wise, off the Linear hot path;
//    the plain instantiation is untouched by it.
//  * GROUPED = true walks the tiles of MANY member GEMMs (one persistent launch for all experts):
//    member shapes and pointers come from device memory (GroupDesc), every member runs at least
//    two K tiles (the K tail is zero-filled anyway) so that the cross-tile pipeline is uniform.
// ABL (diagnostic build only): ablations that keep the instruction stream's shape - 1: no global->LDS
// staging in the main loop, 2: no LDS fragment reads, 3: neither (MFMA + barrier only).  Results are wrong.
// KM (bit 0: P, bit 1: Q): that operand is stored K-MAJOR - element (row, k) at k * ld + row, the layout of the
// reference's un-transposed columnwise MXFP8 payload (ref cublaslt_gemm.cu:179-193, :268-278 hands it to cuBLASLt as a
// non-TN operand).  LDS-DMA lays such a K tile down as [k][256 rows]; the MFMA fragments (8 consecutive k per lane
// and read) come out of it through the transposing LDS load ds_read_b64_tr_b8.
template <int FA /*format of P: 0 e4m3, 1 e5m2*/, int FB /*format of Q*/, bool MX, bool EPI, bool GROUPED = false, int ABL = 0,
          int KM = 0>
__global__ __launch_bounds__(kThreads) void gemm_fp8_kernel(const GemmArgs a) {
  static_assert(!(GROUPED && EPI), "the grouped variant has no extended epilogues");
  static_assert(KM == 0 || !EPI, "K-major operands: basic epilogue only");
  static_assert(MX || (KM & 12) == 0, "scale-layout bits belong to the MXFP8 variants");
  // bits 2 / 3: that K-major operand's e8m0 scales are COMPACT, i.e. stored [k block][row] like the payload (the
  // reference's columnwise scale matrix); otherwise GEMM-swizzled, one format whatever the quantization direction.
  // Compile-time on purpose: a run-time branch in the main loop lets the scheduler tear up the pinned MFMA order
  // (measured: 300-430 spilled registers, 8x slower).
  constexpr bool PKM = (KM & 1) != 0, QKM = (KM & 2) != 0, PKC = PKM && (KM & 4) != 0, QKC = QKM && (KM & 8) != 0;
  // Which form of the LDS-DMA instruction stages the operands (see glds_b128): the assembly form where the compiler's
  // waits cost more than its extra scalar instructions do - both operands K-major (wgrad: 463 -> 371 us per-tensor, 511 ->
  // 442 us MXFP8 at the configs[1] / 8192^3 shapes) and the per-tensor dgrad (346 -> 336 us); the builtin form where
  // they do not (K-contiguous operands: 2 % faster with the builtin; MXFP8 with only Q K-major: 3 %).
#if defined(TE_GEMM_BUILTIN_DMA)
  constexpr bool kAsmDma = false;
#elif defined(TE_GEMM_ASM_DMA_ALL)   // A/B builds: the assembly form in every dense kernel
  constexpr bool kAsmDma = !GROUPED;
#else
  constexpr bool kAsmDma = PKM || (!MX && QKM);
#endif
  // LDS map: [0, 128 KiB) two K-tile slots | MX only: 2 x 2 KiB scale slots | epilogue staging
  constexpr int kScaleBase = 2 * kStageBytes;
  constexpr int kEpiBase = kScaleBase + (MX ? 2 * kScaleSlotBytes : 0);
  constexpr int kEpiRows = MX ? 8 : 16;  // rows per transpose pass (x 64 fp32 columns per wave)
  __shared__ __attribute__((aligned(16))) uint8_t lds[kEpiBase + 8 * kEpiRows * 256];
  typedef __attribute__((address_space(3))) uint8_t lds_u8;

  const unsigned tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned wi = wave >> 2, wj = wave & 3u;
#ifdef TE_GEMM_STAMPS
  unsigned sidx = 0;
  if (a.stamps != nullptr && tid == 0) {
    a.stamps[blockIdx.x * kStampSlots + kStampSlots - 1] = __builtin_amdgcn_s_memtime();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.stamps[blockIdx.x * kStampSlots + kStampSlots - 3] = xcc;
  }
  TE_STAMP();
#endif
  const unsigned tiles_i = (a.n + BM - 1) / BM;
  const unsigned nk = (a.k + BK - 1) / BK;
  unsigned total_tiles = a.ntiles;
  unsigned grp = 0;  // GROUPED: member that holds the tile located last
  // GROUPED: the global tile list is cut into 8 contiguous runs, one per XCD (workgroup id % 8), whose
  // boundaries the setup kernel chose by COST (K tiles + epilogue per tile), not by count: members with a
  // different K (dgrad + wgrad of one layer in one launch) then still finish together.  The table sits
  // behind the tile prefix; run x is [xrun[x], xrun[x + 1]).
  unsigned xrun[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto run_of = [&](unsigned flat, unsigned &pos) {   // flat workgroup-space index -> list position, valid?
    const unsigned x = flat & 7u, loc = flat >> 3;
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (x == static_cast<unsigned>(i)) { lo = xrun[i]; hi = xrun[i + 1]; }
    pos = lo + loc;
    return pos < hi;
  };
  if constexpr (GROUPED) {
    total_tiles = a.group_tile_start[a.num_groups];
#pragma unroll
    for (int i = 0; i < 9; ++i) xrun[i] = a.group_tile_start[a.num_groups + 1 + i];
    unsigned pos0;
    if (!run_of(blockIdx.x, pos0)) return;  // whole workgroup, before any barrier
  }

  // MX: the scale buffer this wave stages (P's for waves 0-3, Q's for waves 4-7)
  const uint8_t *e_base = wave < 4 ? a.eP : a.eQ;
  const unsigned e_bytes = wave < 4 ? a.eP_bytes : a.eQ_bytes;
  const unsigned e_stride = wave < 4 ? a.eP_stride : a.eQ_stride;
  const bool e_swz = (wave < 4 ? a.eP_swz : a.eQ_swz) != 0;

  // Everything that changes from one output tile to the next (all wave-uniform).  The buffer
  // descriptors cover exactly the tile's rows: rows past the matrix end and explicit kOOB
  // offsets read as zero; an invalid tile (past the end of this workgroup's list) gets empty
  // descriptors, so the loads staged "for the next tile" become harmless zero fills.
  struct TileCtx {
    uint8_t *pbase, *qbase;      // first row of the tile in P / Q
    unsigned pbytes, qbytes;     // bytes the tile's descriptors may touch (0 = everything reads 0)
    unsigned i0, j0;
    unsigned ss;                 // MX: scalar part of this wave's scale source address (K tile 0)
    // GROUPED && MX: the member's scale matrix this wave stages from (P's for waves 0-3, Q's for 4-7)
    const uint8_t *eb;
    unsigned ebytes, estride;    // (all wave-uniform: the lane part of the address is rebuilt per K tile)
    // GROUPED only: the member's own problem
    unsigned n, m, k;
    float mult, beta;
    char *D;
    const char *C;
  };
  auto make_tile = [&](unsigned tile, bool valid) {
    TileCtx t;
    unsigned ti, tj;
    if constexpr (GROUPED) {
      {
        // XCD-aware as tile_of: the workgroups of one XCD (id % 8) walk a contiguous run of the global
        // tile list, so an XCD's L2 holds few P / Q panels at a time.
        unsigned pos;
        run_of(tile, pos);
        tile = valid ? pos : 0u;
      }
      // stateless member lookup (the member g with tile_start[g] <= tile < tile_start[g + 1]; empty
      // members have equal bounds and are skipped): with a grid that is not a multiple of 8 (a
      // caller-capped workgroup count) the remapped index of a workgroup's next tile can move
      // backwards, so a running index would be wrong.
      {
        unsigned lo = 0, hi = a.num_groups - 1;
        while (lo < hi) {
          const unsigned mid = (lo + hi) >> 1;
          if (a.group_tile_start[mid + 1] > tile) hi = mid; else lo = mid + 1;
        }
        grp = lo;
      }
      if (!valid) grp = 0;
      const GroupDesc &d = a.groups[grp];
      const unsigned local = tile - a.group_tile_start[grp];
      // tiles in groups of 8 tile rows, column by column inside a group (panel reuse in L2)
      const unsigned m_tiles_i = (d.n + BM - 1) / BM;
      const unsigned per_group = 8 * d.tiles_j;
      const unsigned first_i = (local / per_group) * 8;
      const unsigned gsz = min(m_tiles_i - first_i, 8u);
      const unsigned in_g = local % per_group;
      ti = first_i + in_g % gsz;
      tj = in_g / gsz;
      t.i0 = ti * BM;
      t.j0 = tj * BN;
      t.n = d.n; t.m = d.m; t.k = d.k;
      t.mult = d.mult; t.beta = d.C ? d.beta : 0.f;
      t.D = static_cast<char *>(d.D);
      t.C = static_cast<const char *>(d.C ? d.C : d.D);
      const unsigned rows_p = valid ? min(d.n - t.i0, static_cast<unsigned>(BM)) : 0u;
      const unsigned rows_q = valid ? min(d.m - t.j0, static_cast<unsigned>(BN)) : 0u;
      t.pbase = const_cast<uint8_t *>(d.P) + static_cast<size_t>(t.i0) * d.k;
      t.qbase = const_cast<uint8_t *>(d.Q) + static_cast<size_t>(t.j0) * d.k;
      t.pbytes = rows_p * d.k;
      t.qbytes = rows_q * d.k;
      // K-major member (its row-wise payload, contracted over its rows; every member has the launch's row length
      // a.ldP / a.ldQ): the descriptor starts at the tile's first row inside k-row 0 and ends with the member
      // (a member without rows - an expert without tokens - has k = 0: empty descriptors, its output is zero)
      if constexpr (PKM) {
        t.pbase = const_cast<uint8_t *>(d.P) + t.i0;
        t.pbytes = (valid && d.k != 0) ? d.k * a.ldP - t.i0 : 0u;
      }
      if constexpr (QKM) {
        t.qbase = const_cast<uint8_t *>(d.Q) + t.j0;
        t.qbytes = (valid && d.k != 0) ? d.k * a.ldQ - t.j0 : 0u;
      }
      if constexpr (MX) {
        const GroupScaleDesc &sd = a.group_scales[grp];
        const unsigned stride = wave < 4 ? sd.eP_stride : sd.eQ_stride;
        t.eb = wave < 4 ? sd.eP : sd.eQ;
        t.ebytes = valid ? (wave < 4 ? sd.eP_bytes : sd.eQ_bytes) : 0u;
        t.estride = stride;
        const unsigned tile_row0 = wave < 4 ? t.i0 : t.j0;
        t.ss = e_swz ? ((tile_row0 / 128 + ((wave & 3u) >> 1)) * (stride / 4)) * 512 + (wave & 1u) * 256
                     : (tile_row0 + (wave & 3u) * 64) * stride;
        // compact scales of a K-major member, [k block][row] (see the dense branch below)
        if (wave < 4 ? PKC : QKC) t.ss = tile_row0 + (wave & 3u) * stride;
      }
      return t;
    }
#ifdef TE_GEMM_STAMPS
    if (a.dbg > 0 && a.dbg < 4 && valid) {   // rotate the 4 groups of 8 workgroups of an XCD round
      const unsigned loc = tile / 8;
      tile = (tile % 8) + 8 * ((loc & ~31u) | ((loc + 8u * a.dbg) & 31u));
    }
#endif
#ifdef TE_GEMM_STAMPS
    tile_of(valid ? tile : 0u, a.ntiles, tiles_i, a.tiles_j, ti, tj, (a.dbg >= 4 && a.dbg < 100) ? static_cast<unsigned>(a.dbg) : 8u);
    if (a.dbg == 104) { ti = blockIdx.x % 8; tj = (blockIdx.x / 8) % 4; }   // every round re-reads the same 12 panels
#else
    tile_of(valid ? tile : 0u, a.ntiles, tiles_i, a.tiles_j, ti, tj);
#endif
    t.i0 = ti * BM;
    t.j0 = tj * BN;
    const unsigned rows_p = valid ? min(a.n - t.i0, static_cast<unsigned>(BM)) : 0u;
    const unsigned rows_q = valid ? min(a.m - t.j0, static_cast<unsigned>(BN)) : 0u;
    t.pbase = const_cast<uint8_t *>(a.P) + static_cast<size_t>(t.i0) * a.k;
    t.qbase = const_cast<uint8_t *>(a.Q) + static_cast<size_t>(t.j0) * a.k;
    t.pbytes = rows_p * a.k;
    t.qbytes = rows_q * a.k;
    // K-major operand: the descriptor starts at the tile's first row inside k-row 0 and ends with the matrix, so
    // k-rows past K read as zero; tile rows past the matrix end run into the next k-row (finite bytes that only
    // feed output rows / columns which are never stored)
    if constexpr (PKM) {
      t.pbase = const_cast<uint8_t *>(a.P) + t.i0;
      t.pbytes = valid ? a.k * a.ldP - t.i0 : 0u;
    }
    if constexpr (QKM) {
      t.qbase = const_cast<uint8_t *>(a.Q) + t.j0;
      t.qbytes = valid ? a.k * a.ldQ - t.j0 : 0u;
    }
    if constexpr (MX) {
      // this wave stages the scales of tile rows (wave & 3) * 64 .. + 63 of P (waves 0-3) or Q (4-7)
      const unsigned tile_row0 = wave < 4 ? t.i0 : t.j0;
      t.ss = e_swz ? ((tile_row0 / 128 + ((wave & 3u) >> 1)) * (e_stride / 4)) * 512 + (wave & 1u) * 256
                   : (tile_row0 + (wave & 3u) * 64) * e_stride;
      // compact scales of a K-major operand are stored [k block][row] (the reference's columnwise scale
      // matrix): one 16-byte LDS-DMA instruction of wave 0 (P) / wave 4 (Q) brings the 4 x 256 bytes of a K tile
      // compact scales of a K-major operand, [k block][row]: wave w of the group stages scale ROW w of the K tile
      // (256 bytes = one 4-byte-per-lane instruction, the shape of the other layouts' staging)
      if (wave < 4 ? PKC : QKC) t.ss = tile_row0 + (wave & 3u) * e_stride;
    }
    return t;
  };

  // Staging geometry: wave-instruction q of this wave fills tile rows (wave*4+q)*8 .. +7, lane ->
  // (row = lane >> 3, LDS chunk slot = lane & 7).  The swizzle of a row only depends on
  // (q & 1, lane), so two lane offsets serve the four instructions; the row block and the K
  // position travel in the scalar offset.
  unsigned st_voff[2], st_chunk_k[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const unsigned r = static_cast<unsigned>(q) * 8 + (lane >> 3);
    const unsigned c = (lane & 7u) ^ ((r >> 1) & 7u);
    st_voff[q] = (lane >> 3) * a.k + c * 16;
    st_chunk_k[q] = c * 16;
  }
  // K-MAJOR operand: wave-instruction q fills k-rows (wave*4+q)*4 .. +3 of the [k][256 rows] image, lane ->
  // (k row = lane >> 4, 16-byte unit = lane & 15).  The transposing fragment reads (below) touch, per LDS cycle, 8
  // consecutive k-rows x 2 units x 2 halves: unit index XOR 2 * (k & 7) spreads them over all 64 banks; k & 7 only
  // depends on (q & 1, lane), so two lane offsets serve the four instructions.
  // (instruction parity q & 1 flips bit 3 of the unit index: +-128 bytes, one lane constant for both operands)
  unsigned km_voffP = 0, km_voffQ = 0, km_flip = 0;
  if constexpr (KM != 0) {
    const unsigned kr = lane >> 4;
    const unsigned u = (lane & 15u) ^ (2u * kr);
    km_voffP = kr * a.ldP + u * 16;
    km_voffQ = kr * a.ldQ + u * 16;
    km_flip = (u & 8u) ? static_cast<unsigned>(-128) : 128u;
  }
  // two glds instructions (rows q*8.. of this wave's 32-row share of P and of Q; K-major: k-rows q*4.. of its 16)
  auto stage_P = [&](const TileCtx &t, unsigned slot, unsigned k0, int q, const unsigned(&voff)[2]) {
    uint8_t *base = lds + slot * kStageBytes;
    const unsigned soff = (wave * 4 + q) * 8 * (GROUPED ? t.k : a.k) + k0;
    lds_u8 *dst_p = (lds_u8 *)(base + (wave * 4 + q) * 1024);
    if constexpr (!kAsmDma) {
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(t.pbase, 0, t.pbytes, 0x00020000);
      if constexpr (PKM)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (__attribute__((address_space(3))) void *)dst_p, 16,
                                                 km_voffP + (q & 1) * km_flip, (k0 + (wave * 4 + q) * 4) * a.ldP, 0, kStageAux);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (__attribute__((address_space(3))) void *)dst_p, 16, voff[q & 1], soff, 0, kStageAux);
    } else {
      static_assert(!kAsmDma || kStageAux == 0, "cache-policy experiments need the builtin form (TE_GEMM_BUILTIN_DMA)");
      const v4i_t rp = make_rsrc(t.pbase, t.pbytes);
      const unsigned la = static_cast<unsigned>(reinterpret_cast<uintptr_t>(dst_p));
      if constexpr (PKM)
        glds_b128(rp, la, km_voffP + (q & 1) * km_flip, (k0 + (wave * 4 + q) * 4) * a.ldP);
      else
        glds_b128(rp, la, voff[q & 1], soff);
    }
  };
  auto stage_Q = [&](const TileCtx &t, unsigned slot, unsigned k0, int q, const unsigned(&voff)[2]) {
    uint8_t *base = lds + slot * kStageBytes;
    const unsigned soff = (wave * 4 + q) * 8 * (GROUPED ? t.k : a.k) + k0;
    lds_u8 *dst_q = (lds_u8 *)(base + BM * BK + (wave * 4 + q) * 1024);
    if constexpr (!kAsmDma) {
      const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(t.qbase, 0, t.qbytes, 0x00020000);
      if constexpr (QKM)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void *)dst_q, 16,
                                                 km_voffQ + (q & 1) * km_flip, (k0 + (wave * 4 + q) * 4) * a.ldQ, 0, kStageAux);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void *)dst_q, 16, voff[q & 1], soff, 0, kStageAux);
    } else {
      const v4i_t rq = make_rsrc(t.qbase, t.qbytes);
      const unsigned la = static_cast<unsigned>(reinterpret_cast<uintptr_t>(dst_q));
      if constexpr (QKM)
        glds_b128(rq, la, km_voffQ + (q & 1) * km_flip, (k0 + (wave * 4 + q) * 4) * a.ldQ);
      else
        glds_b128(rq, la, voff[q & 1], soff);
    }
  };
  auto stage_rows = [&](const TileCtx &t, unsigned slot, unsigned k0, int q, const unsigned(&voff)[2], auto hot_tag) {
    // hot = called from the steady-state K loop (not the prologue, not the two K tiles peeled off at a tile boundary)
    constexpr bool BARE = decltype(hot_tag)::value && !MX && !GROUPED;
    if constexpr (kAsmDma) {
      uint8_t *base = lds + slot * kStageBytes;
      const unsigned soff = (wave * 4 + q) * 8 * (GROUPED ? t.k : a.k) + k0;
      const unsigned la_p = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_u8 *)(base + (wave * 4 + q) * 1024)));
      const unsigned la_q = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_u8 *)(base + BM * BK + (wave * 4 + q) * 1024)));
      const unsigned kmo = (k0 + (wave * 4 + q) * 4);
      glds_b128_pair<BARE>(make_rsrc(t.pbase, t.pbytes), la_p, PKM ? km_voffP + (q & 1) * km_flip : voff[q & 1], PKM ? kmo * a.ldP : soff,
                     make_rsrc(t.qbase, t.qbytes), la_q, QKM ? km_voffQ + (q & 1) * km_flip : voff[q & 1], QKM ? kmo * a.ldQ : soff);
    } else {
      stage_P(t, slot, k0, q, voff);
      stage_Q(t, slot, k0, q, voff);
    }
  };
  auto stage_voff = [&](const TileCtx &t, unsigned k0, unsigned(&voff)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if constexpr (GROUPED)
        voff[q] = (k0 + st_chunk_k[q] < t.k) ? (lane >> 3) * t.k + st_chunk_k[q] : kOOB;
      else
        voff[q] = (k0 + st_chunk_k[q] < a.k) ? st_voff[q] : kOOB;
    }
  };
  auto stage = [&](const TileCtx &t, unsigned slot, unsigned k0) {
    unsigned voff[2];
    stage_voff(t, k0, voff);
#pragma unroll
    for (int q = 0; q < 4; ++q) stage_rows(t, slot, k0, q, voff, std::false_type{});
  };

  // Fragment read geometry.  A lane needs chunks c = half + 2q (q = 0..3) of its row; with the
  // swizzle s = (row >> 1) & 7 the chunk sits at c ^ s = (half ^ (s & 1)) + 2 (q ^ (s >> 1)), and
  // both s & 1 and s >> 1 only depend on lane & 31 (rows of different MFMA tiles differ by
  // multiples of 32).  So 4 lane-constant offsets serve every tile; the tile's row block is an
  // immediate.
  const unsigned fr_row = lane & 31u;
  const unsigned fr_half = lane >> 5;
  const unsigned sw = (fr_row >> 1) & 7u;
  unsigned offP[4], offQ[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned chunk = (fr_half ^ (sw & 1u)) + 2u * (static_cast<unsigned>(q) ^ (sw >> 1));
    offP[q] = (wi * 128 + fr_row) * BK + chunk * 16;
    offQ[q] = BM * BK + (wj * 64 + fr_row) * BK + chunk * 16;
  }
  // one MFMA operand fragment: P row block mi / Q column block nj, K step ks of the tile in `slot`
  // (operand register map of v_mfma_*_32x32x64_f8f6f4, measured on gfx950: a lane's first 16
  // bytes are K = 16*half + [0,16), its last 16 bytes K = 32 + 16*half + [0,16); the 32-element
  // block b takes its e8m0 scale from lane half b)
  // K-major image: one ds_read_b64_tr_b8 hands lane l (16-lane group g = l >> 4, i = l & 15) the 8 bytes
  // M[2b + (i >> 3)][i & 7], b = 0..7, of the 16 x 8 byte matrix formed by the 8-byte chunks that the 16 lanes of
  // its group ADDRESS (measured on gfx950, scratch/tr8_probe.hip).  With supplier 2b + p pointed at k-row kbase + b,
  // rows r0 + 8p .. + 7, lane i receives 8 consecutive k of row r0 + i: a quarter of an MFMA operand fragment.
  // Group g covers rows 16 (g & 1) .. + 15 of the 32-row block with the K half h = g >> 1, as the MFMA wants
  // them; four reads (k offsets 0, 8, 32, 40 from 64 ks + 16 h) build the 32-byte fragment.
  // The unit index of row block mi is wi * 8 + mi * 2 + (g & 1) (Q: wj * 4 + nj * 2 + (g & 1)): mi / nj sit in bits
  // of their own, so the block's address is the lane's base XOR (mi << 5) - one lane register per operand.
  unsigned kmP = 0, kmQ = 0, kmS = 0;
  if constexpr (KM != 0) {
    const unsigned g = lane >> 4, i = lane & 15u, b = i >> 1, pp = i & 1u, h = g >> 1;
    const unsigned krow = 16u * h + b;                       // + 64 ks + {0, 8, 32, 40}
    kmP = krow * 256u + (((wi * 8u + (g & 1u)) ^ (2u * b)) * 16u) + 8u * pp;
    kmQ = BM * BK + krow * 256u + (((wj * 4u + (g & 1u)) ^ (2u * b)) * 16u) + 8u * pp;
    // scale image [k block][256 rows] of a K tile: supplier 2b + p -> scale row b & 3, rows 16 (g & 1) + 8p .. + 7;
    // the low dword of the result is then the lane's 4 block scales (rows: + 32 mi, operand: + 1024 for Q)
    kmS = (b & 3u) * 256u + 16u * (g & 1u) + 8u * pp;
  }
  typedef int v2i_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) v2i_t lds_v2i;
  // the four transposing reads of one fragment: k offsets 0, 8, 32, 40 (x 256 bytes per k-row) from `p0`
  auto rd_tr8 = [&](const uint8_t *p0, v8i &f) {
    const v2i_t r0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(p0));
    const v2i_t r1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(p0 + 8 * 256));
    const v2i_t r2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(p0 + 32 * 256));
    const v2i_t r3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(p0 + 40 * 256));
    f = v8i{r0[0], r0[1], r1[0], r1[1], r2[0], r2[1], r3[0], r3[1]};
  };
  auto rdP = [&](unsigned slot, int ks, int mi, v8i &f) {
    if constexpr ((ABL & 2) != 0) { asm volatile("" : "+v"(f)); return; }
    const uint8_t *img = lds + slot * kStageBytes;
    if constexpr (PKM) {
      rd_tr8(img + (kmP ^ (static_cast<unsigned>(mi) << 5)) + ks * 64 * 256, f);
      return;
    }
    const u32x4 lo = *reinterpret_cast<const u32x4 *>(img + offP[2 * ks] + mi * 32 * BK);
    const u32x4 hi = *reinterpret_cast<const u32x4 *>(img + offP[2 * ks + 1] + mi * 32 * BK);
    f = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  auto rdQ = [&](unsigned slot, int ks, int nj, v8i &f) {
    if constexpr ((ABL & 2) != 0) { asm volatile("" : "+v"(f)); return; }
    const uint8_t *img = lds + slot * kStageBytes;
    if constexpr (QKM) {
      rd_tr8(img + (kmQ ^ (static_cast<unsigned>(nj) << 5)) + ks * 64 * 256, f);
      return;
    }
    const u32x4 lo = *reinterpret_cast<const u32x4 *>(img + offQ[2 * ks] + nj * 32 * BK);
    const u32x4 hi = *reinterpret_cast<const u32x4 *>(img + offQ[2 * ks + 1] + nj * 32 * BK);
    f = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };

  v16f acc[4][2];
  auto zero_acc = [&] {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;
  };
  zero_acc();

  // MX scales: one dword = the 4 K blocks (e8m0 bytes) of a K tile for one row.  They travel
  // like the payload: wave w copies the dwords of 64 tile rows global -> LDS (one 4-byte LDS-DMA
  // instruction per K tile) into the scale slot of the K tile, and after the barrier every wave
  // picks the 6 dwords of its own MFMA rows out of LDS.  The descriptors cover exactly the scale
  // buffers: rows of a partial tile past the (padded) buffer read 0, rows past n / m inside the
  // padding only feed outputs that are never stored.
  //   compact layout  : source row*stride + 4 kt,              LDS image [row]            (dwords)
  //   swizzled layout : source 512-byte blocks of 128 rows,     LDS image [row/128][row%32][row%128/32]
  uint32_t scP[4] = {0, 0, 0, 0}, scQ[2] = {0, 0};
  unsigned sv = 0, rsP0 = 0, rsQ0 = 0, rsPstep = 0, rsQstep = 0;
  if constexpr (MX) {
    sv = (e_swz || (wave < 4 ? PKC : QKC)) ? lane * 4 : lane * e_stride;
    if (a.eP_swz) {
      rsP0 = wi * 512 + fr_row * 16;
      rsPstep = 4;
    } else {
      rsP0 = (wi * 128 + fr_row) * 4;
      rsPstep = 128;
    }
    if (a.eQ_swz) {
      rsQ0 = 1024 + (wj >> 1) * 512 + fr_row * 16 + (wj & 1u) * 8;
      rsQstep = 4;
    } else {
      rsQ0 = 1024 + (wj * 64 + fr_row) * 4;
      rsQstep = 128;
    }
  }
  // One 4-byte-per-lane LDS-DMA instruction per wave and K tile, whatever the layout (no branch in the main loop):
  //   compact rowwise  : lane = tile row, source row * stride + 4 kt            -> LDS image [row] (dwords)
  //   GEMM-swizzled    : 512-byte blocks of 128 rows, source + 512 kt
  //   compact K-major  : wave w of the group = scale row w of the K tile, lane = 4 tile rows, source
  //                      (4 kt + w) * stride + tile row 0 + 4 lane             -> LDS image [k block][256 rows]
  const bool e_kc = wave < 4 ? PKC : QKC;
  auto stage_scales = [&](const TileCtx &t, unsigned slot, unsigned kt) {
    if constexpr (MX) {
      lds_u8 *dst = (lds_u8 *)(lds + kScaleBase + slot * kScaleSlotBytes + wave * 256);
      const unsigned stride = GROUPED ? t.estride : e_stride;       // (the member's own in a grouped launch)
      const unsigned e_step = e_swz ? 512u : (e_kc ? 4u * stride : 4u);
      const unsigned lane_off = GROUPED ? ((e_swz || e_kc) ? lane * 4 : lane * stride) : sv;
      if constexpr (!kAsmDma) {
        const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t *>(GROUPED ? t.eb : e_base), 0, GROUPED ? t.ebytes : e_bytes, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(re, (__attribute__((address_space(3))) void *)dst, 4, lane_off,
                                                 t.ss + kt * e_step, 0, 0);
      } else {
        glds_b32(make_rsrc(GROUPED ? t.eb : e_base, GROUPED ? t.ebytes : e_bytes),
                 static_cast<unsigned>(reinterpret_cast<uintptr_t>(dst)), lane_off, t.ss + kt * e_step);
      }
    }
  };
  // Pick this wave's scale dwords of K tile `kt` (already in scale slot `slot`) and pre-shift so
  // that opsel 0 / 2 select this lane's K block.  K blocks past the end of K carry whatever the
  // padded scale buffer holds: force 2^0 there (their payload bytes are zero-filled).
  // Branch-free on purpose: a branch in the main loop lets the optimiser sink the MFMAs out of
  // their place in the pipeline.
  auto commit_scales = [&](unsigned slot, unsigned kt, unsigned kdim) {
    if constexpr (MX) {
      const uint8_t *img = lds + kScaleBase + slot * kScaleSlotBytes;
      const unsigned kbase = min(kt * BK, kdim);
      const unsigned valid = min(4u, (kdim - kbase + 31) / 32);
      const uint32_t keep = valid >= 4 ? 0xFFFFFFFFu : ((1u << (8 * valid)) - 1u);
      const uint32_t fill = 0x7F7F7F7Fu & ~keep;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        uint32_t raw;
        if constexpr (PKC)      // [k block][row] image: one transposing read = this lane's 4 block scales
          raw = static_cast<uint32_t>(__builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(img + kmS + wi * 128 + mi * 32))[0]);
        else
          raw = *reinterpret_cast<const uint32_t *>(img + rsP0 + mi * rsPstep);
        scP[mi] = ((raw & keep) | fill) >> (8 * fr_half);
      }
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) {
        uint32_t raw;
        if constexpr (QKC)
          raw = static_cast<uint32_t>(__builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(img + 1024 + kmS + wj * 64 + nj * 32))[0]);
        else
          raw = *reinterpret_cast<const uint32_t *>(img + rsQ0 + nj * rsQstep);
        scQ[nj] = ((raw & keep) | fill) >> (8 * fr_half);
      }
    }
  };
  // one MFMA: acc[mi][nj] += P fragment x Q fragment (K step KS selects the scale bytes)
  auto mm = [&](auto ks_tag, auto mi_tag, auto nj_tag, const v8i &fp, const v8i &fq) {
    constexpr int KS = decltype(ks_tag)::value;
    constexpr int MI = decltype(mi_tag)::value;
    constexpr int NJ = decltype(nj_tag)::value;
    if constexpr (MX) {
      acc[MI][NJ] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
          fp, fq, acc[MI][NJ], FA, FB, 2 * KS, (int)scP[MI], 2 * KS, (int)scQ[NJ]);
    } else {
      acc[MI][NJ] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp, fq, acc[MI][NJ], FA, FB, 0, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // pin the hand-made interleave of MFMA / LDS / global->LDS
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // Fragment registers: p[mi] holds row block mi, q[ks][nj] column block nj of K step ks.  Each
  // register is refilled from LDS right after the last MFMA that reads it, 5..7 MFMAs before its
  // next use.  MFMA order inside a K tile: K step 0 over the 4 x 2 blocks, then K step 1.
  v8i p[4], q[2][2];
#ifdef TE_GEMM_STAMPS
  unsigned long long w_vm = 0, w_bar = 0, w_tiles = 0, w_first = 0, w_last = 0, w_epi = 0, w_y = 0;
#endif

  // One K tile: 16 MFMAs + the reads of the next K tile's first fragments + the staging of the
  // K tile two ahead (`st` / `st_k0`: which output tile and which K offset that is) + the MX
  // scale hand-over.  `g` counts K tiles across output tiles and picks the LDS slot.
  auto k_tile = [&](unsigned g, const TileCtx &st, unsigned st_k0, unsigned st_kt, unsigned commit_kt,
                    unsigned commit_k, auto prefetch_tag, auto hot_tag) {
    constexpr bool PREFETCH = decltype(prefetch_tag)::value;  // fetch the next K tile's first fragments
    const unsigned cur = g & 1u, nxt = cur ^ 1u;
    mm(I0{}, I0{}, I0{}, p[0], q[0][0]);
    rdQ(cur, 1, 0, q[1][0]);
    mm(I0{}, I0{}, I1{}, p[0], q[0][1]);
    rdP(cur, 1, 0, p[0]);
    mm(I0{}, I1{}, I0{}, p[1], q[0][0]);
    rdQ(cur, 1, 1, q[1][1]);
    mm(I0{}, I1{}, I1{}, p[1], q[0][1]);
    rdP(cur, 1, 1, p[1]);
    mm(I0{}, I2{}, I0{}, p[2], q[0][0]);
    mm(I0{}, I2{}, I1{}, p[2], q[0][1]);
    rdP(cur, 1, 2, p[2]);
    mm(I0{}, I3{}, I0{}, p[3], q[0][0]);
    mm(I0{}, I3{}, I1{}, p[3], q[0][1]);
    rdP(cur, 1, 3, p[3]);
    mm(I1{}, I0{}, I0{}, p[0], q[1][0]);
    unsigned voff[2];
    stage_voff(st, st_k0, voff);
    mm(I1{}, I0{}, I1{}, p[0], q[1][1]);
    // Every wave has now pulled all of this K tile out of slot `cur` (last read: two MFMAs ago);
    // the next K tile (issued most of a tile ago) has landed.  After the barrier the slot is
    // refilled with the K tile two ahead and the first fragments of the next one are fetched.
#ifdef TE_GEMM_STAMPS
    if (a.wstamps != nullptr) {
      // where the wave is parked at the K-tile hand-over: waiting for ITS staged loads (t1 - t0) or for the
      // other waves at the barrier (t2 - t1); s_memtime = shader cycles
      unsigned long long t0, t1, t2;
      asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
      asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
      w_vm += t1 - t0;
      w_bar += t2 - t1;
      w_tiles += 1;
      if (w_first == 0) w_first = t0;
      w_last = t2;
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
    TE_STAMP();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PREFETCH) rdP(nxt, 0, 0, p[0]);
    if constexpr (PREFETCH) rdQ(nxt, 0, 0, q[0][0]);
    mm(I1{}, I1{}, I0{}, p[1], q[1][0]);
    if constexpr (PREFETCH) rdQ(nxt, 0, 1, q[0][1]);
    if constexpr (!(ABL & 1)) stage_rows(st, cur, st_k0, 0, voff, hot_tag);
    mm(I1{}, I1{}, I1{}, p[1], q[1][1]);
    if constexpr (PREFETCH) rdP(nxt, 0, 1, p[1]);
    if constexpr (!(ABL & 1)) stage_rows(st, cur, st_k0, 1, voff, hot_tag);
    mm(I1{}, I2{}, I0{}, p[2], q[1][0]);
    if constexpr (!(ABL & 1)) stage_rows(st, cur, st_k0, 2, voff, hot_tag);
    mm(I1{}, I2{}, I1{}, p[2], q[1][1]);
    if constexpr (PREFETCH) rdP(nxt, 0, 2, p[2]);
    if constexpr (!(ABL & 1)) stage_rows(st, cur, st_k0, 3, voff, hot_tag);
    stage_scales(st, cur, st_kt);
    mm(I1{}, I3{}, I0{}, p[3], q[1][0]);
    mm(I1{}, I3{}, I1{}, p[3], q[1][1]);
    if constexpr (PREFETCH) rdP(nxt, 0, 3, p[3]);
    commit_scales(nxt, commit_kt, commit_k);
    __builtin_amdgcn_sched_barrier(0);
#ifdef TE_GEMM_STAMPS
    if (a.wstamps != nullptr) {   // end of the post-barrier segment (6 MFMAs + staging + next fragments issued)
      unsigned long long t3;
      asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t3) :: "memory");
      w_y += t3 - w_last;
    }
#endif
  };

  // ------------------------------------------------------------ epilogue
  // Wave-private LDS transpose of kEpiRows x 64 fp32 at a time (LDS of its own).
  // MFMA C layout: lane -> column (lane & 31), register r -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
  // Column index XOR 32 on rows whose bit 2 is set: the two lane halves (rows +4) land 32 banks
  // apart, so the b32 writes are conflict-free; each b128 read covers 4 rows x 256 contiguous bytes.
  float *stg = reinterpret_cast<float *>(lds + kEpiBase) + wave * (kEpiRows * 64);
  const unsigned e_row = lane >> 4;          // 0..3
  const unsigned e_col = (lane & 15u) * 4u;  // 0..60
  float mult0 = a.alpha;
  if constexpr (!MX && !GROUPED) {
    if (a.sP) mult0 *= *a.sP;
    if (a.sQ) mult0 *= *a.sQ;
  }
  // FULL = the tile lies inside D and rows are vector-aligned: no per-element guards (and no
  // divergent control flow, which would make the compiler drain every store before the next
  // bias add).  Otherwise the guarded element-wise path.
  // NT: streaming stores for an output too large to stay cached until someone reads it (measured on
  // configs[1]: the following GEMMs run 1-2 % faster when the 67-235 MB outputs do not displace their panels)
  auto epilogue_as = [&](const TileCtx &t, auto *dptr, auto with_c, auto full_tag, auto nt_tag) {
    using OT = std::remove_pointer_t<decltype(dptr)>;
    constexpr bool WITH_C = decltype(with_c)::value;
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr bool NT = decltype(nt_tag)::value;
    using VecT = std::conditional_t<sizeof(OT) == 4, u32x4, u32x2>;
    const unsigned jbase = t.j0 + wj * 64 + e_col;
    // the problem this tile belongs to: the launch's (folded to kernel arguments) or its member's
    const unsigned e_n = GROUPED ? t.n : a.n, e_m = GROUPED ? t.m : a.m, e_ldd = GROUPED ? t.m : a.ldd;
    const float mult = GROUPED ? t.mult : mult0, e_beta = GROUPED ? t.beta : a.beta;
    // where the accumulation input sits relative to the output (0 unless a grouped C != D)
    const ptrdiff_t c_delta = GROUPED ? (t.C - t.D) : 0;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (!GROUPED && a.bias) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (FULL || jbase + x < e_m) bias[x] = load_any(a.bias, jbase + x, a.bias_dtype);
    }
    // first (unconditional) use of the bias registers: the wait for their loads happens here, once
    asm volatile("" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]));
    // wave-uniform 64-bit base of this wave's 128 x 64 block + a 32-bit lane offset
    char *wave_base = reinterpret_cast<char *>(dptr + static_cast<size_t>(t.i0 + wi * 128) * e_ldd + t.j0 + wj * 64);
    const unsigned lane_off = (e_row * e_ldd + e_col) * static_cast<unsigned>(sizeof(OT));
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int part = 0; part < 32 / kEpiRows; ++part) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
          for (int rr = 0; rr < kEpiRows / 2; ++rr) {
            const unsigned l = (rr & 3) + 8 * (rr >> 2) + 4 * fr_half;
            stg[l * 64 + ((nj * 32 + fr_row) ^ (32u * fr_half))] = acc[mi][nj][(kEpiRows / 2) * part + rr];
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < kEpiRows / 4; ++ps) {
          const unsigned l = ps * 4 + e_row;
          const f32x4 v4 = *reinterpret_cast<const f32x4 *>(stg + l * 64 + (e_col ^ (32u * (ps & 1))));
          const unsigned li = mi * 32 + part * kEpiRows + l;  // row inside this wave's 128
          OT *dst = reinterpret_cast<OT *>(wave_base + static_cast<size_t>(mi * 32 + part * kEpiRows + ps * 4) * e_ldd * sizeof(OT) + lane_off);
          const OT *csrc = reinterpret_cast<const OT *>(reinterpret_cast<const char *>(dst) + c_delta);
          float v[4] = {v4[0] * mult, v4[1] * mult, v4[2] * mult, v4[3] * mult};
          if constexpr (FULL) {
            if constexpr (WITH_C) {
              OT old[4];
              *reinterpret_cast<VecT *>(old) = *reinterpret_cast<const VecT *>(csrc);
#pragma unroll
              for (int x = 0; x < 4; ++x) v[x] += e_beta * widen(old[x]);
            }
            OT o[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) narrow(v[x] + bias[x], o[x]);
            if constexpr (NT)
              __builtin_nontemporal_store(*reinterpret_cast<const VecT *>(o), reinterpret_cast<VecT *>(dst));
            else
              *reinterpret_cast<VecT *>(dst) = *reinterpret_cast<const VecT *>(o);
          } else {
            if (t.i0 + wi * 128 + li >= e_n) continue;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              if (jbase + x >= e_m) continue;
              float y = v[x];
              if constexpr (WITH_C) y += e_beta * widen(csrc[x]);
              narrow(y + bias[x], dst[x]);
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  };
  auto epilogue_typed = [&](const TileCtx &t, auto *dptr) {
    using OT = std::remove_pointer_t<decltype(dptr)>;
    const unsigned e_n = GROUPED ? t.n : a.n, e_m = GROUPED ? t.m : a.m, e_ldd = GROUPED ? t.m : a.ldd;
    bool full = t.i0 + BM <= e_n && t.j0 + BN <= e_m && e_ldd % 4 == 0 &&
                reinterpret_cast<uintptr_t>(dptr) % (4 * sizeof(OT)) == 0;
    if constexpr (GROUPED) full = full && reinterpret_cast<uintptr_t>(t.C) % (4 * sizeof(OT)) == 0;
    const bool with_c = (GROUPED ? t.beta : a.beta) != 0.f;
    using No = std::false_type;
    using Yes = std::true_type;
    if (full) {
      if (with_c) epilogue_as(t, dptr, Yes{}, Yes{}, No{});
      else if (a.nt_store) epilogue_as(t, dptr, No{}, Yes{}, Yes{});
      else epilogue_as(t, dptr, No{}, Yes{}, No{});
    } else {
      with_c ? epilogue_as(t, dptr, Yes{}, No{}, No{}) : epilogue_as(t, dptr, No{}, No{}, No{});
    }
  };
  // Extended epilogue: same LDS transpose, then one guarded element at a time:
  //   y = alpha * acc (+ beta * C) (+ bias);  GELU_AUX: aux = y, y = gelu(y);
  //   DGELU: y *= dgelu(aux), column sums for the bias gradient;  FP8 D: amax |y|, y * scale.
  auto epilogue_ext = [&](const TileCtx &t) {
    const unsigned jbase = t.j0 + wj * 64 + e_col;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (jbase + x < a.m) bias[x] = load_any(a.bias, jbase + x, a.bias_dtype);
    }
    const bool fp8_out = a.d_dtype == kNVTEFloat8E4M3 || a.d_dtype == kNVTEFloat8E5M2;
    const float d_scale = (fp8_out && a.d_scale) ? *a.d_scale : 1.f;
    float tmax = 0.f;
    float colsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int part = 0; part < 32 / kEpiRows; ++part) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
          for (int rr = 0; rr < kEpiRows / 2; ++rr) {
            const unsigned l = (rr & 3) + 8 * (rr >> 2) + 4 * fr_half;
            // dynamic mi: copy through a switch-free select (mi loop is not unrolled)
            float v = 0.f;
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2)
              if (m2 == mi) v = acc[m2][nj][(kEpiRows / 2) * part + rr];
            stg[l * 64 + ((nj * 32 + fr_row) ^ (32u * fr_half))] = v;
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < kEpiRows / 4; ++ps) {
          const unsigned l = ps * 4 + e_row;
          const f32x4 v4 = *reinterpret_cast<const f32x4 *>(stg + l * 64 + (e_col ^ (32u * (ps & 1))));
          const unsigned gi = t.i0 + wi * 128 + mi * 32 + part * kEpiRows + l;
          if (gi >= a.n) continue;
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const unsigned gj = jbase + x;
            if (gj >= a.m) continue;
            const size_t idx = static_cast<size_t>(gi) * a.ldd + gj;
            float y = v4[x] * mult0;
            if (a.beta != 0.f) y += a.beta * load_any(a.D, idx, a.d_dtype);
            y += bias[x];
            if (a.epi == kEpiGeluAux) {
              store_any(a.aux, idx, a.aux_dtype, y);
              y = gelu_tanh(y);
            } else if (a.epi == kEpiDgelu) {
              y *= dgelu_tanh(load_any(a.aux, idx, a.aux_dtype));
              colsum[x] += y;
            }
            if (fp8_out) {
              tmax = fmaxf(tmax, fabsf(y));
              y *= d_scale;
            }
            store_any(a.D, idx, a.d_dtype, y);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    if (a.dbias_acc) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        // the 4 lanes that share these columns (lane >> 4 = 0..3) add up first
        float c = colsum[x];
        c += __shfl_xor(c, 16);
        c += __shfl_xor(c, 32);
        if (e_row == 0 && jbase + x < a.m) atomicAdd(a.dbias_acc + jbase + x, c);
      }
    }
    if (fp8_out && a.d_amax) {
      tmax = wave_max(tmax);
      if (lane == 0) atomic_max_nonneg(a.d_amax, tmax);
    }
  };
  auto epilogue = [&](const TileCtx &t) {
    if constexpr (EPI) {
      epilogue_ext(t);
      return;
    }
    void *dbase = a.D;
    if constexpr (GROUPED) dbase = t.D;
    switch (a.d_dtype) {
      case kNVTEFloat32: epilogue_typed(t, static_cast<float *>(dbase)); break;
      case kNVTEBFloat16: epilogue_typed(t, static_cast<bf16_t *>(dbase)); break;
      default: epilogue_typed(t, static_cast<f16_t *>(dbase)); break;
    }
  };

  // ------------------------------------------------------------ the walk over output tiles
  unsigned tile = blockIdx.x;
  TileCtx cur_t = make_tile(tile, true);

  // prologue: K tiles 0 and 1 of the first output tile in flight (K tile 1 is an all-zero fill
  // when nk == 1; the host then launches one workgroup per output tile)
  stage(cur_t, 0, 0);
  stage_scales(cur_t, 0, 0);
  stage(cur_t, 1, BK);
  stage_scales(cur_t, 1, 1);
  if constexpr (MX)
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  commit_scales(0, 0, GROUPED ? cur_t.k : a.k);
  auto first_fragments = [&](unsigned slot) {
    rdP(slot, 0, 0, p[0]);
    rdQ(slot, 0, 0, q[0][0]);
    rdQ(slot, 0, 1, q[0][1]);
    rdP(slot, 0, 1, p[1]);
    rdP(slot, 0, 2, p[2]);
    rdP(slot, 0, 3, p[3]);
  };
  first_fragments(0);
  __builtin_amdgcn_sched_barrier(0);

#ifdef TE_GEMM_STAMPS
  // experiment (diagnostic build only): static priority for the second-dispatched half of the workgroup
  if (a.dbg == 201 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  if (a.dbg == 202 && wave < 4) __builtin_amdgcn_s_setprio(1);
#endif
  unsigned g = 0;
  while (true) {
    const unsigned next = tile + gridDim.x;
    bool has_next = next < total_tiles;
    if constexpr (GROUPED) {
      unsigned pos;
      has_next = run_of(next, pos);
    }
    const TileCtx next_t = make_tile(next, has_next);
    // K tiles of THIS output tile: the launch's, or the member's (never fewer than two)
    const unsigned nk_t = GROUPED ? max((cur_t.k + BK - 1) / BK, 2u) : nk;
    const unsigned k_cur = GROUPED ? cur_t.k : a.k, k_next = GROUPED ? next_t.k : a.k;
    for (unsigned kt = 0; kt + 2 < nk_t; ++kt, ++g)
      k_tile(g, cur_t, (kt + 2) * BK, kt + 2, kt + 1, k_cur, std::true_type{}, std::true_type{});
    if (nk_t >= 2) {
      k_tile(g, next_t, 0, 0, nk_t - 1, k_cur, std::true_type{}, std::false_type{});
      ++g;
    }
    // the last K tile leaves the fragment registers dead, so the epilogue has room to work in
    k_tile(g, next_t, BK, 1, 0, k_next, std::false_type{}, std::false_type{});
    ++g;
    TE_STAMP();
#ifdef TE_GEMM_STAMPS
    const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
    epilogue(cur_t);
#ifdef TE_GEMM_STAMPS
    w_epi += __builtin_amdgcn_s_memtime() - te0;
#endif
    TE_STAMP();
    if (!has_next) break;
    first_fragments(g & 1u);
    zero_acc();
    tile = next;
    cur_t = next_t;
  }
  // the trailing zero-fill stages must be out of LDS before the workgroup ends
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef TE_GEMM_STAMPS
  if (a.wstamps != nullptr && lane == 0) {
    unsigned long long *o = a.wstamps + (static_cast<size_t>(blockIdx.x) * 8 + wave) * 8;
    o[0] = w_vm; o[1] = w_bar; o[2] = w_last - w_first; o[3] = w_tiles; o[4] = w_epi; o[5] = w_y;
  }
  if (a.stamps != nullptr && tid == 0) {
    a.stamps[blockIdx.x * kStampSlots + kStampSlots - 2] = __builtin_amdgcn_s_memtime();
    a.stamps[blockIdx.x * kStampSlots + kStampSlots - 4] = sidx;
  }
#endif
}

// Every (MX, EPI, GROUPED, KM) combination the host code launches: the instantiation list of a format pair.
#ifdef TE_GEMM_DEV_ONE   // codegen experiments: one kernel, seconds to compile (hipcc -S --cuda-device-only -DTE_GEMM_PART=0)
#define TE_GEMM_APPLY(X, ...) X(__VA_ARGS__)
#define TE_GEMM_VARIANTS(X) TE_GEMM_APPLY(X, TE_GEMM_DEV_ONE)
#else
#define TE_GEMM_VARIANTS(X)                                                                         \
  X(false, false, false, 0) X(false, true, false, 0) X(true, false, false, 0) X(true, true, false, 0) \
  X(false, false, true, 0) X(true, false, true, 0)                                                  \
  X(false, false, true, 1) X(false, false, true, 2) X(false, false, true, 3)                        \
  X(false, false, false, 1) X(false, false, false, 2) X(false, false, false, 3)                     \
  X(true, false, false, 2) X(true, false, false, 10) X(true, false, false, 3) X(true, false, false, 7) \
  X(true, false, false, 11) X(true, false, false, 15)                                              \
  X(true, false, true, 2) X(true, false, true, 10) X(true, false, true, 3) X(true, false, true, 15)
#endif
constexpr int variant_key(bool mx, bool epi, bool grouped, int km) {
  return (mx ? 1 : 0) | (epi ? 2 : 0) | (grouped ? 4 : 0) | (km << 3);
}

template <int FA, int FB>
void launch_variant(int key, const void *args_v, hipStream_t stream) {
  const GemmArgs &args = *static_cast<const GemmArgs *>(args_v);
  const dim3 grid(args.grid), block(kThreads);
#ifdef TE_GEMM_STAMPS
  if (key == variant_key(false, false, false, 0) && FA == 0 && FB == 0 && args.dbg >= 101 && args.dbg <= 103) {
    if (args.dbg == 101) hipLaunchKernelGGL((gemm_fp8_kernel<0, 0, false, false, false, 1>), grid, block, 0, stream, args);
    if (args.dbg == 102) hipLaunchKernelGGL((gemm_fp8_kernel<0, 0, false, false, false, 2>), grid, block, 0, stream, args);
    if (args.dbg == 103) hipLaunchKernelGGL((gemm_fp8_kernel<0, 0, false, false, false, 3>), grid, block, 0, stream, args);
    NVTE_CHECK_HIP(hipGetLastError());
    return;
  }
#endif
#define TE_GEMM_CASE(MX_, EPI_, GR_, KM_)                                                                     \
  if (key == variant_key(MX_, EPI_, GR_, KM_)) {                                                              \
    hipLaunchKernelGGL((gemm_fp8_kernel<FA, FB, MX_, EPI_, GR_, 0, KM_>), grid, block, 0, stream, args);      \
    NVTE_CHECK_HIP(hipGetLastError());                                                                        \
    return;                                                                                                   \
  }
  TE_GEMM_VARIANTS(TE_GEMM_CASE)
#undef TE_GEMM_CASE
  NVTE_ERROR("GEMM kernel variant ", key, " is not built");
}

}  // namespace

#if TE_GEMM_HAS_FMT(0)
void gemm_launch_fmt_0(int key, const void *args, hipStream_t stream) { launch_variant<0, 0>(key, args, stream); }
#endif
#if TE_GEMM_HAS_FMT(1)
void gemm_launch_fmt_1(int key, const void *args, hipStream_t stream) { launch_variant<0, 1>(key, args, stream); }
#endif
#if TE_GEMM_HAS_FMT(2)
void gemm_launch_fmt_2(int key, const void *args, hipStream_t stream) { launch_variant<1, 0>(key, args, stream); }
#endif
#if TE_GEMM_HAS_FMT(3)
void gemm_launch_fmt_3(int key, const void *args, hipStream_t stream) { launch_variant<1, 1>(key, args, stream); }
#endif

#if TE_GEMM_HAS_HOST
namespace {

template <bool MX, bool EPI, bool GROUPED = false, int KM = 0>
void launch_by_format(int fa, int fb, const GemmArgs &args_in, hipStream_t stream) {
#ifdef TE_GEMM_STAMPS
  GemmArgs args = args_in;
  args.stamps = g_stamps;
  args.wstamps = g_wstamps;
  args.dbg = std::getenv("TE_GEMM_DBG") ? std::atoi(std::getenv("TE_GEMM_DBG")) : 0;
#else
  const GemmArgs &args = args_in;
#endif
  constexpr int key = variant_key(MX, EPI, GROUPED, KM);
  switch (fa * 2 + fb) {
    case 0: gemm_launch_fmt_0(key, &args, stream); return;
    case 1: gemm_launch_fmt_1(key, &args, stream); return;
    case 2: gemm_launch_fmt_2(key, &args, stream); return;
    case 3: gemm_launch_fmt_3(key, &args, stream); return;
    default: NVTE_ERROR("unsupported FP8 format combination");
  }
}

// NVTE_AMD_GEMM_NT_MB overrides the streaming-store threshold (tuning aid; 0 = always, a huge value = never)
size_t nt_store_bytes() {
  static const size_t v = [] {
    const char *e = std::getenv("NVTE_AMD_GEMM_NT_MB");
    return e ? static_cast<size_t>(std::atoll(e)) << 20 : kNtStoreBytes;
  }();
  return v;
}

int fmt_of(DType t) { return t == kNVTEFloat8E4M3 ? 0 : 1; }

// BGRADB ("bias gradient of input B", ref cublaslt_gemm.cu:667-669): dbias[i] = sum_k B[i][k]
// in dequantized units, one wave per row of the [n, k] K-contiguous FP8 operand.
// scale: per-tensor scale_inv (sc != null) or e8m0 per 32-element block (e != null).
template <Fp8 F>
__global__ __launch_bounds__(256) void rowsum_fp8_kernel(const uint8_t *__restrict__ in, unsigned n, unsigned k,
                                                         const float *sc, const uint8_t *e, unsigned e_stride,
                                                         unsigned e_swz, void *out, int out_dtype) {
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (row >= n) return;
  const uint8_t *src = in + static_cast<size_t>(row) * k;
  float sum = 0.f;
  for (unsigned c = lane * 16; c < k; c += 64 * 16) {  // k % 16 == 0
    const u32x4 v = *reinterpret_cast<const u32x4 *>(src + c);
    float part = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int b = 0; b < 4; ++b) part += fp8_to_float<F>(v[w], b);
    if (e != nullptr) {
      const unsigned blk = c / 32;
      const size_t ei = e_swz ? (static_cast<size_t>(row / 128) * (e_stride / 4) + blk / 4) * 512 +
                                    (row % 32) * 16 + (row % 128 / 32) * 4 + blk % 4
                              : static_cast<size_t>(row) * e_stride + blk;
      part *= e8m0_to_float(e[ei]);
    }
    sum += part;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) store_any(out, row, out_dtype, sc ? sum * *sc : sum);
}

// fp32 column sums (DGELU_BGRAD accumulator) -> bias dtype
__global__ void dbias_convert_kernel(const float *acc, void *out, int out_dtype, unsigned m) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) store_any(out, j, out_dtype, acc[j]);
}

// ref common/common.cu:21-25 (update_tensor_scale_inv_kernel)
__global__ void scale_inv_from_scale_kernel(const float *scale, float *scale_inv) {
  *scale_inv = 1.f / (scale == nullptr ? 1.f : *scale);
}

// One operand after canonicalisation: a K-contiguous byte matrix [rows, k].
struct Operand {
  const uint8_t *data = nullptr;
  DType dtype = kNVTEByte;
  const void *scale_inv = nullptr;
  unsigned scale_stride = 0;
  size_t scale_bytes = 0;
  bool swizzled = false;
  bool k_major = false;   // MXFP8 only: the reference's un-transposed columnwise payload, element (row, k) at k * ld + row
  size_t ld = 0;
};

}  // namespace

// D = alpha * op(A) op(B) + beta * D  (+bias).  Names follow the reference's cublas_gemm.
void gemm_impl(const Tensor &A, const Tensor &B, Tensor &D, const Tensor *bias, const Tensor *pre_gelu,
               bool transa, bool transb, bool grad, const Tensor *workspace, float alpha, float beta,
               hipStream_t stream, int sm_count = 0) {
  const auto [A0, A1] = A.flat2d();
  const auto [B0, B1] = B.flat2d();
  const size_t m = transa ? A0 : A1;
  const size_t k = transa ? A1 : A0;
  const size_t n = transb ? B1 : B0;
  NVTE_CHECK((transb ? B0 : B1) == k, "GEMM inputs have incompatible dimensions (A is ", A0, "x", A1,
             ", B is ", B0, "x", B1, ")");
  if (m == 0 || n == 0) return;
  NVTE_CHECK(k > 0, "GEMM with k == 0 is not supported");
  NVTE_CHECK(A.scaling_mode == B.scaling_mode,
             "Inputs A and B to GEMM need to have compatible scaling modes, but got A.scaling_mode = ",
             scaling_mode_name(A.scaling_mode), ", B.scaling_mode = ", scaling_mode_name(B.scaling_mode));
  NVTE_CHECK(A.has_data() || A.has_columnwise_data(), "Input A does not hold any data!");
  NVTE_CHECK(B.has_data() || B.has_columnwise_data(), "Input B does not hold any data!");
  NVTE_CHECK(D.data.ptr != nullptr, "Output D is not allocated");
  NVTE_CHECK(D.data.numel() == m * n, "Output D has wrong size (expected ", n, "x", m, ", got ",
             D.data.shape, ")");
  const bool fp8_out = is_fp8(D.data.dtype);
  NVTE_CHECK(fp8_out || is_high_precision(D.data.dtype), "GEMM output dtype ", dtype_name(D.data.dtype),
             " is not supported (FP32/BF16/FP16/FP8 only)");
  // Epilogue selection, the table of ref cublaslt_gemm.cu:650-689:
  //   bias & gelu : grad ? DGELU_BGRAD : GELU_AUX_BIAS      bias : grad ? BGRADB : BIAS
  //   gelu        : grad ? DGELU : GELU_AUX
  const bool has_bias = bias != nullptr && bias->data.ptr != nullptr;
  const bool gelu = pre_gelu != nullptr && pre_gelu->data.ptr != nullptr;
  if (gelu) {
    NVTE_CHECK(!is_fp8(pre_gelu->data.dtype), "fp8 Aux output for gemm + gelu fusion not supported!");
    NVTE_CHECK(is_high_precision(pre_gelu->data.dtype), "GELU aux tensor must be FP32/BF16/FP16");
    NVTE_CHECK(pre_gelu->data.numel() == m * n, "GELU aux tensor has wrong size (expected ", n, "x", m,
               ", got ", pre_gelu->data.shape, ")");
  }
  if (has_bias) {
    // BGRADB reduces operand B over k: one entry per column of the column-major D, i.e. n
    const size_t want = (grad && !gelu) ? n : m;
    NVTE_CHECK(bias->data.numel() == want, "Bias has wrong size (expected ", want, ", got ",
               bias->data.numel(), ")");
    NVTE_CHECK(is_high_precision(bias->data.dtype), "Bias dtype must be FP32/BF16/FP16");
  }
  if (fp8_out) {
    NVTE_CHECK(is_tensor_scaling(D.scaling_mode) && is_tensor_scaling(A.scaling_mode),
               "FP8 GEMM output is implemented for per-tensor scaling only");
    NVTE_CHECK(beta == 0.f, "Accumulation mode not supported for FP8 output");
  }

  Operand opA, opB;  // opA rows = m (reference A), opB rows = n (reference B)
  bool mx = false;
  if (is_tensor_scaling(A.scaling_mode)) {
    // Per-tensor scaling.  nvte_is_non_tn_fp8_gemm_supported() == 0, so a non-transposed A
    // (transposed B) is replaced by its pre-transposed "columnwise" copy: every FP8 GEMM is TN.
    // An operand whose contraction runs over its stored rows is read from its pre-transposed copy when the tensor
    // carries one; a tensor that only holds the ROW-WISE payload is read K-major through the transposing LDS loads
    // instead (no transposed copy has to exist: 3 instead of 4 bytes per element on the quantize side, half the FP8
    // memory of that tensor).  The reference takes that route where its GEMM reads non-TN FP8 operands
    // (ref transformer_engine.cpp:1131-1144, pytorch/csrc/quantizer.cpp:342-356).
    auto rowwise_of = [](const Tensor &T, Operand &op, const char *name) {
      NVTE_CHECK(T.has_data(), "Input ", name, " is missing row-wise usage");
      op.data = static_cast<const uint8_t *>(T.data.ptr);
      op.dtype = T.data.dtype;
      op.scale_inv = T.scale_inv.ptr;
    };
    auto contracted_over_rows = [&](const Tensor &T, Operand &op, const char *name, size_t rows) {
      if (T.has_columnwise_data() && is_fp8(T.columnwise_data.dtype)) {
        op.data = static_cast<const uint8_t *>(T.columnwise_data.ptr);
        op.dtype = T.columnwise_data.dtype;
        op.scale_inv = T.columnwise_scale_inv.ptr;
        return;
      }
      NVTE_CHECK(T.has_data() && is_fp8(T.data.dtype), "Input ", name, " is missing column-wise usage (and holds no "
                 "row-wise FP8 payload that could be read K-major)");
      NVTE_CHECK(rows % 16 == 0, "FP8 GEMM: reading the row-wise payload of ", name, " K-major needs its last dimension (",
                 rows, ") to be a multiple of 16; provide the transposed copy instead");
      rowwise_of(T, op, name);
      op.k_major = true;
      op.ld = rows;
    };
    if (transa) rowwise_of(A, opA, "A"); else contracted_over_rows(A, opA, "A", m);
    if (!transb) rowwise_of(B, opB, "B"); else contracted_over_rows(B, opB, "B", n);
    if (opA.k_major || opB.k_major)
      NVTE_CHECK(static_cast<unsigned long long>(k) * std::max(opA.ld, opB.ld) < (1ull << 32),
                 "FP8 GEMM: K-major operand too large for 32-bit offsets");
    NVTE_CHECK(is_fp8(opA.dtype) && is_fp8(opB.dtype),
               "This library implements the FP8 GEMM only; got A dtype ", dtype_name(opA.dtype),
               ", B dtype ", dtype_name(opB.dtype));
    NVTE_CHECK(opA.scale_inv != nullptr && opB.scale_inv != nullptr,
               "FP8 input to GEMM requires inverse of scale!");
    NVTE_CHECK(k % 16 == 0, "Leading dimension requirement on A for FP8 GEMM. Caller must pad.");
  } else if (is_mxfp8(A.scaling_mode)) {
    mx = true;
    // Rowwise data is K-contiguous for a transposed A / non-transposed B.  The other layouts
    // contract over the stored rows: they need the columnwise-quantized copy in the transposed
    // layout of amd_ext.h (the MXFP8 twin of "non-TN FP8 GEMM unsupported -> keep a transpose").
    // `rows` = the operand's non-contracted extent (m for A, n for B)
    auto pick = [](const Tensor &T, bool rowwise, const char *name, size_t rows) {
      Operand op;
      if (rowwise) {
        NVTE_CHECK(T.has_data(), "Input ", name, " is missing row-wise usage");
        op.data = static_cast<const uint8_t *>(T.data.ptr);
        op.dtype = T.data.dtype;
        op.scale_inv = T.scale_inv.ptr;
        op.scale_stride = static_cast<unsigned>(T.scale_inv.shape.data[T.scale_inv.shape.ndim - 1]);
        op.scale_bytes = T.scale_inv.numel();
      } else {
        NVTE_CHECK(T.has_columnwise_data(), "Input ", name, " is missing column-wise usage");
        op.data = static_cast<const uint8_t *>(T.columnwise_data.ptr);
        op.dtype = T.columnwise_data.dtype;
        op.scale_inv = T.columnwise_scale_inv.ptr;
        const NVTEShape &ss = T.columnwise_scale_inv.shape;
        op.scale_bytes = T.columnwise_scale_inv.numel();
        if (T.columnwise_transposed) {
          // amd_ext.h layout: payload [rows, k], scales [pad128(rows), pad4(k / 32)] - K-contiguous like a rowwise operand
          op.scale_stride = static_cast<unsigned>(ss.data[ss.ndim - 1]);
        } else {
          // the reference's layout (ref cublaslt_gemm.cu:179-193, :268-278: the columnwise payload keeps the tensor's own
          // [k, rows] shape): K-MAJOR, read through the transposing LDS loads.  Compact scales [pad4(k / 32), pad128(rows)]
          // are addressed by their row length; GEMM-swizzled ones have the one (row, k block) tile format whichever
          // direction they were quantized in (ref common/swizzle/swizzle.cu), described by pad4(k / 32) blocks per row.
          op.k_major = true;
          op.ld = rows;
          NVTE_CHECK(rows % 16 == 0, "MXFP8 GEMM: the un-transposed columnwise payload of ", name,
                     " is read in 16-byte pieces along its last dimension (", rows, "), which must be a multiple of 16");
          NVTE_CHECK(ss.ndim >= 2, "MXFP8 GEMM: columnwise scale_inv of ", name, " must be 2-D");
          op.scale_stride = static_cast<unsigned>(T.with_gemm_swizzled_scales ? ss.data[ss.ndim - 2] : ss.data[ss.ndim - 1]);
          NVTE_CHECK((reinterpret_cast<uintptr_t>(op.scale_inv) & 15u) == 0, "MXFP8 GEMM: columnwise scale_inv of ", name,
                     " must be 16-byte aligned");
        }
      }
      op.swizzled = T.with_gemm_swizzled_scales;
      return op;
    };
    opA = pick(A, transa, "A", m);
    opB = pick(B, !transb, "B", n);
    if (opA.k_major || opB.k_major) {
      NVTE_CHECK(opA.k_major, "MXFP8 GEMM: a K-major B operand is implemented together with a K-major A only (the NT "
                 "layout); quantize B with the transposed columnwise layout or A without it");
      NVTE_CHECK(static_cast<unsigned long long>(k) * std::max(opA.ld, opB.ld) < (1ull << 32),
                 "MXFP8 GEMM: K-major operand too large for 32-bit offsets");
    }
    NVTE_CHECK(is_fp8(opA.dtype) && is_fp8(opB.dtype), "MXFP8 GEMM needs FP8 payloads");
    NVTE_CHECK(opA.scale_inv != nullptr && opB.scale_inv != nullptr,
               "FP8 input to GEMM requires inverse of scale!");
    NVTE_CHECK(k % 32 == 0, "MXFP8 GEMM requires K to be a multiple of 32, got ", k);
  } else {
    NVTE_ERROR("Not implemented scaling modes: ", scaling_mode_name(A.scaling_mode), " and ",
               scaling_mode_name(B.scaling_mode), ".");
  }
  NVTE_CHECK((reinterpret_cast<uintptr_t>(opA.data) & 15u) == 0 &&
                 (reinterpret_cast<uintptr_t>(opB.data) & 15u) == 0,
             "FP8 GEMM operands must be 16-byte aligned");
  NVTE_CHECK(k <= (1u << 23), "K too large for this kernel's 32-bit tile offsets");
  NVTE_CHECK(n < (1ull << 31) && m < (1ull << 31), "matrix too large");

  (void)device_info();  // throws without a usable GPU

  GemmArgs g;
  std::memset(&g, 0, sizeof(g));
  g.P = opB.data;
  g.Q = opA.data;
  g.D = D.data.ptr;
  NVTE_CHECK(g.P != nullptr && g.Q != nullptr, "GEMM operands have shapes but no data pointers");
  NVTE_CHECK(g.D != nullptr, "GEMM output D has a shape but no data pointer");
  g.bias_dtype = has_bias ? bias->data.dtype : kNVTEFloat32;
  if (mx) {
    g.eP = static_cast<const uint8_t *>(opB.scale_inv);
    g.eQ = static_cast<const uint8_t *>(opA.scale_inv);
    g.eP_stride = opB.scale_stride;
    g.eQ_stride = opA.scale_stride;
    g.eP_swz = opB.swizzled;
    g.eP_bytes = static_cast<unsigned>(std::min<size_t>(opB.scale_bytes, 0xFFFFFFFFu));
    g.eQ_bytes = static_cast<unsigned>(std::min<size_t>(opA.scale_bytes, 0xFFFFFFFFu));
    g.eQ_swz = opA.swizzled;
  } else {
    g.sP = static_cast<const float *>(opB.scale_inv);
    g.sQ = static_cast<const float *>(opA.scale_inv);
  }
  g.n = static_cast<unsigned>(n);
  g.m = static_cast<unsigned>(m);
  g.k = static_cast<unsigned>(k);
  g.ldd = static_cast<unsigned>(m);
  g.ldP = static_cast<unsigned>(opB.ld);
  g.ldQ = static_cast<unsigned>(opA.ld);
  g.alpha = alpha;
  g.beta = beta;
  g.d_dtype = D.data.dtype;
  g.tiles_j = static_cast<unsigned>(ceil_div(m, BN));
  g.ntiles = static_cast<unsigned>(ceil_div(n, BM)) * g.tiles_j;
  // Persistent launch: one workgroup per CU walks the tile list; a K of a single tile cannot
  // pipeline across output tiles, so it gets one workgroup per tile.
  // `sm_count` > 0 (math_sm_count / kNVTEMatmulConfigSMCount, ref pytorch/csrc/extensions/gemm.cpp:281-298:
  // the framework leaves NVTE_EXT_MARGIN_SM compute units to the communication kernels of an overlapped
  // step) caps the persistent grid: the other CUs stay free for whatever runs beside this launch.
  unsigned cus = static_cast<unsigned>(device_info().num_cus);
  if (sm_count > 0) cus = std::min(cus, static_cast<unsigned>(sm_count));
  g.grid = (k > BK) ? std::min(g.ntiles, cus) : g.ntiles;
  g.nt_store = (m * n * dtype_bits(D.data.dtype) / 8 > nt_store_bytes()) ? 1u : 0u;
  const bool bias_in_kernel = has_bias && !grad;
  const bool dgelu_bgrad = has_bias && grad && gelu;
  const bool bgradb = has_bias && grad && !gelu;
  g.bias = bias_in_kernel ? bias->data.ptr : nullptr;
  const bool ext = gelu || fp8_out;
  if (ext) {
    g.epi = gelu ? (grad ? kEpiDgelu : kEpiGeluAux) : kEpiNone;
    g.aux = gelu ? pre_gelu->data.ptr : nullptr;
    g.aux_dtype = gelu ? pre_gelu->data.dtype : kNVTEFloat32;
    if (fp8_out) {
      g.d_scale = static_cast<const float *>(D.scale.ptr);
      g.d_amax = static_cast<float *>(D.amax.ptr);
      if (g.d_amax) zero_async(g.d_amax, sizeof(float), stream);
    }
    if (dgelu_bgrad) {
      // fp32 column sums by atomics: straight into an fp32 bias, else into the caller's workspace
      if (bias->data.dtype == kNVTEFloat32) {
        g.dbias_acc = static_cast<float *>(bias->data.ptr);
      } else {
        const size_t ws_bytes = workspace ? workspace->data.numel() * dtype_bits(workspace->data.dtype) / 8 : 0;
        uintptr_t ws = workspace ? reinterpret_cast<uintptr_t>(workspace->data.ptr) : 0;
        const uintptr_t ws_al = (ws + 255) & ~static_cast<uintptr_t>(255);
        NVTE_CHECK(ws != 0 && ws_al - ws + m * sizeof(float) <= ws_bytes,
                   "DGELU_BGRAD epilogue with a ", dtype_name(bias->data.dtype), " bias gradient needs ",
                   m * sizeof(float) + 256, " bytes of workspace for the FP32 column sums, got ", ws_bytes);
        g.dbias_acc = reinterpret_cast<float *>(ws_al);
      }
      zero_async(g.dbias_acc, m * sizeof(float), stream);
    }
  }
  const int fp = fmt_of(opB.dtype), fq = fmt_of(opA.dtype);
  if (mx && (opA.k_major || opB.k_major)) {
    NVTE_CHECK(!ext, "MXFP8 GEMM: the GELU / FP8-output epilogues read K-contiguous operands only; quantize with the "
               "transposed columnwise layout (nvte_amd_set_columnwise_transposed) for this call");
    // kernel variant: bit 0 / 1 = P / Q K-major, bit 2 / 3 = that operand's scales are compact ([k block][row])
    const int km = (opB.k_major ? 1 : 0) | 2 | ((opB.k_major && !opB.swizzled) ? 4 : 0) | (!opA.swizzled ? 8 : 0);
    switch (km) {
      case 2: launch_by_format<true, false, false, 2>(fp, fq, g, stream); break;
      case 10: launch_by_format<true, false, false, 10>(fp, fq, g, stream); break;
      case 3: launch_by_format<true, false, false, 3>(fp, fq, g, stream); break;
      case 7: launch_by_format<true, false, false, 7>(fp, fq, g, stream); break;
      case 11: launch_by_format<true, false, false, 11>(fp, fq, g, stream); break;
      default: launch_by_format<true, false, false, 15>(fp, fq, g, stream); break;
    }
  } else if (mx) {
    ext ? launch_by_format<true, true>(fp, fq, g, stream) : launch_by_format<true, false>(fp, fq, g, stream);
  } else if (opA.k_major || opB.k_major) {
    NVTE_CHECK(!ext, "FP8 GEMM: the GELU / FP8-output epilogues read K-contiguous operands only; provide the transposed "
               "copies (column-wise usage) for this call");
    switch ((opB.k_major ? 1 : 0) | (opA.k_major ? 2 : 0)) {
      case 1: launch_by_format<false, false, false, 1>(fp, fq, g, stream); break;
      case 2: launch_by_format<false, false, false, 2>(fp, fq, g, stream); break;
      default: launch_by_format<false, false, false, 3>(fp, fq, g, stream); break;
    }
  } else
    ext ? launch_by_format<false, true>(fp, fq, g, stream) : launch_by_format<false, false>(fp, fq, g, stream);

  if (dgelu_bgrad && bias->data.dtype != kNVTEFloat32) {
    hipLaunchKernelGGL(dbias_convert_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, stream, g.dbias_acc,
                       bias->data.ptr, static_cast<int>(bias->data.dtype), static_cast<unsigned>(m));
    NVTE_CHECK_HIP(hipGetLastError());
  }
  if (bgradb) {
    NVTE_CHECK(!opB.k_major, "FP8 GEMM: the bias-gradient side kernel sums a K-contiguous operand B; give B its "
               "transposed copy (MXFP8: nvte_amd_set_columnwise_transposed) or compute dbias in the quantize "
               "(nvte_quantize_dbias)");
    // operand B as the kernel reads it: [n, k] K-contiguous
    const dim3 grid(static_cast<unsigned>(ceil_div(n, 4))), block(256);
    const float *sc = mx ? nullptr : static_cast<const float *>(opB.scale_inv);
    const uint8_t *e = mx ? static_cast<const uint8_t *>(opB.scale_inv) : nullptr;
    if (opB.dtype == kNVTEFloat8E4M3)
      hipLaunchKernelGGL(rowsum_fp8_kernel<Fp8::E4M3>, grid, block, 0, stream, opB.data, g.n, g.k, sc, e,
                         opB.scale_stride, opB.swizzled ? 1u : 0u, bias->data.ptr, static_cast<int>(bias->data.dtype));
    else
      hipLaunchKernelGGL(rowsum_fp8_kernel<Fp8::E5M2>, grid, block, 0, stream, opB.data, g.n, g.k, sc, e,
                         opB.scale_stride, opB.swizzled ? 1u : 0u, bias->data.ptr, static_cast<int>(bias->data.dtype));
    NVTE_CHECK_HIP(hipGetLastError());
  }
  // ref cublaslt_gemm.cu:785-791
  if (fp8_out && D.scale_inv.ptr != nullptr) {
    hipLaunchKernelGGL(scale_inv_from_scale_kernel, dim3(1), dim3(1), 0, stream,
                       static_cast<const float *>(D.scale.ptr), static_cast<float *>(D.scale_inv.ptr));
    NVTE_CHECK_HIP(hipGetLastError());
  }
}


// ------------------------------------------------------------------------------------------
// Grouped GEMM (ref common/gemm/cublaslt_grouped_gemm.cu): D[i] = alpha[i] op(A[i]) op(B[i]) +
// beta[i] C[i] for all members in ONE persistent launch.  A device-side setup kernel turns the
// grouped tensors' metadata (first_dims / last_dims / tensor_offsets, all device arrays) into
// per-member GroupDesc records + a tile prefix in workspace_setup; the main kernel walks the
// global tile list.  No host read of any member shape: HIP-graph safe.
// ------------------------------------------------------------------------------------------
namespace {

constexpr int kMaxListed = 64;  // discrete A / C / D lists travel in the kernel arguments

struct GroupedOperandMeta {
  const uint8_t *base = nullptr;     // payload of the storage picked (null: discrete list)
  const float *scale_inv = nullptr;  // per-tensor scaling: one entry per member
  const uint8_t *e8m0 = nullptr;     // MXFP8: the members' scale matrices back to back
  int k_is_last = 1;                 // MXFP8: 1 = rowwise storage, 0 = per-member transposed columnwise
  int k_major = 0;                   // the payload of an operand contracted over its ROWS is read as it is stored: per-tensor
                                     // FP8 row-wise data, MXFP8 the reference's un-transposed columnwise data
  unsigned ld = 0;                   // ... and its row length (bytes between consecutive k), one for all members
  int km_compact = 0;                // MXFP8 K-major: compact scales, stored [k block][row] per member
  GroupShapes shapes;
};
struct GroupedOutMeta {
  char *base = nullptr;  // null: discrete list (or absent, for C)
  GroupShapes shapes;
};
struct ListedInputs {
  const uint8_t *ptr[kMaxListed];
  const float *scale_inv[kMaxListed];
  unsigned first[kMaxListed], last[kMaxListed];  // LOGICAL dims of the member
};
struct ListedOutputs {
  char *d[kMaxListed];
  const char *c[kMaxListed];
  unsigned first[kMaxListed], last[kMaxListed];
};

__device__ __forceinline__ void member_dims(const GroupShapes &g, unsigned i, int64_t &first, int64_t &last,
                                            int64_t &off) {
  first = g.first_dims ? g.first_dims[i] : g.uniform_first;
  last = g.last_dims ? g.last_dims[i] : g.uniform_last;
  off = g.offsets ? g.offsets[i] : static_cast<int64_t>(i) * g.uniform_first * g.uniform_last;
}

// Tail of the setup kernels (one wave; lane l owns members [lo_i, hi_i), whose records it has just written):
// the tile prefix tile_start[0..G] and, behind it, the 9 boundaries of the per-XCD runs of the global tile list.
// A tile costs its K tiles (never fewer than the two the pipeline runs) + 2 for the epilogue; boundary x is
// where the running cost passes x/8 of the total.  With one K for every member this is the equal-count split;
// with a workgroup count that is not a multiple of 8 (a workgroup then changes XCD run from tile to tile) the
// equal-count split is forced: it is the one for which "flat index < total" and "position inside its run" agree.
__device__ __forceinline__ void finish_group_setup(const GroupDesc *descs, unsigned *tile_start, unsigned G,
                                                   unsigned lo_i, unsigned hi_i, unsigned grid) {
  const unsigned lane = threadIdx.x;
  unsigned local = 0;
  unsigned long long local_cost = 0;
  for (unsigned i = lo_i; i < hi_i; ++i) {
    const unsigned tiles = ((descs[i].n + BM - 1) / BM) * descs[i].tiles_j;
    local += tiles;
    local_cost += static_cast<unsigned long long>(tiles) * (max((descs[i].k + BK - 1) / BK, 2u) + 2u);
  }
  unsigned incl = local;
  unsigned long long incl_cost = local_cost;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned up = __shfl_up(incl, off, 64);
    const unsigned long long upc = __shfl_up(incl_cost, off, 64);
    if (lane >= static_cast<unsigned>(off)) {
      incl += up;
      incl_cost += upc;
    }
  }
  const unsigned total = __shfl(incl, 63, 64);
  const unsigned long long total_cost = __shfl(incl_cost, 63, 64);
  unsigned run = incl - local;
  for (unsigned i = lo_i; i < hi_i; ++i) {
    tile_start[i] = run;
    run += ((descs[i].n + BM - 1) / BM) * descs[i].tiles_j;
  }
  unsigned *xrun = tile_start + G + 1;
  if (lane == 63) {
    tile_start[G] = total;
    xrun[0] = 0;
    xrun[8] = total;
  }
  if (grid % 8 != 0 || total_cost == 0) {
    if (lane >= 1 && lane < 8) {
      const unsigned q8 = total / 8, r8 = total % 8;
      xrun[lane] = lane < r8 ? lane * (q8 + 1) : r8 * (q8 + 1) + (lane - r8) * q8;
    }
    return;
  }
  // the lane whose members straddle a boundary's cost knows where that boundary ideally falls ...
  unsigned ideal[8];
#pragma unroll
  for (int x = 0; x < 8; ++x) ideal[x] = 0xFFFFFFFFu;
  unsigned long long cost_run = incl_cost - local_cost;
  unsigned tile_run = incl - local;
  for (unsigned i = lo_i; i < hi_i; ++i) {
    const unsigned tiles = ((descs[i].n + BM - 1) / BM) * descs[i].tiles_j;
    const unsigned long long c = max((descs[i].k + BK - 1) / BK, 2u) + 2u;
    const unsigned long long end = cost_run + tiles * c;
#pragma unroll
    for (unsigned x = 1; x < 8; ++x) {
      const unsigned long long target = total_cost * x / 8;
      if (tiles != 0 && target >= cost_run && target < end)
        ideal[x] = tile_run + static_cast<unsigned>((target - cost_run + c / 2) / c);
    }
    cost_run = end;
    tile_run += tiles;
  }
  // ... every lane learns all seven (wave minimum: exactly one lane holds each), and lane 0 snaps the run
  // LENGTHS to whole rounds of the XCD's grid / 8 workgroups: a run of 129 tiles costs its 32 workgroups five
  // rounds, one of 128 four (the last run takes what the rounding leaves)
#pragma unroll
  for (int x = 1; x < 8; ++x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ideal[x] = min(ideal[x], static_cast<unsigned>(__shfl_xor(ideal[x], off, 64)));
  }
  if (lane == 0) {
    const unsigned wpx = grid / 8;
    unsigned prev = 0;
#pragma unroll
    for (int x = 1; x < 8; ++x) {
      const unsigned want = min(max(ideal[x], prev), total);
      unsigned b = prev + (want - prev + wpx / 2) / wpx * wpx;
      b = min(b, total);
      xrun[x] = b;
      prev = b;
    }
  }
}

// One wave: lane l builds the records of members [l * per, (l + 1) * per) and the tile prefix.
__global__ __launch_bounds__(64) void grouped_setup_kernel(GroupDesc *descs, unsigned *tile_start,
                                                           const GroupedOperandMeta A, const GroupedOperandMeta B,
                                                           const GroupedOutMeta C, const GroupedOutMeta D,
                                                           const ListedInputs la, const ListedOutputs lo,
                                                           const float *alpha, int alpha_n, const float *beta,
                                                           int beta_n, unsigned G, int transa, int transb,
                                                           unsigned d_elem_bytes, int has_c, GroupScaleDesc *sdescs,
                                                           unsigned grid) {
  const unsigned lane = threadIdx.x;
  const unsigned per = (G + 63) / 64, lo_i = lane * per, hi_i = min(lo_i + per, G);
  const bool mx = sdescs != nullptr;
  // MXFP8: member i's scale matrix starts at the sum of the PADDED sizes of the members before it
  // (ref cublaslt_grouped_gemm.cu:1117-1198 compute_grouped_scale_inv_offset): [pad128(rows), pad4(k/32)]
  // with rows x k the member's K-contiguous view (rowwise storage, or the transposed columnwise one).
  auto scale_dims = [](const GroupedOperandMeta &o, int64_t first, int64_t last, unsigned &stride, unsigned &bytes) {
    const int64_t rows = o.k_is_last ? first : last, kk = o.k_is_last ? last : first;
    stride = static_cast<unsigned>(((kk + 31) / 32 + 3) / 4 * 4);
    bytes = static_cast<unsigned>((rows + 127) / 128 * 128) * stride;
    // the reference's columnwise scale matrix of a member, compact: [pad4(first / 32), pad128(last)] (same bytes)
    if (o.km_compact) stride = static_cast<unsigned>((rows + 127) / 128 * 128);
  };
  unsigned a_sc_run = 0, b_sc_run = 0;
  // MXFP8 contract (amd_ext.h): every member's FIRST dim is a multiple of 128.  The quantize kernels write a
  // member's scales at its global, unpadded row offset while the offsets here are sums of padded sizes; the
  // two agree only on that contract, so one member off it misplaces the scales of every LATER member too.
  // `mx_bad_run` = some member up to and including the current one breaks the contract: those members
  // become NaN like the other inconsistent shapes instead of silently mis-scaled numbers.
  unsigned mx_bad_run = 0;
  if (mx) {
    unsigned a_loc = 0, b_loc = 0, bad_loc = 0;
    for (unsigned i = lo_i; i < hi_i; ++i) {
      int64_t f, l, o;
      unsigned st, by;
      member_dims(A.shapes, i, f, l, o);
      scale_dims(A, f, l, st, by);
      a_loc += by;
      bad_loc |= (f % 128 != 0) ? 1u : 0u;
      member_dims(B.shapes, i, f, l, o);
      scale_dims(B, f, l, st, by);
      b_loc += by;
      bad_loc |= (f % 128 != 0) ? 1u : 0u;
    }
    unsigned a_incl = a_loc, b_incl = b_loc, bad_incl = bad_loc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned ua = __shfl_up(a_incl, off, 64), ub = __shfl_up(b_incl, off, 64);
      const unsigned ubad = __shfl_up(bad_incl, off, 64);
      if (lane >= static_cast<unsigned>(off)) { a_incl += ua; b_incl += ub; bad_incl |= ubad; }
    }
    a_sc_run = a_incl - a_loc;
    b_sc_run = b_incl - b_loc;
    const unsigned bad_before = __shfl_up(bad_incl, 1, 64);
    mx_bad_run = lane > 0 ? bad_before : 0u;
  }
  for (unsigned i = lo_i; i < hi_i; ++i) {
    int64_t a_first, a_last, a_off = 0, b_first, b_last, b_off, d_first, d_last, d_off = 0;
    const uint8_t *q_ptr;
    const float *q_scale;
    if (A.base) {
      member_dims(A.shapes, i, a_first, a_last, a_off);
      q_ptr = A.base + a_off;
      q_scale = mx ? nullptr : A.scale_inv + i;
    } else {
      a_first = la.first[i];
      a_last = la.last[i];
      q_ptr = la.ptr[i];
      q_scale = la.scale_inv[i];
    }
    member_dims(B.shapes, i, b_first, b_last, b_off);
    char *d_ptr;
    const char *c_ptr = nullptr;
    if (D.base) {
      member_dims(D.shapes, i, d_first, d_last, d_off);
      d_ptr = D.base + d_off * d_elem_bytes;
      if (has_c) {
        int64_t c_first, c_last, c_off;
        member_dims(C.base ? C.shapes : D.shapes, i, c_first, c_last, c_off);
        c_ptr = (C.base ? C.base : D.base) + c_off * d_elem_bytes;
      }
    } else {
      d_first = lo.first[i];
      d_last = lo.last[i];
      d_ptr = lo.d[i];
      if (has_c) c_ptr = lo.c[i];
    }
    const int64_t m = transa ? a_first : a_last, ka = transa ? a_last : a_first;
    const int64_t n = transb ? b_last : b_first, kb = transb ? b_first : b_last;
    // the stores follow D's own shape; anything inconsistent (or a K the 16-byte loads cannot take)
    // turns the member's output into NaN instead of reading out of bounds
    if (mx) mx_bad_run |= (a_first % 128 != 0 || b_first % 128 != 0) ? 1u : 0u;
    // K-major operands are staged in 16-byte units along their rows, inside 32-bit descriptors
    const bool km_ok = (!B.k_major || (n % 16 == 0 && static_cast<uint64_t>(kb) * B.ld < (1ull << 32))) &&
                       (!A.k_major || (m % 16 == 0 && static_cast<uint64_t>(ka) * A.ld < (1ull << 32)));
    const bool ok = d_first == n && d_last == m && ka == kb && ka % (mx ? 32 : 16) == 0 && ka <= (1 << 23) &&
                    mx_bad_run == 0 && km_ok;
    GroupDesc d;
    d.P = B.base + b_off;
    d.Q = q_ptr;
    d.D = d_ptr;
    d.C = c_ptr;
    d.n = static_cast<unsigned>(d_first);
    d.m = static_cast<unsigned>(d_last);
    d.k = ok ? static_cast<unsigned>(ka) : 0u;
    if (mx) {
      GroupScaleDesc sd;
      scale_dims(B, b_first, b_last, sd.eP_stride, sd.eP_bytes);
      scale_dims(A, a_first, a_last, sd.eQ_stride, sd.eQ_bytes);
      sd.eP = B.e8m0 + b_sc_run;
      sd.eQ = A.e8m0 + a_sc_run;
      b_sc_run += sd.eP_bytes;
      a_sc_run += sd.eQ_bytes;
      sdescs[i] = sd;
      d.mult = ok ? alpha[alpha_n > 1 ? i : 0] : __builtin_nanf("");
    } else {
      d.mult = ok ? alpha[alpha_n > 1 ? i : 0] * B.scale_inv[i] * *q_scale : __builtin_nanf("");
    }
    d.beta = has_c ? beta[beta_n > 1 ? i : 0] : 0.f;
    d.tiles_j = (d.m + BN - 1) / BN;
    descs[i] = d;
  }
  finish_group_setup(descs, tile_start, G, lo_i, hi_i, grid);
}

// records | tile prefix + the 9 XCD run boundaries | MXFP8 scale records (+ 256 to align the caller's pointer)
constexpr size_t cround_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
constexpr size_t group_prefix_bytes(size_t G) { return cround_up((G + 1 + 9) * sizeof(unsigned), 256); }
constexpr size_t grouped_setup_bytes(size_t G) {
  return cround_up(G * sizeof(GroupDesc), 256) + group_prefix_bytes(G) + cround_up(G * sizeof(GroupScaleDesc), 256) + 256;
}

// Pick the K-contiguous storage of an FP8 grouped operand ("TN everywhere", like the ungrouped path
// and ref cublaslt_grouped_gemm.cu:551-565): rowwise when the contraction runs along the member's last
// dim, else the per-member transposed ("columnwise") copy.
GroupedOperandMeta pick_grouped_operand(const GroupedTensor &t, bool k_is_last, const char *name, DType &dtype) {
  const bool mx = is_mxfp8(t.scaling_mode);
  NVTE_CHECK(is_tensor_scaling(t.scaling_mode) || mx,
             "Grouped GEMM: FP8 per-tensor scaling and MXFP8 are implemented in this library, ", name, " has ",
             scaling_mode_name(t.scaling_mode));
  GroupedOperandMeta m;
  const void *sinv;
  if (k_is_last) {
    NVTE_CHECK(t.has_data(), "Grouped GEMM: ", name, " is missing row-wise data");
    m.base = static_cast<const uint8_t *>(t.data.ptr);
    sinv = t.scale_inv.ptr;
    dtype = t.data.dtype;
  } else if (!mx && !t.has_columnwise_data() && t.has_data() && t.last_dims.ptr == nullptr) {
    // per-tensor FP8 without transposed copies: the row-wise payload is read K-major (ds_read_b64_tr_b8), as the
    // dense GEMM does; the members share one row length
    const GroupShapes gs = group_shapes_of(t);
    NVTE_CHECK(gs.uniform_last > 0 && gs.uniform_last % 16 == 0 && gs.uniform_last < (1ll << 31), "Grouped GEMM: ", name,
               " has no column-wise data and its last dimension (", gs.uniform_last,
               ") cannot be read K-major (multiple of 16 needed)");
    m.base = static_cast<const uint8_t *>(t.data.ptr);
    sinv = t.scale_inv.ptr;
    dtype = t.data.dtype;
    m.k_major = 1;
    m.ld = static_cast<unsigned>(gs.uniform_last);
  } else {
    NVTE_CHECK(t.has_columnwise_data(), "Grouped GEMM: ", name, " is missing column-wise data needed for TN layout");
    m.base = static_cast<const uint8_t *>(t.columnwise_data.ptr);
    sinv = t.columnwise_scale_inv.ptr;
    dtype = t.columnwise_data.dtype;
    if (mx && !t.columnwise_transposed) {
      // the reference's columnwise layout (ref cublaslt_grouped_gemm.cu:551-565 hands it to cuBLASLt as a non-TN
      // operand): members un-transposed [first_i, last], scales per member [pad4(first_i / 32), pad128(last)]
      // (compact) or GEMM-swizzled as [pad128(last), pad4(first_i / 32)] - read K-major like the dense GEMM does
      const GroupShapes gs = group_shapes_of(t);
      NVTE_CHECK(t.last_dims.ptr == nullptr && gs.uniform_last > 0 && gs.uniform_last % 16 == 0 &&
                     gs.uniform_last < (1ll << 31),
                 "Grouped GEMM: MXFP8 ", name, " in the reference's columnwise layout is read K-major and needs one "
                 "last dimension for all members, a multiple of 16 (or quantize it with the transposed columnwise "
                 "layout, nvte_amd_set_grouped_columnwise_transposed)");
      m.k_major = 1;
      m.ld = static_cast<unsigned>(gs.uniform_last);
      m.km_compact = t.with_gemm_swizzled_scales ? 0 : 1;
    }
  }
  m.k_is_last = k_is_last ? 1 : 0;
  if (mx) m.e8m0 = static_cast<const uint8_t *>(sinv); else m.scale_inv = static_cast<const float *>(sinv);
  NVTE_CHECK(is_fp8(dtype), "Grouped GEMM: this library implements FP8 operands only, ", name, " is ",
             dtype_name(dtype));
  NVTE_CHECK(sinv != nullptr, "Grouped GEMM: FP8 input ", name, " requires inverse of scale!");
  NVTE_CHECK((reinterpret_cast<uintptr_t>(m.base) & 15u) == 0, "Grouped GEMM: ", name, " must be 16-byte aligned");
  m.shapes = group_shapes_of(t);
  return m;
}

struct GroupedCall {
  const GroupedTensor *A = nullptr;  // or a_list
  const NVTETensor *a_list = nullptr;
  const GroupedTensor *B = nullptr;
  const GroupedTensor *C = nullptr;  // or c_list, or neither
  GroupedTensor *D = nullptr;        // or d_list
  const NVTETensor *c_list = nullptr;
  NVTETensor *d_list = nullptr;
  size_t num_c = 0;
};

// `ws_ptr` / `ws_bytes`: device memory for the GroupDesc records + tile prefix (the caller's
// workspace_setup, or this library's static scratch array for the entry point that allocates nothing).
void grouped_gemm_impl(const char *api, const GroupedCall &call, size_t G, bool transa, bool transb,
                       const NVTETensor alpha_, const NVTETensor beta_, void *ws_ptr, size_t ws_bytes,
                       NVTEGroupedMatmulConfig config, hipStream_t stream) {
  NVTE_CHECK(G >= 1, "Grouped GEMM: number of tensors must be at least 1");
  NVTE_CHECK(G <= 1024, "Grouped GEMM: at most 1024 member tensors are supported, got ", G);
  const Tensor &alpha = *lookup_checked(alpha_);
  const Tensor &beta = *lookup_checked(beta_);
  NVTE_CHECK(alpha.data.dtype == kNVTEFloat32 && beta.data.dtype == kNVTEFloat32 && alpha.data.ptr && beta.data.ptr,
             "Grouped GEMM: alpha and beta must be FP32 device tensors");
  NVTE_CHECK(alpha.data.numel() == G || alpha.data.numel() == 1, "Grouped GEMM: alpha must have ", G,
             " (or 1) elements, got ", alpha.data.numel());
  NVTE_CHECK(beta.data.numel() == G || beta.data.numel() == 1, "Grouped GEMM: beta must have ", G,
             " (or 1) elements, got ", beta.data.numel());

  DType a_dtype = kNVTEByte, b_dtype = kNVTEByte;
  GroupedOperandMeta A, B;
  ListedInputs la;
  ListedOutputs lo;
  std::memset(&la, 0, sizeof(la));
  std::memset(&lo, 0, sizeof(lo));
  NVTE_CHECK(call.B->num_tensors == G, "Grouped GEMM: B must have num_tensors = ", G);
  check_grouped_input(*call.B, "B");
  B = pick_grouped_operand(*call.B, !transb, "B", b_dtype);
  const bool mx = is_mxfp8(call.B->scaling_mode);
  bool a_swz = false;
  if (call.A != nullptr) {
    NVTE_CHECK(call.A->num_tensors == G, "Grouped GEMM: A must have num_tensors = ", G);
    check_grouped_input(*call.A, "A");
    NVTE_CHECK(is_mxfp8(call.A->scaling_mode) == mx, "Grouped GEMM: A and B must both use MXFP8 scaling or both not.");
    A = pick_grouped_operand(*call.A, transa, "A", a_dtype);
    a_swz = call.A->with_gemm_swizzled_scales;
  } else {
    NVTE_CHECK(!mx, "Grouped GEMM: a discrete A_list is implemented for FP8 per-tensor scaling only");
    NVTE_CHECK(G <= static_cast<size_t>(kMaxListed), "Grouped GEMM: A_list supports at most ", kMaxListed,
               " tensors, got ", G);
    for (size_t i = 0; i < G; ++i) {
      const Tensor &t = *lookup_checked(call.a_list[i]);
      NVTE_CHECK(is_tensor_scaling(t.scaling_mode), "Grouped GEMM: A_list tensor ", i,
                 " must use FP8 per-tensor scaling");
      // no transposed copy: the row-wise payload is read K-major (all list members then need one row length)
      const bool a_km = !transa && t.columnwise_data.ptr == nullptr && t.data.ptr != nullptr;
      const Buf &payload = (transa || a_km) ? t.data : t.columnwise_data;
      const Buf &sinv = (transa || a_km) ? t.scale_inv : t.columnwise_scale_inv;
      NVTE_CHECK(payload.ptr != nullptr, "Grouped GEMM: A_list tensor ", i, " is missing ",
                 transa ? "row-wise" : "column-wise", " data");
      NVTE_CHECK(is_fp8(payload.dtype), "Grouped GEMM: A_list tensor ", i, " must be FP8");
      NVTE_CHECK(sinv.ptr != nullptr, "Grouped GEMM: A_list tensor ", i, " is missing scale_inv");
      NVTE_CHECK((reinterpret_cast<uintptr_t>(payload.ptr) & 15u) == 0, "Grouped GEMM: A_list tensor ", i,
                 " must be 16-byte aligned");
      if (i == 0) a_dtype = payload.dtype;
      NVTE_CHECK(payload.dtype == a_dtype, "Grouped GEMM: A_list tensors must share one dtype");
      const auto dims = t.flat2d();  // logical [first, last]
      NVTE_CHECK(dims[0] < (1ull << 31) && dims[1] < (1ull << 31), "Grouped GEMM: A_list tensor ", i, " too large");
      NVTE_CHECK(i == 0 || (a_km ? 1 : 0) == A.k_major, "Grouped GEMM: A_list tensors must all carry column-wise data or none");
      if (a_km) {
        NVTE_CHECK(dims[1] % 16 == 0 && (i == 0 || A.ld == dims[1]), "Grouped GEMM: A_list tensors without column-wise "
                   "data are read K-major and need one last dimension, a multiple of 16 (tensor ", i, ": ", dims[1], ")");
        A.k_major = 1;
        A.ld = static_cast<unsigned>(dims[1]);
      }
      la.ptr[i] = static_cast<const uint8_t *>(payload.ptr);
      la.scale_inv[i] = static_cast<const float *>(sinv.ptr);
      la.first[i] = static_cast<unsigned>(dims[0]);
      la.last[i] = static_cast<unsigned>(dims[1]);
    }
  }

  GroupedOutMeta C, D;
  DType d_dtype;
  int has_c = 0;
  if (call.D != nullptr) {
    NVTE_CHECK(call.D->num_tensors == G, "Grouped GEMM: D must have num_tensors = ", G);
    check_grouped_output(*call.D, "D", false);
    NVTE_CHECK(call.D->has_data(), "Grouped GEMM: D is missing row-wise data");
    d_dtype = call.D->data.dtype;
    D.base = static_cast<char *>(call.D->data.ptr);
    D.shapes = group_shapes_of(*call.D);
    if (call.C != nullptr) {
      NVTE_CHECK(call.C->num_tensors == G, "Grouped GEMM: C must have num_tensors = ", G);
      NVTE_CHECK(call.C->has_data(), "Grouped GEMM: C is missing row-wise data");
      NVTE_CHECK(call.C->data.dtype == d_dtype, "Grouped GEMM: C and D must have the same dtype");
      C.base = static_cast<char *>(call.C->data.ptr);
      C.shapes = group_shapes_of(*call.C);
      has_c = 1;
    }
  } else {
    NVTE_CHECK(G <= static_cast<size_t>(kMaxListed), "Grouped GEMM: D_list supports at most ", kMaxListed,
               " tensors, got ", G);
    has_c = call.num_c > 0;
    d_dtype = lookup_checked(call.d_list[0])->data.dtype;
    for (size_t i = 0; i < G; ++i) {
      const Tensor &t = *lookup_checked(call.d_list[i]);
      NVTE_CHECK(t.has_data(), "Grouped GEMM: D_list tensor ", i, " has no data");
      NVTE_CHECK(t.data.dtype == d_dtype, "Grouped GEMM: D_list tensor ", i, " has a different dtype");
      NVTE_CHECK(t.data.shape.ndim == 2, "Grouped GEMM: D_list tensor ", i, " must be 2D.");
      lo.d[i] = static_cast<char *>(t.data.ptr);
      lo.first[i] = static_cast<unsigned>(t.data.shape.data[0]);
      lo.last[i] = static_cast<unsigned>(t.data.shape.data[1]);
      if (has_c) {
        const Tensor &c = *lookup_checked(call.c_list[i]);
        NVTE_CHECK(c.has_data(), "Grouped GEMM: C_list tensor ", i, " has no data");
        NVTE_CHECK(c.data.dtype == d_dtype, "Grouped GEMM: C_list tensor ", i, " has a different dtype");
        NVTE_CHECK(c.data.numel() == t.data.numel(), "Grouped GEMM: C_list tensor ", i, " has a different size");
        lo.c[i] = static_cast<const char *>(c.data.ptr);
      }
    }
  }
  NVTE_CHECK(is_high_precision(d_dtype), "Grouped GEMM: D must be BF16, FP16, or FP32.");

  NVTE_CHECK(ws_ptr != nullptr && ws_bytes >= grouped_setup_bytes(G), "Grouped GEMM: Insufficient setup workspace (",
             ws_bytes, " bytes, need ", grouped_setup_bytes(G), ": nvte_get_grouped_gemm_setup_workspace_size)");
  const uintptr_t ws_al = (reinterpret_cast<uintptr_t>(ws_ptr) + 255) & ~static_cast<uintptr_t>(255);
  GroupDesc *descs = reinterpret_cast<GroupDesc *>(ws_al);
  unsigned *tile_start = reinterpret_cast<unsigned *>(ws_al + round_up(G * sizeof(GroupDesc), 256));
  GroupScaleDesc *sdescs = mx ? reinterpret_cast<GroupScaleDesc *>(ws_al + round_up(G * sizeof(GroupDesc), 256) +
                                                                   group_prefix_bytes(G))
                              : nullptr;

  const DeviceInfo &di = device_info();
  unsigned cus = static_cast<unsigned>(di.num_cus);
  if (config != nullptr) {
    const int want = static_cast<const GroupedMatmulConfig *>(config)->sm_count;
    if (want > 0) cus = std::min(cus, static_cast<unsigned>(want));
  }
  hipLaunchKernelGGL(grouped_setup_kernel, dim3(1), dim3(64), 0, stream, descs, tile_start, A, B, C, D, la, lo,
                     static_cast<const float *>(alpha.data.ptr), static_cast<int>(alpha.data.numel()),
                     static_cast<const float *>(beta.data.ptr), static_cast<int>(beta.data.numel()),
                     static_cast<unsigned>(G), transa ? 1 : 0, transb ? 1 : 0,
                     static_cast<unsigned>(dtype_bits(d_dtype) / 8), has_c, sdescs, cus);
  NVTE_CHECK_HIP(hipGetLastError());

  GemmArgs g;
  std::memset(&g, 0, sizeof(g));
  g.groups = descs;
  g.group_scales = sdescs;
  g.eP_swz = (mx && call.B->with_gemm_swizzled_scales) ? 1u : 0u;
  g.eQ_swz = (mx && a_swz) ? 1u : 0u;
  g.group_tile_start = tile_start;
  g.num_groups = static_cast<unsigned>(G);
  g.d_dtype = d_dtype;
  g.grid = cus;  // persistent: the tile count is only known on the device
  (void)api;
  // as gemm_impl: an output too large to stay cached until someone reads it leaves through streaming stores
  // (members that accumulate into C read it first and keep the plain stores)
  size_t out_bytes = 0;
  if (call.D != nullptr) {
    out_bytes = call.D->data.numel() * dtype_bits(d_dtype) / 8;
  } else {
    for (size_t i = 0; i < G; ++i) out_bytes += lookup_checked(call.d_list[i])->data.numel() * dtype_bits(d_dtype) / 8;
  }
  g.nt_store = out_bytes > nt_store_bytes() ? 1u : 0u;
  g.ldP = B.ld;
  g.ldQ = A.ld;
  if (mx && (A.k_major || B.k_major)) {
    // (bit 0 / 1: P / Q K-major, bit 2 / 3: that operand's scales are compact - as in gemm_impl)
    NVTE_CHECK(A.k_major, "Grouped GEMM: MXFP8 B in the reference's columnwise layout needs A in it too (wgrad); "
               "quantize B with the transposed columnwise layout (nvte_amd_set_grouped_columnwise_transposed)");
    NVTE_CHECK(!B.k_major || A.km_compact == B.km_compact, "Grouped GEMM: MXFP8 A and B in the reference's columnwise "
               "layout must both carry compact or both GEMM-swizzled scales");
    const int km = (B.k_major ? 1 : 0) | 2 | ((B.k_major && B.km_compact) ? 4 : 0) | (A.km_compact ? 8 : 0);
    switch (km) {
      case 2: launch_by_format<true, false, true, 2>(fmt_of(b_dtype), fmt_of(a_dtype), g, stream); break;
      case 10: launch_by_format<true, false, true, 10>(fmt_of(b_dtype), fmt_of(a_dtype), g, stream); break;
      case 3: launch_by_format<true, false, true, 3>(fmt_of(b_dtype), fmt_of(a_dtype), g, stream); break;
      default: launch_by_format<true, false, true, 15>(fmt_of(b_dtype), fmt_of(a_dtype), g, stream); break;
    }
  } else if (mx) {
    launch_by_format<true, false, true>(fmt_of(b_dtype), fmt_of(a_dtype), g, stream);
  } else {
    switch ((B.k_major ? 1 : 0) | (A.k_major ? 2 : 0)) {
      case 0: launch_by_format<false, false, true>(fmt_of(b_dtype), fmt_of(a_dtype), g, stream); break;
      case 1: launch_by_format<false, false, true, 1>(fmt_of(b_dtype), fmt_of(a_dtype), g, stream); break;
      case 2: launch_by_format<false, false, true, 2>(fmt_of(b_dtype), fmt_of(a_dtype), g, stream); break;
      default: launch_by_format<false, false, true, 3>(fmt_of(b_dtype), fmt_of(a_dtype), g, stream); break;
    }
  }
}

// ---- nvte_amd_multi_gemm (amd_ext.h): a LIST of independent GEMMs in one persistent launch --------------
// dgrad and wgrad of one layer are independent and read the same dY; launched one after the other, wgrad's
// 896 tiles leave half the chip idle for its last half round (3.5 rounds of 256 workgroups).  As two members of
// one grouped launch the 512 + 896 tiles are 5.5 rounds: the cost-weighted XCD runs of finish_group_setup give
// every XCD the same number of K tiles although the two members have a different K.
constexpr int kMaxMulti = 8;
struct MultiEntry {
  const uint8_t *P, *Q;
  const float *sP, *sQ;
  void *D;
  unsigned n, m, k;
  float beta;
};
struct MultiParams {
  MultiEntry e[kMaxMulti];
};
__global__ __launch_bounds__(64) void multi_setup_kernel(GroupDesc *descs, unsigned *tile_start, const MultiParams p,
                                                         unsigned G, unsigned grid) {
  const unsigned lane = threadIdx.x;
  if (lane < G) {
    const MultiEntry &e = p.e[lane];
    GroupDesc d;
    d.P = e.P;
    d.Q = e.Q;
    d.D = e.D;
    d.C = e.beta != 0.f ? e.D : nullptr;
    d.mult = *e.sP * *e.sQ;
    d.beta = e.beta;
    d.n = e.n;
    d.m = e.m;
    d.k = e.k;
    d.tiles_j = (e.m + BN - 1) / BN;
    descs[lane] = d;
  }
  finish_group_setup(descs, tile_start, G, min(lane, G), min(lane + 1, G), grid);
}

// the K-contiguous FP8 storage of a per-tensor-scaled operand ("TN everywhere", as gemm_impl)
bool pick_tensor_scaling_operand(const Tensor &T, bool use_rowwise, Operand &op) {
  if (!is_tensor_scaling(T.scaling_mode)) return false;
  if (use_rowwise) {
    if (!T.has_data()) return false;
    op.data = static_cast<const uint8_t *>(T.data.ptr);
    op.dtype = T.data.dtype;
    op.scale_inv = T.scale_inv.ptr;
  } else {
    if (!T.has_columnwise_data()) return false;
    op.data = static_cast<const uint8_t *>(T.columnwise_data.ptr);
    op.dtype = T.columnwise_data.dtype;
    op.scale_inv = T.columnwise_scale_inv.ptr;
  }
  return is_fp8(op.dtype) && op.data != nullptr && op.scale_inv != nullptr &&
         (reinterpret_cast<uintptr_t>(op.data) & 15u) == 0;
}

void multi_gemm_impl(const NVTETensor *A, const NVTETensor *B, NVTETensor *D, const int *transa, const int *transb,
                     int num, bool grad, const Tensor *workspace, const int *accumulate, hipStream_t stream) {
  NVTE_CHECK(num >= 0, "nvte_amd_multi_gemm: negative num_gemms");
  NVTE_CHECK(num == 0 || (A && B && D && transa && transb), "nvte_amd_multi_gemm: null argument list");
  // one launch when every GEMM is a plain per-tensor-scaled FP8 GEMM of the same formats and output dtype;
  // otherwise (and for a single GEMM) the list runs as nvte_cublas_gemm calls, one after the other
  MultiParams p;
  std::memset(&p, 0, sizeof(p));
  bool fuse = num >= 2 && num <= kMaxMulti && workspace != nullptr && workspace->data.ptr != nullptr;
  int fp = -1, fq = -1;
  DType d_dtype = kNVTEFloat32;
  size_t min_out_bytes = ~static_cast<size_t>(0);
  for (int i = 0; fuse && i < num; ++i) {
    const Tensor &a = *lookup_checked(A[i]);
    const Tensor &b = *lookup_checked(B[i]);
    const Tensor &d = *lookup_checked(D[i]);
    const auto [A0, A1] = a.flat2d();
    const auto [B0, B1] = b.flat2d();
    const size_t m = transa[i] ? A0 : A1, k = transa[i] ? A1 : A0, n = transb[i] ? B1 : B0;
    Operand opA, opB;
    fuse = (transb[i] ? B0 : B1) == k && m > 0 && n > 0 && k >= 16 && k % 16 == 0 && k <= (1u << 23) &&
           n < (1ull << 31) && m < (1ull << 31) && pick_tensor_scaling_operand(a, transa[i] != 0, opA) &&
           pick_tensor_scaling_operand(b, transb[i] == 0, opB) && d.data.ptr != nullptr && d.data.numel() == m * n &&
           is_high_precision(d.data.dtype);
    if (!fuse) break;
    if (i == 0) {
      fp = fmt_of(opB.dtype);
      fq = fmt_of(opA.dtype);
      d_dtype = d.data.dtype;
    }
    fuse = fp == fmt_of(opB.dtype) && fq == fmt_of(opA.dtype) && d_dtype == d.data.dtype;
    p.e[i] = MultiEntry{opB.data, opA.data, static_cast<const float *>(opB.scale_inv),
                        static_cast<const float *>(opA.scale_inv), d.data.ptr, static_cast<unsigned>(n),
                        static_cast<unsigned>(m), static_cast<unsigned>(k), (accumulate && accumulate[i]) ? 1.f : 0.f};
    min_out_bytes = std::min(min_out_bytes, m * n * dtype_bits(d.data.dtype) / 8);
  }
  const size_t ws_bytes = workspace ? workspace->data.numel() * dtype_bits(workspace->data.dtype) / 8 : 0;
  fuse = fuse && ws_bytes >= grouped_setup_bytes(static_cast<size_t>(num));
  if (!fuse) {
    for (int i = 0; i < num; ++i)
      gemm_impl(*lookup_checked(A[i]), *lookup_checked(B[i]), *lookup_checked(D[i]), nullptr, nullptr, transa[i] != 0,
                transb[i] != 0, grad, workspace, 1.f, (accumulate && accumulate[i]) ? 1.f : 0.f, stream);
    return;
  }
  const size_t G = static_cast<size_t>(num);
  const uintptr_t ws_al = (reinterpret_cast<uintptr_t>(workspace->data.ptr) + 255) & ~static_cast<uintptr_t>(255);
  GroupDesc *descs = reinterpret_cast<GroupDesc *>(ws_al);
  unsigned *tile_start = reinterpret_cast<unsigned *>(ws_al + round_up(G * sizeof(GroupDesc), 256));
  const unsigned cus = static_cast<unsigned>(device_info().num_cus);
  hipLaunchKernelGGL(multi_setup_kernel, dim3(1), dim3(64), 0, stream, descs, tile_start, p, static_cast<unsigned>(G), cus);
  NVTE_CHECK_HIP(hipGetLastError());
  GemmArgs g;
  std::memset(&g, 0, sizeof(g));
  g.groups = descs;
  g.group_tile_start = tile_start;
  g.num_groups = static_cast<unsigned>(G);
  g.d_dtype = d_dtype;
  g.grid = cus;
  g.nt_store = min_out_bytes > nt_store_bytes() ? 1u : 0u;   // as gemm_impl: outputs nobody re-reads from cache
  launch_by_format<false, false, true>(fp, fq, g, stream);
}

// ---- static scratch for nvte_amd_grouped_gemm_noalloc (amd_ext.h) --------------------------------
// The GroupDesc records + tile prefix of ONE grouped launch of up to kNoAllocMaxGroups members live in
// a device array of this code object instead of a caller workspace.  Every GPU the library runs on
// holds its own copy (about 26 KiB) until the process ends.  The array's name is a host-side stand-in:
// host code only ever uses the per-device address obtained below.
constexpr size_t kNoAllocMaxGroups = 256;
// = what nvte_get_grouped_gemm_setup_workspace_size() asks a caller for: GroupDesc records, tile prefix
// ((G + 1) tile starts + 9 XCD run boundaries, 4 bytes each), GroupScaleDesc records, 256 to align the base
constexpr size_t kNoAllocScratchBytes = grouped_setup_bytes(kNoAllocMaxGroups);
static_assert(kNoAllocScratchBytes <= 32 * 1024, "the static scratch array is meant to stay modest");
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex scratch_mu;
void *scratch_ptr_of_device[kMaxDevices] = {};

// Device address of `scratch` on the CURRENT device (looked up once per device).  The caller's stream must
// belong to the current device, as for every other entry point of this library (kernels are launched on
// `stream` with pointers of the current device; HIP rejects a launch on another device's stream).
void *scratch_ptr() {
  int dev = 0;
  NVTE_CHECK_HIP(hipGetDevice(&dev));
  NVTE_CHECK(dev >= 0 && dev < kMaxDevices, "device ordinal out of range");
  std::lock_guard<std::mutex> g(scratch_mu);
  if (scratch_ptr_of_device[dev] == nullptr) {
    void *p = nullptr;
    NVTE_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    NVTE_CHECK(p != nullptr, "hipGetSymbolAddress returned a null address for the static scratch array");
    scratch_ptr_of_device[dev] = p;
  }
  return scratch_ptr_of_device[dev];
}

void ws_of(NVTETensor ws_setup_, void *&ptr, size_t &bytes) {
  Tensor &ws = *lookup_checked(ws_setup_);
  ptr = ws.data.ptr;
  bytes = ws.data.numel() * dtype_bits(ws.data.dtype) / 8;
}

}  // namespace
}  // namespace te

using namespace te;

extern "C" {

// No BLAS handle exists here; the call is the library's per-device set-up point: it resolves the
// address of the static scratch array on the current device (so that a later
// nvte_amd_grouped_gemm_noalloc under graph capture does not have to).
void nvte_cublas_handle_init(void) {
  api_guard("nvte_cublas_handle_init", [&] { (void)scratch_ptr(); });
}

#ifdef TE_GEMM_STAMPS
// diagnostic build only (scratch/): where the next GEMM launches write their in-kernel time stamps
void nvte_amd_debug_set_gemm_stamps(void *buf) { te::g_stamps = static_cast<unsigned long long *>(buf); }
void nvte_amd_debug_set_gemm_wait_stamps(void *buf) { te::g_wstamps = static_cast<unsigned long long *>(buf); }
#endif

size_t nvte_amd_grouped_gemm_noalloc_max_tensors(void) { return kNoAllocMaxGroups; }

void nvte_amd_grouped_gemm_noalloc(const NVTEGroupedTensor A, int transa, const NVTEGroupedTensor B, int transb,
                                   const NVTEGroupedTensor C, NVTEGroupedTensor D, const NVTETensor alpha,
                                   const NVTETensor beta, NVTEGroupedMatmulConfig config, hipStream_t stream) {
  api_guard("nvte_amd_grouped_gemm_noalloc", [&] {
    GroupedCall call;
    call.A = lookup_grouped_checked(A);
    call.B = lookup_grouped_checked(B);
    call.C = C ? lookup_grouped_checked(C) : nullptr;
    call.D = lookup_grouped_checked(D);
    const size_t G = call.A->num_tensors;
    NVTE_CHECK(G <= kNoAllocMaxGroups, "nvte_amd_grouped_gemm_noalloc: ", G, " member tensors exceed the static scratch "
               "limit of ", kNoAllocMaxGroups, "; use nvte_grouped_gemm with a caller workspace");
    static_assert(kNoAllocScratchBytes >= grouped_setup_bytes(kNoAllocMaxGroups), "scratch too small");
    // Nothing in the scratch has to start at zero: grouped_setup_kernel rewrites every record and
    // prefix entry of this call's G members before the GEMM kernel (same stream) reads them.
    grouped_gemm_impl("nvte_amd_grouped_gemm_noalloc", call, G, transa != 0, transb != 0, alpha, beta,
                      scratch_ptr(), kNoAllocScratchBytes, config, stream);
  });
}

void nvte_amd_multi_gemm(const NVTETensor *A, const NVTETensor *B, NVTETensor *D, const int *transa,
                         const int *transb, int num_gemms, int grad, NVTETensor workspace, const int *accumulate,
                         hipStream_t stream) {
  api_guard("nvte_amd_multi_gemm", [&] {
    multi_gemm_impl(A, B, D, transa, transb, num_gemms, grad != 0, lookup(workspace), accumulate, stream);
  });
}

void nvte_cublas_gemm(const NVTETensor A, const NVTETensor B, NVTETensor D, const NVTETensor bias,
                      NVTETensor pre_gelu_out, bool transa, bool transb, bool grad,
                      NVTETensor workspace, bool accumulate, bool use_split_accumulator,
                      int math_sm_count, hipStream_t stream) {
  api_guard("nvte_cublas_gemm", [&] {
    (void)use_split_accumulator;
    gemm_impl(*lookup_checked(A), *lookup_checked(B), *lookup_checked(D), lookup(bias),
              lookup(pre_gelu_out), transa, transb, grad, lookup(workspace), 1.f, accumulate ? 1.f : 0.f,
              stream, math_sm_count);
  });
}

void nvte_cublas_gemm_scaled(const NVTETensor A, const NVTETensor B, NVTETensor D,
                             const NVTETensor bias, NVTETensor pre_gelu_out, bool transa,
                             bool transb, bool grad, NVTETensor workspace, float alpha, float beta,
                             bool use_split_accumulator, int math_sm_count, hipStream_t stream) {
  api_guard("nvte_cublas_gemm_scaled", [&] {
    (void)use_split_accumulator;
    gemm_impl(*lookup_checked(A), *lookup_checked(B), *lookup_checked(D), lookup(bias),
              lookup(pre_gelu_out), transa, transb, grad, lookup(workspace), alpha, beta, stream, math_sm_count);
  });
}

void nvte_cublas_gemm_v2(int transa, int transb, const float *alpha, const NVTETensor A,
                         const NVTETensor B, const float *beta, const NVTETensor C, NVTETensor D,
                         NVTETensor workspace, NVTEMatmulConfig config, hipStream_t stream) {
  api_guard("nvte_cublas_gemm_v2", [&] {
    const Tensor &a = *lookup_checked(A);
    const Tensor &b = *lookup_checked(B);
    const Tensor *c = lookup_checked(C);
    Tensor *d = lookup_checked(D);
    NVTE_CHECK(c == d, "Currently nvte_cublas_gemm_v2 does not support different C and D tensors.");
    NVTE_CHECK(alpha != nullptr && beta != nullptr, "alpha and beta must be valid host pointers");
    MatmulConfig cfg;
    if (config != nullptr) cfg = *static_cast<const MatmulConfig *>(config);
    const bool grad_epi = cfg.dbias_tensor != nullptr || cfg.with_dgelu_epilogue;
    if (grad_epi) {
      NVTE_CHECK(cfg.bias_tensor == nullptr && !cfg.with_gelu_epilogue, "Invalid epilogue (bias=",
                 cfg.bias_tensor != nullptr, ", dbias=", cfg.dbias_tensor != nullptr,
                 ", gelu=", cfg.with_gelu_epilogue, ", dgelu=", cfg.with_dgelu_epilogue, ").");
    }
    const Tensor *bias_t = nullptr;
    if (!grad_epi && cfg.bias_tensor != nullptr) bias_t = lookup_checked(cfg.bias_tensor);
    if (grad_epi && cfg.dbias_tensor != nullptr) bias_t = lookup_checked(cfg.dbias_tensor);
    const Tensor *aux = nullptr;
    if (cfg.with_gelu_epilogue || cfg.with_dgelu_epilogue) {
      NVTE_CHECK(cfg.epilogue_aux_tensor != nullptr, "Requested epilogue without providing aux tensor.");
      aux = lookup(cfg.epilogue_aux_tensor);
    }
    gemm_impl(a, b, *d, bias_t, aux, transa != 0, transb != 0, grad_epi, lookup(workspace), *alpha, *beta,
              stream, cfg.sm_count);
  });
}

size_t nvte_get_grouped_gemm_setup_workspace_size(size_t num_tensors) { return grouped_setup_bytes(num_tensors); }

void nvte_grouped_gemm(const NVTEGroupedTensor A, int transa, const NVTEGroupedTensor B, int transb,
                       const NVTEGroupedTensor C, NVTEGroupedTensor D, const NVTETensor alpha, const NVTETensor beta,
                       NVTETensor workspace_setup, NVTETensor workspace_cublas, NVTEGroupedMatmulConfig config,
                       hipStream_t stream) {
  api_guard("nvte_grouped_gemm", [&] {
    (void)workspace_cublas;
    GroupedCall call;
    call.A = lookup_grouped_checked(A);
    call.B = lookup_grouped_checked(B);
    call.C = C ? lookup_grouped_checked(C) : nullptr;
    call.D = lookup_grouped_checked(D);
    void *wp; size_t wb;
    ws_of(workspace_setup, wp, wb);
    grouped_gemm_impl("nvte_grouped_gemm", call, call.A->num_tensors, transa != 0, transb != 0, alpha, beta,
                      wp, wb, config, stream);
  });
}

void nvte_grouped_gemm_with_discrete_inputA(const NVTETensor *A_list, size_t num_a_tensors, int transa,
                                            const NVTEGroupedTensor B, int transb, const NVTEGroupedTensor C,
                                            NVTEGroupedTensor D, const NVTETensor alpha, const NVTETensor beta,
                                            NVTETensor workspace_setup, NVTETensor workspace_cublas,
                                            NVTEGroupedMatmulConfig config, hipStream_t stream) {
  api_guard("nvte_grouped_gemm_with_discrete_inputA", [&] {
    (void)workspace_cublas;
    NVTE_CHECK(A_list != nullptr, "Grouped GEMM: A_list is null.");
    NVTE_CHECK(num_a_tensors > 0, "Grouped GEMM: num_a_tensors must be > 0.");
    GroupedCall call;
    call.a_list = A_list;
    call.B = lookup_grouped_checked(B);
    call.C = C ? lookup_grouped_checked(C) : nullptr;
    call.D = lookup_grouped_checked(D);
    NVTE_CHECK(call.B->num_tensors == num_a_tensors, "Grouped GEMM: A_list must have num_tensors (",
               call.B->num_tensors, ") entries, got ", num_a_tensors);
    void *wp; size_t wb;
    ws_of(workspace_setup, wp, wb);
    grouped_gemm_impl("nvte_grouped_gemm_with_discrete_inputA", call, num_a_tensors, transa != 0, transb != 0,
                      alpha, beta, wp, wb, config, stream);
  });
}

void nvte_grouped_gemm_with_discrete_out(const NVTEGroupedTensor A, int transa, const NVTEGroupedTensor B,
                                         int transb, const NVTETensor *C_list, size_t num_c_tensors,
                                         NVTETensor *D_list, size_t num_d_tensors, const NVTETensor alpha,
                                         const NVTETensor beta, NVTETensor workspace_setup,
                                         NVTETensor workspace_cublas, NVTEGroupedMatmulConfig config,
                                         hipStream_t stream) {
  api_guard("nvte_grouped_gemm_with_discrete_out", [&] {
    (void)workspace_cublas;
    NVTE_CHECK(D_list != nullptr, "Grouped GEMM: D_list is null.");
    NVTE_CHECK(num_d_tensors > 0, "Grouped GEMM: num_d_tensors must be > 0.");
    if (num_c_tensors > 0) NVTE_CHECK(C_list != nullptr, "Grouped GEMM: C_list is null but num_c_tensors > 0.");
    GroupedCall call;
    call.A = lookup_grouped_checked(A);
    call.B = lookup_grouped_checked(B);
    call.c_list = C_list;
    call.d_list = D_list;
    call.num_c = num_c_tensors;
    const size_t G = call.A->num_tensors;
    NVTE_CHECK(num_d_tensors == G, "Grouped GEMM: D_list must have num_tensors (", G, ") entries, got ",
               num_d_tensors);
    if (num_c_tensors > 0)
      NVTE_CHECK(num_c_tensors == G, "Grouped GEMM: C_list must have num_tensors (", G, ") entries, got ",
                 num_c_tensors);
    void *wp; size_t wb;
    ws_of(workspace_setup, wp, wb);
    grouped_gemm_impl("nvte_grouped_gemm_with_discrete_out", call, G, transa != 0, transb != 0, alpha, beta,
                      wp, wb, config, stream);
  });
}

void nvte_multi_tensor_gemm(const NVTETensor *A, const NVTETensor *B, NVTETensor *D,
                            const NVTETensor *bias, NVTETensor *pre_gelu_out, const int num_gemms,
                            bool transa, bool transb, bool grad, NVTETensor *workspace,
                            bool accumulate, bool use_split_accumulator, int math_sm_count,
                            hipStream_t stream) {
  api_guard("nvte_multi_tensor_gemm", [&] {
    (void)use_split_accumulator;
    if (num_gemms <= 0) return;
    // The reference fans the list over 4 side streams with event fork / join because one cuBLAS GEMM
    // of an expert-sized problem does not fill an H100 (ref cublaslt_gemm.cu:970-1023).  Here every
    // member is a persistent launch that already spreads over all 256 CUs (a member smaller than the
    // chip leaves CUs to the next launch: launches of one stream overlap at their tails), so by default
    // the list runs in order on the CALLER's stream: no cross-stream events on the hot path, nothing to
    // join, and capturable in a HIP graph as a plain chain.  NVTE_AMD_MULTI_STREAM=1 restores the fork /
    // join over the pool (MultiStreamScope, common.h; DESIGN 3.12 has the history of that switch).
    MultiStreamScope ms(stream, static_cast<size_t>(num_gemms));
    for (int i = 0; i < num_gemms; ++i) {
      gemm_impl(*lookup_checked(A[i]), *lookup_checked(B[i]), *lookup_checked(D[i]),
                bias ? lookup(bias[i]) : nullptr, pre_gelu_out ? lookup(pre_gelu_out[i]) : nullptr,
                transa, transb, grad, workspace ? lookup(workspace[i % nvte_get_num_compute_streams()]) : nullptr,
                1.f, accumulate ? 1.f : 0.f, ms.stream_of(static_cast<size_t>(i)), math_sm_count);
    }
    ms.join();
  });
}

}  // extern "C"

#else   // TE_GEMM_PART 0..3: kernels only
}  // namespace te
#endif  // TE_GEMM_HAS_HOST
