// A caller written against the reference's C++ API (transformer_engine::TensorWrapper,
// QuantizationConfigWrapper, MatmulConfigWrapper, transformer_engine::cuda::sm_count ...), compiled with
// plain g++ against include/ and linked with libtransformer_engine.so: what INTEGRATION.md section 1 asks
// a maintainer of the reference's framework shim to do (ref pytorch/csrc/extensions/cast.cpp:52,
// gemm.cpp:144).  Without a GPU it exercises the host side only and prints "HOST-ONLY OK"; with one it
// quantizes, dequantizes and multiplies small-integer matrices and checks them exactly ("GPU OK").
#include <transformer_engine/cast.h>
#include <transformer_engine/gemm.h>
#include <transformer_engine/recipe.h>
#include <transformer_engine/runtime.h>
#include <transformer_engine/transformer_engine.h>
#include <transformer_engine/transpose.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

using namespace transformer_engine;

#define REQUIRE(cond)                                                        \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                              \
    }                                                                        \
  } while (0)
#define HIP_OK(expr) REQUIRE((expr) == hipSuccess)

static uint16_t bf16_of(float f) {  // exact for the small integers used here
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return static_cast<uint16_t>(u >> 16);
}

int main() {
  // ---------------- host side: handles, wrappers, getters (no device needed)
  {
    float dummy[4];
    TensorWrapper t(dummy, std::vector<size_t>{2, 3, 4}, DType::kBFloat16);
    REQUIRE(t.ndim() == 3 && t.numel() == 24 && t.size(1) == 3 && t.dtype() == DType::kBFloat16);
    REQUIRE(t.dptr() == dummy && t.scale_inv() == nullptr && t.element_size_bits() == 16 && t.bytes() == 48);
    REQUIRE(t.scaling_mode() == NVTE_DELAYED_TENSOR_SCALING);
    t.set_columnwise_data(dummy, DType::kFloat8E4M3, std::vector<size_t>{12, 2});
    REQUIRE(t.columnwise_dptr() == dummy && t.get_columnwise_data().shape.data[0] == 12);
    TensorWrapper moved(std::move(t));
    REQUIRE(t.data() == nullptr && t.numel() == 0 && moved.numel() == 24);
    TensorWrapper mx(NVTE_MXFP8_1D_SCALING);
    mx.set_with_gemm_swizzled_scales(true);
    REQUIRE(mx.get_with_gemm_swizzled_scales() && mx.scaling_mode() == NVTE_MXFP8_1D_SCALING);
    MultiTensorWrapper many(5);
    REQUIRE(many.size() == 5 && many[4] != nullptr);
    GroupedTensorWrapper g(3, std::vector<size_t>{96, 64});
    REQUIRE(g.num_tensors() == 3 && g.logical_shape().data[0] == 96);
    QuantizationConfigWrapper qc;
    qc.set_force_pow_2_scales(true);
    qc.set_amax_epsilon(0.25f);
    float eps = 0.f;
    nvte_get_quantization_config_attribute(qc, kNVTEQuantizationConfigAmaxEpsilon, &eps, sizeof(eps), nullptr);
    REQUIRE(eps == 0.25f);
    MatmulConfigWrapper mc;
    mc.set_use_split_accumulator(true);
    mc.set_sm_count(17);
    int smc = 0;
    nvte_get_matmul_config_attribute(mc, kNVTEMatmulConfigSMCount, &smc, sizeof(smc), nullptr);
    REQUIRE(smc == 17);
    GroupedMatmulConfigWrapper gmc;
    gmc.set_sm_count(10);
    REQUIRE(typeToSize(DType::kBFloat16) == 2 && typeToNumBits(DType::kFloat4E2M1) == 4);
    REQUIRE(is_fp8_dtype(DType::kFloat8E5M2) && !is_fp8_dtype(DType::kFloat8E8M0));
    REQUIRE(nvte_is_non_tn_fp8_gemm_supported() == 0);
    // errors are C++ exceptions, as in the reference (ref common/util/logging.h:38-51)
    bool threw = false;
    try {
      TensorWrapper in(dummy, std::vector<size_t>{4}, DType::kFloat32), out;
      nvte_quantize(in.data(), out.data(), nullptr);
    } catch (const std::exception &e) {
      threw = std::strstr(e.what(), "output data need to be allocated") != nullptr;
    }
    REQUIRE(threw);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::puts("HOST-ONLY OK");
    return 0;
  }

  // ---------------- device side
  REQUIRE(cuda::sm_count() >= 1 && cuda::sm_arch() == 95 && cuda::current_device() == 0);
  REQUIRE(!cuda::supports_multicast() && cuda::cublas_version() == 0);
  const size_t M = 256, K = 128, N = 384;
  std::vector<float> xf(M * K), wf(N * K);
  std::vector<uint16_t> xb(M * K), wb(N * K);
  for (size_t i = 0; i < M * K; ++i) { xf[i] = static_cast<float>(static_cast<int>((i * 7 + i / K) % 7) - 3); xb[i] = bf16_of(xf[i]); }
  for (size_t i = 0; i < N * K; ++i) { wf[i] = static_cast<float>(static_cast<int>((i * 5 + i / K) % 5) - 2); wb[i] = bf16_of(wf[i]); }
  void *dx, *dw, *x8, *x8t, *w8, *w8t, *dy, *dxq;
  float *meta;  // scale_x, amax_x, sinv_x, scale_w, amax_w, sinv_w
  HIP_OK(hipMalloc(&dx, M * K * 2)); HIP_OK(hipMalloc(&dw, N * K * 2));
  HIP_OK(hipMalloc(&x8, M * K)); HIP_OK(hipMalloc(&x8t, M * K)); HIP_OK(hipMalloc(&w8, N * K)); HIP_OK(hipMalloc(&w8t, N * K));
  HIP_OK(hipMalloc(&dy, M * N * 4)); HIP_OK(hipMalloc(&dxq, M * K * 4));
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&meta), 6 * sizeof(float)));
  const float init[6] = {2.f, 0.f, 0.f, 4.f, 0.f, 0.f};
  HIP_OK(hipMemcpy(meta, init, sizeof(init), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dx, xb.data(), M * K * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dw, wb.data(), N * K * 2, hipMemcpyHostToDevice));
  hipStream_t stream = nullptr;
  try {
    const std::vector<size_t> one{1};
    TensorWrapper tx(dx, std::vector<size_t>{M, K}, DType::kBFloat16), tw(dw, std::vector<size_t>{N, K}, DType::kBFloat16);
    // the pointer wiring of the reference's Float8Quantizer (ref pytorch/csrc/quantizer.cpp:336-432, :613-625)
    TensorWrapper qx(NVTE_DELAYED_TENSOR_SCALING), qw(NVTE_DELAYED_TENSOR_SCALING);
    qx.set_rowwise_data(x8, DType::kFloat8E4M3, std::vector<size_t>{M, K}).set_columnwise_data(x8t, DType::kFloat8E4M3, std::vector<size_t>{K, M});
    qx.set_scale(meta + 0, DType::kFloat32, one).set_amax(meta + 1, DType::kFloat32, one);
    qx.set_rowwise_scale_inv(meta + 2, DType::kFloat32, one).set_columnwise_scale_inv(meta + 2, DType::kFloat32, one);
    qw.set_rowwise_data(w8, DType::kFloat8E4M3, std::vector<size_t>{N, K}).set_columnwise_data(w8t, DType::kFloat8E4M3, std::vector<size_t>{K, N});
    qw.set_scale(meta + 3, DType::kFloat32, one).set_amax(meta + 4, DType::kFloat32, one);
    qw.set_rowwise_scale_inv(meta + 5, DType::kFloat32, one).set_columnwise_scale_inv(meta + 5, DType::kFloat32, one);
    QuantizationConfigWrapper qc;
    nvte_quantize_v2(tx.data(), qx.data(), qc, stream);
    nvte_quantize_v2(tw.data(), qw.data(), qc, stream);
    // dequantize X back and multiply: Y[M, N] = X W^T  (general_gemm(W, X, "TN"), ref gemm.cpp:144)
    TensorWrapper tdq(dxq, std::vector<size_t>{M, K}, DType::kFloat32), ty(dy, std::vector<size_t>{M, N}, DType::kFloat32), ws;
    nvte_dequantize(qx.data(), tdq.data(), stream);
    MatmulConfigWrapper mc;
    const float alpha = 1.f, beta = 0.f;
    nvte_cublas_gemm_v2(1, 0, &alpha, qw.data(), qx.data(), &beta, ty.data(), ty.data(), ws.data(), mc, stream);
    HIP_OK(hipStreamSynchronize(stream));
  } catch (const std::exception &e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 1;
  }
  std::vector<float> got_meta(6), got_dq(M * K), got_y(M * N);
  std::vector<uint8_t> hx8(M * K), hx8t(M * K);
  HIP_OK(hipMemcpy(got_meta.data(), meta, 6 * sizeof(float), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(got_dq.data(), dxq, M * K * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(got_y.data(), dy, M * N * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hx8.data(), x8, M * K, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hx8t.data(), x8t, M * K, hipMemcpyDeviceToHost));
  REQUIRE(got_meta[1] == 3.f && got_meta[2] == 0.5f && got_meta[4] == 2.f && got_meta[5] == 0.25f);  // amax, 1/scale
  for (size_t i = 0; i < M * K; ++i) REQUIRE(got_dq[i] == xf[i]);                                    // exact round trip
  for (size_t r = 0; r < M; ++r)
    for (size_t c = 0; c < K; ++c) REQUIRE(hx8t[c * M + r] == hx8[r * K + c]);                       // transposed copy
  for (size_t i = 0; i < M; i += 37)
    for (size_t j = 0; j < N; j += 11) {
      float ref = 0.f;
      for (size_t k = 0; k < K; ++k) ref += xf[i * K + k] * wf[j * K + k];
      REQUIRE(got_y[i * N + j] == ref);                                                              // exact: small integers
    }
  // ---------------- the C++-signature recipe entry point (std::vector by value, mangled symbol; ref recipe.h:71-74,
  // delayed_scaling.cu:162-253), called as pytorch/csrc/extensions/recipe.cpp does: packed reduction buffer,
  // one [H, S] history and one [S] scale per layer.  Values are powers of two, so the expectations are exact.
  {
    //                     layer A: H = 4, S = 2                       layer B: H = 4, S = 1
    const float histA[8] = {0.f, 0.f, 2.f, 1.f, 4.f, 0.5f, 1.f, 8.f}, histB[4] = {0.f, 32.f, 0.25f, 64.f};
    const float scaleA[2] = {5.f, 3.f}, scaleB[1] = {9.f};
    const float reduced[3] = {16.f, 0.f, 7.f};   // A/0 has a new amax, A/1 none (entry 0: no roll), B/0 = 7
    float *dhA, *dhB, *dsA, *dsB, *dred;
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&dhA), sizeof(histA))); HIP_OK(hipMalloc(reinterpret_cast<void **>(&dhB), sizeof(histB)));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&dsA), sizeof(scaleA))); HIP_OK(hipMalloc(reinterpret_cast<void **>(&dsB), sizeof(scaleB)));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&dred), sizeof(reduced)));
    HIP_OK(hipMemcpy(dhA, histA, sizeof(histA), hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dhB, histB, sizeof(histB), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dsA, scaleA, sizeof(scaleA), hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dsB, scaleB, sizeof(scaleB), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dred, reduced, sizeof(reduced), hipMemcpyHostToDevice));
    try {
      TensorWrapper thA(dhA, std::vector<size_t>{4, 2}, DType::kFloat32), thB(dhB, std::vector<size_t>{4, 1}, DType::kFloat32);
      TensorWrapper tsA(dsA, std::vector<size_t>{2}, DType::kFloat32), tsB(dsB, std::vector<size_t>{1}, DType::kFloat32);
      TensorWrapper tred(dred, std::vector<size_t>{3}, DType::kFloat32);
      nvte_delayed_scaling_recipe_amax_and_scale_update_after_reduction(
          tred.data(), std::vector<NVTETensor>{thA.data(), thB.data()}, std::vector<NVTETensor>{tsA.data(), tsB.data()},
          "max", static_cast<NVTEDType>(DType::kFloat8E4M3), 0.f, stream);
      HIP_OK(hipStreamSynchronize(stream));
    } catch (const std::exception &e) {
      std::fprintf(stderr, "exception: %s\n", e.what());
      return 1;
    }
    float gA[8], gB[4], gsA[2], gsB[1];
    HIP_OK(hipMemcpy(gA, dhA, sizeof(gA), hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(gB, dhB, sizeof(gB), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(gsA, dsA, sizeof(gsA), hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(gsB, dsB, sizeof(gsB), hipMemcpyDeviceToHost));
    // A/0: last = 16, amax = max(2, 4, 1, 16) = 16 -> scale 448 / 16, history rolled {0, 4, 1, 16}
    REQUIRE(gsA[0] == 28.f && gA[0] == 0.f && gA[2] == 4.f && gA[4] == 1.f && gA[6] == 16.f);
    // A/1: reduced entry 0 and history row 0 is 0 -> no roll, amax 0 -> the scale stays
    REQUIRE(gsA[1] == 3.f && gA[1] == 0.f && gA[3] == 1.f && gA[5] == 0.5f && gA[7] == 8.f);
    // B/0: last = 7, amax = max(32, 0.25, 64, 7) = 64 -> scale 7, history {0, 0.25, 64, 7}
    REQUIRE(gsB[0] == 7.f && gB[0] == 0.f && gB[1] == 0.25f && gB[2] == 64.f && gB[3] == 7.f);
  }
  std::puts("GPU OK");
  return 0;
}
