// C entry points for quantize / cast_transpose / dequantize and their list forms,
// plus the compute-stream pool.
//
// Replaces: ref common/cast/cast.cu:21-89, common/cast/dispatch/quantize.cuh:33-197,
//           common/transpose/cast_transpose.cu:348-363, multi_cast_transpose.cu:335,
//           common/util/multi_stream.cpp:22-67.
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "transformer_engine/cast.h"
#include "transformer_engine/cast_transpose_noop.h"
#include "transformer_engine/multi_stream.h"
#include "transformer_engine/transpose.h"

namespace te {

bool dbias_workspace_query(const Tensor &in, Tensor &workspace);                             // dbias.hip
void dbias_launch(const Tensor &in, Tensor &dbias, Tensor &workspace, hipStream_t stream);   // dbias.hip
// activation.hip: cast (+transpose) and the column sums in ONE pass, FP8 per-tensor-scaling and MXFP8 outputs
void fused_cast_dbias_launch(const Tensor &in, Tensor &out, Tensor &dbias, Tensor &workspace, hipStream_t stream);
static bool fused_dbias_ok(const Tensor &in, const Tensor &out) {
  if (!is_fp8(out.dtype()) || !is_high_precision(in.data.dtype)) return false;
  if (is_tensor_scaling(out.scaling_mode)) return out.has_data();
  // MXFP8: whole 128x128 tiles take the fused tile kernel (other shapes keep the two-pass route)
  return is_mxfp8(out.scaling_mode) && in.rows() % 128 == 0 && in.cols() % 128 == 0 && in.data.numel() > 0;
}

// The OUTPUT tensor's format picks the kernel (ref dispatch/quantize.cuh:63-93).
void quantize_dispatch(const Tensor &in, const float *noop, Tensor &out, hipStream_t stream) {
  NVTE_CHECK(out.has_data() || out.has_columnwise_data(),
             "Either rowwise or columnwise output data need to be allocated.");
  switch (out.scaling_mode) {
    case NVTE_DELAYED_TENSOR_SCALING:
      if (out.has_columnwise_data()) {
        NVTE_CHECK(out.has_data(), "Quantizing in only the columnwise direction not supported yet!");
        cast_transpose_launch(in, noop, out, stream);
      } else {
        cast_rowwise_launch(in, noop, out, stream);
      }
      break;
    case NVTE_MXFP8_1D_SCALING:
      mxfp8_quantize_launch(in, noop, out, stream);
      break;
    default:
      NVTE_ERROR("Not implemented scaling mode: ", scaling_mode_name(out.scaling_mode), ".");
  }
}

namespace {

constexpr int kNumStreams = 4;  // ref multi_stream.cpp:22
struct StreamPool {
  hipStream_t s[kNumStreams];
  hipEvent_t e[kNumStreams];
  hipEvent_t fork;  // the fork event of MultiStreamScope: never one of the join events e[]
  bool ready = false;
};
int env_int(const char *name, int dflt) {
  const char *v = std::getenv(name);
  return (v != nullptr && *v != 0) ? std::atoi(v) : dflt;
}
// one pool per device ordinal.  The pool streams have the DEFAULT priority: the reference creates them with
// priority -1 (multi_stream.cpp:35), which on ROCm puts them on a separate set of high-priority hardware queues
// whose dispatches pre-empt whatever the framework's own streams run; NVTE_AMD_POOL_PRIORITY=-1 asks for that.
StreamPool &pool() {
  static std::mutex mu;
  static StreamPool pools[64];
  int dev = 0;
  NVTE_CHECK_HIP(hipGetDevice(&dev));
  NVTE_CHECK(dev >= 0 && dev < 64, "device ordinal out of range");
  std::lock_guard<std::mutex> g(mu);
  StreamPool &p = pools[dev];
  if (!p.ready) {
    const int prio = env_int("NVTE_AMD_POOL_PRIORITY", 0);
    for (int i = 0; i < kNumStreams; ++i) {
      NVTE_CHECK_HIP(hipStreamCreateWithPriority(&p.s[i], hipStreamNonBlocking, prio));
      NVTE_CHECK_HIP(hipEventCreateWithFlags(&p.e[i], hipEventDisableTiming));
    }
    NVTE_CHECK_HIP(hipEventCreateWithFlags(&p.fork, hipEventDisableTiming));
    p.ready = true;
  }
  return p;
}

}  // namespace

// NVTE_AMD_MULTI_STREAM: 0 (default) = list entry points run in order on the caller's stream;
// 1 = fork / join over the pool with a dedicated fork event; 2 = the shape of the reference
// (cublaslt_gemm.cu:994-1045, cast.cu:65-88): join event 0 doubles as the fork event.
MultiStreamScope::MultiStreamScope(hipStream_t caller, size_t num_items) : caller_(caller) {
  mode_ = env_int("NVTE_AMD_MULTI_STREAM", 0);
  if (mode_ == 0 || num_items == 0) { mode_ = 0; return; }
  StreamPool &p = pool();
  used_ = static_cast<int>(std::min<size_t>(num_items, kNumStreams));
  hipEvent_t fork = mode_ == 2 ? p.e[0] : p.fork;
  NVTE_CHECK_HIP(hipEventRecord(fork, caller_));
  for (int s = 0; s < used_; ++s) NVTE_CHECK_HIP(hipStreamWaitEvent(p.s[s], fork, 0));
}
hipStream_t MultiStreamScope::stream_of(size_t item) const {
  return mode_ == 0 ? caller_ : pool().s[item % kNumStreams];
}
void MultiStreamScope::join() {
  if (mode_ == 0 || joined_) return;
  joined_ = true;
  StreamPool &p = pool();
  for (int s = 0; s < used_; ++s) {
    NVTE_CHECK_HIP(hipEventRecord(p.e[s], p.s[s]));
    NVTE_CHECK_HIP(hipStreamWaitEvent(caller_, p.e[s], 0));
  }
}

namespace {

void with_config(const NVTETensor input, NVTETensor output, const QuantConfig *cfg,
                 hipStream_t stream) {
  const Tensor &in = *lookup_checked(input);
  Tensor &out = *lookup_checked(output);
  if (cfg != nullptr && cfg->stochastic_rounding) {
    NVTE_CHECK(out.scaling_mode == NVTE_NVFP4_1D_SCALING,
               "Stochastic rounding is only supported for NVFP4 quantization.");
  }
  quantize_dispatch(in, noop_ptr_of(cfg), out, stream);
}

}  // namespace
}  // namespace te

using namespace te;

extern "C" {

void nvte_quantize(const NVTETensor input, NVTETensor output, hipStream_t stream) {
  api_guard("nvte_quantize", [&] { with_config(input, output, nullptr, stream); });
}

void nvte_quantize_noop(const NVTETensor input, NVTETensor output, NVTETensor noop,
                        hipStream_t stream) {
  api_guard("nvte_quantize_noop", [&] {
    QuantConfig cfg;
    cfg.noop_tensor = noop;
    with_config(input, output, &cfg, stream);
  });
}

void nvte_quantize_v2(const NVTETensor input, NVTETensor output,
                      const NVTEQuantizationConfig quant_config, hipStream_t stream) {
  api_guard("nvte_quantize_v2", [&] {
    with_config(input, output, static_cast<const QuantConfig *>(quant_config), stream);
  });
}

void nvte_quantize_dbias(const NVTETensor input, NVTETensor output, NVTETensor dbias,
                         NVTETensor workspace, hipStream_t stream) {
  api_guard("nvte_quantize_dbias", [&] {
    const Tensor &in = *lookup_checked(input);
    Tensor &out = *lookup_checked(output);
    Tensor &db = *lookup_checked(dbias);
    Tensor &ws = *lookup_checked(workspace);
    check_input(in, "cast_input");
    if (fused_dbias_ok(in, out)) {
      fused_cast_dbias_launch(in, out, db, ws, stream);
      return;
    }
    if (dbias_workspace_query(in, ws)) return;
    quantize_dispatch(in, nullptr, out, stream);
    dbias_launch(in, db, ws, stream);
  });
}

void nvte_dequantize(const NVTETensor input, NVTETensor output, hipStream_t stream) {
  api_guard("nvte_dequantize",
            [&] { dequantize_launch(*lookup_checked(input), *lookup_checked(output), stream); });
}

void nvte_multi_tensor_quantize(const NVTETensor *inputs, NVTETensor *outputs,
                                const NVTEQuantizationConfig quant_config, const size_t num_tensors,
                                hipStream_t stream) {
  api_guard("nvte_multi_tensor_quantize", [&] {
    // In order on the caller's stream by default (the reference forks over 4 side streams, ref cast.cu:65-88):
    // every quantize kernel here is a full-chip HBM-bound launch, so side streams add event traffic and
    // nothing else; NVTE_AMD_MULTI_STREAM=1 restores the fork / join (common.h MultiStreamScope).
    MultiStreamScope ms(stream, num_tensors);
    for (size_t i = 0; i < num_tensors; ++i)
      with_config(inputs[i], outputs[i], static_cast<const QuantConfig *>(quant_config), ms.stream_of(i));
    ms.join();
  });
}

void nvte_cast_transpose(const NVTETensor input, NVTETensor output, hipStream_t stream) {
  api_guard("nvte_cast_transpose", [&] {
    cast_transpose_launch(*lookup_checked(input), nullptr, *lookup_checked(output), stream);
  });
}

void nvte_cast_transpose_with_noop(const NVTETensor input, const NVTETensor noop, NVTETensor output,
                                   hipStream_t stream) {
  api_guard("nvte_cast_transpose_with_noop", [&] {
    const Tensor &n = *lookup_checked(noop);
    check_noop(n, "cast_transpose_noop");
    cast_transpose_launch(*lookup_checked(input), static_cast<const float *>(n.data.ptr),
                          *lookup_checked(output), stream);
  });
}

void nvte_cast_transpose_dbias(const NVTETensor input, NVTETensor output, NVTETensor dbias,
                               NVTETensor workspace, hipStream_t stream) {
  api_guard("nvte_cast_transpose_dbias", [&] {
    const Tensor &in = *lookup_checked(input);
    Tensor &out = *lookup_checked(output);
    Tensor &db = *lookup_checked(dbias);
    Tensor &ws = *lookup_checked(workspace);
    check_input(in, "cast_transpose_input");
    if (fused_dbias_ok(in, out)) {
      fused_cast_dbias_launch(in, out, db, ws, stream);
      return;
    }
    if (dbias_workspace_query(in, ws)) return;
    cast_transpose_launch(in, nullptr, out, stream);
    dbias_launch(in, db, ws, stream);
  });
}

void nvte_multi_cast_transpose(size_t num_tensors, const NVTETensor *input_list,
                               NVTETensor *output_list, hipStream_t stream) {
  api_guard("nvte_multi_cast_transpose", [&] {
    for (size_t i = 0; i < num_tensors; ++i) {
      cast_transpose_launch(*lookup_checked(input_list[i]), nullptr, *lookup_checked(output_list[i]),
                            stream);
    }
  });
}

int nvte_get_num_compute_streams(void) {
  api_enter("nvte_get_num_compute_streams");
  return kNumStreams;
}

hipStream_t nvte_get_compute_stream(const int idx) {
  return api_guard("nvte_get_compute_stream", [&]() -> hipStream_t {
    NVTE_CHECK(0 <= idx && idx < kNumStreams, "Invalid compute stream (requested idx ", idx,
               ", but there are ", kNumStreams, " streams)");
    return pool().s[idx];
  });
}

hipEvent_t nvte_get_compute_stream_event(const int idx) {
  return api_guard("nvte_get_compute_stream_event", [&]() -> hipEvent_t {
    NVTE_CHECK(0 <= idx && idx < kNumStreams, "Invalid compute stream (requested idx ", idx,
               ", but there are ", kNumStreams, " streams)");
    return pool().e[idx];
  });
}

}  // extern "C"
