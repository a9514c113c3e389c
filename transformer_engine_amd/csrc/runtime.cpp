// C++ runtime helpers exported next to the C ABI (include/transformer_engine/runtime.h).
//
// Replaces: ref common/util/cuda_runtime.cpp (current_device, sm_arch, sm_count, supports_multicast,
//           stream_priority_range, cudart_version, cublas_version) and common/common.h:1160-1161
//           (typeToSize / typeToNumBits) - the transformer_engine:: symbols of the reference's version
//           script (common/libtransformer_engine.version:5-17) that its framework shims link.
#include <mutex>
#include <vector>

#include "common.h"
#include "transformer_engine/runtime.h"

namespace transformer_engine {

size_t typeToNumBits(const DType type) { return te::dtype_bits(static_cast<NVTEDType>(type)); }

size_t typeToSize(const DType type) {
  NVTE_CHECK(type != DType::kFloat4E2M1, "typeToSize() Does not support FP4 data type.");
  return typeToNumBits(type) / 8;
}

namespace cuda {

namespace {
int resolve(int device_id) {
  if (device_id < 0) return current_device();
  int n = 0;
  NVTE_CHECK_HIP(hipGetDeviceCount(&n));
  NVTE_CHECK(device_id < n, "invalid HIP device ID (", device_id, " of ", n, ")");
  return device_id;
}
const hipDeviceProp_t &props(int device_id) {
  static std::mutex mu;
  static std::vector<hipDeviceProp_t> cache;
  static std::vector<char> have;
  std::lock_guard<std::mutex> g(mu);
  if (cache.size() <= static_cast<size_t>(device_id)) {
    cache.resize(device_id + 1);
    have.resize(device_id + 1, 0);
  }
  if (!have[device_id]) {
    NVTE_CHECK_HIP(hipGetDeviceProperties(&cache[device_id], device_id));
    have[device_id] = 1;
  }
  return cache[device_id];
}
}  // namespace

int current_device() {
  int dev = 0;
  NVTE_CHECK_HIP(hipGetDevice(&dev));
  return dev;
}

int sm_arch(int device_id) {
  const hipDeviceProp_t &p = props(resolve(device_id));
  return 10 * p.major + p.minor;   // gfx950: 9.5 -> 95
}

int sm_count(int device_id) { return props(resolve(device_id)).multiProcessorCount; }

bool supports_multicast(int) { return false; }

void stream_priority_range(int *low_priority, int *high_priority, int device_id) {
  const int dev = resolve(device_id), prev = current_device();
  if (dev != prev) NVTE_CHECK_HIP(hipSetDevice(dev));
  int lo = 0, hi = 0;
  const hipError_t err = hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (dev != prev) NVTE_CHECK_HIP(hipSetDevice(prev));
  NVTE_CHECK_HIP(err);
  if (low_priority != nullptr) *low_priority = lo;
  if (high_priority != nullptr) *high_priority = hi;
}

int cudart_version() {
  int v = 0;
  NVTE_CHECK_HIP(hipRuntimeGetVersion(&v));
  return v;
}

size_t cublas_version() { return 0; }

}  // namespace cuda
}  // namespace transformer_engine
