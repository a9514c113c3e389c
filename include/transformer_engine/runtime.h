/* C++ runtime helpers the reference's framework shims link besides nvte_* (ref
 * common/libtransformer_engine.version:5-17; declared there in common/util/cuda_runtime.h:22-95 and
 * common/common.h:1160-1161).  The namespace keeps the name `cuda` because callers bind these symbols by
 * their mangled names; everything behind them is HIP on gfx950.
 */
#ifndef TE_AMD_RUNTIME_H_
#define TE_AMD_RUNTIME_H_

#include "transformer_engine.h"

#ifdef __cplusplus
namespace transformer_engine {

/* ref common/common.h:1160-1161: storage size of one element (typeToSize: whole bytes, FP4 is an error;
 * typeToNumBits: bits) */
size_t typeToSize(const DType type);
size_t typeToNumBits(const DType type);

namespace cuda {
/* ref cuda_runtime.h:22: ordinal of the current HIP device */
int current_device();
/* ref cuda_runtime.h:31: "compute capability" as 10 * major + minor.  gfx950 reports 95: above every
 * threshold the reference's shims test for FP8 / MXFP8 support (89, 90, 100), which this library
 * implements; the NVFP4 / FP8-blockwise recipes those thresholds also enable raise from nvte_quantize. */
int sm_arch(int device_id = -1);
/* ref cuda_runtime.h:39: number of compute units */
int sm_count(int device_id = -1);
/* ref cuda_runtime.h:57: no multicast objects on xGMI */
bool supports_multicast(int device_id = -1);
/* ref cuda_runtime.h:75: [lowest, highest] stream priority */
void stream_priority_range(int *low_priority, int *high_priority, int device_id = -1);
/* ref cuda_runtime.h:89: HIP runtime version (hipRuntimeGetVersion) */
int cudart_version();
/* ref cuda_runtime.h:95: no BLAS library is linked; returns 0 */
size_t cublas_version();
}  // namespace cuda
}  // namespace transformer_engine
#endif
#endif
