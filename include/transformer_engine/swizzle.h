/* Replaces reference .../include/transformer_engine/swizzle.h
 * (definition: common/swizzle/swizzle.cu:1957).  Converts compact MXFP8 e8m0
 * scale matrices into the "GEMM-ready" layout.  The reference's layout is
 * cuBLAS' 128x4 interleave (ref common/cast/mxfp8/swizzle.cuh:30-41); this
 * library produces exactly that byte permutation, and its own GEMM consumes it.
 */
#ifndef TE_AMD_SWIZZLE_H_
#define TE_AMD_SWIZZLE_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref swizzle.h:31 */
void nvte_swizzle_scaling_factors(const NVTETensor input, NVTETensor output, hipStream_t stream);
/* ref swizzle.h:45 */
void nvte_multi_tensor_swizzle_scaling_factors(const NVTETensor *inputs, NVTETensor *outputs,
                                               const size_t num_tensors, hipStream_t stream);
/* ref swizzle.h:70 */
void nvte_unswizzle_scaling_factors(const NVTETensor input, NVTETensor output, hipStream_t stream);
/* ref swizzle.h:84 */
void nvte_multi_tensor_unswizzle_scaling_factors(const NVTETensor *inputs, NVTETensor *outputs,
                                                 const size_t num_tensors, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
