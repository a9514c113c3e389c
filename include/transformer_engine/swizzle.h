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

/* ref swizzle.h:118,134.  Every member of a SAME-SHAPE MXFP8 grouped tensor: scale matrices padded to (128, 4)
 * row-wise / (4, 128) column-wise, stored back to back per member; output's with_gemm_swizzled_scales flag
 * must say the layout asked for.  Column-wise matrices in the transposed layout of amd_ext.h are handled too. */
void nvte_swizzle_grouped_scaling_factors(const NVTEGroupedTensor input, NVTEGroupedTensor output, hipStream_t stream);
void nvte_unswizzle_grouped_scaling_factors(const NVTEGroupedTensor input, NVTEGroupedTensor output, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
