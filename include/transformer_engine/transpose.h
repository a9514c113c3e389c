/* Cast+transpose and transpose entry points.
 * Replaces reference transformer_engine/common/include/transformer_engine/transpose.h
 * (definitions: common/transpose/cast_transpose.cu:348, transpose.cu:292,
 * multi_cast_transpose.cu:335, cast_transpose_fusion.cu:1260).
 */
#ifndef TE_AMD_TRANSPOSE_H_
#define TE_AMD_TRANSPOSE_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref transpose.h:32 */
void nvte_cast_transpose(const NVTETensor input, NVTETensor output, hipStream_t stream);
/* ref transpose.h:40 */
void nvte_transpose(const NVTETensor input, NVTETensor transposed_output, hipStream_t stream);
/* ref transpose.h:60 */
void nvte_cast_transpose_dbias(const NVTETensor input, NVTETensor output, NVTETensor dbias,
                               NVTETensor workspace, hipStream_t stream);
/* ref transpose.h:93 */
void nvte_multi_cast_transpose(size_t num_tensors, const NVTETensor *input_list,
                               NVTETensor *output_list, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
