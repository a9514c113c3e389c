/* Replaces reference .../include/transformer_engine/cast_transpose_noop.h
 * (definitions: common/transpose/transpose.cu:299, cast_transpose.cu:356). */
#ifndef TE_AMD_CAST_TRANSPOSE_NOOP_H_
#define TE_AMD_CAST_TRANSPOSE_NOOP_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref cast_transpose_noop.h:27 */
void nvte_transpose_with_noop(const NVTETensor input, const NVTETensor noop, NVTETensor output,
                              hipStream_t stream);
/* ref cast_transpose_noop.h:37 */
void nvte_cast_transpose_with_noop(const NVTETensor input, const NVTETensor noop, NVTETensor output,
                                   hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
