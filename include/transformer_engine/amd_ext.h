/* Extensions with NO counterpart in the reference.  They exist because this
 * library is also bound from languages that cannot catch C++ exceptions or build
 * std::vector arguments (ctypes, cgo, JNI ...).
 */
#ifndef TE_AMD_EXT_H_
#define TE_AMD_EXT_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Error mode.  0 (default) = reference behaviour: a failed check throws
 * std::runtime_error through the C entry point (ref common/util/logging.h:38-51).
 * 1 = "record": the entry point catches the exception, stores its text in a
 * thread-local slot and returns normally; the caller polls nvte_amd_last_error(). */
void nvte_amd_set_error_mode(int mode);
int nvte_amd_get_error_mode(void);
/* Text of the last recorded error on this thread, or NULL.  Valid until the next
 * nvte_* call on the same thread. */
const char *nvte_amd_last_error(void);
void nvte_amd_clear_error(void);

/* Plain-C twin of nvte_delayed_scaling_recipe_amax_and_scale_update_after_reduction
 * (ref recipe.h:71-74, whose std::vector arguments make the symbol C++-mangled). */
void nvte_amd_delayed_scaling_after_reduction(const NVTETensor amax_reduction_buffer,
                                              const NVTETensor *amax_histories,
                                              const NVTETensor *scales, size_t num_tensors,
                                              const char *amax_compute_algo,
                                              enum NVTEDType fp8_dtype, float margin,
                                              hipStream_t stream);

/* Library identification: "transformerengine_amd <version> gfx950". */
const char *nvte_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif
