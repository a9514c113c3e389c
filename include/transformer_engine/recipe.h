/* Recipe step: amax-history roll + scale update (delayed scaling) and
 * amax / scale-from-amax (current scaling).
 * Replaces reference .../include/transformer_engine/recipe.h
 * (definitions: common/recipe/delayed_scaling.cu:393-420, current_scaling.cu:247-336).
 */
#ifndef TE_AMD_RECIPE_H_
#define TE_AMD_RECIPE_H_

#include "transformer_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ref recipe.h:41.  amax_history fp32 [H, S]; scale fp32 [S]; may run in place.
 * new_hist[i] = hist[i+1] (i < H-1), new_hist[H-1] = hist[0], then new_hist[0] = 0;
 * amax = max over the rolled window ("max") or hist[0] ("most_recent");
 * scale = fp8_max * 2^-margin / amax when amax is finite and > 0, else unchanged;
 * an infinite quotient becomes FLT_MAX. */
void nvte_delayed_scaling_recipe_amax_and_scale_update(
    const NVTETensor amax_history, const NVTETensor scale, NVTETensor updated_amax_history,
    NVTETensor updated_scale, const char *amax_compute_algo, enum NVTEDType fp8_dtype, float margin,
    hipStream_t stream);

/* ref recipe.h:85 */
void nvte_compute_amax(const NVTETensor input, NVTETensor output, hipStream_t stream);
/* ref recipe.h:99 */
void nvte_compute_amax_with_config(const NVTETensor input, NVTETensor output,
                                   const NVTEQuantizationConfig config, hipStream_t stream);
/* ref recipe.h:125 */
void nvte_compute_scale_from_amax(NVTETensor output, const NVTEQuantizationConfig config,
                                  hipStream_t stream);

/* ref recipe.h:113.  One amax per member tensor of `input` into output's amax buffer (zeroed
 * first); member shapes come from OUTPUT's device-side metadata. */
void nvte_group_compute_amax_with_config(const NVTEGroupedTensor input, NVTEGroupedTensor output,
                                         const NVTEQuantizationConfig config, hipStream_t stream);
/* ref recipe.h:139.  scale[t] = fp8_max / amax[t] (epsilon / pow2 options), scale_inv[t] = 1 / scale[t]. */
void nvte_group_compute_scale_from_amax(NVTEGroupedTensor output, const NVTEQuantizationConfig config,
                                        hipStream_t stream);

/* ref recipe.h:298 (common/recipe/mxfp8_scaling.cu:233).  `input` is the 1-D shard
 * [start_offset, start_offset + len) of a row-major [rows, cols] master weight (rows, cols
 * multiples of 32).  amax_rowwise [pad128(rows), pad4(cols/32)] and amax_colwise
 * [pad4(rows/32), pad128(cols)], both of the input dtype, receive the maximum |x| of every 1x32 /
 * 32x1 block over the elements the shard holds (0 for blocks it does not touch), ready for a MAX
 * all-reduce over the data-parallel group. */
void nvte_mxfp8_scaling_compute_partial_amax(const NVTETensor input, NVTETensor amax_rowwise,
                                             NVTETensor amax_colwise, int rows, int cols,
                                             size_t start_offset, hipStream_t stream);
/* ref recipe.h:318 (mxfp8_scaling.cu:243).  Casts the shard to E4M3 twice, with the rowwise and
 * the columnwise e8m0 decode scales of the FULL matrix (out = x * 2^(127 - e), saturating);
 * outputs are 1-D byte shards of the same length as `input`. */
void nvte_mxfp8_scaling_partial_cast(const NVTETensor input, NVTETensor output_rowwise,
                                     NVTETensor output_colwise, const NVTETensor scale_inv_rowwise,
                                     const NVTETensor scale_inv_colwise, int rows, int cols,
                                     size_t start_offset, hipStream_t stream);

#ifdef __cplusplus
} /* extern "C" */

#include <vector>
/* ref recipe.h:71.  NOTE: like the reference this one has a C++ signature
 * (std::vector by value) even though it sits beside the C functions; the symbol
 * is therefore C++-mangled.  The plain-C twin for FFI callers is
 * nvte_amd_delayed_scaling_after_reduction() in amd_ext.h. */
void nvte_delayed_scaling_recipe_amax_and_scale_update_after_reduction(
    const NVTETensor amax_reduction_buffer, std::vector<NVTETensor> amax_histories,
    std::vector<NVTETensor> scales, const char *amax_compute_algo, NVTEDType fp8_dtype,
    float margin, hipStream_t stream);
#endif

#endif
