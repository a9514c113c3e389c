/* Replaces reference .../include/transformer_engine/multi_stream.h
 * (definitions: common/util/multi_stream.cpp:22-67): a lazily created pool of
 * non-blocking HIP streams + events used to fan a list of independent
 * quantizes / GEMMs out and join them back on the caller's stream. */
#ifndef TE_AMD_MULTI_STREAM_H_
#define TE_AMD_MULTI_STREAM_H_

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ref multi_stream.h:21 */ int nvte_get_num_compute_streams(void);
/* ref multi_stream.h:31 */ hipStream_t nvte_get_compute_stream(const int idx);
/* ref multi_stream.h:41 */ hipEvent_t nvte_get_compute_stream_event(const int idx);

#ifdef __cplusplus
}
#endif
#endif
