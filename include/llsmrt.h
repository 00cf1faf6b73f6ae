/*
 * llsmrt.h -- real-time (pull-loop) synthesis interface of libllsm2_amd.
 * Replaces the reference's llsmrt.h:33-54 (implementation llsmrt.c:157-602)
 * for the harmonic-model path (options->use_l1 == 0).  Written from scratch;
 * same names, argument meaning and return conventions.
 *
 * One producer calls llsm_rtsynth_buffer_feed (one frame = one hop of audio),
 * one consumer calls the fetch functions (one sample per call, non-blocking,
 * returns 1 on success and 0 when no sample is ready).  The per-hop DSP
 * (harmonic frame, noise-envelope frames, excitation mix, FFT noise filter)
 * runs on the GPU; the output rings live in host memory so a fetch never
 * touches the device.
 */
#ifndef LLSM_AMD_LLSMRT_H
#define LLSM_AMD_LLSMRT_H

#include "llsm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void llsm_rtsynth_buffer;

/* replaces llsmrt.h:34-35: NULL when conf lacks NCHANNEL / THOP / CHANFREQ
 * (llsmrt.c:163), when options->use_l1 != 0 (pulse-by-pulse synthesis is out
 * of scope here), or when no usable GPU exists. */
llsm_rtsynth_buffer* llsm_create_rtsynth_buffer(llsm_soptions* options,
  llsm_container* conf, int capacity_samples);
void llsm_delete_rtsynth_buffer(llsm_rtsynth_buffer* dst);      /* llsmrt.h:38 */
int  llsm_rtsynth_buffer_getlatency(llsm_rtsynth_buffer* src);  /* llsmrt.h:41 */
int  llsm_rtsynth_buffer_numoutput(llsm_rtsynth_buffer* src);   /* llsmrt.h:43 */
/* does not take ownership of `frame` (llsmrt.c:513-516); blocks while the
 * output ring is full (llsmrt.c:489-493). */
void llsm_rtsynth_buffer_feed(llsm_rtsynth_buffer* dst, llsm_container* frame);
int  llsm_rtsynth_buffer_fetch(llsm_rtsynth_buffer* src, FP_TYPE* dst);
int  llsm_rtsynth_buffer_fetch_decomposed(llsm_rtsynth_buffer* src,
  FP_TYPE* dst_p, FP_TYPE* dst_ap);
void llsm_rtsynth_buffer_clear(llsm_rtsynth_buffer* dst);       /* llsmrt.h:54 */

#ifdef __cplusplus
}
#endif
#endif
