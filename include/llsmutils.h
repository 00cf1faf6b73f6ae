/*
 * llsmutils.h -- LLSM-specific utilities; replaces the reference's installed llsmutils.h (llsmutils.h:30-46).
 * Written from scratch.  The reference takes the `lfmodel` type from ciglet, which is neither vendored nor
 * pinned: the struct below is this library's own (the five fields libllsm2 itself reads, llsmutils.c:24-43;
 * te, tp, ta relative to T0), so it is source-compatible with host code that fills it field by field and
 * NOT guaranteed layout-compatible with a ciglet build.
 */
#ifndef LLSM_AMD_LLSMUTILS_H
#define LLSM_AMD_LLSMUTILS_H

#include "llsm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  FP_TYPE T0;   /* period (s) */
  FP_TYPE te;   /* instant of the main excitation, relative to T0 */
  FP_TYPE tp;   /* instant of the flow maximum, relative to T0 */
  FP_TYPE ta;   /* return-phase time constant, relative to T0 */
  FP_TYPE Ee;   /* excitation amplitude */
} lfmodel;

/* ciglet's lfmodel_from_rd / lfmodel_spectrum as this library defines them (csrc/lfmodel.h);
 * lfmodel_spectrum returns malloc'ed magnitudes and fills dst_phase when non-NULL */
lfmodel  llsm_lfmodel_from_rd(FP_TYPE rd, FP_TYPE T0, FP_TYPE Ee);
FP_TYPE* llsm_lfmodel_spectrum(lfmodel model, FP_TYPE* freq, int nf, FP_TYPE* dst_phase);

llsm_gfm llsm_lfmodel_to_gfm(lfmodel src);                         /* replaces llsmutils.h:31 */
lfmodel  llsm_gfm_to_lfmodel(llsm_gfm src);                        /* replaces llsmutils.h:34 */
/* replaces llsmutils.h:38-39 */
FP_TYPE* llsm_synthesize_harmonic_frame_auto(llsm_soptions* options, FP_TYPE* ampl, FP_TYPE* phse,
  int nhar, FP_TYPE f0, int nx);
/* replaces llsmutils.h:44-46; src: a frame with LLSM_FRAME_F0 / RD / VTMAGN / VSPHSE */
FP_TYPE* llsm_make_filtered_pulse(llsm_container* src, lfmodel* sources, FP_TYPE* offsets, int num_pulses,
  int pre_rotate, int size, FP_TYPE fnyq, FP_TYPE lip_radius, FP_TYPE fs);

#ifdef __cplusplus
}
#endif
#endif
