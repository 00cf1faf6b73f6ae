/*
 * dsputils.h -- per-frame DSP entry points of libllsm2_amd.
 *
 * Written from scratch; replaces the reference's installed dsputils.h (dsputils.h:25-132, makefile:132-135):
 * same names, argument meaning, ownership (returned arrays are malloc'ed, the caller frees them).  Heavy
 * work runs in HIP kernels on the default device (csrc/frame_kernels.hip, csrc/l1_kernels.hip and the batch
 * kernels); there is no CPU fallback: without a device the functions leave their outputs zeroed / return
 * NULL and llsm_gpu_last_error() says why.  These are ONE-frame conveniences -- thousands of frames per
 * call go through llsm_gpu.h.
 */
#ifndef LLSM_AMD_DSPUTILS_H
#define LLSM_AMD_DSPUTILS_H

#include "llsm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* replaces dsputils.h:26-27; overwrites f0[] */
void llsm_refine_f0(FP_TYPE* x, int nx, FP_TYPE fs, FP_TYPE* f0, int nfrm, FP_TYPE thop);
/* replaces dsputils.h:30-31; wintype "hanning" | "blackman"; dst_spec / dst_phse: nfrm rows of nfft / 2 + 1
 * values allocated by the caller (dst_phse may be NULL) */
void llsm_compute_spectrogram(FP_TYPE* x, int nx, int* center, int* winsize, int nfrm, int nfft,
  char* wintype, FP_TYPE** dst_spec, FP_TYPE** dst_phse);
/* replaces dsputils.h:34-35 */
void llsm_compute_dc(FP_TYPE* x, int nx, int* center, int* winsize, int nfrm, FP_TYPE* dst_dc);
/* replaces dsputils.h:38-40; spectrum is a LOG-magnitude spectrum (dsputils.c:209-211) */
void llsm_harmonic_peakpicking(FP_TYPE* spectrum, FP_TYPE* phase, int nfft, FP_TYPE fs, int nhar,
  FP_TYPE f0, FP_TYPE* dst_ampl, FP_TYPE* dst_phse);
/* replaces dsputils.h:43-44 */
void llsm_harmonic_czt(FP_TYPE* x, int nx, FP_TYPE f0, FP_TYPE fs, int nhar, FP_TYPE* dst_ampl,
  FP_TYPE* dst_phse);
/* replaces dsputils.h:49-51; dst_ampl[i] / dst_phse[i] are malloc'ed for voiced frames, NULL otherwise */
void llsm_harmonic_analysis(FP_TYPE* x, int nx, FP_TYPE fs, FP_TYPE* f0, int nfrm, FP_TYPE thop,
  FP_TYPE rel_winsize, int maxnhar, int method, int* dst_nhar, FP_TYPE** dst_ampl, FP_TYPE** dst_phse);
/* replaces dsputils.h:54; fmin / fmax relative to the sampling rate */
FP_TYPE* llsm_subband_energy(FP_TYPE* x, int nx, FP_TYPE fmin, FP_TYPE fmax);
/* replaces dsputils.h:57-58 (elementwise, host) */
void llsm_fft_to_psd(FP_TYPE* X_re, FP_TYPE* X_im, int nfft, FP_TYPE wsqr, FP_TYPE* dst_psd);
/* replaces dsputils.h:61 */
void llsm_estimate_psd(FP_TYPE* x, int nx, int nfft, FP_TYPE* dst_psd);
/* replace dsputils.h:64-75 (index arithmetic and linear interpolation on short vectors, host) */
FP_TYPE* llsm_warp_frequency(FP_TYPE fmin, FP_TYPE fmax, int n, FP_TYPE warp_const);
FP_TYPE* llsm_spectral_mean(FP_TYPE* spectrum, int nspec, FP_TYPE fnyq, FP_TYPE* freq, int nfreq);
FP_TYPE* llsm_spectrum_from_envelope(FP_TYPE* freq, FP_TYPE* ampl, int nfreq, int nspec, FP_TYPE fnyq);
/* replaces dsputils.h:78 */
int llsm_get_fftsize(FP_TYPE* f0, int nfrm, FP_TYPE fs, FP_TYPE rel_winsize);
/* replace dsputils.h:81-86; f0 relative to the sampling rate; both return the same signal here
 * (the recurrent sinusoid bank and the ICZT compute one function; test/test-harmonic.c:40-47) */
FP_TYPE* llsm_synthesize_harmonic_frame(FP_TYPE* ampl, FP_TYPE* phse, int nhar, FP_TYPE f0, int nx);
FP_TYPE* llsm_synthesize_harmonic_frame_iczt(FP_TYPE* ampl, FP_TYPE* phse, int nhar, FP_TYPE f0, int nx);
/* replace dsputils.h:89-92; Gaussian samples from the counter generator of DESIGN.md section 6 */
FP_TYPE* llsm_generate_white_noise(int nx);
FP_TYPE* llsm_generate_bandlimited_noise(int nx, FP_TYPE fmin, FP_TYPE fmax);
/* replace dsputils.h:95-100 (closed form per harmonic, host) */
void llsm_lipfilter(FP_TYPE radius, FP_TYPE f0, int nhar, FP_TYPE* dst_ampl, FP_TYPE* dst_phse, int inverse);
void llsm_lipfilter_reim(FP_TYPE radius, FP_TYPE f0, int nhar, FP_TYPE* dst_re, FP_TYPE* dst_im, int inverse);
/* replace dsputils.h:104-112; f0 relative to the sampling rate */
FP_TYPE* llsm_harmonic_spectrum(FP_TYPE* ampl, int nhar, FP_TYPE f0, int nfft);
FP_TYPE* llsm_harmonic_envelope(FP_TYPE* ampl, int nhar, FP_TYPE f0, int nfft);
FP_TYPE* llsm_harmonic_minphase(FP_TYPE* ampl, int nhar);
/* replace dsputils.h:114-128 */
typedef void llsm_cached_glottal_model;
llsm_cached_glottal_model* llsm_create_cached_glottal_model(FP_TYPE* param, int nparam, int nhar);
void llsm_delete_cached_glottal_model(llsm_cached_glottal_model* dst);
FP_TYPE llsm_spectral_glottal_fitting(FP_TYPE* ampl, int nhar, llsm_cached_glottal_model* model);
/* replaces dsputils.h:131-132 (host) */
FP_TYPE* llsm_smoothing_filter(FP_TYPE* x, int nx, int order);

#ifdef __cplusplus
}
#endif
#endif
