// kernels.h -- host-side view of the HIP kernels (kernels.hip): plain structs
// and one launch function per kernel.  engine.cpp sequences them.
#ifndef LLSM_AMD_KERNELS_H
#define LLSM_AMD_KERNELS_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include "cheby.h"                                // IIR_SEG
#include "packed.h"                               // LlsmPackedLayout

// One 4th-order Chebyshev-I section in transposed direct form II (cheby.h)
// with everything the wave-parallel block recursion needs, all float64:
// steady-state initial conditions zi, the state-transition powers
// M[d] = (A^IIR_SEG)^(2^d) (row-major 4x4) and the zero-input output rows
// H[i] = e0^T A^i.
struct FiltSectionD {
  double b[5];
  double a[5];
  double zi[4];
  double M[6][16];
  double H[IIR_SEG][4];
};

// One zero-phase filtering job: src -> [sec0] -> (mid -> [sec1]) -> dst.
#ifndef KAL_CHUNK
#define KAL_CHUNK 8                                  // frames per chunk of k_kalman: checkpoint spacing (engine.cpp sizes the checkpoint rows with it) and rows per request
#endif
struct FiltJob {
  const float* src;
  float* dst;
  float* mid;    // only used when sec1 >= 0
  float* tmp;    // n + 30 floats (forward-pass output)
  int n;
  int sec0;
  int sec1;      // -1: single section
  int square;    // dst = y*y (llsm_subband_energy)
  int pad;       // odd-extension length of filtfilt (0: the default 15)
  int fused;     // two sections per pass (k_filtfilt): only the interior [wlo, whi) is valid
  int wlo, whi;  // samples of dst this job writes; whi <= wlo: all n
};

// switchable conventions of the ciglet primitives the reference cannot confirm
struct DevConventions {
  int mavg_half;       // moving_avg(x, n, 3): half width 3 (7 taps, default) or 1 (3 taps)
  int interp1u_excl;   // interp1u's right end: 0 inclusive (default), 1 exclusive
  int kalman_init;     // kalmanf1d at the first frame: 0 x0 = z0, P0 = R0 (default); 1 x0 = z0, P0 = the filter update of the prior P = R0
  float lobe_bias;     // cig_spec2env: constant added to the log envelope (layer 1); default 0.13397922601295542
  int lf_rd_clamp;     // lfmodel_from_rd: 0 extension formulas outside 0.21 <= Rd <= 2.7 (default); 1 Rd limited to 0.3 .. 2.7
};
int llsm_l1_kernels_set_conventions(const DevConventions& c);
int llsm_kernels_set_conventions(const DevConventions& c);

// Device-resident description of a batch (all pointers are device pointers).
struct BatchDev {
  int n_utt, nframes;
  int maxnhar, maxnhar_e, npsd, nchannel;
  float thop, fs, rel_winsize;
  // per utterance
  const int* x_off; const int* nx; const int* frm_off; const int* nfrm;
  // per frame
  const int* frm_utt;
  float* f0; int* nhar; float* ampl; float* phse;
  float* psd; float* psdres; int* has_psdres; float* edc; int* nhar_e;
  float* eenv_ampl; float* eenv_phse;
  // waveforms
  const float* x;
  // frame pairs sharing one complex transform in the PSD / envelope kernels: (g0, g1 or -1),
  // both of one utterance, so that a frame's result does not depend on its batch neighbours.
  // NULL = global pairing (2p, 2p + 1).
  const int2* pairs; int npairs;
  // 16-aligned blocks of every utterance's frames: (first global frame, frames in the block); the units of
  // k_harm_speech_tile.  NULL: no tiles.
  const int2* hblocks; int nhblocks;
  // k_synth_ola4: phasor tables shared by the frames of one F0 (0: every frame runs its own recurrences; same bits)
  int synth_tables;
};

// Transforms beyond the LDS kernels (N > LLSM_LDS_FFT_MAX points: a 25 ms hop at 96 kHz, peak picking below 21.6 Hz): the
// in-place FFT of dev_common.h run on per-workgroup GLOBAL scratch with twiddles from a table of its own
// (e^{-2 pi i m / tw_big_nmax}); engine.cpp llsm_engine_big_fft provides both before such a launch.
#define LLSM_LDS_FFT_MAX 8192
#define LLSM_BIG_FFT_MAX (1 << 17)
#define LLSM_BIG_FFT_GRID 512                          // persistent workgroups of a big-transform launch, at most
// ... fewer when a workgroup's scratch slice is large: the scratch of a launch stays within 256 MB (a 2^17-point peak-picking
// launch -- a batch whose lowest F0 is unknown provisions for it -- gets 170 workgroups instead of 512 x 1.5 MB)
static inline int llsm_big_fft_grid(size_t elems_per_wg) {
  const size_t cap = ((size_t)256 << 20) / sizeof(float2) / (elems_per_wg ? elems_per_wg : 1);
  return (int)(cap < 16 ? 16 : (cap > LLSM_BIG_FFT_GRID ? LLSM_BIG_FFT_GRID : cap));
}

struct LaunchCtx {
  hipStream_t stream;
  void (*prof_begin)(void* user, const char* name, hipStream_t stream);   // (events go on the stream of the launch)
  void (*prof_end)(void* user, hipStream_t stream);
  void* prof_user;
  const float2* tw_big = nullptr; int tw_big_nmax = 0; float2* big_scratch = nullptr; size_t big_scratch_elems = 0;
};

int launch_refine_f0(LaunchCtx* P, const BatchDev& d);
int launch_harm_speech(LaunchCtx* P, const BatchDev& d, float min_f0);
int launch_harm_env(LaunchCtx* P, const BatchDev& d, const float* ce, size_t ce_stride);
int launch_synth_frames(LaunchCtx* P, const BatchDev& d, int nwin, const float* win,
  const float* cyc_shift, float* frames, int lds_harmonics);
int synth_ola_group_units(void);
int synth_ola_unit_div(int nwin, int lds_harmonics);
int launch_synth_ola(LaunchCtx* P, const BatchDev& d, const int4* units, int nunits, int halo,
  int nwin, const float* win, int lds_harmonics, const int* out_off, const int* out_len,
  const float* x, float* out, int mode, float* mix);
int launch_harm_pp_big(LaunchCtx* P, const BatchDev& d, const float* sig, size_t sig_stride, int nsig,
  const int* nfft_u, int maxnhar, float norm_base, int lds_n, int nmax, int* nhar_out, float* ampl, float* phse);
int launch_filtfilt(LaunchCtx* P, const FiltJob* jobs, int njobs, const FiltSectionD* sections);
int launch_wf_selftest(LaunchCtx* P, int logN, const float2* in, float2* out, int count, int inverse);
int launch_spgm_env(LaunchCtx* P, const BatchDev& d, int nwin_psd, int N, int logN,
  int nfft_psd, float norm_base, const float2* tw, int tw_nmax, float* env_out,
  int2* fix_list, int* fix_count, int which = 3);   // fix_list [number of frame pairs] / fix_count: pairs whose DC / Nyquist bin the second launch recomputes exactly (NULL: off)
int launch_psd_frames(LaunchCtx* P, const BatchDev& d, const float* xres, int nwin,
  const float* win, float inv_wpow, int N, int logN, const float2* tw, int tw_nmax,
  float* psd_log);
int launch_kalman(LaunchCtx* P, const BatchDev& d, const float* env, const float* psd_log,
  float* ck, int nspec);
int launch_white(LaunchCtx* P, const BatchDev& d, float* white, int ntemplate_ext,
  const int* out_len, unsigned long long seed);
int launch_env_frames(LaunchCtx* P, const BatchDev& d, float fs_syn, int nwin,
  const float* win, float* envf);
#define LLSM_EXC_HITS 3                              // envelope frames that can cover one output sample
int launch_env_params(LaunchCtx* P, const BatchDev& d, float2* cplx);
int launch_pack_frames(LaunchCtx* P, const BatchDev& d, const LlsmPackedLayout& L, float* out, float* const* dst_tab);
int launch_unpack_frames(LaunchCtx* P, const BatchDev& d, const LlsmPackedLayout& L, const float* in, const float* const* src_tab);
int launch_scatter_outputs(LaunchCtx* P, int n_utt, int max_ny, const float* y, const float* ysin, const float* ynoise,
  const int* y_off, const int* ny, float* const* tab);
int launch_env_plan(LaunchCtx* P, int max_ny, int max_nfrm, int nwin_env, float thop, float fs, int2* hits, int* over);
int launch_excite_env(LaunchCtx* P, const BatchDev& d, const float* colored, int ntemplate_ext,
  const int2* hits, const float2* cplx, int nwin_env, const float* win, int nch_active,
  const int* out_off, const int* out_len, int max_len, float fs_syn, float* yexc);
int launch_noise_filter(LaunchCtx* P, const BatchDev& d, const float* yexc,
  const int* out_off, const int* out_len, float fnyq_conf, float fs_syn, int nwin,
  const float* win, float inv_wsqr, int N, int logN, const float2* tw, int tw_nmax,
  float* nframes_out, int* live, int rt);
int launch_noise_filter_ola(LaunchCtx* P, const BatchDev& d, const int4* units, int nunits, int halo,
  const float* yexc, const int* out_off, const int* out_len, float fnyq_conf, float fs_syn, int nwin,
  const float* win, int wsym, float inv_wsqr, int logN, float* ynoise);   // wsym: win[j] == win[wsym - j] (0: no such symmetry)
int launch_ola_noise_mix(LaunchCtx* P, const BatchDev& d, const float* nframes_in,
  const int* live, int N, const int* out_off, const int* out_len,
  int max_len, float fs_syn, const float* ysin, float* ynoise, float* y);

int launch_utt_fftsize(LaunchCtx* P, const BatchDev& d, int nmax, int* nfft_u);
int launch_harm_pp(LaunchCtx* P, const BatchDev& d, const float* sig, size_t sig_stride, int nsig,
  const int* nfft_u, int maxnhar, float norm_base, const float2* tw, int tw_nmax, int lds_n,
  int* nhar_out, float* ampl, float* phse);
int launch_rt_template(LaunchCtx* P, const float* colored, int ntemplate_ext, int nch, int nch_active,
  int ntemplate, int S, float* tpl);
int launch_rt_rings(LaunchCtx* P, int S, float* mod, float* sinr, float* noiser, int cap, int nch,
  int mod_curr, int sin_curr, int noise_curr, int nhop, int nwin, const float* envf,
  const float* frames_sin, const float* f0, const int* has_nm, const int* nhar);
int launch_rt_rings_excite(LaunchCtx* P, int S, float* mod, float* sinr, float* noiser, int cap, int nch,
  int mod_curr, int sin_curr, int noise_curr, int nhop, int nwin, const float* envf, const float* frames_sin,
  const float* f0, const int* has_nm, const int* nhar, const float* tpl, float* excr, int ntemplate, int exc_curr,
  int exc_cycle, float* exc_frame);
int launch_rt_excite(LaunchCtx* P, int S, const float* mod, const float* tpl, float* excr, int cap,
  int nch, int ntemplate, int mod_curr, int exc_curr, int exc_cycle, int curr_nhop, int nx,
  int nwin_frame, float* exc_frame);
int launch_rt_mix(LaunchCtx* P, int S, float* noiser, const float* sinr, int cap, int noise_curr,
  int sin_curr, int sin_pos, int nfft, const float* nframes_in, const int* live, int next_nhop,
  int out_stride, float* out);

// llsmrt: one hop in two launches (k_rt_front: envelope frames beside the harmonic frame, ring adds, excitation; k_rt_back:
// noise filter of a pair of streams on four wavefronts, noise-ring add, the hop's output samples)
// the per-stream parameter rows of a hop where the host packed them (pinned, device-mapped); f0 == nullptr: none
struct RtPbpOp;
struct RtRows {
  const float *f0, *cyc, *ampl, *phse, *edc, *eamp, *ephs, *psd;
  const int *nhar, *nhar_e, *has_nm;
  const float* f0sin; const RtPbpOp* ops;           // pulse-by-pulse buffers (else nullptr): -> the device f0_sin / ops rows
};
// pulse-by-pulse bookkeeping of the hop inside k_rt_front / k_rt_hop (k_rt_pbp's arguments); ops == nullptr: none
struct RtPbpArgs { RtPbpOp* ops; float* frwd; float* bkwd; int dual_curr; const float* pulse_out; int pulse_stride; };
// host != nullptr: every workgroup first moves its stream's rows (as many harmonics as the frame has) from *host into
// the device rows of d / cyc_shift / has_nm, then works on those
int launch_rt_front(LaunchCtx* P, const BatchDev& d, int nwin, const float* win, const float* f0_sin, const float* cyc_shift,
  float* envf, float* frames_sin, int lds_harmonics, float* mod, float* sinr, float* noiser, int cap, int mod_curr,
  int sin_curr, int noise_curr, int nhop, const int* has_nm, const float* tpl, float* excr, int ntemplate, int exc_curr,
  int exc_cycle, float* exc_frame, const RtRows* host = nullptr, const RtPbpArgs* pbp = nullptr);
int launch_rt_hop(LaunchCtx* P, const BatchDev& d, int nwin, const float* win, const float* f0_sin, const float* cyc_shift,
  float* envf, float* frames_sin, int lds_harmonics, float* mod, float* sinr, float* noiser, int cap, int mod_curr,
  int sin_curr, int noise_curr, int nhop, const int* has_nm, const float* tpl, float* excr, int ntemplate, int exc_curr,
  int exc_cycle, float* exc_frame, const RtRows* host, float fnyq_conf, float inv_wsqr, int N, int logN, const float2* tw,
  int tw_nmax, float* nframes, int* live, int sin_pos, int next_nhop, int out_stride, float* out, const RtPbpArgs* pbp = nullptr,
  bool on_chip = true);
int launch_rt_back(LaunchCtx* P, const BatchDev& d, const float* exc_frame, float fnyq_conf, float fs_syn, int nwin,
  const float* win, float inv_wsqr, int N, int logN, const float2* tw, int tw_nmax, float* nframes, int* live,
  float* noiser, const float* sinr, int cap, int noise_curr, int sin_curr, int sin_pos, int next_nhop, int out_stride,
  float* out);


// ---- layer 1 / pulse-by-pulse synthesis (l1_kernels.hip) ----
struct AlphaCache { double* alpha; float* rd; float* f0; };   // alpha: 9 float64 per frame, the whole lf::Solved (l1_kernels.hip lf_solve_cached)
struct L1Dev {
  int nframes, maxnhar, nspec;
  float fnyq, lip_radius;
  const float* f0; int* nhar; float* ampl; float* phse;
  float* rd; float* vtmagn; float* vsphse; int* nvsphse; int* has_hm;
  // tolayer1 only (may be NULL / 0): scratch rows [nframes][maxnhar] for the source-removed amplitudes, and the
  // per-utterance frame pairs (BatchDev::pairs) of the two-frames-per-transform envelope kernel
  float* src_ampl; const int2* pairs; int npairs;
  // alpha of the frame's LF model (the wavefront search of lf_solve_wave), kept per frame with the (Rd, F0) it was solved
  // for: the four kernels of a layer-1 step that need it solve it once (acache.alpha NULL: no cache)
  AlphaCache acache;
};
// one pulse group = the pulses of one frame (llsm_make_filtered_pulse's arguments, llsmutils.c:132-134)
struct PbpJob {
  int frame;        // global frame index (rows of f0 / rd / vtmagn / vsphse)
  int first, npulse;// pulses[first .. first + npulse)
  int size;         // pulse_size (power of two)
  int pre_rotate;
  int out_off;      // offset of this group's `size` samples in the pulse buffer
  int start;        // output sample of pulse sample 0 (within the utterance / ring)
  int zero_extra;   // pulse sample that ALSO lands on output sample 0 through (int) truncation, or -1
};
struct PbpPulse { double T0, te, tp, ta, Ee; float offset; float pad; };
// one stretch of the HM <-> PbP cross-fade curve (layer0.c:240-262)
struct PbpSeg { double state, rate; int j0, j1, dir, len; int out_off, pad; };

// llsmrt, PbP path: per-stream operations of one hop on the dual (pulse) buffer and the sinusoid ring
struct RtPbpOp { int add_off, add_size, rd_off, rd_on, term_off, term_size, pad0, pad1; };
int launch_rt_pbp(LaunchCtx* P, int S, const RtPbpOp* ops, float* frwd, float* bkwd, int cap, int dual_curr,
  float* sinr, int sin_curr, int nhop, const float* win, const float* pulse_out, int pulse_stride);
int launch_coder_encode(LaunchCtx* P, int order_spec, int order_bap, int ns, int npsd, float fnyq, float liprad,
  const float* melaxis, int nframes, const float* f0, const float* rd, const float* psd, const float* vtmagn,
  const int* nvsphse, float* enc);
int launch_coder_decode(LaunchCtx* P, int order_spec, int order_bap, int ns, int npsd, int maxnhar, float fnyq, float liprad,
  const float* melaxis, float mel_floor, float mel_ceil, int nframes, const float* enc, int use_l1, const float2* tw,
  int tw_nmax, float* f0, float* rd, int* nhar, float* ampl, float* phse, float* psd, float* vtmagn, float* vsphse,
  int* nvsphse, int* has_hm);
int l1_minphase_nmax(int maxnhar);
int launch_l1_rd_fit(LaunchCtx* P, const L1Dev& d, const float* model_power, const float* model_param,
  const double* inv_t, const double* cumlog_t, float* rd_raw);   // inv_t / cumlog_t: [nh (+ 1)][64] float64 tables, NULL: per-term form
int launch_l1_rd_smooth(LaunchCtx* P, int n_utt, const int* frm_off, const int* nfrm, int order,
  const float* rd_raw, int* prev_idx, int* next_idx, float* cont, float* rd_out);
int launch_l1_frame(LaunchCtx* P, const L1Dev& d, int nfft, const float2* tw, int tw_nmax);
int launch_l1_to_l0(LaunchCtx* P, const L1Dev& d, int maxnhar_conf, int only_missing, const int* select,
  const float2* tw, int tw_nmax);
int launch_pbp_pulse(LaunchCtx* P, const L1Dev& d, const PbpJob* jobs, int njobs, const PbpPulse* pulses,
  int size_max, float fs, const float2* tw, int tw_nmax, float* out);
// next-cycle projection of every layer-1 frame (layer0.c:181-191), float64
int launch_l1_projection(LaunchCtx* P, const L1Dev& d, double fs, double* proj);
int launch_l1_mixcurve(LaunchCtx* P, const PbpSeg* segs, int nsegs, float* mixw);
int launch_pbp_mix(LaunchCtx* P, int n_utt, int max_len, const int* out_off, const int* out_len, const int* frm_off,
  const int* nfrm, float thop, float fs, int nwin, const float* hm_frames, const float* f0_hm, const PbpJob* jobs,
  const int2* blk_jobs, const int* blk_off, const float* pulse_buf, const float* mixw, const float* ynoise,
  float* ysin, float* y);

#endif
