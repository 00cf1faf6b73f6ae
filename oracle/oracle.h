/*
 * oracle.h -- CPU restatement of libllsm2's layer-0 analysis/synthesis path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it, and only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (Sleepwalking/libllsm2 @ 2.1.0) cannot be
 * built here because all of its arithmetic lives in the un-vendored,
 * un-pinned dependency `ciglet`, which is absent.  The reference tree holds no
 * golden vectors.  This oracle therefore restates the reference's own code
 * (layer0.c, dsputils.c, llsmutils.c, llsmrt.c, buffer.h, frame.c) plus OUR
 * documented definition of each ciglet primitive the path uses (DESIGN.md
 * section "ciglet contract").  It is pinned only by the reference's
 * re-statable known-answer tests (tests/test_oracle_kat.py) and by
 * numpy/scipy cross-checks of every primitive (tests/test_oracle_dsp.py).
 *
 * Build: `make -C oracle` -> liboracle_f32.so (OFP=float, the FP_TYPE the
 * reference ships with) and liboracle_f64.so (OFP=double, the precision
 * yardstick used by the parity tests).
 */
#ifndef LLSM_ORACLE_H
#define LLSM_ORACLE_H

#ifndef OFP
#define OFP double
#endif
typedef OFP fp;

#ifdef __cplusplus
extern "C" {
#endif

/* ---- analysis / synthesis options (mirror llsm_aoptions/llsm_soptions,
 *      llsm.h:260-272, 290-299; defaults layer0.c:27-43, 78-87) ---- */
typedef struct {
  fp thop;
  int maxnhar;
  int maxnhar_e;
  int npsd;
  int nchannel;
  fp chanfreq[8];   /* nchannel-1 used */
  fp lip_radius;
  int f0_refine;
  int hm_method;    /* 0 = peak picking, 1 = CZT */
  fp rel_winsize;
} o_aoptions;

typedef struct {
  fp fs;
  int use_iczt;
  int use_l1;
  fp iczt_param_a;
  fp iczt_param_b;
} o_soptions;

/* ---- flat (SoA) layer-0 parameter set of one utterance ----
 * Row i describes frame i.  Rows of unvoiced frames have nhar = nhar_e = 0.
 * This is the same layout the GPU library uses in HBM. */
typedef struct {
  int nfrm, maxnhar, maxnhar_e, npsd, nchannel;
  fp thop, fnyq;
  fp chanfreq[8];
  fp* f0;         /* [nfrm] */
  int* nhar;      /* [nfrm] */
  fp* ampl;       /* [nfrm][maxnhar] */
  fp* phse;       /* [nfrm][maxnhar] */
  fp* psd;        /* [nfrm][npsd]  dB */
  fp* psdres;     /* [nfrm][npsd]  (LLSM_FRAME_PSDRES), may be NULL */
  fp* edc;        /* [nfrm][nchannel] */
  int* nhar_e;    /* [nfrm] */
  fp* eenv_ampl;  /* [nfrm][nchannel][maxnhar_e] */
  fp* eenv_phse;  /* [nfrm][nchannel][maxnhar_e] */
} o_params;

/* ---- index plan shared convention: every sample-index
 * product is evaluated in IEEE float32, left to right, regardless of OFP ---- */
int o_idx_center(int i, float thop, float fs);          /* round(i*thop*fs) */
int o_idx_nwin_sin(float thop, float fs);               /* round(thop*fs)*2 */
int o_idx_nwin_env(float thop, float fs);               /* round(thop*2.0*fs) */
int o_idx_nwin_filt(float thop, float fs);              /* round(thop*fs*2) */
int o_idx_nwin_psd(float thop, float fs);               /* round(thop*4*fs) */
int o_idx_ny(int nfrm, float thop, float fs);           /* round((nfrm+1)*thop*fs) */
int o_idx_hwin(float f0, float fs, float rel_winsize);  /* round(fs/f0*rel/2)*2 */
int o_idx_nhar(float f0, float fs, int maxnhar);        /* min(floor(fs/f0/2),max) */
int o_idx_env_ola(int i, int j, float thop, float fs);  /* round((i-1)*thop*fs+j) */
int o_idx_dcwin(float f0, float thop, float fs);
int o_idx_spgmwin(float f0, float fs, int nwin_psd);
int o_nextpow2(double x);                               /* pow(2,ceil(log2 x)) */
float o_idx_rawfrac(int i, float thop, float fs, int* baseidx);

/* ---- ciglet-contract primitives (our definitions) ---- */
void o_fft(fp* re, fp* im, int n, int inverse);   /* in place; inverse scales 1/n */
void o_hanning(fp* w, int n);                     /* symmetric */
void o_hanning_ola(fp* w, int n);                 /* overlap-add Hann: symmetric unless "hann_periodic" */
/* convention switches (same names / values as llsm_gpu_set_convention, llsm_gpu.h) */
int o_set_convention(const char* name, int value);
int o_conv_mavg_half(void);
int o_conv_lf_rd_clamp(void);
int o_conv_interp1u_excl(void);
double o_conv_lobe_bias(void);
void o_blackman(fp* w, int n);                    /* symmetric, 0.42/0.5/0.08 */
void o_fetch_frame(const fp* x, int nx, int center, int nf, fp* out);
void o_czt(const fp* x, int n, fp omega0, int nout, fp* yr, fp* yi);
void o_iczt(const fp* xr, const fp* xi, int nbin, fp omega0, int n, fp* y);
void o_set_czt_mode(int bluestein);               /* 0 direct (default), 1 FFT-based */
void o_gensins(const fp* freq, const fp* ampl, const fp* phse, int nsin,
  fp fs, int n, fp* y);
void o_interp1(const fp* xi, const fp* yi, int ni, const fp* xq, int nq, fp* yq);
void o_interp1u(fp x0, fp x1, const fp* yi, int ni, const fp* xq, int nq, fp* yq);
void o_moving_avg(const fp* x, int n, int halford, fp* y);
void o_kalmanf1d(const fp* z, const fp* Q, const fp* R, int n, fp* P, fp* y);
void o_kalmans1d(const fp* y, const fp* P, const fp* Q, int n, fp* s);
void o_cheby1(int order, double rp_db, double wn, int highpass, double* b, double* a);
void o_get_chebyshev_filter(fp cutoff, int highpass, fp* a, fp* b); /* 5 + 5 */
void o_filtfilt(const fp* b, int nb, const fp* a, int na, const fp* x, int nx, fp* y);
void o_chebyfilt(const fp* x, int nx, fp c1, fp c2, fp* y);
void o_spec2env(const fp* S, int nfft, fp f0, fp* env);   /* log-magnitude envelope */
void o_stft_frame(const fp* x, int nx, int center, int winsize, int nfft,
  int blackman, fp* magn, fp* phse, fp* wsum);
fp o_wrap(fp x);
fp o_rng_normal(unsigned long long seed, unsigned long long idx);

/* ---- llsm layer-0 building blocks (dsputils.c / llsmutils.c) ---- */
void o_harmonic_czt(const fp* x, int nx, fp f0, fp fs, int nhar, fp* ampl, fp* phse);
void o_harmonic_analysis(const fp* x, int nx, fp fs, const fp* f0, int nfrm,
  fp thop, fp rel_winsize, int maxnhar, int method, int stride,
  int* nhar, fp* ampl, fp* phse);
void o_compute_spectrogram(const fp* x, int nx, const int* center,
  const int* winsize, int nfrm, int nfft, int blackman, fp* spec, fp* phse);
void o_compute_dc(const fp* x, int nx, const int* center, const int* winsize,
  int nfrm, fp* dc);
void o_subband_energy(const fp* x, int nx, fp fmin, fp fmax, fp* y);
void o_estimate_psd(const fp* x, int nx, int nfft, fp* psd);
void o_synth_harmonic_frame(const fp* ampl, const fp* phse, int nhar, fp f0, int nx, fp* y);
void o_synth_harmonic_frame_iczt(const fp* ampl, const fp* phse, int nhar, fp f0, int nx, fp* y);
int  o_synth_harmonic_frame_auto(const o_soptions* opt, const fp* ampl,
  const fp* phse, int nhar, fp f0, int nx, fp* y);  /* returns 1 if ICZT chosen */
void o_generate_white_noise(int nx, unsigned long long seed, fp* y);
void o_generate_bandlimited_noise(int nx, fp fmin, fp fmax,
  unsigned long long seed, const fp* white_in, fp* y);
void o_spectrum_from_envelope(const fp* freq, const fp* ampl, int nfreq,
  int nspec, fp fnyq, fp* out);
void o_refine_f0(const fp* x, int nx, fp fs, fp* f0, int nfrm, fp thop);

/* ---- layer-0 entry points (layer0.c:478-511, 636-664) ---- */
void o_default_aoptions(o_aoptions* o);
void o_default_soptions(o_soptions* o, fp fs);
/* p must be pre-allocated for nfrm frames; x_res (nx) may be NULL. */
int o_analyze(const o_aoptions* opt, const fp* x, int nx, fp fs, fp* f0,
  int nfrm, o_params* p, fp* x_res);
/* y, y_sin, y_noise: ny = o_idx_ny(...) samples each. white: optional
 * [nchannel][min(20000,ny)+128] injected Gaussian templates, else seeded RNG. */
int o_synthesize(const o_soptions* opt, const o_params* p,
  unsigned long long seed, const fp* white, fp* y, fp* y_sin, fp* y_noise);
void o_chunk_phasepropagate(o_params* p, int sign);
void o_chunk_phasesync_rps(o_params* p);

/* ---- layer 1 (source-filter) conversion and pulse-by-pulse synthesis (l1_oracle.c) ---- */
typedef struct { fp T0, te, tp, ta, Ee; } o_lfmodel;    /* te, tp, ta relative to T0 (llsmutils.c:24-43) */
typedef struct { fp Fa, Rk, Rg, T0, Ee; } o_gfm;        /* llsm_gfm, llsm.h:181-187 */
/* llsm_fgfm (llsm.h:190-191); `frame` = index of the source frame instead of the container */
typedef void (*o_fgfm)(o_gfm* dst, fp* delta_t, void* info, int frame);
/* flat layer-1 members of one utterance, beside o_params */
typedef struct {
  int nfrm, nspec, maxnhar;     /* maxnhar: row width of vsphse (== o_params.maxnhar) */
  fp lip_radius;
  fp* rd;        /* [nfrm]          LLSM_FRAME_RD (every frame) */
  fp* vtmagn;    /* [nfrm][nspec]   LLSM_FRAME_VTMAGN, dB */
  fp* vsphse;    /* [nfrm][maxnhar] LLSM_FRAME_VSPHSE */
  int* nvsphse;  /* [nfrm]          length of the VSPHSE array */
  int* has_l1;   /* [nfrm]          VTMAGN / VSPHSE present */
  int* has_hm;   /* [nfrm]          LLSM_FRAME_HM present (rows of o_params valid) */
  int* pbpsyn;   /* [nfrm]          LLSM_FRAME_PBPSYN == 1 */
  int* has_eff;  /* [nfrm]          LLSM_FRAME_PBPEFF attached */
  fp* dbg_y_hm; fp* dbg_y_pbp; fp* dbg_y_mix;   /* optional [ny] taps of the three internal signals */
} o_l1params;

o_lfmodel o_lfmodel_from_rd(fp rd, fp T0, fp Ee);
void o_lfmodel_spectrum(o_lfmodel m, const fp* freq, int nf, fp* magn, fp* phase);
void o_lfmodel_waveform(o_lfmodel m, const double* t, int n, double* out);
o_gfm o_lfmodel_to_gfm(o_lfmodel s);
o_lfmodel o_gfm_to_lfmodel(o_gfm s);
void o_interp1u_excl(fp x0, fp x1, const fp* yi, int ni, const fp* xq, int nq, fp* yq);
void o_interp_in_blank(const fp* x, int n, fp blank, fp* y);
void o_minphase(const fp* logmag, int nfft, fp* phase);
void o_lipfilter(fp radius, fp f0, int nhar, fp* ampl, fp* phse, int inverse);
void o_lipfilter_reim(fp radius, fp f0, int nhar, fp* re, fp* im, int inverse);
void o_harmonic_spectrum(const fp* ampl, int nhar, fp f0, int nfft, fp* X);
void o_harmonic_envelope(const fp* ampl, int nhar, fp f0, int nfft, fp* env_db);
int  o_minphase_fftsize(int nhar);
void o_harmonic_minphase(const fp* ampl, int nhar, fp* phse);
void* o_glottal_create(const fp* param, int nparam, int nhar);
void o_glottal_delete(void* g);
fp   o_glottal_fit(const fp* ampl, int nhar, void* g);
void o_smoothing_filter(const fp* x, int nx, int order, fp* y);
void o_analyze_rd(const o_params* p, fp lip_radius, fp* rd_smooth);
void o_chunk_tolayer1(const o_params* p, o_l1params* q, int nfft);
void o_frame_tolayer0(o_params* p, const o_l1params* q, int i, int maxnhar_conf);
void o_chunk_tolayer0(o_params* p, const o_l1params* q, int maxnhar_conf);
void o_l1_phaseshift(o_l1params* q, int i, fp theta);
fp   o_pulse_projection(fp rd, fp f0, fp vsphse0, fp fs, fp origin);
void o_make_filtered_pulse(fp rd, fp f0, const fp* vtmagn, int nspec, const fp* vsphse, int nhar,
  const o_lfmodel* sources, const fp* offsets, int num_pulses, int pre_rotate, int size, fp fnyq,
  fp lip_radius, fp fs, fp* y);
void o_synthesize_harmonics_l1(const o_soptions* opt, o_params* p, o_l1params* q, int maxnhar_conf,
  o_fgfm effect, void* effect_info, fp* y_mix, int ny);
int o_synthesize_l1(const o_soptions* opt, o_params* p, o_l1params* q, int maxnhar_conf,
  o_fgfm effect, void* effect_info, unsigned long long seed, const fp* white,
  fp* y, fp* y_sin, fp* y_noise);

/* ---- frame coder (coder_oracle.c; coder.c:44-292) ---- */
typedef struct o_coder o_coder;
o_coder* o_coder_create(fp fnyq, int nchannel, int nhar_e, int npsd, int nspec, fp liprad, int order_spec, int order_bap);
void o_coder_delete(o_coder* c);
void o_coder_encode(const o_coder* c, fp f0, fp rd, const fp* psd, const fp* vtmagn, fp* enc);
void o_coder_decode(const o_coder* c, const fp* src, int use_layer1, fp* f0_out, fp* rd_out, int* nhar_out, fp* psd_out,
  fp* vtmagn, fp* vsphse, fp* ampl_out, fp* phse_out, int maxnhar);

/* ---- llsmrt streaming synthesis (llsmrt.c) ---- */
typedef struct o_rtsynth o_rtsynth;
o_rtsynth* o_rt_create(const o_soptions* opt, const o_params* conf,
  int capacity, unsigned long long seed);
void o_rt_delete(o_rtsynth* s);
int  o_rt_latency(o_rtsynth* s);
int  o_rt_numoutput(o_rtsynth* s);
void o_rt_feed(o_rtsynth* s, const o_params* p, int frame);
void o_rt_feed_l1(o_rtsynth* s, o_params* p, o_l1params* q, int frame, int maxnhar_conf,
  o_fgfm effect, void* effect_info);
int  o_rt_fetch(o_rtsynth* s, fp* p_out, fp* ap_out);

#ifdef __cplusplus
}
#endif
#endif
