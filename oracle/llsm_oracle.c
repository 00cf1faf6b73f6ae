/*
 * llsm_oracle.c -- oracle restatement of libllsm2 layer-0 analysis and offline
 * synthesis (reference: layer0.c, dsputils.c, llsmutils.c, frame.c @ 2.1.0).
 * TEST INFRASTRUCTURE ONLY; "parity unpinned" (see oracle.h).
 *
 * Each function cites the reference file:line it follows.  Data are flat
 * arrays (o_params) instead of llsm_container trees; the container ABI is
 * host logic of the product and is tested on its own (tests/test_structs.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* constants.h:4-13 */
#define LOG2IN(x) ((x) / 2.3025851 * 10.0)
#define DB2LOG(x) ((x) * 2.3025851 / 20.0)
#define EULERGAMMA 0.57721566
#define LOGCHI2VAR (M_PI * M_PI / 6.0)
#define LOGRESBIAS 0.375

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline fp fpmax(fp a, fp b) { return a > b ? a : b; }

void o_default_aoptions(o_aoptions* o) {              /* layer0.c:27-43 */
  memset(o, 0, sizeof(*o));
  o -> thop = (fp)0.005; o -> maxnhar = 100; o -> maxnhar_e = 4;
  o -> npsd = 256; o -> nchannel = 4;
  o -> chanfreq[0] = 2000; o -> chanfreq[1] = 4000; o -> chanfreq[2] = 8000;
  o -> lip_radius = (fp)1.5; o -> f0_refine = 1; o -> hm_method = 1;
  o -> rel_winsize = 4;
}
void o_default_soptions(o_soptions* o, fp fs) {       /* layer0.c:78-87 */
  o -> fs = fs; o -> use_iczt = 1; o -> use_l1 = 0;
  o -> iczt_param_a = (fp)0.275; o -> iczt_param_b = (fp)2.26;
}

/* dsputils.c:145-169 */
void o_harmonic_czt(const fp* x, int nx, fp f0, fp fs, int nhar, fp* ampl, fp* phse) {
  fp* w = malloc(sizeof(fp) * nx);
  fp* tr = malloc(sizeof(fp) * (nhar + 1));
  fp* ti = malloc(sizeof(fp) * (nhar + 1));
  int shift = nx / 2;
  o_blackman(w, nx);
  fp winsum = 0;
  for(int i = 0; i < nx; i ++) winsum += w[i];
  for(int i = 0; i < nx; i ++) w[i] *= x[i];
  fp omega0 = (fp)(2.0 * M_PI * f0 / fs);
  o_czt(w, nx, omega0, nhar + 1, tr, ti);
  for(int i = 0; i < nhar; i ++) {
    fp ishift = (fp)(shift * 2.0 * M_PI * f0 / fs * (i + 1.0));
    fp s_re = (fp)cos((double)ishift), s_im = (fp)sin((double)ishift);
    fp d_re = tr[i + 1] * s_re - ti[i + 1] * s_im;
    fp d_im = tr[i + 1] * s_im + ti[i + 1] * s_re;
    ampl[i] = (fp)(sqrt((double)(d_re * d_re + d_im * d_im)) * 2.0 / winsum);
    phse[i] = (fp)atan2((double)d_im, (double)d_re);
  }
  free(w); free(tr); free(ti);
}

/* dsputils.c:96-115.  `weight` of cig_stft_forward is taken to be the sum of
 * the window (OUR reading; only a per-frame scale of an envelope that feeds
 * a variance, or of HMPP amplitudes). */
void o_compute_spectrogram(const fp* x, int nx, const int* center,
  const int* winsize, int nfrm, int nfft, int blackman, fp* spec, fp* phse) {
  int ns = nfft / 2 + 1;
  int standard_winsize = 1024;
  fp* w = malloc(sizeof(fp) * standard_winsize);
  if(blackman) o_blackman(w, standard_winsize); else o_hanning(w, standard_winsize);
  fp standard_normalizer = 0;
  for(int i = 0; i < standard_winsize; i ++) standard_normalizer += w[i];
  standard_normalizer *= (fp)0.5;
  free(w);
  for(int i = 0; i < nfrm; i ++) {
    o_stft_frame(x, nx, center[i], winsize[i], nfft, blackman,
      spec + (size_t)i * ns, phse ? phse + (size_t)i * ns : NULL, NULL);
    fp normalizer = standard_winsize / standard_normalizer / winsize[i];
    for(int j = 0; j < ns; j ++) spec[(size_t)i * ns + j] *= normalizer;
  }
}

/* dsputils.c:117-124 */
void o_compute_dc(const fp* x, int nx, const int* center, const int* winsize,
  int nfrm, fp* dc) {
  for(int i = 0; i < nfrm; i ++) {
    int n = winsize[i];
    fp* f = malloc(sizeof(fp) * imax(n, 1));
    o_fetch_frame(x, nx, center[i], n, f);
    fp acc = 0;
    for(int j = 0; j < n; j ++) acc += f[j];
    dc[i] = acc / n;
    free(f);
  }
}

/* parabolic peak interpolation on a log-magnitude spectrum (qifft) */
static fp qifft(const fp* s, int k, fp* freq) {
  fp a = s[k - 1], b = s[k], c = s[k + 1];
  fp a1 = (a + c) / (fp)2.0 - b;
  fp a2 = c - b - a1;
  fp x = (a1 == 0) ? 0 : - a2 / a1 * (fp)0.5;
  if(x < -1 || x > 1) x = 0;
  *freq = k + x;
  return a1 * x * x + a2 * x + b;
}

/* dsputils.c:126-143 (HMPP) */
static void harmonic_peakpicking(const fp* spectrum, const fp* phase, int nfft,
  fp fs, int nhar, fp f0, fp* ampl, fp* phse) {
  const fp tolerance = (fp)0.3;
  for(int i = 1; i <= nhar; i ++) {
    int l_idx = (int)round((double)(f0 * (i - tolerance) / fs * nfft));
    int u_idx = (int)round((double)(f0 * (i + tolerance) / fs * nfft));
    l_idx = imax(1, l_idx);
    u_idx = imin(nfft / 2 - 1, u_idx);
    int peak = l_idx;
    for(int j = l_idx; j <= u_idx; j ++) if(spectrum[j] > spectrum[peak]) peak = j;
    fp pf, pa;
    pa = qifft(spectrum, peak, & pf);
    ampl[i - 1] = (fp)exp((double)pa);
    int k = (int)pf;
    fp r = (fp)fmod((double)pf, 1.0);
    phse[i - 1] = phase[k] + (phase[k + 1] - phase[k]) * r;  /* no unwrapping */
  }
}

/* dsputils.c:318-326 */
static int get_fftsize(const fp* f0, int nfrm, fp fs, fp rel_winsize) {
  fp minf0 = 1000;
  for(int i = 0; i < nfrm; i ++) if(f0[i] > 0 && f0[i] < minf0) minf0 = f0[i];
  int max_winsize = o_idx_hwin((float)minf0, (float)fs, (float)rel_winsize);
  return o_nextpow2(max_winsize);
}

/* dsputils.c:175-228.  Results go to row i of [nfrm][stride] arrays. */
void o_harmonic_analysis(const fp* x, int nx, fp fs, const fp* f0, int nfrm,
  fp thop, fp rel_winsize, int maxnhar, int method, int stride,
  int* nhar, fp* ampl, fp* phse) {
  int nfft = get_fftsize(f0, nfrm, fs, rel_winsize);
  int ns = nfft / 2 + 1;
  for(int i = 0; i < nfrm; i ++) {
    nhar[i] = 0;
    if(! (f0[i] > 0)) continue;
    int winsize = o_idx_hwin((float)f0[i], (float)fs, (float)rel_winsize);
    int center = o_idx_center(i, (float)thop, (float)fs);
    int nh = o_idx_nhar((float)f0[i], (float)fs, maxnhar);
    nhar[i] = nh;
    fp* a = ampl + (size_t)i * stride;
    fp* p = phse + (size_t)i * stride;
    if(method == 0) {
      fp* magn = malloc(sizeof(fp) * ns);
      fp* ph = malloc(sizeof(fp) * ns);
      o_compute_spectrogram(x, nx, & center, & winsize, 1, nfft, 1, magn, ph);
      for(int j = 0; j < ns; j ++) magn[j] = (fp)log((double)magn[j] + 1e-8);
      harmonic_peakpicking(magn, ph, nfft, fs, nh, f0[i], a, p);
      free(magn); free(ph);
    } else {
      fp* xfrm = malloc(sizeof(fp) * winsize);
      o_fetch_frame(x, nx, center, winsize, xfrm);
      o_harmonic_czt(xfrm, winsize, f0[i], fs, nh, a, p);
      free(xfrm);
    }
  }
}

/* dsputils.c:230-235 */
void o_subband_energy(const fp* x, int nx, fp fmin, fp fmax, fp* y) {
  o_chebyfilt(x, nx, fmin, fmax, y);
  for(int i = 0; i < nx; i ++) y[i] *= y[i];
}

/* dsputils.c:237-265 */
void o_estimate_psd(const fp* x, int nx, int nfft, fp* psd) {
  fp* w = malloc(sizeof(fp) * nx);
  fp* re = calloc(nfft, sizeof(fp));
  fp* im = calloc(nfft, sizeof(fp));
  o_blackman(w, nx);
  fp win_power = 0;
  for(int i = 0; i < nx; i ++) {
    re[i] = w[i] * x[i];
    win_power += w[i] * w[i];
  }
  o_fft(re, im, nfft, 0);
  for(int i = 0; i < nfft / 2 + 1; i ++)
    psd[i] = (re[i] * re[i] + im[i] * im[i]) / win_power;
  free(w); free(re); free(im);
}

/* dsputils.c:328-336 */
void o_synth_harmonic_frame(const fp* ampl, const fp* phse, int nhar, fp f0, int nx, fp* y) {
  fp* freq = malloc(sizeof(fp) * imax(nhar, 1));
  for(int i = 0; i < nhar; i ++) freq[i] = (fp)(f0 * (i + 1.0));
  o_gensins(freq, ampl, phse, nhar, 1, nx, y);
  free(freq);
}

/* dsputils.c:338-351 (cos_2/sin_2 fast-math replaced by libm) */
void o_synth_harmonic_frame_iczt(const fp* ampl, const fp* phse, int nhar, fp f0, int nx, fp* y) {
  int m = imax(nhar + 1, nx);
  fp* re = calloc((size_t)m * 2, sizeof(fp));
  fp* im = re + m;
  fp omega0 = (fp)(2.0 * M_PI * f0);
  for(int i = 0; i < nhar; i ++) {
    double ph = (double)phse[i] - nx / 2 * (1.0 + i) * (double)omega0;
    re[i + 1] = (fp)((double)ampl[i] * cos(ph) * nx);
    im[i + 1] = (fp)((double)ampl[i] * sin(ph) * nx);
  }
  o_iczt(re, im, nhar + 1, omega0, nx, y);
  free(re);
}

/* llsmutils.c:45-58 (log_1 fast-math replaced by libm log) */
int o_synth_harmonic_frame_auto(const o_soptions* opt, const fp* ampl,
  const fp* phse, int nhar, fp f0, int nx, fp* y) {
  if(opt == NULL || ! opt -> use_iczt) {
    o_synth_harmonic_frame(ampl, phse, nhar, f0, nx, y);
    return 0;
  }
  if(nhar > 0 && log((double)nx) * opt -> iczt_param_a < log((double)nhar) - opt -> iczt_param_b) {
    o_synth_harmonic_frame_iczt(ampl, phse, nhar, f0, nx, y);
    return 1;
  }
  o_synth_harmonic_frame(ampl, phse, nhar, f0, nx, y);
  return 0;
}

/* dsputils.c:353-361; randn(0,1) replaced by the counter generator. */
void o_generate_white_noise(int nx, unsigned long long seed, fp* y) {
  int ntemplate = imin(20000, nx);
  for(int i = 0; i < ntemplate; i ++) y[i] = o_rng_normal(seed, (unsigned long long)i);
  for(int i = ntemplate; i < nx; i ++) y[i] = y[(i - ntemplate) % ntemplate];
}

/* dsputils.c:363-383 */
static void stretch_stationary_noise(const fp* x, int nx, int ny, int overlap, fp* y) {
  for(int i = 0; i < ny; i ++) y[i] = 0;
  for(int i = 0; i < imin(nx, ny); i ++) y[i] = x[i];
  if(ny <= nx) return;
  int head = nx;
  while(1) {
    for(int i = 0; i < overlap; i ++) {
      fp r = (fp)i / overlap;
      y[head - overlap + i] *= (fp)1.0 - r;
      y[head - overlap + i] += x[i] * r;
      y[head - overlap + i] /= (fp)sqrt((double)(2 * r * (r - 1) + 1));
    }
    for(int i = 0; i < nx - overlap; i ++) {
      if(head + i >= ny) return;
      y[head + i] = x[i + overlap];
    }
    head += nx - overlap;
  }
}

/* dsputils.c:385-394.  white_in (ntemplate+128 samples) overrides the RNG. */
void o_generate_bandlimited_noise(int nx, fp fmin, fp fmax,
  unsigned long long seed, const fp* white_in, fp* y) {
  int ntemplate = imin(20000, nx);
  int extension = 128;
  int n = ntemplate + extension;
  fp* white = malloc(sizeof(fp) * n);
  fp* colored = malloc(sizeof(fp) * n);
  if(white_in) memcpy(white, white_in, sizeof(fp) * n);
  else o_generate_white_noise(n, seed, white);
  o_chebyfilt(white, n, fmin, fmax, colored);
  stretch_stationary_noise(colored, ntemplate, nx, 128, y);
  free(white); free(colored);
}

/* dsputils.c:308-316 */
void o_spectrum_from_envelope(const fp* freq, const fp* ampl, int nfreq,
  int nspec, fp fnyq, fp* out) {
  fp* faxis = malloc(sizeof(fp) * nspec);
  for(int i = 0; i < nspec; i ++) faxis[i] = (fp)i * fnyq / nspec;
  o_interp1(freq, ampl, nfreq, faxis, nspec, out);
  free(faxis);
}

/* dsputils.c:72-94.  ciglet's ifdetector is opaque; OUR estimator: the phase
 * advance between two Hann-windowed single-bin DFTs one sample apart, window
 * span 4/fres samples (DESIGN.md "F0 refinement"). */
void o_refine_f0(const fp* x, int nx, fp fs, fp* f0, int nfrm, fp thop) {
  for(int i = 0; i < nfrm; i ++) {
    if(f0[i] == 0) continue;
    fp favg = 0; int nfavg = 0;
    for(int j = 1; j <= 3; j ++) {
      double fc = (double)f0[i] / fs * j, fres = (double)f0[i] / fs;
      int nh = (int)round(4.0 / fres);
      fp* xfrm = malloc(sizeof(fp) * nh);
      fp* w = malloc(sizeof(fp) * nh);
      o_fetch_frame(x, nx, o_idx_center(i, (float)thop, (float)fs), nh, xfrm);
      o_hanning(w, nh - 1);
      double c0r = 0, c0i = 0, c1r = 0, c1i = 0;
      for(int t = 0; t < nh - 1; t ++) {
        double ph = 2.0 * M_PI * fc * t, c = cos(ph), s = sin(ph);
        c0r += w[t] * xfrm[t] * c;     c0i -= w[t] * xfrm[t] * s;
        c1r += w[t] * xfrm[t + 1] * c; c1i -= w[t] * xfrm[t + 1] * s;
      }
      double pr = c1r * c0r + c1i * c0i, pi = c1i * c0r - c1r * c0i;
      fp f_j = (fp)(atan2(pi, pr) / (2.0 * M_PI) / j);
      if(fabs((double)(f_j - f0[i] / fs)) < f0[i] * 0.1 / fs) { favg += f_j; nfavg ++; }
      free(xfrm); free(w);
    }
    if(nfavg > 0) f0[i] = favg / nfavg * fs;
  }
}

/* layer0.c:117-146 (analysis calls it with options == NULL: sinusoid bank) */
static void synthesize_harmonics_l0(const o_soptions* opt, const o_params* p,
  fp fs, int ny, fp* y) {
  const int maxnhar = 2048;
  float thop = (float)p -> thop;
  for(int i = 0; i < ny; i ++) y[i] = 0;
  int nwin = o_idx_nwin_sin(thop, (float)fs);
  fp* w = malloc(sizeof(fp) * nwin);
  fp* yi = malloc(sizeof(fp) * nwin);
  fp* phase = malloc(sizeof(fp) * maxnhar);
  o_hanning_ola(w, nwin);
  for(int i = 0; i < p -> nfrm; i ++) {
    if(p -> f0[i] == 0) continue;
    const fp* ampl = p -> ampl + (size_t)i * p -> maxnhar;
    const fp* phse = p -> phse + (size_t)i * p -> maxnhar;
    int baseidx;
    float frac = o_idx_rawfrac(i, thop, (float)fs, & baseidx);
    fp phase_correction = (fp)((double)(fp)(frac * 2) * M_PI / fs * p -> f0[i]);
    int nhar = imin(maxnhar, p -> nhar[i]);
    for(int k = 0; k < nhar; k ++)
      phase[k] = (fp)(phse[k] - phase_correction * (k + 1.0));
    o_synth_harmonic_frame_auto(opt, ampl, phase, nhar, p -> f0[i] / fs, nwin, yi);
    for(int j = 0; j < nwin; j ++) {
      yi[j] *= w[j];
      int idx = baseidx + j - nwin / 2;
      if(idx >= 0 && idx < ny) y[idx] += yi[j];
    }
  }
  free(w); free(yi); free(phase);
}

/* Test-bench hook (tools/psd_bisect.py; not part of the restatement): called with every intermediate array of
   analyze_noise_psd so that a caller can CAPTURE it or REPLACE it in place -- e.g. run the float64 build with ONE stage
   taken from the float32 build, to find which stage's rounding moves a smoothed-PSD value.  Stages: 0 residual x_res [nx],
   1 spectrogram magnitudes [nfrm][nfft_spgm/2+1], 2 resampled log envelope [nfrm][nspec], 3 log periodogram of the
   residual [nspec][nfrm], 4 process variance Q of bin `index` [nfrm], 5 filtered means of that bin, 6 their variances,
   7 smoothed means. */
typedef void (*o_stage_hook_t)(int stage, int index, fp* data, long n);
static o_stage_hook_t g_stage_hook = NULL;
void o_set_stage_hook(o_stage_hook_t h) { g_stage_hook = h; }
#define STAGE(id, idx, ptr, n) do { if(g_stage_hook) g_stage_hook(id, idx, ptr, (long)(n)); } while(0)

/* layer0.c:318-415 */
static void analyze_noise_psd(const o_aoptions* opt, const fp* x, const fp* x_res,
  int nx, fp fs, o_params* p) {
  int nfrm = p -> nfrm;
  float thop = (float)opt -> thop;
  int nwin = o_idx_nwin_psd(thop, (float)fs);
  int nfft = o_nextpow2(nwin);
  int nspec = nfft / 2 + 1;
  int nfft_spgm = o_nextpow2(0.03 * fs);
  int ns_spgm = nfft_spgm / 2 + 1;
  fp* spgm = malloc(sizeof(fp) * (size_t)nfrm * ns_spgm);
  int* center = malloc(sizeof(int) * nfrm);
  int* winsize_spgm = malloc(sizeof(int) * nfrm);
  for(int i = 0; i < nfrm; i ++) {
    winsize_spgm[i] = o_idx_spgmwin((float)p -> f0[i], (float)fs, nwin);
    center[i] = o_idx_center(i, thop, (float)fs);
  }
  o_compute_spectrogram(x, nx, center, winsize_spgm, nfrm, nfft_spgm, 0, spgm, NULL);
  STAGE(1, 0, spgm, (size_t)nfrm * ns_spgm);
  /* The reference writes the resampled envelope back over the spectrogram row (layer0.c:339-343:
     spgm[i][j] = env[idx] * 2, j < nspec).  That row has nfft_spgm / 2 + 1 entries, so for hops long enough that
     nfft > nfft_spgm (4 thop fs > 2^ceil(log2(0.03 fs)), e.g. thop = 16 ms at 8 kHz) the reference writes past its
     row; the envelope goes to its own [nfrm][nspec] plane here, which is what the reference computes whenever it
     stays inside its buffers. */
  fp* env = malloc(sizeof(fp) * ns_spgm);
  fp* envq = malloc(sizeof(fp) * (size_t)nfrm * nspec);
  for(int i = 0; i < nfrm; i ++) {
    fp f0_scaled = (p -> f0[i] == 0 ? 200 : p -> f0[i]) / fs;
    o_spec2env(spgm + (size_t)i * ns_spgm, nfft_spgm, f0_scaled, env);
    for(int j = 0; j < nspec; j ++) {
      int idx = j * nfft_spgm / nfft;
      envq[(size_t)i * nspec + j] = env[idx] * 2;
    }
  }
  free(env); free(winsize_spgm);
  STAGE(2, 0, envq, (size_t)nfrm * nspec);

  fp* spgm_psd = malloc(sizeof(fp) * (size_t)nspec * nfrm);  /* [nspec][nfrm] */
  fp* spgm_res = malloc(sizeof(fp) * (size_t)nfrm * nspec);  /* [nfrm][nspec] */
  fp* psdvec = malloc(sizeof(fp) * nspec);
  fp* xfrm = malloc(sizeof(fp) * nwin);
  for(int i = 0; i < nfrm; i ++) {
    o_fetch_frame(x_res, nx, center[i], nwin, xfrm);
    o_estimate_psd(xfrm, nwin, nfft, psdvec);
    for(int j = 0; j < nspec; j ++)
      spgm_psd[(size_t)j * nfrm + i] = (fp)log((double)fpmax((fp)1e-10, psdvec[j]));
  }
  free(xfrm);
  STAGE(3, 0, spgm_psd, (size_t)nspec * nfrm);
  fp* Q = malloc(sizeof(fp) * nfrm); fp* R = malloc(sizeof(fp) * nfrm);
  fp* P = malloc(sizeof(fp) * nfrm);
  fp* yk = malloc(sizeof(fp) * nfrm); fp* sk = malloc(sizeof(fp) * nfrm);
  for(int i = 0; i < nfrm; i ++) R[i] = (fp)LOGCHI2VAR;
  for(int j = 0; j < nspec; j ++) {
    for(int i = 0; i < nfrm; i ++) {
      fp m1 = 0, m2 = 0;
      for(int k = -1; k <= 1; k ++) {
        int idx = imin(nfrm - 1, imax(0, i + k));
        fp v = envq[(size_t)idx * nspec + j];
        m1 += v; m2 += v * v;
      }
      Q[i] = fpmax((fp)1e-8, m2 / 3 - m1 * m1 / 9);
    }
    STAGE(4, j, Q, nfrm);
    o_kalmanf1d(spgm_psd + (size_t)j * nfrm, Q, R, nfrm, P, yk);
    STAGE(5, j, yk, nfrm); STAGE(6, j, P, nfrm);
    o_kalmans1d(yk, P, Q, nfrm, sk);
    STAGE(7, j, sk, nfrm);
    for(int i = 0; i < nfrm; i ++) {
      spgm_res[(size_t)i * nspec + j] = spgm_psd[(size_t)j * nfrm + i] - sk[i];
      spgm_psd[(size_t)j * nfrm + i] = (fp)(sk[i] + EULERGAMMA);
    }
  }
  free(P); free(Q); free(R); free(yk); free(sk);

  int npsd = opt -> npsd;
  fp* dst_axis = malloc(sizeof(fp) * npsd);
  for(int j = 0; j < npsd; j ++)
    dst_axis[j] = (fp)((fs / 2.0) * j / (npsd - 1));      /* linspace(0, fs/2, npsd) */
  fp* dst_psd = malloc(sizeof(fp) * npsd);
  fp* dst_res = malloc(sizeof(fp) * npsd);
  for(int i = 0; i < nfrm; i ++) {
    for(int j = 0; j < nspec; j ++) psdvec[j] = spgm_psd[(size_t)j * nfrm + i];
    if(o_conv_interp1u_excl()) {
      o_interp1u_excl(0, (fp)(fs / 2.0), psdvec, nspec, dst_axis, npsd, dst_psd);
      o_interp1u_excl(0, (fp)(fs / 2.0), spgm_res + (size_t)i * nspec, nspec, dst_axis, npsd, dst_res);
    } else {
      o_interp1u(0, (fp)(fs / 2.0), psdvec, nspec, dst_axis, npsd, dst_psd);
      o_interp1u(0, (fp)(fs / 2.0), spgm_res + (size_t)i * nspec, nspec, dst_axis, npsd, dst_res);
    }
    for(int j = 0; j < npsd; j ++) {
      p -> psdres[(size_t)i * npsd + j] = (fp)LOG2IN(dst_res[j]);
      fp e = (fp)exp((double)dst_psd[j]);
      p -> psd[(size_t)i * npsd + j] = (fp)(10.0 * log10((double)(e * 44100 / fs) + 1e-12));
    }
  }
  free(dst_axis); free(dst_psd); free(dst_res);
  free(spgm_psd); free(spgm_res); free(spgm); free(envq); free(center); free(psdvec);
}

/* layer0.c:417-469 */
static void analyze_noise_envelope(const o_aoptions* opt, const fp* x,
  const fp* x_res, int nx, fp fs, const fp* f0, o_params* p) {
  int nfrm = p -> nfrm, nch = opt -> nchannel, me = opt -> maxnhar_e;
  float thop = (float)opt -> thop;
  int* tmp_nhar = malloc(sizeof(int) * nfrm);
  fp* tmp_ampl = malloc(sizeof(fp) * (size_t)nfrm * imax(me, 1));
  fp* tmp_phse = malloc(sizeof(fp) * (size_t)nfrm * imax(me, 1));
  fp* tmp_dc = malloc(sizeof(fp) * nfrm);
  int* center = malloc(sizeof(int) * nfrm);
  int* nwin = malloc(sizeof(int) * nfrm);
  fp* ce = malloc(sizeof(fp) * nx);
  for(int i = 0; i < nfrm; i ++) {
    center[i] = o_idx_center(i, thop, (float)fs);
    nwin[i] = o_idx_dcwin((float)f0[i], thop, (float)fs);
  }
  for(int c = 0; c < nch; c ++) {
    fp fmin = c == 0 ? 0 : opt -> chanfreq[c - 1];
    fp fmax = c == nch - 1 ? (fp)(fs / 2.0) : opt -> chanfreq[c];
    o_subband_energy(fmin > 6000.0 ? x : x_res, nx, fmin / fs, fmax / fs, ce);
    o_harmonic_analysis(ce, nx, fs, f0, nfrm, opt -> thop, opt -> rel_winsize,
      me, opt -> hm_method, imax(me, 1), tmp_nhar, tmp_ampl, tmp_phse);
    o_compute_dc(ce, nx, center, nwin, nfrm, tmp_dc);
    for(int i = 0; i < nfrm; i ++) {
      p -> edc[(size_t)i * nch + c] = tmp_dc[i];
      if(f0[i] == 0) continue;
      p -> nhar_e[i] = tmp_nhar[i];
      for(int k = 0; k < tmp_nhar[i]; k ++) {
        p -> eenv_ampl[((size_t)i * nch + c) * me + k] = tmp_ampl[(size_t)i * me + k];
        p -> eenv_phse[((size_t)i * nch + c) * me + k] = tmp_phse[(size_t)i * me + k];
      }
    }
  }
  free(tmp_nhar); free(tmp_ampl); free(tmp_phse); free(tmp_dc);
  free(center); free(nwin); free(ce);
}

/* layer0.c:478-511 */
int o_analyze(const o_aoptions* opt, const fp* x, int nx, fp fs, fp* f0,
  int nfrm, o_params* p, fp* x_res_out) {
  p -> nfrm = nfrm; p -> maxnhar = opt -> maxnhar; p -> maxnhar_e = opt -> maxnhar_e;
  p -> npsd = opt -> npsd; p -> nchannel = opt -> nchannel;
  p -> thop = opt -> thop; p -> fnyq = (fp)(fs / 2.0);
  for(int c = 0; c < opt -> nchannel - 1; c ++) p -> chanfreq[c] = opt -> chanfreq[c];
  /* llsm_create_frame defaults, frame.c:79-88 */
  for(size_t i = 0; i < (size_t)nfrm * opt -> npsd; i ++) { p -> psd[i] = -120; p -> psdres[i] = 0; }
  for(size_t i = 0; i < (size_t)nfrm * opt -> nchannel; i ++) p -> edc[i] = (fp)1e-5;
  memset(p -> ampl, 0, sizeof(fp) * (size_t)nfrm * opt -> maxnhar);
  memset(p -> phse, 0, sizeof(fp) * (size_t)nfrm * opt -> maxnhar);
  memset(p -> eenv_ampl, 0, sizeof(fp) * (size_t)nfrm * opt -> nchannel * opt -> maxnhar_e);
  memset(p -> eenv_phse, 0, sizeof(fp) * (size_t)nfrm * opt -> nchannel * opt -> maxnhar_e);
  for(int i = 0; i < nfrm; i ++) { p -> nhar[i] = 0; p -> nhar_e[i] = 0; }

  if(opt -> f0_refine) o_refine_f0(x, nx, fs, f0, nfrm, opt -> thop);
  for(int i = 0; i < nfrm; i ++) p -> f0[i] = f0[i];

  o_harmonic_analysis(x, nx, fs, f0, nfrm, opt -> thop, opt -> rel_winsize,
    opt -> maxnhar, opt -> hm_method, opt -> maxnhar, p -> nhar, p -> ampl, p -> phse);
  fp* x_sin = malloc(sizeof(fp) * nx);
  fp* x_res = malloc(sizeof(fp) * nx);
  synthesize_harmonics_l0(NULL, p, fs, nx, x_sin);
  for(int i = 0; i < nx; i ++) x_res[i] = x[i] - x_sin[i];
  free(x_sin);
  if(x_res_out) memcpy(x_res_out, x_res, sizeof(fp) * nx);
  STAGE(0, 0, x_res, nx);

  analyze_noise_psd(opt, x, x_res, nx, fs, p);
  analyze_noise_envelope(opt, x, x_res, nx, fs, f0, p);
  free(x_res);
  return 0;
}

/* layer0.c:289-316 */
static void synthesize_noise_envelope(const o_soptions* opt, const o_params* p,
  int channel, fp fs, int ny, fp* y) {
  float thop = (float)p -> thop;
  int nch = p -> nchannel, me = p -> maxnhar_e;
  for(int i = 0; i < ny; i ++) y[i] = 0;
  int nwin = o_idx_nwin_env(thop, (float)fs);
  fp* w = malloc(sizeof(fp) * nwin);
  fp* yi = malloc(sizeof(fp) * nwin);
  o_hanning_ola(w, nwin);
  for(int i = 0; i < p -> nfrm; i ++) {
    int nhar = p -> f0[i] > 0 ? p -> nhar_e[i] : 0;
    const fp* a = p -> eenv_ampl + ((size_t)i * nch + channel) * me;
    const fp* ph = p -> eenv_phse + ((size_t)i * nch + channel) * me;
    o_synth_harmonic_frame_auto(opt, a, ph, nhar, p -> f0[i] / fs, nwin, yi);
    fp offset = p -> edc[(size_t)i * nch + channel];
    for(int j = 0; j < nwin; j ++) yi[j] = fpmax(yi[j] + offset, (fp)1e-8);
    for(int j = 0; j < nwin; j ++) {
      yi[j] *= w[j];
      int idx = o_idx_env_ola(i, j, thop, (float)fs);
      if(idx >= 0 && idx < ny) y[idx] += yi[j];
    }
  }
  free(w); free(yi);
}

/* layer0.c:535-555 */
static void synthesize_noise_excitation(const o_soptions* opt, const o_params* p,
  fp fs, int ny, unsigned long long seed, const fp* white, fp* y) {
  int nch = p -> nchannel;
  int ntpl = imin(20000, ny) + 128;
  fp* xn = malloc(sizeof(fp) * ny);
  fp* env = malloc(sizeof(fp) * ny);
  for(int i = 0; i < ny; i ++) y[i] = 0;
  for(int c = 0; c < nch; c ++) {
    fp fmin = c == 0 ? 0 : p -> chanfreq[c - 1];
    fp fmax = c == nch - 1 ? (fp)(fs / 2.0) : p -> chanfreq[c];
    if(fmin >= fs / 2.0) break;
    o_generate_bandlimited_noise(ny, fmin / fs, fmax / fs,
      seed * 16 + (unsigned long long)c, white ? white + (size_t)c * ntpl : NULL, xn);
    synthesize_noise_envelope(opt, p, c, fs, ny, env);
    for(int i = 0; i < ny; i ++) {
      xn[i] *= (fp)sqrt((double)env[i]);
      y[i] += xn[i];
    }
  }
  free(xn); free(env);
}

/* layer0.c:557-634 */
static void filter_noise(const o_params* p, fp fs, const fp* x, int nx, fp* y) {
  const int nfade = 16;
  float thop = (float)p -> thop;
  int nwin = o_idx_nwin_filt(thop, (float)fs);
  fp* w = malloc(sizeof(fp) * nwin);
  o_hanning_ola(w, nwin);
  fp wsqr = 0;
  for(int i = 0; i < nwin; i ++) wsqr += w[i] * w[i];
  int nfft = o_nextpow2(nwin * 1.2 + nfade * 2);
  int nspec = nfft / 2 + 1;
  int npsd = p -> npsd;
  fp* psd = malloc(sizeof(fp) * nspec);
  fp* envs = malloc(sizeof(fp) * nspec);
  fp* H = malloc(sizeof(fp) * nspec);
  fp* x_re = malloc(sizeof(fp) * nfft);
  fp* x_im = malloc(sizeof(fp) * nfft);
  fp* xfrm = malloc(sizeof(fp) * nwin);
  fp* src_axis = malloc(sizeof(fp) * npsd);
  fp* src_psd = malloc(sizeof(fp) * npsd);
  for(int j = 0; j < npsd; j ++) src_axis[j] = (fp)((double)p -> fnyq * j / (npsd - 1));
  for(int i = 0; i < nx; i ++) y[i] = 0;
  for(int i = 0; i < p -> nfrm; i ++) {
    const fp* npsdv = p -> psd + (size_t)i * npsd;
    fp peak = npsdv[0];
    for(int j = 1; j < npsd; j ++) if(npsdv[j] > peak) peak = npsdv[j];
    if(peak < -100) continue;
    int center = o_idx_center(i, thop, (float)fs);
    o_fetch_frame(x, nx, center, nwin, xfrm);
    for(int j = 0; j < nfft; j ++) { x_re[j] = 0; x_im[j] = 0; }
    for(int j = 0; j < nwin; j ++) x_re[j - nwin / 2 + nfft / 2] = xfrm[j] * w[j];
    o_fft(x_re, x_im, nfft, 0);
    for(int j = 0; j < nspec; j ++) psd[j] = (x_re[j] * x_re[j] + x_im[j] * x_im[j]) / wsqr;
    o_moving_avg(psd, nspec, o_conv_mavg_half(), envs);
    for(int j = 0; j < npsd; j ++) src_psd[j] = npsdv[j];
    if(p -> psdres)
      for(int j = 0; j < npsd; j ++)
        src_psd[j] += (fp)(p -> psdres[(size_t)i * npsd + j] - LOG2IN(LOGRESBIAS));
    o_spectrum_from_envelope(src_axis, src_psd, npsd, nspec - 1, (fp)(fs / 2.0), H);
    for(int j = 0; j < nspec - 1; j ++)
      H[j] = (fp)(exp(DB2LOG((double)H[j])) / sqrt((double)(envs[j] * 44100 / fs) + 1e-8));
    for(int j = 0; j < nspec - 1; j ++) { x_re[j] *= H[j]; x_im[j] *= H[j]; }
    x_re[nspec - 1] = x_re[nspec - 2];
    x_im[nspec - 1] = x_im[nspec - 2];
    for(int k = 1; k < nfft / 2; k ++) {         /* complete_symm / complete_asymm */
      x_re[nfft - k] = x_re[k];
      x_im[nfft - k] = -x_im[k];
    }
    o_fft(x_re, x_im, nfft, 1);                  /* real part kept */
    for(int j = 0; j < nfade; j ++) {
      x_re[j] *= (fp)j / nfade;
      x_re[nfft - j - 1] *= (fp)(1.0 - (fp)j / nfade);
    }
    for(int j = 0; j < nfft; j ++) {
      int idx = center + j - nfft / 2;
      if(idx >= 0 && idx < nx) y[idx] += x_re[j];
    }
  }
  free(w); free(psd); free(envs); free(H); free(x_re); free(x_im);
  free(xfrm); free(src_axis); free(src_psd);
}

/* layer0.c:636-664 (use_l1 == 0 path) */
int o_synthesize(const o_soptions* opt, const o_params* p,
  unsigned long long seed, const fp* white, fp* y, fp* y_sin, fp* y_noise) {
  fp fs = opt -> fs;
  int ny = o_idx_ny(p -> nfrm, (float)p -> thop, (float)fs);
  synthesize_harmonics_l0(opt, p, fs, ny, y_sin);
  fp* y_exc = malloc(sizeof(fp) * ny);
  synthesize_noise_excitation(opt, p, fs, ny, seed, white, y_exc);
  filter_noise(p, fs, y_exc, ny, y_noise);
  for(int i = 0; i < ny; i ++) y[i] = y_sin[i] + y_noise[i];
  free(y_exc);
  return ny;
}

/* layer0.c:636-664 with options->use_l1 = 1: the harmonic part comes from the PbP / HM cross-fade state
 * machine (l1_oracle.c), the noise part is the same */
int o_synthesize_l1(const o_soptions* opt, o_params* p, o_l1params* q, int maxnhar_conf,
  o_fgfm effect, void* effect_info, unsigned long long seed, const fp* white,
  fp* y, fp* y_sin, fp* y_noise) {
  fp fs = opt -> fs;
  int ny = o_idx_ny(p -> nfrm, (float)p -> thop, (float)fs);
  o_synthesize_harmonics_l1(opt, p, q, maxnhar_conf, effect, effect_info, y_sin, ny);
  fp* y_exc = malloc(sizeof(fp) * ny);
  synthesize_noise_excitation(opt, p, fs, ny, seed, white, y_exc);
  filter_noise(p, fs, y_exc, ny, y_noise);
  for(int i = 0; i < ny; i ++) y[i] = y_sin[i] + y_noise[i];
  free(y_exc);
  return ny;
}

/* frame.c:57-60, 152-166 applied over a chunk; layer0.c:687-706 */
static void frame_phaseshift(o_params* p, int i, fp theta) {
  for(int k = 0; k < p -> nhar[i]; k ++) {
    fp* ph = p -> phse + (size_t)i * p -> maxnhar + k;
    *ph = o_wrap((fp)(*ph + theta * (k + 1.0)));
  }
  for(int c = 0; c < p -> nchannel; c ++)
    for(int k = 0; k < p -> nhar_e[i]; k ++) {
      fp* ph = p -> eenv_phse + ((size_t)i * p -> nchannel + c) * p -> maxnhar_e + k;
      *ph = o_wrap((fp)(*ph + theta * (k + 1.0)));
    }
}
void o_chunk_phasesync_rps(o_params* p) {
  for(int i = 0; i < p -> nfrm; i ++) {
    fp ref = p -> nhar[i] > 0 ? p -> phse[(size_t)i * p -> maxnhar] : 0;
    frame_phaseshift(p, i, -ref);
  }
}
void o_chunk_phasepropagate(o_params* p, int sign) {
  fp acc = 0;
  for(int i = 0; i < p -> nfrm; i ++) {
    acc += p -> f0[i];                         /* cumsum (inclusive) */
    fp d = (fp)(acc * (p -> thop * sign * 2.0 * M_PI));
    frame_phaseshift(p, i, d);
  }
}

/* ---- CPU baseline helper for bench.py (test infrastructure, like the rest of this file):
 * analyse + resynthesise `n_utt` utterances (utterance u uses x[(u % n_distinct) * nx ..]), one
 * utterance per OpenMP thread, `nthreads` threads (1 = the single-core leg).  Returns seconds of
 * wall clock; *frames_out = frames processed.  Runs the same o_analyze / o_synthesize the parity
 * tests use (czt mode as currently set by o_set_czt_mode). */
#include <omp.h>
#include <stdlib.h>
static void bench_alloc_params(o_params* p, const o_aoptions* a, int nfrm, fp fs) {
  memset(p, 0, sizeof(*p));
  p -> nfrm = nfrm; p -> maxnhar = a -> maxnhar; p -> maxnhar_e = a -> maxnhar_e;
  p -> npsd = a -> npsd; p -> nchannel = a -> nchannel; p -> thop = a -> thop; p -> fnyq = fs / 2;
  for(int c = 0; c < 8; c ++) p -> chanfreq[c] = a -> chanfreq[c];
  int me = a -> maxnhar_e > 0 ? a -> maxnhar_e : 1;
  p -> f0 = calloc(nfrm, sizeof(fp)); p -> nhar = calloc(nfrm, sizeof(int));
  p -> ampl = calloc((size_t)nfrm * a -> maxnhar, sizeof(fp)); p -> phse = calloc((size_t)nfrm * a -> maxnhar, sizeof(fp));
  p -> psd = calloc((size_t)nfrm * a -> npsd, sizeof(fp)); p -> psdres = calloc((size_t)nfrm * a -> npsd, sizeof(fp));
  p -> edc = calloc((size_t)nfrm * a -> nchannel, sizeof(fp)); p -> nhar_e = calloc(nfrm, sizeof(int));
  p -> eenv_ampl = calloc((size_t)nfrm * a -> nchannel * me, sizeof(fp));
  p -> eenv_phse = calloc((size_t)nfrm * a -> nchannel * me, sizeof(fp));
}
static void bench_free_params(o_params* p) {
  free(p -> f0); free(p -> nhar); free(p -> ampl); free(p -> phse); free(p -> psd); free(p -> psdres);
  free(p -> edc); free(p -> nhar_e); free(p -> eenv_ampl); free(p -> eenv_phse);
}
double o_bench_anasynth(const o_aoptions* aopt, const o_soptions* sopt, const fp* x, int nx,
  int n_distinct, fp fs, const fp* f0, int nfrm, int n_utt, int nthreads, long long* frames_out) {
  if(nthreads < 1) nthreads = 1;
  const int ny = o_idx_ny(nfrm, (float)aopt -> thop, (float)sopt -> fs);
  double t0 = omp_get_wtime();
  #pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
  for(int u = 0; u < n_utt; u ++) {
    o_params p; bench_alloc_params(& p, aopt, nfrm, fs);
    fp* f = malloc(sizeof(fp) * nfrm); memcpy(f, f0, sizeof(fp) * nfrm);
    fp* y = malloc(sizeof(fp) * 3 * (size_t)(ny > 0 ? ny : 1));
    o_analyze(aopt, x + (size_t)(u % n_distinct) * nx, nx, fs, f, nfrm, & p, NULL);
    o_synthesize(sopt, & p, 1000 + (unsigned long long)u, NULL, y, y + ny, y + 2 * (size_t)ny);
    free(y); free(f); bench_free_params(& p);
  }
  double dt = omp_get_wtime() - t0;
  if(frames_out) *frames_out = (long long)n_utt * nfrm;
  return dt;
}
