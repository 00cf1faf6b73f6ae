/*
 * l1_oracle.c -- CPU restatement of libllsm2's layer-1 (source-filter) conversion and of the
 * pulse-by-pulse (PbP) harmonic synthesis.  TEST INFRASTRUCTURE ONLY; "parity unpinned" (see oracle.h: own LF model).
 *
 * Follows, function by function:
 *   layer1.c:48-84    llsm_analyze_rd          -> analyze_rd
 *   layer1.c:90-127   llsm_frame_tolayer1      -> frame_tolayer1
 *   layer1.c:129-149  llsm_chunk_tolayer1      -> o_chunk_tolayer1
 *   layer1.c:151-195  llsm_frame_tolayer0      -> o_frame_tolayer0
 *   dsputils.c:396-431 lip radiation filter    -> o_lipfilter, o_lipfilter_reim
 *   dsputils.c:433-456 llsm_harmonic_spectrum  -> o_harmonic_spectrum
 *   dsputils.c:458-484 llsm_harmonic_envelope  -> o_harmonic_envelope
 *   dsputils.c:486-510 llsm_harmonic_minphase  -> o_harmonic_minphase
 *   dsputils.c:512-579 cached glottal model + spectral fitting -> o_glottal_*
 *   dsputils.c:582-608 llsm_smoothing_filter   -> o_smoothing_filter
 *   llsmutils.c:24-43  lfmodel <-> gfm         -> o_lfmodel_to_gfm, o_gfm_to_lfmodel
 *   llsmutils.c:60-131 make_filtered_pulse_spectrum, :132-201 llsm_make_filtered_pulse
 *   layer0.c:148-287   llsm_synthesize_harmonics (use_l1 = 1 branch) -> o_synthesize_harmonics_l1
 *
 * ciglet primitives this path adds, OUR definitions (ciglet is absent and unpinned):
 *   lfmodel_from_rd   Fant 1995 Rd -> (Ra, Rk, Rg) regression, with the usual extension outside 0.21 <= Rd <= 2.7
 *   lfmodel_spectrum  analytic Fourier transform (e^{-j 2 pi f t}, t = 0 at the glottal opening) of the
 *                     LF flow-DERIVATIVE waveform; checked against numerical integration in tests
 *   minphase          cepstral folding of a log-magnitude half spectrum
 *   interp_in_blank   linear interpolation across entries equal to `blank`, nearest value at the ends
 *   itakura_saito     mean over bins of P/Pm - log(P/Pm) - 1
 *   find_minima       index of the global minimum in [lo, hi]
 *   phase_diff(a, b)  wrap(a - b)
 *   safe_aliased_sinc(T, w) = sin(T w / 2) / sin(w / 2), T at the singularity (periodic sinc, peak T)
 *   cig_spec2env      the cepstral sinc-lifter envelope o_spec2env defines (log spectrum averaged over one
 *                     f0), plus the constant that makes llsm_harmonic_envelope reproduce harmonic amplitudes:
 *                     the 3-period Hann lobes llsm_harmonic_spectrum draws peak at 1.5 a_k and their log
 *                     averages -0.53944 below the peak over one harmonic spacing, so a flat comb would come
 *                     back 0.13398 nepers (1.16 dB) low; VTMAGN is later read AT the harmonics as their
 *                     amplitude (layer1.c:177-180), so the envelope must pass through them (nhar unused)
 *   interp1u          dsputils.c:495-498 only make sense with an EXCLUSIVE right end (sample k at
 *                     x0 + k (x1 - x0) / ni): o_interp1u_excl.  (layer0.c:393 reads exactly with the
 *                     inclusive end; each call site keeps the reading that makes the reference's own
 *                     code self-consistent -- UNVERIFIED.)
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define DB2LOG(x) ((x) * 2.3025851 / 20.0)
#define LOG2DB(x) ((x) / 2.3025851 * 20.0)
/* -(log 1.5 + (1/3) int_{-1.5}^{1.5} log(sinc(x) / (1 - x^2)) dx): see "cig_spec2env" above */
#define O_SPEC2ENV_LOBE_BIAS (o_conv_lobe_bias())     /* default 0.13397922601295542; switch "spec2env_lobe_1e6" */

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------ LF model */
o_lfmodel o_lfmodel_from_rd(fp rd_, fp T0, fp Ee) {
  double rd = rd_, Rap, Rkp, Rgp;
  if(o_conv_lf_rd_clamp()) rd = rd < 0.3 ? 0.3 : (rd > 2.7 ? 2.7 : rd);   /* convention "lf_rd_clamp": Fant's fitted range only */
  if(rd < 0.21) Rap = 1e-6;
  else if(rd <= 2.7) Rap = (4.8 * rd - 1.0) / 100.0;
  else Rap = (32.3 / rd) / 100.0;
  if(rd <= 2.7) {
    Rkp = (22.4 + 11.8 * rd) / 100.0;
    Rgp = Rkp / (4.0 * (0.11 * rd / (0.5 + 1.2 * Rkp) - Rap));
  } else {
    double OQupp = 1.0 - 1.0 / (2.17 * rd);
    Rgp = 9.3552e-3 + 596e-2 / (7.96 - 2.0 * OQupp);
    Rkp = 2.0 * Rgp * OQupp - 1.04;
  }
  o_lfmodel m;
  m.T0 = T0; m.Ee = Ee;
  m.tp = (fp)(1.0 / (2.0 * Rgp));
  m.te = (fp)((1.0 / (2.0 * Rgp)) * (Rkp + 1.0));
  m.ta = (fp)Rap;
  return m;
}

o_gfm o_lfmodel_to_gfm(o_lfmodel s) {                   /* llsmutils.c:24-32 */
  o_gfm r;
  r.Fa = (fp)(1.0 / (s.ta * s.T0)); r.Rk = (s.te - s.tp) / s.tp; r.Rg = (fp)(0.5 / s.tp);
  r.T0 = s.T0; r.Ee = s.Ee;
  return r;
}
o_lfmodel o_gfm_to_lfmodel(o_gfm s) {                   /* llsmutils.c:34-43 */
  o_lfmodel r;
  r.ta = (fp)(1.0 / s.Fa / s.T0); r.tp = (fp)(0.5 / s.Rg); r.te = r.tp + r.tp * s.Rk;
  r.T0 = s.T0; r.Ee = s.Ee;
  return r;
}

/* Implicit LF parameters in seconds: epsilon (return phase), alpha (open phase growth).
 *   E(t) = E0 e^{alpha t} sin(wg t)                                  0 <= t <= Te
 *   E(t) = -(Ee / (eps Ta)) (e^{-eps (t - Te)} - e^{-eps (T0 - Te)})   Te < t <= T0
 *   eps Ta = 1 - e^{-eps (T0 - Te)};  alpha from zero net flow;  E0 = -Ee / (e^{alpha Te} sin(wg Te)). */
typedef struct { double Te, Tp, Ta, T0, Ee, wg, eps, alpha; } lf_solved;

static double lf_return_area(const lf_solved* s) {
  double D = s -> T0 - s -> Te;
  return -(s -> Ee / (s -> eps * s -> Ta)) * ((1.0 - exp(-s -> eps * D)) / s -> eps - D * exp(-s -> eps * D));
}
static double lf_open_area(const lf_solved* s, double a) {
  double sw = sin(s -> wg * s -> Te), cw = cos(s -> wg * s -> Te);
  return -s -> Ee * (a - s -> wg * cw / sw + s -> wg * exp(-a * s -> Te) / sw) / (a * a + s -> wg * s -> wg);
}
static lf_solved lf_solve(o_lfmodel m) {
  lf_solved s;
  s.T0 = m.T0; s.Te = (double)m.te * m.T0; s.Tp = (double)m.tp * m.T0; s.Ta = (double)m.ta * m.T0; s.Ee = m.Ee;
  if(s.Te > 0.999 * s.T0) s.Te = 0.999 * s.T0;
  if(s.Ta < 1e-9 * s.T0) s.Ta = 1e-9 * s.T0;
  s.wg = M_PI / s.Tp;
  double D = s.T0 - s.Te;
  /* epsilon by fixed point + Newton on g(e) = e Ta - 1 + exp(-e D) */
  double e = 1.0 / s.Ta;
  for(int it = 0; it < 100; it ++) {
    double g = e * s.Ta - 1.0 + exp(-e * D), dg = s.Ta - D * exp(-e * D);
    double step = g / dg;
    e -= step;
    if(fabs(step) < 1e-15 * fabs(e)) break;
  }
  s.eps = e;
  /* alpha: bracket a sign change of the net flow on a grid in units of 1 / Te, then bisect */
  double Ar = lf_return_area(& s);
  double lo = 0, hi = 0, flo = 0; int found = 0;
  double prev = lf_open_area(& s, -60.0 / s.Te) + Ar;
  for(int k = -59; k <= 60 && ! found; k ++) {
    double a = k / s.Te, f = lf_open_area(& s, a) + Ar;
    if((prev <= 0 && f > 0) || (prev >= 0 && f < 0)) { lo = (k - 1) / s.Te; hi = a; flo = prev; found = 1; }
    prev = f;
  }
  if(! found) { s.alpha = 0; return s; }
  for(int it = 0; it < 200; it ++) {
    double mid = 0.5 * (lo + hi), f = lf_open_area(& s, mid) + Ar;
    if((f <= 0) == (flo <= 0)) { lo = mid; flo = f; } else hi = mid;
    if(hi - lo < 1e-15 * fmax(fabs(lo), fabs(hi))) break;
  }
  s.alpha = 0.5 * (lo + hi);
  return s;
}

/* flow-derivative waveform sampled at t[] (seconds from the glottal opening); for the tests */
void o_lfmodel_waveform(o_lfmodel m, const double* t, int n, double* out) {
  lf_solved s = lf_solve(m);
  double E0 = -s.Ee / (exp(s.alpha * s.Te) * sin(s.wg * s.Te));
  for(int i = 0; i < n; i ++) {
    if(t[i] < 0 || t[i] > s.T0) out[i] = 0;
    else if(t[i] <= s.Te) out[i] = E0 * exp(s.alpha * t[i]) * sin(s.wg * t[i]);
    else out[i] = -(s.Ee / (s.eps * s.Ta)) * (exp(-s.eps * (t[i] - s.Te)) - exp(-s.eps * (s.T0 - s.Te)));
  }
}

void o_lfmodel_spectrum(o_lfmodel m, const fp* freq, int nf, fp* magn, fp* phase) {
  lf_solved s = lf_solve(m);
  double sw = sin(s.wg * s.Te), cw = cos(s.wg * s.Te), D = s.T0 - s.Te;
  for(int i = 0; i < nf; i ++) {
    double complex sj = I * 2.0 * M_PI * (double)freq[i];
    double complex as = s.alpha - sj;
    /* open phase; E0 e^{alpha Te} = -Ee / sin(wg Te) keeps the exponentials bounded */
    double complex O = (-s.Ee / sw) * (cexp(-sj * s.Te) * (as * sw - s.wg * cw) + s.wg * exp(-s.alpha * s.Te)) /
                       (as * as + s.wg * s.wg);
    double complex R;
    if(freq[i] == 0)
      R = -(s.Ee / (s.eps * s.Ta)) * ((1.0 - exp(-s.eps * D)) / s.eps - D * exp(-s.eps * D));
    else
      R = -(s.Ee / (s.eps * s.Ta)) * cexp(-sj * s.Te) *
          ((1.0 - cexp(-(s.eps + sj) * D)) / (s.eps + sj) - exp(-s.eps * D) * (1.0 - cexp(-sj * D)) / sj);
    double complex G = O + R;
    if(magn) magn[i] = (fp)cabs(G);
    if(phase) phase[i] = (fp)carg(G);
  }
}

/* ------------------------------------------------------------------ small primitives */
void o_interp1u_excl(fp x0, fp x1, const fp* yi, int ni, const fp* xq, int nq, fp* yq) {
  for(int q = 0; q < nq; q ++) {
    fp pos = (xq[q] - x0) / (x1 - x0) * ni;
    int k = (int)floor((double)pos);
    if(k < 0) { yq[q] = yi[0]; continue; }
    if(k >= ni - 1) { yq[q] = yi[ni - 1]; continue; }
    fp r = pos - k;
    yq[q] = yi[k] + (yi[k + 1] - yi[k]) * r;
  }
}

void o_interp_in_blank(const fp* x, int n, fp blank, fp* y) {
  int prev = -1;
  for(int i = 0; i < n; i ++) y[i] = x[i];
  for(int i = 0; i < n; i ++) {
    if(x[i] == blank) continue;
    if(prev < 0) for(int j = 0; j < i; j ++) y[j] = x[i];
    else for(int j = prev + 1; j < i; j ++) y[j] = x[prev] + (x[i] - x[prev]) * (fp)(j - prev) / (fp)(i - prev);
    prev = i;
  }
  if(prev >= 0) for(int j = prev + 1; j < n; j ++) y[j] = x[prev];
}

void o_minphase(const fp* logmag, int nfft, fp* phase) {
  fp* re = malloc(sizeof(fp) * nfft); fp* im = calloc(nfft, sizeof(fp));
  for(int i = 0; i <= nfft / 2; i ++) re[i] = logmag[i];
  for(int i = 1; i < nfft / 2; i ++) re[nfft - i] = logmag[i];
  o_fft(re, im, nfft, 1);                                  /* real cepstrum */
  for(int i = 1; i < nfft / 2; i ++) { re[i] *= 2; re[nfft - i] = 0; }
  for(int i = 0; i < nfft; i ++) im[i] = 0;
  o_fft(re, im, nfft, 0);
  for(int i = 0; i <= nfft / 2; i ++) phase[i] = im[i];
  free(re); free(im);
}

static fp aliased_sinc(int T, fp w) {
  double d = sin(0.5 * (double)w);
  if(fabs(d) < 1e-12) return (fp)T;
  return (fp)(sin(0.5 * T * (double)w) / d);
}

/* ------------------------------------------------------------------ dsputils.c:396-510 */
void o_lipfilter(fp radius, fp f0, int nhar, fp* ampl, fp* phse, int inverse) {
  fp Rr = (fp)(128.0 / 9.0 / M_PI / M_PI);
  fp Lr = (fp)(8.0 * radius / 100.0 / 3.0 / M_PI / 340.0);
  for(int i = 0; i < nhar; i ++) {
    fp omega = (fp)(f0 * (1.0 + i) * 2.0 * M_PI);
    double complex r = I * ((double)(omega * Lr * Rr) / ((double)Rr + I * (double)(omega * Lr)));
    if(inverse) {
      if(ampl) ampl[i] /= (fp)cabs(r);
      if(phse) phse[i] -= (fp)carg(r);
    } else {
      if(ampl) ampl[i] *= (fp)cabs(r);
      if(phse) phse[i] += (fp)carg(r);
    }
  }
}
void o_lipfilter_reim(fp radius, fp f0, int nhar, fp* re, fp* im, int inverse) {
  fp Rr = (fp)(128.0 / 9.0 / M_PI / M_PI);
  fp Lr = (fp)(8.0 * radius / 100.0 / 3.0 / M_PI / 340.0);
  for(int i = 0; i < nhar; i ++) {
    fp omega = (fp)(f0 * (1.0 + i) * 2.0 * M_PI);
    double complex r = I * ((double)(omega * Lr * Rr) / ((double)Rr + I * (double)(omega * Lr)));
    double complex y = (double)re[i] + I * (double)im[i];
    y = inverse ? y / r : y * r;
    re[i] = (fp)creal(y); im[i] = (fp)cimag(y);
  }
}

void o_harmonic_spectrum(const fp* ampl, int nhar, fp f0, int nfft, fp* X) {
  int nX = nfft / 2 + 1;
  int T = (int)(3.0 / f0);
  int width = (int)ceil((double)(f0 * nfft * 1.5));
  for(int j = 0; j < nX; j ++) X[j] = 0;
  for(int i = 0; i < nhar; i ++) {
    fp ifreq = (fp)(f0 * (1.0 + i));
    int center = (int)round((double)(ifreq * nfft));
    for(int j = imax(0, center - width); j < imin(nX, center + width + 1); j ++) {
      fp omega = (fp)(((fp)j / nfft - ifreq) * 2.0 * M_PI);
      fp resp = (fp)(0.5 * aliased_sinc(T, omega) + 0.25 * aliased_sinc(T, (fp)(omega - 2.0 * M_PI / T)) +
                     0.25 * aliased_sinc(T, (fp)(omega + 2.0 * M_PI / T)));
      X[j] = (fp)fmax((double)X[j], (double)(resp * ampl[i]));
    }
  }
  for(int j = 0; j < nX; j ++) X[j] *= f0;
}

static fp compress_logspectrum(fp x) { return x > -10 ? x : (fp)((x + 10.0) / 2 - 10.0); }
static fp decompress_logspectrum(fp x) { return x > -10 ? x : (fp)((x + 10.0) * 2 - 10.0); }

void o_harmonic_envelope(const fp* ampl, int nhar, fp f0, int nfft, fp* env_db) {
  int nX = nfft / 2 + 1;
  fp* ca = malloc(sizeof(fp) * imax(nhar, 1));
  fp mx = nhar > 0 ? ampl[0] : 1;
  for(int i = 1; i < nhar; i ++) if(ampl[i] > mx) mx = ampl[i];
  fp peak = (fp)log((double)mx);
  for(int i = 0; i < nhar; i ++) ca[i] = (fp)exp((double)compress_logspectrum((fp)(log((double)ampl[i]) - peak)));
  fp* X = malloc(sizeof(fp) * nX);
  o_harmonic_spectrum(ca, nhar, f0, nfft, X);
  o_spec2env(X, nfft, f0, env_db);                         /* cig_spec2env(X, nfft, f0, nhar, NULL) */
  for(int i = 0; i < nX; i ++) env_db[i] += (fp)O_SPEC2ENV_LOBE_BIAS;
  for(int i = 0; i < nX; i ++) env_db[i] = (fp)LOG2DB(decompress_logspectrum(env_db[i]) + peak);
  free(ca); free(X);
}

int o_minphase_fftsize(int nhar) {
  return imax(64, (int)pow(2.0, ceil(log2((double)nhar) + 2)));
}

void o_harmonic_minphase(const fp* ampl, int nhar, fp* har_phse_out) {
  int nfft = o_minphase_fftsize(nhar);
  int ns = nfft / 2 + 1;
  fp* har_idx = calloc(nhar + 1, sizeof(fp)); fp* har_ampl = calloc(nhar + 1, sizeof(fp));
  fp* fft_idx = calloc(ns, sizeof(fp)); fp* spectrum = calloc(ns, sizeof(fp)); fp* sphase = calloc(ns, sizeof(fp));
  fp* hp = calloc(nhar + 1, sizeof(fp));
  for(int i = 0; i < nhar; i ++) {
    har_idx[i + 1] = (fp)((i + 1.0) / (nhar + 1.0) * nfft / 2.0);
    har_ampl[i + 1] = (fp)log((double)ampl[i] + 1e-10);
  }
  har_ampl[0] = har_ampl[1];
  for(int i = 0; i < ns; i ++) fft_idx[i] = (fp)i;
  o_interp1u_excl(0, har_idx[nhar] * 2 - har_idx[nhar - 1], har_ampl, nhar + 1, fft_idx, ns, spectrum);
  o_minphase(spectrum, nfft, sphase);
  o_interp1u_excl(0, (fp)(nfft / 2 + 1), sphase, ns, har_idx, nhar + 1, hp);
  for(int i = 1; i < nhar; i ++) hp[i - 1] = hp[i];          /* dsputils.c:505-506 (sic: i < nhar) */
  for(int i = 0; i < nhar; i ++) har_phse_out[i] = hp[i];
  free(har_idx); free(har_ampl); free(fft_idx); free(spectrum); free(sphase); free(hp);
}

/* ------------------------------------------------------------------ dsputils.c:512-579 */
typedef struct { fp** power; fp* param; int nhar, nresp; } glottal_cache;

void* o_glottal_create(const fp* param, int nparam, int nhar) {
  glottal_cache* g = malloc(sizeof(glottal_cache));
  g -> nresp = nparam; g -> nhar = nhar;
  g -> power = calloc(nparam, sizeof(fp*)); g -> param = calloc(nparam, sizeof(fp));
  fp f0 = 200.0;
  fp* freq = calloc(nhar, sizeof(fp));
  for(int i = 0; i < nhar; i ++) freq[i] = (fp)(f0 * (1.0 + i));
  for(int i = 0; i < nparam; i ++) {
    g -> param[i] = param[i];
    o_lfmodel lf = o_lfmodel_from_rd(param[i], (fp)(1.0 / f0), 1.0);
    g -> power[i] = calloc(nhar, sizeof(fp));
    o_lfmodel_spectrum(lf, freq, nhar, g -> power[i], NULL);
    for(int j = 0; j < nhar; j ++) { g -> power[i][j] /= (fp)(j + 1.0); g -> power[i][j] *= g -> power[i][j]; }
  }
  free(freq);
  return g;
}
void o_glottal_delete(void* g_) {
  glottal_cache* g = g_;
  if(! g) return;
  for(int i = 0; i < g -> nresp; i ++) free(g -> power[i]);
  free(g -> power); free(g -> param); free(g);
}
static fp qifft_parabola(const fp* s, int k, fp* pos) {      /* 3-point parabola through k-1, k, k+1 */
  fp a = s[k - 1], b = s[k], c = s[k + 1];
  fp den = a - 2 * b + c;
  fp d = den == 0 ? 0 : (fp)(0.5 * (a - c) / den);
  *pos = k + d;
  return (fp)(b - 0.25 * (a - c) * d);
}
fp o_glottal_fit(const fp* ampl, int nhar, void* g_) {
  glottal_cache* g = g_;
  nhar = imin(nhar, g -> nhar);
  fp* power = calloc(imax(nhar, 1), sizeof(fp)); fp* pm = calloc(imax(nhar, 1), sizeof(fp));
  fp* dist = calloc(g -> nresp, sizeof(fp));
  for(int i = 0; i < nhar; i ++) power[i] = ampl[i] * ampl[i];
  for(int i = 0; i < g -> nresp; i ++) {
    fp gain = power[0] / g -> power[i][0];
    double is = 0;
    for(int j = 0; j < nhar; j ++) {
      pm[j] = g -> power[i][j] * gain;
      double r = (double)power[j] / (double)pm[j];
      is += r - log(r) - 1.0;
    }
    dist[i] = (fp)exp(is / nhar);
  }
  int valley = 0;
  for(int i = 1; i < g -> nresp; i ++) if(dist[i] < dist[valley]) valley = i;
  fp refined = g -> param[valley];
  if(valley > 0 && valley < g -> nresp - 1) {
    qifft_parabola(dist, valley, & refined);
    int k = (int)refined;
    refined = g -> param[k] + (g -> param[k + 1] - g -> param[k]) * (fp)fmod((double)refined, 1.0);
  }
  free(power); free(pm); free(dist);
  return refined;
}

void o_smoothing_filter(const fp* x, int nx, int order, fp* y) {     /* dsputils.c:582-608 */
  if(nx < order) { memcpy(y, x, sizeof(fp) * nx); return; }
  fp m0 = 0, m1 = 0;
  for(int i = 0; i < order; i ++) { m0 += x[i]; m1 += x[nx - order + i]; }
  m0 /= order; m1 /= order;
  for(int i = 0; i < nx; i ++) y[i] = 0;
  for(int i = 0; i < order / 2; i ++) { y[i] = m0; y[nx - i - 1] = m1; }
  for(int i = order / 2; i < nx - order / 2; i ++) {
    int lo = i - order / 2, hi = lo + order;
    fp mean = 0;
    for(int j = lo; j < hi; j ++) mean += x[j];
    mean /= order;
    int npos = 0, nneg = 0; fp dtot = 0;
    for(int j = lo; j < hi; j ++) {
      npos += x[j] >= mean; nneg += x[j] <= mean;
      dtot += x[j] - mean > 0 ? x[j] - mean : 0;
    }
    y[i] = mean + (npos - nneg) * dtot / order / order;
  }
}

/* ------------------------------------------------------------------ layer1.c */
static void lf_vs_ampl(fp rd, fp f0, int nhar, fp* vs_ampl) {   /* layer1.c:104-107, 172-175 */
  fp* freq = calloc(imax(nhar, 1), sizeof(fp));
  for(int i = 0; i < nhar; i ++) freq[i] = (fp)(f0 * (i + 1.0));
  o_lfmodel glott = o_lfmodel_from_rd(rd, (fp)(1.0 / f0), 1.0);
  o_lfmodel_spectrum(glott, freq, nhar, vs_ampl, NULL);
  for(int i = 1; i < nhar; i ++) vs_ampl[i] /= (fp)((1.0 + i) * vs_ampl[0]);
  if(nhar > 0) vs_ampl[0] = 1.0;
  free(freq);
}

void o_analyze_rd(const o_params* p, fp lip_radius, fp* rd_smooth) {     /* layer1.c:48-84 */
  int nfrm = p -> nfrm, ncand = 64;
  fp* rd_list = malloc(sizeof(fp) * ncand);
  for(int i = 0; i < ncand; i ++) rd_list[i] = (fp)(0.02 + (3.0 - 0.02) * i / (ncand - 1));
  void* cgm = o_glottal_create(rd_list, ncand, 80);
  fp* rd = calloc(nfrm, sizeof(fp)); fp* cont = calloc(nfrm, sizeof(fp));
  for(int i = 0; i < nfrm; i ++) {
    fp f0 = p -> f0[i];
    if(f0 == 0) continue;
    int nhar = imin(p -> nhar[i], (int)round(8000.0 / (double)f0));
    fp* ampl = calloc(imax(nhar, 1), sizeof(fp));
    memcpy(ampl, p -> ampl + (size_t)i * p -> maxnhar, sizeof(fp) * nhar);
    o_lipfilter(lip_radius, f0, nhar, ampl, NULL, 1);
    rd[i] = o_glottal_fit(ampl, nhar, cgm);
    free(ampl);
  }
  o_glottal_delete(cgm); free(rd_list);
  o_interp_in_blank(rd, nfrm, 0, cont);
  o_smoothing_filter(cont, nfrm, (int)round(0.02 / (double)p -> thop), rd_smooth);
  free(rd); free(cont);
}

static void frame_tolayer1(const o_params* p, o_l1params* q, int i, int nfft) {   /* layer1.c:90-127 */
  int nspec = nfft / 2 + 1, nhar = p -> nhar[i];
  fp rd = q -> rd[i], f0 = p -> f0[i];
  fp* ampl = calloc(imax(nhar, 1), sizeof(fp)); fp* phse = calloc(imax(nhar, 1), sizeof(fp));
  fp* vs_ampl = calloc(imax(nhar, 1), sizeof(fp)); fp* vt_phse = calloc(imax(nhar, 1), sizeof(fp));
  memcpy(ampl, p -> ampl + (size_t)i * p -> maxnhar, sizeof(fp) * nhar);
  memcpy(phse, p -> phse + (size_t)i * p -> maxnhar, sizeof(fp) * nhar);
  lf_vs_ampl(rd, f0, nhar, vs_ampl);
  o_lipfilter(q -> lip_radius, f0, nhar, ampl, phse, 1);
  for(int k = 0; k < nhar; k ++) ampl[k] /= vs_ampl[k];
  o_harmonic_minphase(ampl, nhar, vt_phse);
  for(int k = 0; k < nhar; k ++) q -> vsphse[(size_t)i * q -> maxnhar + k] = phse[k] - vt_phse[k];
  q -> nvsphse[i] = nhar;
  o_harmonic_envelope(ampl, nhar, (fp)(f0 / p -> fnyq / 2.0), nfft, q -> vtmagn + (size_t)i * nspec);
  q -> has_l1[i] = 1;
  free(ampl); free(phse); free(vs_ampl); free(vt_phse);
}

void o_chunk_tolayer1(const o_params* p, o_l1params* q, int nfft) {       /* layer1.c:129-149 */
  q -> nspec = nfft / 2 + 1;
  o_analyze_rd(p, q -> lip_radius, q -> rd);
  for(int i = 0; i < p -> nfrm; i ++) {
    q -> has_l1[i] = 0; q -> nvsphse[i] = 0;
    if(p -> f0[i] == 0) continue;
    frame_tolayer1(p, q, i, nfft);
  }
}

/* layer1.c:151-195; writes row i of p (nhar / ampl / phse).  maxnhar_conf < 0: conf has no MAXNHAR. */
void o_frame_tolayer0(o_params* p, const o_l1params* q, int i, int maxnhar_conf) {
  fp f0 = p -> f0[i];
  if(f0 == 0 || ! q -> has_l1[i]) return;
  int nspec = q -> nspec, nhar = q -> nvsphse[i];
  if(maxnhar_conf >= 0) nhar = imin(nhar, maxnhar_conf);
  nhar = imin(nhar, (int)(p -> fnyq / f0));
  nhar = imin(nhar, p -> maxnhar);                          /* row width of the flat layout */
  const fp* spec_env = q -> vtmagn + (size_t)i * nspec;
  const fp* vs_phse = q -> vsphse + (size_t)i * q -> maxnhar;
  fp* freq = calloc(imax(nhar, 1), sizeof(fp)); fp* vs_ampl = calloc(imax(nhar, 1), sizeof(fp));
  fp* faxis = calloc(nspec, sizeof(fp)); fp* vt_ampl = calloc(imax(nhar, 1), sizeof(fp));
  fp* vt_phse = calloc(imax(nhar, 1), sizeof(fp));
  for(int k = 0; k < nhar; k ++) freq[k] = (fp)(f0 * (k + 1.0));
  lf_vs_ampl(q -> rd[i], f0, nhar, vs_ampl);
  for(int k = 0; k < nspec; k ++) faxis[k] = (fp)((double)p -> fnyq * k / (nspec - 1));
  o_interp1(faxis, spec_env, nspec, freq, nhar, vt_ampl);
  for(int k = 0; k < nhar; k ++) vt_ampl[k] = (fp)exp(DB2LOG((double)vt_ampl[k]));
  if(nhar > 0) o_harmonic_minphase(vt_ampl, nhar, vt_phse);
  fp* a = p -> ampl + (size_t)i * p -> maxnhar; fp* ph = p -> phse + (size_t)i * p -> maxnhar;
  for(int k = 0; k < nhar; k ++) { a[k] = vt_ampl[k] * vs_ampl[k]; ph[k] = vt_phse[k] + vs_phse[k]; }
  o_lipfilter(q -> lip_radius, f0, nhar, a, ph, 0);
  p -> nhar[i] = nhar;
  q -> has_hm[i] = 1;
  free(freq); free(vs_ampl); free(faxis); free(vt_ampl); free(vt_phse);
}

void o_chunk_tolayer0(o_params* p, const o_l1params* q, int maxnhar_conf) {
  for(int i = 0; i < p -> nfrm; i ++) o_frame_tolayer0(p, q, i, maxnhar_conf);
}

/* frame.c:152-166 on the layer-1 member: vs_phse[k] = wrap(vs_phse[k] + theta (k + 1)) */
void o_l1_phaseshift(o_l1params* q, int i, fp theta) {
  fp* v = q -> vsphse + (size_t)i * q -> maxnhar;
  for(int k = 0; k < q -> nvsphse[i]; k ++) v[k] = o_wrap((fp)(v[k] + theta * (k + 1.0)));
}

/* ------------------------------------------------------------------ llsmutils.c:60-201 */
static void complete_symm(fp* x, int n) { for(int k = 1; k < n / 2; k ++) x[n - k] = x[k]; }
static void complete_asymm(fp* x, int n) { for(int k = 1; k < n / 2; k ++) x[n - k] = -x[k]; }

static void make_filtered_pulse_spectrum(fp rd, fp f0, const fp* vsphse, o_lfmodel source, fp phase_shift,
  int size, fp fnyq, const fp* vt_harphse, int nhar, const fp* freq_axis, fp* dst_re, fp* dst_im) {
  int halfsize = size / 2 + 1;
  fp* freq_har = calloc(nhar + 1, sizeof(fp)); fp* phse_har = calloc(nhar + 1, sizeof(fp));
  for(int i = 0; i <= nhar; i ++) freq_har[i] = i * f0;
  o_lfmodel source_orig = o_lfmodel_from_rd(rd, (fp)(1.0 / f0), 1.0);
  o_lfmodel_spectrum(source_orig, freq_har + 1, nhar, NULL, phse_har + 1);
  fp vsshift = (fp)(vsphse[0] - (phse_har[1] - 0.5 * M_PI));
  for(int i = 1; i <= nhar; i ++) {
    phse_har[i] -= (fp)(0.5 * M_PI);
    phse_har[i] = o_wrap(vsphse[i - 1] - phse_har[i] - vsshift * i);
  }
  for(int i = 0; i < nhar; i ++) phse_har[i + 1] += vt_harphse[i];
  fp* pre = calloc(nhar + 1, sizeof(fp)); fp* pim = calloc(nhar + 1, sizeof(fp));
  for(int i = 0; i < nhar + 1; i ++) { pre[i] = (fp)cos((double)phse_har[i]); pim[i] = (fp)sin((double)phse_har[i]); }
  fp* dre = calloc(halfsize, sizeof(fp)); fp* delta = calloc(halfsize, sizeof(fp));
  o_interp1(freq_har, pre, nhar + 1, freq_axis, halfsize, dre);
  o_interp1(freq_har, pim, nhar + 1, freq_axis, halfsize, delta);
  for(int i = 0; i < halfsize; i ++) delta[i] = (fp)atan2((double)delta[i], (double)dre[i]);
  fp* lfphse = calloc(halfsize, sizeof(fp)); fp* lfmagn = calloc(halfsize, sizeof(fp));
  fp lfmagnf0 = 0;
  o_lfmodel_spectrum(source_orig, & f0, 1, & lfmagnf0, NULL);
  o_lfmodel_spectrum(source, freq_axis, halfsize, lfmagn, lfphse);
  lfmagn[0] = 0; lfphse[0] = 0;
  for(int i = 1; i < halfsize; i ++) {
    lfmagn[i] *= (fnyq / freq_axis[i]) / lfmagnf0;
    lfphse[i] += (fp)(phase_shift * i * 2 * M_PI / size);
    lfphse[i] += (fp)(delta[i] - 0.5 * M_PI);
    dst_re[i] += (fp)(lfmagn[i] * cos((double)lfphse[i]));
    dst_im[i] += (fp)(lfmagn[i] * sin((double)lfphse[i]));
  }
  free(freq_har); free(phse_har); free(pre); free(pim); free(dre); free(delta); free(lfphse); free(lfmagn);
}

/* y: `size` samples.  vtmagn: nspec dB values on linspace(0, fnyq, nspec); vsphse: nhar values. */
void o_make_filtered_pulse(fp rd, fp f0, const fp* vtmagn, int nspec, const fp* vsphse, int nhar,
  const o_lfmodel* sources, const fp* offsets, int num_pulses, int pre_rotate, int size, fp fnyq,
  fp lip_radius, fp fs, fp* y) {
  int halfsize = size / 2 + 1;
  fp* freq_axis = calloc(size, sizeof(fp)); fp* re = calloc(size, sizeof(fp)); fp* im = calloc(size, sizeof(fp));
  for(int i = 0; i < halfsize; i ++) freq_axis[i] = i * fs / size;
  fp* vtaxis = calloc(nspec, sizeof(fp));
  for(int k = 0; k < nspec; k ++) vtaxis[k] = (fp)((double)fnyq * k / (nspec - 1));
  fp* freq_har = calloc(nhar + 1, sizeof(fp));
  for(int i = 0; i <= nhar; i ++) freq_har[i] = i * f0;
  fp* vtamplhar = calloc(imax(nhar, 1), sizeof(fp)); fp* vt_phse = calloc(imax(nhar, 1), sizeof(fp));
  o_interp1(vtaxis, vtmagn, nspec, freq_har + 1, nhar, vtamplhar);
  for(int i = 0; i < nhar; i ++) vtamplhar[i] = (fp)exp(DB2LOG((double)vtamplhar[i]));
  o_harmonic_minphase(vtamplhar, nhar, vt_phse);
  for(int i = 0; i < num_pulses; i ++)
    make_filtered_pulse_spectrum(rd, f0, vsphse, sources[i], -offsets[i] - pre_rotate, size, fnyq, vt_phse, nhar,
      freq_axis, re, im);
  o_lipfilter_reim(lip_radius, fs / size, halfsize, re, im, 0);
  fp* vts = calloc(halfsize, sizeof(fp));
  o_interp1(vtaxis, vtmagn, nspec, freq_axis, halfsize, vts);
  for(int i = 0; i < halfsize; i ++) { fp g = (fp)exp(DB2LOG((double)vts[i])); re[i] *= g; im[i] *= g; }
  complete_symm(re, size); complete_asymm(im, size);
  o_fft(re, im, size, 1);
  int fadein = imin(256, pre_rotate), fadeout = imin(256, size);
  for(int i = 0; i < size; i ++) y[i] = re[i];
  for(int i = 0; i < fadein; i ++) y[i] *= (fp)i / fadein;
  for(int i = size - fadeout; i < size; i ++) y[i] *= (fp)(size - i) / fadeout;
  free(freq_axis); free(re); free(im); free(vtaxis); free(freq_har); free(vtamplhar); free(vt_phse); free(vts);
}

/* ------------------------------------------------------------------ layer0.c:148-287, use_l1 = 1 */
/* First-harmonic pulse tracker of one frame (layer0.c:181-198 / llsmrt.c:316-333): where the next glottal
 * cycle begins, relative to `origin` (samples). */
fp o_pulse_projection(fp rd, fp f0, fp vsphse0, fp fs, fp origin) {
  fp len_period = fs / f0;
  o_lfmodel sm = o_lfmodel_from_rd(rd, (fp)(1.0 / f0), 1.0);
  fp source_p0 = 0;
  o_lfmodel_spectrum(sm, & f0, 1, NULL, & source_p0);
  source_p0 -= (fp)(0.5 * M_PI);
  fp p0 = o_wrap(vsphse0);
  fp p0_dist = o_wrap(source_p0 - p0);                     /* phase_diff(source_p0, p0) */
  if(p0_dist < 0) p0_dist += (fp)(2.0 * M_PI);
  return (fp)(origin + p0_dist / 2 / M_PI * len_period);
}

void o_synthesize_harmonics_l1(const o_soptions* opt, o_params* p, o_l1params* q, int maxnhar_conf,
  o_fgfm effect, void* effect_info, fp* y_mix, int ny) {
  const fp fs = opt -> fs, thop = p -> thop;
  const int nfrm = p -> nfrm, maxnhar = 2048;
  fp* y_hm = calloc(ny, sizeof(fp)); fp* y_pbp = calloc(ny, sizeof(fp));
  for(int i = 0; i < ny; i ++) y_mix[i] = 0;
  int nwin = o_idx_nwin_sin((float)thop, (float)fs);
  fp* w = malloc(sizeof(fp) * nwin);
  o_hanning_ola(w, nwin);
  fp pulse_previous = 0; int pbp_periods = 0; const int pbp_periods_thrd = 3;
  fp pbp_switch_rate = 0, pbp_switch_state = 0; int baseidx_prev = 0;
  for(int i = 0; i < nfrm; i ++) {
    fp f0 = p -> f0[i];
    if(f0 == 0) continue;
    int baseidx = (int)((float)((float)((float)i * (float)thop) * (float)fs));   /* int baseidx = i * thop * fs; */
    if(! q -> has_l1[i]) continue;
    const fp* vsphse = q -> vsphse + (size_t)i * q -> maxnhar;
    const fp* vtmagn = q -> vtmagn + (size_t)i * q -> nspec;
    int pbp_on = q -> pbpsyn[i] == 1;
    int nspec = q -> nspec;
    fp len_period = fs / f0;
    o_lfmodel source_model = o_lfmodel_from_rd(q -> rd[i], (fp)(1.0 / f0), 1.0);
    fp pulse_projected = o_pulse_projection(q -> rd[i], f0, vsphse[0], fs, (fp)baseidx);
    int len_reset = (int)(fmax((double)len_period, (double)(thop * fs)) * 2);
    if(pulse_projected - pulse_previous > len_reset) pulse_previous = pulse_projected - len_reset;
    int num_periods = (int)round((double)((pulse_projected - pulse_previous) / len_period));
    len_period = (pulse_projected - pulse_previous) / num_periods;
    if(pbp_on || pbp_periods > 0) {
      if(num_periods > 0) {
        int pulse_size = (int)pow(2.0, ceil(log2(fmax((double)len_period * 2, (double)nspec))));
        fp* offsets = calloc(num_periods, sizeof(fp));
        o_lfmodel* sources = calloc(num_periods, sizeof(o_lfmodel));
        for(int j = 0; j < num_periods; j ++) {
          fp delta_t = 0;
          if(effect != NULL && q -> has_eff[i]) {
            o_gfm g = o_lfmodel_to_gfm(source_model);
            effect(& g, & delta_t, effect_info, i);
            sources[j] = o_gfm_to_lfmodel(g);
          } else sources[j] = source_model;
          offsets[j] = pulse_previous + j * len_period + delta_t * fs;
        }
        int pulse_base = (int)offsets[0];
        for(int j = 0; j < num_periods; j ++) offsets[j] -= pulse_base;
        fp* yp = calloc(pulse_size, sizeof(fp));
        o_make_filtered_pulse(q -> rd[i], f0, vtmagn, nspec, vsphse, q -> nvsphse[i], sources, offsets, num_periods,
          (int)len_period, pulse_size, p -> fnyq, q -> lip_radius, fs, yp);
        for(int k = 0; k < pulse_size; k ++) {
          int idx = (int)(pulse_base + k - len_period);
          if(idx >= 0 && idx < ny) y_pbp[idx] += yp[k];
        }
        free(yp); free(offsets); free(sources);
        pbp_periods += pbp_on ? num_periods : -num_periods;
        pbp_periods = imin(pbp_periods, pbp_periods_thrd);
        pbp_periods = imax(pbp_periods, 0);
      }
    }
    pulse_previous = pulse_projected;
    {
      fp hop = thop * fs;
      pbp_switch_rate = (fp)(1.0 / (len_period < hop ? len_period : hop));
    }
    int require_hm = 0;
    if(pbp_on && pbp_periods == pbp_periods_thrd) {
      for(int j = baseidx_prev; j < baseidx; j ++) {
        if(pbp_switch_state < 1.0) { pbp_switch_state += pbp_switch_rate; require_hm = 1; }
        if(j >= 0 && j < ny) y_mix[j] = pbp_switch_state;
      }
    } else if(! pbp_on && pbp_periods == 0) {
      for(int j = baseidx_prev; j < baseidx; j ++) {
        if(pbp_switch_state > 0) { pbp_switch_state -= pbp_switch_rate; require_hm = 1; }
        if(j >= 0 && j < ny) y_mix[j] = pbp_switch_state;
      }
    } else {
      for(int j = baseidx_prev; j < baseidx; j ++) if(j >= 0 && j < ny) y_mix[j] = pbp_switch_state;
    }
    baseidx_prev = baseidx;
    if(pbp_on && pbp_periods == pbp_periods_thrd && (! require_hm)) continue;
    if(! q -> has_hm[i]) o_frame_tolayer0(p, q, i, maxnhar_conf);
    if(! q -> has_hm[i]) continue;
    int nhar = imin(maxnhar, p -> nhar[i]);
    fp* yi = malloc(sizeof(fp) * nwin);
    o_synth_harmonic_frame_auto(opt, p -> ampl + (size_t)i * p -> maxnhar, p -> phse + (size_t)i * p -> maxnhar,
      nhar, f0 / fs, nwin, yi);
    for(int j = 0; j < nwin; j ++) {
      int idx = baseidx + j - nwin / 2;
      if(idx >= 0 && idx < ny) y_hm[idx] += yi[j] * w[j];
    }
    free(yi);
  }
  free(w);
  if(q -> dbg_y_hm) memcpy(q -> dbg_y_hm, y_hm, sizeof(fp) * ny);
  if(q -> dbg_y_pbp) memcpy(q -> dbg_y_pbp, y_pbp, sizeof(fp) * ny);
  if(q -> dbg_y_mix) memcpy(q -> dbg_y_mix, y_mix, sizeof(fp) * ny);
  for(int i = 0; i < ny; i ++) y_mix[i] = (fp)(y_hm[i] * (1.0 - y_mix[i]) + y_pbp[i] * y_mix[i]);
  free(y_hm); free(y_pbp);
}
