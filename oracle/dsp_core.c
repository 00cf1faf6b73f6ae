/*
 * dsp_core.c -- oracle restatement of the slice of `ciglet` that libllsm2's
 * layer-0 path calls.  TEST INFRASTRUCTURE ONLY.
 *
 * ciglet (github.com/Sleepwalking/ciglet, unpinned, absent from the
 * reference tree) cannot be consulted here; every function below is OUR
 * definition, chosen to satisfy the constraints the reference's call sites
 * impose (cited per function) and cross-checked against numpy/scipy in
 * tests/test_oracle_dsp.py.  "parity unpinned" applies (see oracle.h).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------ */
/* Index plan.  SURVEY Appendix B: every sample-index product is      */
/* evaluated left to right in IEEE float32 (FP_TYPE=float, strict     */
/* evaluation), literals `2.0` promote to double exactly as in the    */
/* reference expression.  Compiled with -ffp-contract=off.            */
/* ------------------------------------------------------------------ */
int o_idx_center(int i, float thop, float fs) {      /* dsputils.c:191, layer0.c:332,429,588 */
  volatile float a = (float)i * thop;
  volatile float b = a * fs;
  return (int)round((double)b);
}
float o_idx_rawfrac(int i, float thop, float fs, int* baseidx) { /* layer0.c:127-129 */
  volatile float a = (float)i * thop;
  volatile float raw = a * fs;
  int base = (int)round((double)raw);
  if(baseidx) *baseidx = base;
  volatile float d = raw - (float)base;
  return d;
}
int o_idx_nwin_sin(float thop, float fs) {            /* layer0.c:121 */
  volatile float a = thop * fs;
  return (int)(round((double)a) * 2);
}
int o_idx_nwin_env(float thop, float fs) {            /* layer0.c:293 (2.0 is double) */
  volatile double a = (double)thop * 2.0;
  volatile double b = a * (double)fs;
  return (int)round(b);
}
int o_idx_nwin_filt(float thop, float fs) {           /* layer0.c:560 */
  volatile float a = thop * fs;
  volatile float b = a * 2.0f;
  return (int)round((double)b);
}
int o_idx_nwin_psd(float thop, float fs) {            /* layer0.c:320 */
  volatile float a = thop * 4.0f;
  volatile float b = a * fs;
  return (int)round((double)b);
}
int o_idx_ny(int nfrm, float thop, float fs) {        /* layer0.c:643 */
  volatile float a = (float)(nfrm + 1) * thop;
  volatile float b = a * fs;
  return (int)round((double)b);
}
int o_idx_hwin(float f0, float fs, float rel) {       /* dsputils.c:190 */
  volatile float a = fs / f0;
  volatile float b = a * rel;
  volatile float c = b / 2.0f;
  return (int)(round((double)c) * 2);
}
int o_idx_nhar(float f0, float fs, int maxnhar) {     /* dsputils.c:171-173, 206, 218 */
  volatile float a = fs / f0;
  volatile float b = a / 2.0f;
  return imin((int)floor((double)b), maxnhar);
}
int o_idx_env_ola(int i, int j, float thop, float fs) { /* layer0.c:307 */
  volatile float a = (float)(i - 1) * thop;
  volatile float b = a * fs;
  volatile float c = b + (float)j;
  return (int)round((double)c);
}
int o_idx_dcwin(float f0, float thop, float fs) {     /* layer0.c:430 (ternary promotes to double) */
  if(f0 == 0) {
    volatile float a = thop * 2.0f;
    volatile double b = (double)a * (double)fs;
    return (int)round(b);
  }
  volatile double a = 2.0 / (double)f0;
  volatile double b = a * (double)fs;
  return (int)round(b);
}
int o_idx_spgmwin(float f0, float fs, int nwin_psd) { /* layer0.c:331 (int truncation) */
  if(f0 == 0) return nwin_psd;
  volatile float a = fs / f0;
  volatile float b = a * 3.0f;
  return (int)b;
}
int o_nextpow2(double x) {                            /* pow(2, ceil(log2(x))) */
  return (int)pow(2.0, ceil(log2(x)));
}

/* ------------------------------------------------------------------ */
/* FFT: forward unnormalised (e^{-j}), inverse scaled by 1/n.         */
/* Constraint: PSD divides only by sum(w^2) (dsputils.c:240-243);     */
/* unity OLA gain through fft->ifft (layer0.c:593-624).               */
/* ------------------------------------------------------------------ */
#define FFT_CACHE 6
static __thread int fft_cache_n[FFT_CACHE];
static __thread fp* fft_cache_c[FFT_CACHE];
static __thread fp* fft_cache_s[FFT_CACHE];

static void fft_twiddles(int n, fp** c, fp** s) {
  for(int i = 0; i < FFT_CACHE; i ++)
    if(fft_cache_n[i] == n) { *c = fft_cache_c[i]; *s = fft_cache_s[i]; return; }
  int slot = 0;
  for(int i = 0; i < FFT_CACHE; i ++) if(fft_cache_n[i] == 0) { slot = i; break; }
  free(fft_cache_c[slot]); free(fft_cache_s[slot]);
  fft_cache_c[slot] = malloc(sizeof(fp) * (n / 2 + 1));
  fft_cache_s[slot] = malloc(sizeof(fp) * (n / 2 + 1));
  for(int k = 0; k < n / 2; k ++) {
    fft_cache_c[slot][k] = (fp)cos(2.0 * M_PI * k / n);
    fft_cache_s[slot][k] = (fp)sin(2.0 * M_PI * k / n);
  }
  fft_cache_n[slot] = n;
  *c = fft_cache_c[slot]; *s = fft_cache_s[slot];
}

void o_fft(fp* re, fp* im, int n, int inverse) {
  if(n <= 1) return;
  fp* tc; fp* ts;
  fft_twiddles(n, & tc, & ts);
  /* bit reversal */
  for(int i = 1, j = 0; i < n; i ++) {
    int bit = n >> 1;
    for(; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if(i < j) {
      fp t = re[i]; re[i] = re[j]; re[j] = t;
      t = im[i]; im[i] = im[j]; im[j] = t;
    }
  }
  for(int len = 2; len <= n; len <<= 1) {
    int half = len >> 1, step = n / len;
    for(int i = 0; i < n; i += len) {
      for(int k = 0; k < half; k ++) {
        fp wr = tc[k * step];
        fp wi = inverse ? ts[k * step] : -ts[k * step];
        fp ur = re[i + k], ui = im[i + k];
        fp vr = re[i + k + half] * wr - im[i + k + half] * wi;
        fp vi = re[i + k + half] * wi + im[i + k + half] * wr;
        re[i + k] = ur + vr; im[i + k] = ui + vi;
        re[i + k + half] = ur - vr; im[i + k + half] = ui - vi;
      }
    }
  }
  if(inverse) {
    fp s = (fp)1.0 / n;
    for(int i = 0; i < n; i ++) { re[i] *= s; im[i] *= s; }
  }
}

/* ------------------------------------------------------------------ */
/* Windows: symmetric (denominator n-1), MATLAB-style naming.          */
/* UNVERIFIED against ciglet; Blackman cancels in  */
/* dsputils.c:153,163 and :257,261; Hann does not cancel in the OLA.   */
/* ------------------------------------------------------------------ */
/* ---- switchable conventions (mirrors llsm_gpu_set_convention of the product) ---- */
static int conv_hann_periodic = 0, conv_mavg_half = 3, conv_filtfilt_pad = 15, conv_interp1u_excl = 0, conv_kalman_init = 0, conv_lobe_1e6 = 133979, conv_lf_rd_clamp = 0;
int o_set_convention(const char* name, int value) {
  if(! strcmp(name, "hann_periodic")) conv_hann_periodic = value;
  else if(! strcmp(name, "moving_avg_half")) conv_mavg_half = value;
  else if(! strcmp(name, "filtfilt_pad")) conv_filtfilt_pad = value;
  else if(! strcmp(name, "interp1u_exclusive")) conv_interp1u_excl = value;
  else if(! strcmp(name, "kalman_init")) conv_kalman_init = value;
  else if(! strcmp(name, "spec2env_lobe_1e6")) conv_lobe_1e6 = value;
  else if(! strcmp(name, "lf_rd_clamp")) conv_lf_rd_clamp = value;
  else return -1;
  return 0;
}
int o_conv_mavg_half(void) { return conv_mavg_half; }
int o_conv_lf_rd_clamp(void) { return conv_lf_rd_clamp; }
int o_conv_interp1u_excl(void) { return conv_interp1u_excl; }
/* cig_spec2env's constant (l1_oracle.c): units of 1e-6, 133979 = the calibrated value itself */
double o_conv_lobe_bias(void) { return conv_lobe_1e6 == 133979 ? 0.13397922601295542 : conv_lobe_1e6 * 1e-6; }
/* overlap-add Hann (layer0.c:122, 294, 561; llsmrt.c:120): symmetric unless the convention says periodic */
void o_hanning_ola(fp* w, int n) {
  if(n == 1) { w[0] = 1; return; }
  for(int i = 0; i < n; i ++)
    w[i] = (fp)(0.5 - 0.5 * cos(2.0 * M_PI * i / (conv_hann_periodic ? n : n - 1)));
}
void o_hanning(fp* w, int n) {
  if(n == 1) { w[0] = 1; return; }
  for(int i = 0; i < n; i ++)
    w[i] = (fp)(0.5 - 0.5 * cos(2.0 * M_PI * i / (n - 1)));
}
void o_blackman(fp* w, int n) {
  if(n == 1) { w[0] = 1; return; }
  for(int i = 0; i < n; i ++) {
    double t = 2.0 * M_PI * i / (n - 1);
    w[i] = (fp)(0.42 - 0.5 * cos(t) + 0.08 * cos(2.0 * t));
  }
}

/* out[j] = x[center - nf/2 + j], zero outside [0,nx).  Required by the
 * analysis/OLA alignment layer0.c:589-592 vs 620-623 and the shift = nx/2
 * phase reference dsputils.c:151-158. */
void o_fetch_frame(const fp* x, int nx, int center, int nf, fp* out) {
  int base = center - nf / 2;
  for(int j = 0; j < nf; j ++) {
    int idx = base + j;
    out[j] = (idx >= 0 && idx < nx) ? x[idx] : 0;
  }
}

/* ------------------------------------------------------------------ */
/* CZT / ICZT.  Y[k] = sum_t x[t] e^{-j w0 k t}, k = 0..nout-1        */
/* (needed so the e^{+j (n/2) w0 k} rotation of dsputils.c:157-164     */
/* yields cosine phase at the window centre).                          */
/* y[t] = (1/n) Re sum_k X[k] e^{+j w0 k t} (dsputils.c:345-348).      */
/* mode 0: direct evaluation with double-precision phase; mode 1:      */
/* Bluestein chirp convolution through o_fft (the cost structure of a  */
/* CZT as ciglet implements it; used for the CPU baseline timing).     */
/* ------------------------------------------------------------------ */
static int czt_mode = 0;
void o_set_czt_mode(int bluestein) { czt_mode = bluestein; }

static void czt_direct(const fp* x, int n, fp omega0, int nout, fp* yr, fp* yi) {
  double w0 = (double)omega0;
  for(int k = 0; k < nout; k ++) {
    double sr = 0, si = 0;
    double wk = w0 * k;
    for(int t = 0; t < n; t ++) {
      double ph = wk * t;
      sr += (double)x[t] * cos(ph);
      si -= (double)x[t] * sin(ph);
    }
    yr[k] = (fp)sr; yi[k] = (fp)si;
  }
}

/* chirp[m] = e^{-j w0 m^2 / 2}, phase reduced in double */
static void chirp(double w0, int m, double* c, double* s) {
  double ph = fmod(0.5 * w0 * (double)m * (double)m, 2.0 * M_PI);
  *c = cos(ph); *s = -sin(ph);
}

static void czt_bluestein(const fp* xr, const fp* xi, int n, fp omega0,
  int nout, fp* yr, fp* yi, int conj_kernel) {
  /* computes sum_t x[t] W^{tk}, W = e^{-j w0} (conj_kernel=0) or e^{+j w0} */
  double w0 = conj_kernel ? -(double)omega0 : (double)omega0;
  int L = 1; while(L < n + nout - 1) L <<= 1;
  fp* ar = calloc(L, sizeof(fp)); fp* ai = calloc(L, sizeof(fp));
  fp* br = calloc(L, sizeof(fp)); fp* bi = calloc(L, sizeof(fp));
  for(int t = 0; t < n; t ++) {
    double c, s; chirp(w0, t, & c, & s);
    double vr = xr[t], vi = xi ? xi[t] : 0;
    ar[t] = (fp)(vr * c - vi * s); ai[t] = (fp)(vr * s + vi * c);
  }
  int m = imax(n, nout);
  for(int t = 0; t < m; t ++) {
    double c, s; chirp(w0, t, & c, & s);
    if(t < nout) { br[t] = (fp)c; bi[t] = (fp)-s; }
    if(t > 0 && t < n) { br[L - t] = (fp)c; bi[L - t] = (fp)-s; }
  }
  o_fft(ar, ai, L, 0); o_fft(br, bi, L, 0);
  for(int i = 0; i < L; i ++) {
    fp r = ar[i] * br[i] - ai[i] * bi[i];
    fp q = ar[i] * bi[i] + ai[i] * br[i];
    ar[i] = r; ai[i] = q;
  }
  o_fft(ar, ai, L, 1);
  for(int k = 0; k < nout; k ++) {
    double c, s; chirp(w0, k, & c, & s);
    if(yr) yr[k] = (fp)(ar[k] * c - ai[k] * s);
    if(yi) yi[k] = (fp)(ar[k] * s + ai[k] * c);
  }
  free(ar); free(ai); free(br); free(bi);
}

void o_czt(const fp* x, int n, fp omega0, int nout, fp* yr, fp* yi) {
  if(czt_mode) czt_bluestein(x, NULL, n, omega0, nout, yr, yi, 0);
  else czt_direct(x, n, omega0, nout, yr, yi);
}

void o_iczt(const fp* xr, const fp* xi, int nbin, fp omega0, int n, fp* y) {
  if(czt_mode) {
    czt_bluestein(xr, xi, nbin, omega0, n, y, NULL, 1);
    for(int t = 0; t < n; t ++) y[t] /= n;
    return;
  }
  double w0 = (double)omega0;
  for(int t = 0; t < n; t ++) {
    double acc = 0;
    for(int k = 0; k < nbin; k ++) {
      if(xr[k] == 0 && xi[k] == 0) continue;
      double ph = w0 * k * t;
      acc += (double)xr[k] * cos(ph) - (double)xi[k] * sin(ph);
    }
    y[t] = (fp)(acc / n);
  }
}

/* y[t] = sum_i a_i cos(2 pi f_i / fs (t - n/2) + p_i); INFERRED from
 * test/test-harmonic.c:40-47 (must equal the ICZT frame).  Evaluated with the
 * two-term cosine recurrence ("the recurrent method", llsm.h:292-294). */
void o_gensins(const fp* freq, const fp* ampl, const fp* phse, int nsin,
  fp fs, int n, fp* y) {
  for(int t = 0; t < n; t ++) y[t] = 0;
  int half = n / 2;
  for(int i = 0; i < nsin; i ++) {
    double w = 2.0 * M_PI * (double)freq[i] / (double)fs;
    fp c2 = (fp)(2.0 * cos(w));
    fp y1 = (fp)((double)ampl[i] * cos(w * (-half - 1) + (double)phse[i]));
    fp y2 = (fp)((double)ampl[i] * cos(w * (-half - 2) + (double)phse[i]));
    for(int t = 0; t < n; t ++) {
      fp y0 = c2 * y1 - y2;
      y[t] += y0;
      y2 = y1; y1 = y0;
    }
  }
}

/* piecewise-linear interpolation; queries outside the support clamp to the
 * end values (never happens on the hot path: layer0.c:388-396, dsputils.c:311). */
void o_interp1(const fp* xi, const fp* yi, int ni, const fp* xq, int nq, fp* yq) {
  int k = 0;
  for(int q = 0; q < nq; q ++) {
    fp x = xq[q];
    if(x <= xi[0]) { yq[q] = yi[0]; continue; }
    if(x >= xi[ni - 1]) { yq[q] = yi[ni - 1]; continue; }
    if(xi[k] > x) k = 0;
    while(k < ni - 2 && xi[k + 1] <= x) k ++;
    fp r = (x - xi[k]) / (xi[k + 1] - xi[k]);
    yq[q] = yi[k] + (yi[k + 1] - yi[k]) * r;
  }
}
void o_interp1u(fp x0, fp x1, const fp* yi, int ni, const fp* xq, int nq, fp* yq) {
  for(int q = 0; q < nq; q ++) {
    fp pos = (xq[q] - x0) / (x1 - x0) * (ni - 1);
    int k = (int)floor((double)pos);
    if(k < 0) { yq[q] = yi[0]; continue; }
    if(k >= ni - 1) { yq[q] = yi[ni - 1]; continue; }
    fp r = pos - k;
    yq[q] = yi[k] + (yi[k + 1] - yi[k]) * r;
  }
}

/* centred moving average with half-order h: mean of x[i-h .. i+h], window
 * shrinking at the edges.  UNVERIFIED (whether ciglet's third argument is a
 * tap count or a half-order); layer0.c:597 passes 3. */
void o_moving_avg(const fp* x, int n, int h, fp* y) {
  for(int i = 0; i < n; i ++) {
    int lo = imax(0, i - h), hi = imin(n - 1, i + h);
    fp acc = 0;
    for(int j = lo; j <= hi; j ++) acc += x[j];
    y[i] = acc / (hi - lo + 1);
  }
}

/* scalar random-walk Kalman filter + RTS smoother (layer0.c:361-385).
 * x_0 = z_0, P_0 = R_0 (UNVERIFIED initialisation). */
void o_kalmanf1d(const fp* z, const fp* Q, const fp* R, int n, fp* P, fp* y) {
  fp x = z[0], p = R[0];                              /* the first observation is the state ... */
  if(conv_kalman_init == 1) {                         /* ... or prior (z0, R0) and the filter update of frame 0 as well */
    fp pp = p + Q[0];
    p = ((fp)1.0 - pp / (pp + R[0])) * pp;
  }
  y[0] = x; P[0] = p;
  for(int i = 1; i < n; i ++) {
    fp pp = p + Q[i];
    fp k = pp / (pp + R[i]);
    x = x + k * (z[i] - x);
    p = ((fp)1.0 - k) * pp;
    y[i] = x; P[i] = p;
  }
}
void o_kalmans1d(const fp* y, const fp* P, const fp* Q, int n, fp* s) {
  s[n - 1] = y[n - 1];
  for(int i = n - 2; i >= 0; i --) {
    fp c = P[i] / (P[i] + Q[i + 1]);
    s[i] = y[i] + c * (s[i + 1] - y[i]);
  }
}

/* ------------------------------------------------------------------ */
/* Chebyshev type-I design == scipy.signal.cheby1(order, rp, wn, btype)*/
/* (filter-coef.h identified as cheby1(4, 0.5, (i+1)*0.02)). */
/* ------------------------------------------------------------------ */
typedef struct { double re, im; } cpx;
static cpx cmul(cpx a, cpx b) { cpx r = {a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re}; return r; }
static cpx cdiv(cpx a, cpx b) {
  double d = b.re*b.re + b.im*b.im;
  cpx r = {(a.re*b.re + a.im*b.im)/d, (a.im*b.re - a.re*b.im)/d}; return r;
}
static void poly_from_roots(const cpx* r, int n, double* out) {
  cpx c[16]; c[0].re = 1; c[0].im = 0;
  for(int i = 0; i < n; i ++) {
    c[i + 1].re = 0; c[i + 1].im = 0;
    for(int j = i + 1; j >= 1; j --) {
      cpx t = cmul(c[j - 1], r[i]);
      c[j].re -= t.re; c[j].im -= t.im;
    }
  }
  for(int i = 0; i <= n; i ++) out[i] = c[i].re;
}
void o_cheby1(int N, double rp, double wn, int highpass, double* b, double* a) {
  double eps = sqrt(pow(10.0, 0.1 * rp) - 1.0);
  double mu = asinh(1.0 / eps) / N;
  cpx p[16];
  cpx kprod = {1, 0};
  for(int i = 0; i < N; i ++) {
    double theta = M_PI * (-N + 1 + 2 * i) / (2.0 * N);
    /* p = -sinh(mu + j theta) */
    p[i].re = -sinh(mu) * cos(theta);
    p[i].im = -cosh(mu) * sin(theta);
    cpx np = {-p[i].re, -p[i].im};
    kprod = cmul(kprod, np);
  }
  double k = kprod.re;
  if(N % 2 == 0) k /= sqrt(1.0 + eps * eps);
  double fs2 = 4.0;                      /* 2*fs with fs = 2 */
  double warped = fs2 * tan(M_PI * wn / 2.0);
  cpx z[16];
  cpx den = {1, 0};
  if(! highpass) {
    for(int i = 0; i < N; i ++) { p[i].re *= warped; p[i].im *= warped; }
    k *= pow(warped, N);
    for(int i = 0; i < N; i ++) {
      cpx d = {fs2 - p[i].re, -p[i].im};
      den = cmul(den, d);
      z[i].re = -1; z[i].im = 0;
    }
    cpx one = {1, 0};
    k *= cdiv(one, den).re;
  } else {
    cpx pn = {1, 0};
    for(int i = 0; i < N; i ++) {
      cpx np = {-p[i].re, -p[i].im};
      pn = cmul(pn, np);
    }
    cpx one = {1, 0};
    k *= cdiv(one, pn).re;               /* real(prod(-z)/prod(-p)), z empty */
    for(int i = 0; i < N; i ++) {
      cpx w = {warped, 0};
      p[i] = cdiv(w, p[i]);
    }
    for(int i = 0; i < N; i ++) {
      cpx d = {fs2 - p[i].re, -p[i].im};
      den = cmul(den, d);
      z[i].re = 1; z[i].im = 0;
    }
    cpx num = {pow(fs2, N), 0};
    k *= cdiv(num, den).re;
  }
  for(int i = 0; i < N; i ++) {
    cpx nu = {fs2 + p[i].re, p[i].im};
    cpx de = {fs2 - p[i].re, -p[i].im};
    p[i] = cdiv(nu, de);
  }
  poly_from_roots(z, N, b);
  poly_from_roots(p, N, a);
  for(int i = 0; i <= N; i ++) b[i] *= k;
}

/* dsputils.c:28-49: index = max(0, round(cutoff*2/0.02 - 1)) clamped to 47;
 * row i of the table is cheby1(4, 0.5, (i+1)*0.02). */
void o_get_chebyshev_filter(fp cutoff, int highpass, fp* a, fp* b) {
  const float step_freq = 0.02f;        /* `static const FP_TYPE step_freq`, FP_TYPE=float */
  /* The row is an INDEX: like the frame centres (o_idx_*) it follows the reference's FP_TYPE = float evaluation in
   * both builds of this oracle -- the normalised cutoff is a float there (dsputils.c:28, layer0.c:440).  With a
   * float64 cutoff a band edge on a half row (3080 Hz at 8 kHz: 38.5 - 1) picks the neighbouring filter: 24 - 55 %
   * difference in the band energies (found by tools/fuzz_soak.py). */
  const float cutoff_f = (float)cutoff;
  int index = imax(0, (int)round((double)cutoff_f * 2.0 / (double)step_freq - 1));
  if(index >= 48) index = 47;
  double bd[5], ad[5];
  o_cheby1(4, 0.5, (index + 1) * 0.02, highpass, bd, ad);
  for(int i = 0; i < 5; i ++) { a[i] = (fp)ad[i]; b[i] = (fp)bd[i]; }
}

/* forward-backward IIR == scipy.signal.filtfilt(b, a, x) defaults:
 * odd extension by padlen = 3*max(na,nb), steady-state initial conditions
 * (lfilter_zi) scaled by the first sample of each pass.  UNVERIFIED against
 * ciglet's edge handling. */
static void lfilter_zi(const double* b, const double* a, int n, double* zi) {
  /* solve (I - A^T) zi = B with A = companion(a); closed form as in scipy */
  int m = n - 1;
  double asum = 0, csum = 0;
  for(int i = 0; i < n; i ++) asum += a[i];
  for(int i = 1; i < n; i ++) csum += b[i] - a[i] * b[0];
  zi[0] = csum / asum;
  double acc = 1.0, cs = 0;
  for(int i = 1; i < m; i ++) {
    acc += a[i];
    cs += b[i] - a[i] * b[0];
    zi[i] = acc * zi[0] - cs;
  }
}
static void lfilter_df2t(const fp* b, const fp* a, int n, const fp* x, int nx,
  fp* z, fp* y, int reverse) {
  for(int t = 0; t < nx; t ++) {
    int idx = reverse ? nx - 1 - t : t;
    fp xi = x[idx];
    fp yi = b[0] * xi + z[0];
    for(int i = 0; i < n - 2; i ++)
      z[i] = b[i + 1] * xi + z[i + 1] - a[i + 1] * yi;
    z[n - 2] = b[n - 1] * xi - a[n - 1] * yi;
    y[idx] = yi;
  }
}
void o_filtfilt(const fp* b, int nb, const fp* a, int na, const fp* x, int nx, fp* y) {
  int n = imax(na, nb);
  int pad = imin(conv_filtfilt_pad > 0 ? conv_filtfilt_pad : 3 * n, nx - 1);
  int ne = nx + 2 * pad;
  fp* ext = malloc(sizeof(fp) * ne);
  fp* tmp = malloc(sizeof(fp) * ne);
  for(int j = 0; j < pad; j ++) {
    ext[j] = (fp)2.0 * x[0] - x[pad - j];
    ext[pad + nx + j] = (fp)2.0 * x[nx - 1] - x[nx - 2 - j];
  }
  memcpy(ext + pad, x, sizeof(fp) * nx);
  double bd[8] = {0}, ad[8] = {0}, zid[8] = {0};
  for(int i = 0; i < nb; i ++) bd[i] = b[i] / a[0];
  for(int i = 0; i < na; i ++) ad[i] = a[i] / a[0];
  lfilter_zi(bd, ad, n, zid);
  fp bb[8] = {0}, aa[8] = {0}, z[8];
  for(int i = 0; i < n; i ++) { bb[i] = (fp)bd[i]; aa[i] = (fp)ad[i]; }
  for(int i = 0; i < n - 1; i ++) z[i] = (fp)zid[i] * ext[0];
  lfilter_df2t(bb, aa, n, ext, ne, z, tmp, 0);
  for(int i = 0; i < n - 1; i ++) z[i] = (fp)zid[i] * tmp[ne - 1];
  lfilter_df2t(bb, aa, n, tmp, ne, z, ext, 1);
  memcpy(y, ext + pad, sizeof(fp) * nx);
  free(ext); free(tmp);
}

/* dsputils.c:51-70 */
void o_chebyfilt(const fp* x, int nx, fp c1, fp c2, fp* y) {
  if(c1 < 0) c1 = 0;
  if(c2 > (fp)0.5) c2 = (fp)0.5;
  if(c1 != 0 && c2 < (fp)0.5) {
    fp* x1 = malloc(sizeof(fp) * nx);
    o_chebyfilt(x, nx, c1, (fp)0.5, x1);
    o_chebyfilt(x1, nx, 0, c2, y);
    free(x1);
    return;
  }
  fp a[5], b[5];
  if(c1 == 0) o_get_chebyshev_filter(c2, 0, a, b);
  else        o_get_chebyshev_filter(c1, 1, a, b);
  o_filtfilt(b, 5, a, 5, x, nx, y);
}

/* smooth log-magnitude envelope of a linear magnitude spectrum: cepstral
 * liftering with sinc(q*f0), i.e. the log spectrum smoothed over one f0.
 * OUR definition (ciglet's spec2env is opaque); it only feeds the Kalman
 * process variance Q (layer0.c:339-376), low parity sensitivity. */
void o_spec2env(const fp* S, int nfft, fp f0, fp* env) {
  int ns = nfft / 2 + 1;
  fp* re = malloc(sizeof(fp) * nfft);
  fp* im = calloc(nfft, sizeof(fp));
  for(int i = 0; i < ns; i ++) re[i] = (fp)log((double)S[i] + 1e-10);
  for(int i = 1; i < nfft / 2; i ++) re[nfft - i] = re[i];
  o_fft(re, im, nfft, 1);
  for(int q = 1; q <= nfft / 2; q ++) {
    double a = M_PI * q * (double)f0;
    fp l = (fp)(sin(a) / a);
    re[q] *= l; im[q] = 0;
    if(q < nfft / 2) { re[nfft - q] *= l; im[nfft - q] = 0; }
  }
  im[0] = 0;
  o_fft(re, im, nfft, 0);
  for(int i = 0; i < ns; i ++) env[i] = re[i];
  free(re); free(im);
}

/* One STFT frame as cig_stft_forward is used by dsputils.c:96-115: window of
 * `winsize` centred on `center`, zero-phase placement (frame centre at buffer
 * index 0, time-aliased if winsize > nfft), linear magnitude and phase on
 * nfft/2+1 bins. */
void o_stft_frame(const fp* x, int nx, int center, int winsize, int nfft,
  int blackman, fp* magn, fp* phse, fp* wsum) {
  fp* w = malloc(sizeof(fp) * winsize);
  fp* f = malloc(sizeof(fp) * winsize);
  fp* re = calloc(nfft, sizeof(fp));
  fp* im = calloc(nfft, sizeof(fp));
  if(blackman) o_blackman(w, winsize); else o_hanning(w, winsize);
  o_fetch_frame(x, nx, center, winsize, f);
  fp ws = 0;
  for(int j = 0; j < winsize; j ++) {
    int pos = ((j - winsize / 2) % nfft + nfft) % nfft;
    re[pos] += f[j] * w[j];
    ws += w[j];
  }
  o_fft(re, im, nfft, 0);
  for(int k = 0; k <= nfft / 2; k ++) {
    if(magn) magn[k] = (fp)sqrt((double)(re[k] * re[k] + im[k] * im[k]));
    if(phse) phse[k] = (fp)atan2((double)im[k], (double)re[k]);
  }
  if(wsum) *wsum = ws;
  free(w); free(f); free(re); free(im);
}

fp o_wrap(fp x) {                        /* to (-pi, pi], frame.c:59 */
  double y = (double)x - 2.0 * M_PI * floor(((double)x + M_PI) / (2.0 * M_PI));
  if(y <= -M_PI) y += 2.0 * M_PI;
  if(y > M_PI) y -= 2.0 * M_PI;
  return (fp)y;
}

/* Counter-based Gaussian generator (replaces ciglet randn over libc rand(),
 * dsputils.c:353-361): splitmix64 hash of (seed, idx) -> two 24-bit uniforms
 * -> Box-Muller (cosine branch).  Bit-identical integer stage on CPU and GPU. */
fp o_rng_normal(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = idx + seed * 0x9E3779B97F4A7C15ULL + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  float u1 = (float)((z >> 40) + 1) * (1.0f / 16777216.0f);
  float u2 = (float)((z >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
  double r = sqrt(-2.0 * log((double)u1));
  return (fp)(r * cos(2.0 * M_PI * (double)u2));
}
