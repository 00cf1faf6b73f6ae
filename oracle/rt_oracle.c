/*
 * rt_oracle.c -- oracle restatement of libllsm2's real-time synthesis buffer
 * (reference: llsmrt.c:32-602, buffer.h:32-138 @ 2.1.0), harmonic-model path
 * only (use_l1 == 0; the PbP branch llsmrt.c:305-420 is out of scope, see
 * DESIGN.md).  TEST INFRASTRUCTURE ONLY; "parity unpinned" (oracle.h).
 *
 * Hop bookkeeping (cycle / curr_nhop / next_nhop) is always float32, as in
 * the FP_TYPE=float reference, so both oracle builds agree on every index.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define LOG2IN(x) ((x) / 2.3025851 * 10.0)
#define DB2LOG(x) ((x) * 2.3025851 / 20.0)
#define LOGRESBIAS 0.375

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline fp fpmax(fp a, fp b) { return a > b ? a : b; }

/* ---- buffer.h:32-138 ---- */
typedef struct { fp* data; int capacity; int curr; } ring;
static ring* ring_create(int capacity) {
  ring* r = malloc(sizeof(ring));
  r -> capacity = capacity; r -> curr = 0;
  r -> data = calloc(capacity, sizeof(fp));
  return r;
}
static void ring_delete(ring* r) { if(r) { free(r -> data); free(r); } }
static fp ring_read(ring* r, int idx) {
  return r -> data[(r -> curr + idx + r -> capacity) % r -> capacity];
}
static void ring_append(ring* r, fp x) {
  r -> data[r -> curr] = x; r -> curr = (r -> curr + 1) % r -> capacity;
}
static void ring_forward(ring* r, int size) { r -> curr = (r -> curr + size) % r -> capacity; }
static void ring_readchunk(ring* r, int lag, int size, fp* dst) {
  int base = r -> curr + r -> capacity;
  for(int i = 0; i < size; i ++) dst[i] = r -> data[(base + lag + i) % r -> capacity];
}
static void ring_writechunk(ring* r, int lag, int size, const fp* src) {
  int base = r -> curr + r -> capacity;
  for(int i = 0; i < size; i ++) r -> data[(base + lag + i) % r -> capacity] = src[i];
}
static void ring_addchunk(ring* r, int lag, int size, const fp* src) {
  int base = r -> curr + r -> capacity;
  for(int i = 0; i < size; i ++) r -> data[(base + lag + i) % r -> capacity] += src[i];
}
static void ring_appendchunk(ring* r, int size, const fp* src) {
  ring_forward(r, size); ring_writechunk(r, -size, size, src);
}
static void ring_appendblank(ring* r, int size) {
  ring_forward(r, size);
  int base = r -> curr + r -> capacity;
  for(int i = 0; i < size; i ++) r -> data[(base - size + i) % r -> capacity] = 0;
}

/* ---- llsmrt.c:32-78 ---- */
struct o_rtsynth {
  int nout, nchannel, ntemplate, ninternal;
  int npsd, maxnhar_e;
  o_soptions opt;
  float fs, thop, cycle;
  fp fnyq;
  int curr_nhop, next_nhop, exc_cycle, sin_pos, nfft;
  fp* win;
  int has_prev; fp* prev_psd;
  ring* out_p; ring* out_ap;
  fp** tpl; ring** mod;
  ring* exc_mix; ring* noise; ring* sin;
  fp* psd_axis;
};

/* llsmrt.c:80-91 */
static fp circular_noise(const fp* x, int nx, int i) {
  int overlap = 32;
  i = i % nx;
  if(i >= overlap) return x[i];
  fp y = x[i];
  fp r = (fp)i / overlap;
  y *= (fp)1.0 - r;
  y += x[nx - overlap + i] * r;
  y /= (fp)sqrt((double)(2 * r * (r - 1) + 1));
  return y;
}

/* llsmrt.c:110-129 */
static void update_cycle(o_rtsynth* s) {
  int prev_nhop = s -> curr_nhop;
  volatile float c = s -> cycle + s -> thop;
  volatile float cf = c * s -> fs;
  s -> curr_nhop = (int)floor((double)cf);
  volatile float d = (float)prev_nhop / s -> fs;
  volatile float c2 = c - d;
  s -> cycle = c2;
  int nwin = s -> curr_nhop * 2;
  free(s -> win); s -> win = malloc(sizeof(fp) * nwin);
  o_hanning(s -> win, nwin);
  volatile float e = s -> cycle + s -> thop;
  volatile float ef = e * s -> fs;
  s -> next_nhop = (int)floor((double)ef);
  for(int ch = 0; ch < s -> nchannel; ch ++) ring_appendblank(s -> mod[ch], s -> curr_nhop);
  ring_appendblank(s -> sin, s -> curr_nhop);
  ring_appendblank(s -> noise, s -> curr_nhop);
}

/* llsmrt.c:134-147 */
static void run_excitation_buffers(o_rtsynth* s, int nx) {
  fp* x = calloc(nx, sizeof(fp));
  fp* m = malloc(sizeof(fp) * nx);
  for(int c = 0; c < s -> nchannel; c ++) {
    ring_readchunk(s -> mod[c], -s -> curr_nhop - nx, nx, m);
    for(int i = 0; i < nx; i ++)
      x[i] += (fp)sqrt((double)m[i]) * s -> tpl[c][(s -> exc_cycle + i) % s -> ntemplate];
  }
  ring_appendchunk(s -> exc_mix, nx, x);
  s -> exc_cycle = (s -> exc_cycle + nx) % s -> ntemplate;
  free(x); free(m);
}

/* llsmrt.c:157-223 */
o_rtsynth* o_rt_create(const o_soptions* opt, const o_params* conf,
  int capacity, unsigned long long seed) {
  o_rtsynth* s = calloc(1, sizeof(o_rtsynth));
  s -> nchannel = conf -> nchannel;
  s -> ntemplate = (int)opt -> fs;
  s -> ninternal = (int)(opt -> fs * 0.2);
  s -> npsd = conf -> npsd; s -> maxnhar_e = conf -> maxnhar_e;
  s -> opt = *opt;
  s -> fs = (float)opt -> fs; s -> thop = (float)conf -> thop;
  s -> fnyq = conf -> fnyq;
  volatile float hf = s -> thop * s -> fs;
  s -> nfft = o_nextpow2((double)hf * 2.2 + 32);
  s -> out_p = ring_create(capacity); s -> out_ap = ring_create(capacity);
  s -> exc_mix = ring_create(s -> ninternal);
  s -> noise = ring_create(s -> ninternal);
  s -> sin = ring_create(s -> ninternal);
  s -> tpl = malloc(sizeof(fp*) * s -> nchannel);
  s -> mod = malloc(sizeof(ring*) * s -> nchannel);
  for(int c = 0; c < s -> nchannel; c ++) {
    s -> tpl[c] = calloc(s -> ntemplate, sizeof(fp));
    s -> mod[c] = ring_create(s -> ninternal);
  }
  s -> psd_axis = malloc(sizeof(fp) * s -> npsd);
  for(int j = 0; j < s -> npsd; j ++)
    s -> psd_axis[j] = (fp)((double)s -> fnyq * j / (s -> npsd - 1));
  s -> prev_psd = malloc(sizeof(fp) * s -> npsd);
  s -> curr_nhop = 1;
  update_cycle(s);
  s -> cycle = 0;
  s -> sin_pos = -s -> curr_nhop * 2 - s -> nfft / 2;
  /* llsm_make_exc_template, llsmrt.c:93-107 */
  fp fs = opt -> fs;
  fp* x = malloc(sizeof(fp) * s -> ntemplate);
  for(int c = 0; c < s -> nchannel; c ++) {
    fp fmin = c == 0 ? 0 : conf -> chanfreq[c - 1];
    fp fmax = c == s -> nchannel - 1 ? (fp)(fs / 2.0) : conf -> chanfreq[c];
    if(fmin >= fs / 2.0) break;
    o_generate_bandlimited_noise(s -> ntemplate, fmin / fs, fmax / fs,
      seed * 16 + (unsigned long long)c, NULL, x);
    for(int j = 0; j < s -> ntemplate; j ++)
      s -> tpl[c][j] = circular_noise(x, s -> ntemplate, j);
  }
  free(x);
  /* llsm_fill_excitation_buffers, llsmrt.c:149-155 */
  for(int i = 0; i < s -> ninternal - 1; i ++)
    for(int c = 0; c < s -> nchannel; c ++) ring_append(s -> mod[c], (fp)1e-5);
  for(int i = 0; i < 5; i ++) run_excitation_buffers(s, s -> ninternal / 5);
  return s;
}

void o_rt_delete(o_rtsynth* s) {
  if(! s) return;
  ring_delete(s -> out_p); ring_delete(s -> out_ap);
  ring_delete(s -> exc_mix); ring_delete(s -> noise); ring_delete(s -> sin);
  for(int c = 0; c < s -> nchannel; c ++) { free(s -> tpl[c]); ring_delete(s -> mod[c]); }
  free(s -> tpl); free(s -> mod); free(s -> win); free(s -> psd_axis);
  free(s -> prev_psd); free(s);
}

int o_rt_latency(o_rtsynth* s) { return -s -> sin_pos - s -> curr_nhop; }  /* llsmrt.c:568-571 */
int o_rt_numoutput(o_rtsynth* s) { return s -> nout; }

/* llsmrt.c:422-478 */
static void feed_filter(o_rtsynth* s) {
  const int nfade = 16;
  int nfft = s -> nfft, nspec = nfft / 2 + 1, nhop = s -> curr_nhop, nwin = nhop * 2;
  if(! s -> has_prev) return;
  fp wsqr = 0;
  for(int i = 0; i < nwin; i ++) wsqr += s -> win[i] * s -> win[i];
  fp peak = s -> prev_psd[0];
  for(int j = 1; j < s -> npsd; j ++) if(s -> prev_psd[j] > peak) peak = s -> prev_psd[j];
  if(peak < -100) return;
  fp* x_re = calloc(nfft, sizeof(fp)); fp* x_im = calloc(nfft, sizeof(fp));
  fp* psd = malloc(sizeof(fp) * nspec); fp* env = malloc(sizeof(fp) * nspec);
  fp* H = malloc(sizeof(fp) * nspec);
  ring_readchunk(s -> exc_mix, -nhop * 2, nwin, x_re + nfft / 2 - nhop);
  for(int i = 0; i < nwin; i ++) x_re[i - nhop + nfft / 2] *= s -> win[i];
  o_fft(x_re, x_im, nfft, 0);
  for(int j = 0; j < nspec; j ++) psd[j] = (x_re[j] * x_re[j] + x_im[j] * x_im[j]) / wsqr;
  o_moving_avg(psd, nspec, 3, env);
  o_spectrum_from_envelope(s -> psd_axis, s -> prev_psd, s -> npsd, nspec - 1,
    (fp)(s -> fs / 2.0), H);
  for(int j = 0; j < nspec - 1; j ++)
    H[j] = (fp)(exp(DB2LOG((double)H[j])) / sqrt((double)(env[j] * 44100 / s -> fs) + 1e-8));
  for(int j = 0; j < nspec - 1; j ++) { x_re[j] *= H[j]; x_im[j] *= H[j]; }
  x_re[nspec - 1] = x_re[nspec - 2]; x_im[nspec - 1] = x_im[nspec - 2];
  for(int k = 1; k < nfft / 2; k ++) { x_re[nfft - k] = x_re[k]; x_im[nfft - k] = -x_im[k]; }
  o_fft(x_re, x_im, nfft, 1);
  for(int i = 0; i < nfade; i ++) {
    x_re[i] *= (fp)i / nfade;
    x_re[nfft - i - 1] *= (fp)(1.0 - (fp)i / nfade);
  }
  ring_addchunk(s -> noise, -nfft, nfft, x_re);
  free(x_re); free(x_im); free(psd); free(env); free(H);
}

/* llsmrt.c:505-521 with :255-291 and :480-503 */
void o_rt_feed(o_rtsynth* s, const o_params* p, int i) {
  update_cycle(s);
  int nch = s -> nchannel, me = p -> maxnhar_e;
  int nwin = s -> curr_nhop * 2;
  fp f0 = p -> f0[i];
  fp* x = malloc(sizeof(fp) * nwin);
  /* feed_modcomps */
  for(int c = 0; c < nch; c ++) {
    int nh = f0 > 0 ? p -> nhar_e[i] : 0;
    o_synth_harmonic_frame_auto(& s -> opt, p -> eenv_ampl + ((size_t)i * nch + c) * me,
      p -> eenv_phse + ((size_t)i * nch + c) * me, nh, f0 / s -> fs, nwin, x);
    fp offset = p -> edc[(size_t)i * nch + c];
    for(int j = 0; j < nwin; j ++) x[j] = fpmax(x[j] + offset, (fp)1e-8) * s -> win[j];
    ring_addchunk(s -> mod[c], -nwin, nwin, x);
  }
  /* feed_sinusoids */
  if(f0 > 0) {
    fp phase_shift = (fp)((double)(float)(s -> cycle * 2) * M_PI * f0);
    int nhar = imin(p -> nhar[i], s -> nfft);
    fp* phase = malloc(sizeof(fp) * (nhar > 0 ? nhar : 1));
    for(int k = 0; k < nhar; k ++)
      phase[k] = (fp)(p -> phse[(size_t)i * p -> maxnhar + k] - phase_shift * (k + 1.0));
    o_synth_harmonic_frame_auto(& s -> opt, p -> ampl + (size_t)i * p -> maxnhar,
      phase, nhar, f0 / s -> fs, nwin, x);
    for(int j = 0; j < nwin; j ++) x[j] *= s -> win[j];
    ring_addchunk(s -> sin, -nwin, nwin, x);
    free(phase);
  }
  free(x);
  run_excitation_buffers(s, s -> curr_nhop);
  feed_filter(s);
  /* feed_mix */
  fp* x_nos = malloc(sizeof(fp) * s -> next_nhop);
  fp* x_sin = malloc(sizeof(fp) * s -> next_nhop);
  ring_readchunk(s -> noise, -s -> nfft, s -> next_nhop, x_nos);
  ring_readchunk(s -> sin, s -> sin_pos, s -> next_nhop, x_sin);
  ring_appendchunk(s -> out_p, s -> next_nhop, x_sin);
  ring_appendchunk(s -> out_ap, s -> next_nhop, x_nos);
  s -> nout += s -> next_nhop;
  free(x_nos); free(x_sin);
  /* prev_nm with PSDRES folded in, llsmrt.c:513-520 */
  s -> has_prev = 1;
  for(int j = 0; j < s -> npsd; j ++) {
    s -> prev_psd[j] = p -> psd[(size_t)i * s -> npsd + j];
    if(p -> psdres)
      s -> prev_psd[j] += (fp)(p -> psdres[(size_t)i * s -> npsd + j] - LOG2IN(LOGRESBIAS));
  }
}

/* llsmrt.c:545-566 */
int o_rt_fetch(o_rtsynth* s, fp* p_out, fp* ap_out) {
  if(s -> nout <= 0) return 0;
  *p_out = ring_read(s -> out_p, -s -> nout);
  *ap_out = ring_read(s -> out_ap, -s -> nout);
  s -> nout --;
  return 1;
}
