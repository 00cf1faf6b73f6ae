/*
 * rt_oracle.c -- oracle restatement of libllsm2's real-time synthesis buffer
 * (reference: llsmrt.c:32-602, buffer.h:32-217 @ 2.1.0): the harmonic-model
 * path (o_rt_feed) and the pulse-by-pulse path of use_l1 == 1 (o_rt_feed_l1,
 * llsmrt.c:295-420).  TEST INFRASTRUCTURE ONLY; "parity unpinned" (oracle.h).
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

/* ---- buffer.h:140-217 ---- */
typedef struct { fp* frwd; fp* bkwd; int capacity; int curr; } dual;
static dual* dual_create(int capacity) {
  dual* d = malloc(sizeof(dual));
  d -> capacity = capacity; d -> curr = 0;
  d -> frwd = calloc(capacity, sizeof(fp)); d -> bkwd = calloc(capacity, sizeof(fp));
  return d;
}
static void dual_delete(dual* d) { if(d) { free(d -> frwd); free(d -> bkwd); free(d); } }
static void dual_readchunk(dual* d, int offset, int size, fp* dst) {
  int before = offset > 0 ? 0 : -offset;
  if(before > size) before = size;
  int base = d -> curr + d -> capacity;
  for(int i = 0; i < before; i ++) dst[i] = d -> bkwd[(base + offset + i) % d -> capacity];
  for(int i = before; i < size; i ++) dst[i] = d -> frwd[(base + offset + i) % d -> capacity];
}
static void dual_forward(dual* d, int size) {
  for(int i = 0; i < size; i ++) {
    d -> bkwd[d -> curr] = d -> frwd[d -> curr];
    d -> frwd[d -> curr] = 0;
    d -> curr = (d -> curr + 1) % d -> capacity;
  }
}
static void dual_addchunk(dual* d, int offset, int size, const fp* src) {
  int before = offset > 0 ? 0 : -offset;
  if(before > size) before = size;
  int base = d -> curr + d -> capacity;
  for(int i = 0; i < before; i ++) d -> bkwd[(base + offset + i) % d -> capacity] += src[i];
  for(int i = before; i < size; i ++) d -> frwd[(base + offset + i) % d -> capacity] += src[i];
}

/* ---- llsmrt.c:32-78 ---- */
struct o_rtsynth {
  fp pulse; int pbp_offset, pbp_state; dual* pulse_buf;
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
  o_hanning_ola(s -> win, nwin);
  volatile float e = s -> cycle + s -> thop;
  volatile float ef = e * s -> fs;
  s -> next_nhop = (int)floor((double)ef);
  s -> pulse -= prev_nhop;                                   /* llsmrt.c:116-118 */
  if(s -> pbp_state && s -> pbp_offset > s -> sin_pos + s -> curr_nhop) s -> pbp_offset -= prev_nhop;
  dual_forward(s -> pulse_buf, s -> curr_nhop);
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
  s -> pulse_buf = dual_create(s -> ninternal);
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
  ring_delete(s -> exc_mix); ring_delete(s -> noise); ring_delete(s -> sin); dual_delete(s -> pulse_buf);
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
  o_moving_avg(psd, nspec, o_conv_mavg_half(), env);
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

/* feed_sinusoids (llsmrt.c:273-291) on row i of p */
static void feed_sinusoids(o_rtsynth* s, const o_params* p, int i) {
  fp f0 = p -> f0[i];
  if(!(f0 > 0)) return;
  int nwin = s -> curr_nhop * 2;
  fp* x = malloc(sizeof(fp) * nwin);
  fp phase_shift = (fp)((double)(float)(s -> cycle * 2) * M_PI * f0);
  int nhar = imin(p -> nhar[i], s -> nfft);
  fp* phase = malloc(sizeof(fp) * (nhar > 0 ? nhar : 1));
  for(int k = 0; k < nhar; k ++)
    phase[k] = (fp)(p -> phse[(size_t)i * p -> maxnhar + k] - phase_shift * (k + 1.0));
  o_synth_harmonic_frame_auto(& s -> opt, p -> ampl + (size_t)i * p -> maxnhar, phase, nhar, f0 / s -> fs, nwin, x);
  for(int j = 0; j < nwin; j ++) x[j] *= s -> win[j];
  ring_addchunk(s -> sin, -nwin, nwin, x);
  free(phase); free(x);
}

/* llsm_rtsynth_buffer_feed with options.use_l1 = 1: llsmrt.c:505-521 with the deterministic part of
 * :295-420 (pulse tracker, onset / termination, dual-buffer overlap-add, HM hand-over) */
void o_rt_feed_l1(o_rtsynth* s, o_params* p, o_l1params* q, int i, int maxnhar_conf,
  o_fgfm effect, void* effect_info) {
  update_cycle(s);
  int nch = s -> nchannel, me = p -> maxnhar_e;
  int nhop = s -> curr_nhop, nwin = nhop * 2;
  fp f0 = p -> f0[i];
  fp* x = malloc(sizeof(fp) * nwin);
  for(int c = 0; c < nch; c ++) {                              /* feed_modcomps */
    int nh = f0 > 0 ? p -> nhar_e[i] : 0;
    o_synth_harmonic_frame_auto(& s -> opt, p -> eenv_ampl + ((size_t)i * nch + c) * me,
      p -> eenv_phse + ((size_t)i * nch + c) * me, nh, f0 / s -> fs, nwin, x);
    fp offset = p -> edc[(size_t)i * nch + c];
    for(int j = 0; j < nwin; j ++) x[j] = fpmax(x[j] + offset, (fp)1e-8) * s -> win[j];
    ring_addchunk(s -> mod[c], -nwin, nwin, x);
  }
  free(x);
  if(q -> has_l1[i] && f0 != 0) {
    const fp* vsphse = q -> vsphse + (size_t)i * q -> maxnhar;
    const fp* vtmagn = q -> vtmagn + (size_t)i * q -> nspec;
    int pbp_on = q -> pbpsyn[i] == 1, nspec = q -> nspec;
    fp fs = s -> fs;
    fp len_period = fs / f0;
    o_lfmodel source_model = o_lfmodel_from_rd(q -> rd[i], (fp)(1.0 / f0), 1.0);
    fp pulse_projected = o_pulse_projection(q -> rd[i], f0, vsphse[0], fs, 0);
    int len_reset = (int)(fmax((double)len_period, (double)nhop) * 2);
    if(pulse_projected - s -> pulse > len_reset) s -> pulse = pulse_projected - len_reset;
    int num_periods = (int)round((double)((pulse_projected - s -> pulse) / len_period));
    if(num_periods > 0) len_period = (pulse_projected - s -> pulse) / num_periods;
    int pulse_size = (int)pow(2.0, ceil(log2(fmax((double)len_period * 2, (double)nspec))));
    int pbp_onset = 0, pbp_termination = 0;
    if(pbp_on && ! s -> pbp_state) {
      pbp_onset = 1; s -> pbp_state = 1; s -> pbp_offset = -nhop;
      if(! q -> has_hm[i]) o_frame_tolayer0(p, q, i, maxnhar_conf);
      feed_sinusoids(s, p, i);
    }
    if(! pbp_on && s -> pbp_state) {
      pbp_termination = 1; s -> pbp_state = 0;
      num_periods += (int)ceil((double)((-s -> pbp_offset) / len_period));
    }
    if(s -> pbp_state || pbp_termination) {
      int period_begin = pbp_onset ? -2 : 0, period_end = num_periods;
      int num_pulses = period_end - period_begin;
      int pre_rotate = (int)(len_period < nhop * 2 ? len_period : (fp)(nhop * 2));
      if(num_pulses > 0) {
        fp* offsets = calloc(num_pulses, sizeof(fp));
        o_lfmodel* sources = calloc(num_pulses, sizeof(o_lfmodel));
        for(int k = 0; k < num_pulses; k ++) {
          fp delta_t = 0;
          if(effect != NULL && q -> has_eff[i]) {
            o_gfm g = o_lfmodel_to_gfm(source_model);
            effect(& g, & delta_t, effect_info, i);
            sources[k] = o_gfm_to_lfmodel(g);
          } else sources[k] = source_model;
          offsets[k] = s -> pulse + (k + period_begin) * len_period + delta_t * fs;
        }
        int pulse_base = (int)offsets[0];
        for(int k = 0; k < num_pulses; k ++) offsets[k] -= pulse_base;
        fp* y = calloc(pulse_size, sizeof(fp));
        o_make_filtered_pulse(q -> rd[i], f0, vtmagn, nspec, vsphse, q -> nvsphse[i], sources, offsets, num_pulses,
          pre_rotate, pulse_size, s -> fnyq, q -> lip_radius, fs, y);
        dual_addchunk(s -> pulse_buf, pulse_base - pre_rotate - nhop, pulse_size, y);
        free(y); free(offsets); free(sources);
      }
    }
    if(! s -> pbp_state) {
      if(! q -> has_hm[i]) o_frame_tolayer0(p, q, i, maxnhar_conf);
      feed_sinusoids(s, p, i);
    }
    s -> pulse = pulse_projected;
    if(s -> pbp_state && s -> pbp_offset <= s -> sin_pos + nhop) {
      fp* xx = calloc(nhop * 2, sizeof(fp));
      dual_readchunk(s -> pulse_buf, s -> pbp_offset, nhop * 2, xx);
      for(int j = 0; j < nhop * 2; j ++) xx[j] *= s -> win[j];
      ring_addchunk(s -> sin, s -> pbp_offset, nhop * 2, xx);
      free(xx);
    }
    if(pbp_termination) {
      int size = -nhop - s -> pbp_offset;
      if(size > 0) {
        fp* xx = calloc(size, sizeof(fp));
        dual_readchunk(s -> pulse_buf, s -> pbp_offset, size, xx);
        for(int j = 0; j < nhop; j ++) { xx[j] *= s -> win[j]; xx[size - nhop + j] *= s -> win[j + nhop]; }
        ring_addchunk(s -> sin, s -> pbp_offset, size, xx);
        free(xx);
      }
    }
  }
  run_excitation_buffers(s, s -> curr_nhop);
  feed_filter(s);
  fp* x_nos = malloc(sizeof(fp) * s -> next_nhop);
  fp* x_sin = malloc(sizeof(fp) * s -> next_nhop);
  ring_readchunk(s -> noise, -s -> nfft, s -> next_nhop, x_nos);
  ring_readchunk(s -> sin, s -> sin_pos, s -> next_nhop, x_sin);
  ring_appendchunk(s -> out_p, s -> next_nhop, x_sin);
  ring_appendchunk(s -> out_ap, s -> next_nhop, x_nos);
  s -> nout += s -> next_nhop;
  free(x_nos); free(x_sin);
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
