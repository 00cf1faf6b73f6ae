/* C99 host: a two-fold time stretch done the way a libllsm2 user edits a model -- through the container API only.
 * The workflow is the one the reference's demo program walks (test/demo-stretch.c: analyse, go to layer 1, undo the
 * phase propagation, build a chunk with twice the frames by copying and blending neighbouring frames, back to layer 0,
 * propagate phases, synthesise); the code below is this repository's own: linear blends of F0 / Rd / vocal-tract
 * magnitude / noise PSD / band energies, circular blends of the phase rows, the envelope harmonics of the nearer frame.
 * It exercises what a drop-in has to survive: llsm_copy_container deep copies, llsm_container_attach replacing
 * members (with destructors and copy constructors), a conf with an edited NFRM, llsm_create_chunk without frames,
 * llsm_chunk_tolayer1 / tolayer0 / phasepropagate on edited frames, llsm_synthesize of a chunk no analysis produced.
 * Checks: twice the length, finite, same level and same long-term spectrum as the unstretched resynthesis, F0 kept.
 * Built and run by tests/test_c_host.py (gcc -std=c99 -Wall -Wextra -Werror -pedantic). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "llsm.h"
#include "llsm_gpu.h"

#define CHECK(c) do { if(!(c)) { fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #c, __LINE__, llsm_gpu_last_error()); return 1; } } while(0)
static const double PI = 3.14159265358979323846;

static FP_TYPE mixf(FP_TYPE a, FP_TYPE b, FP_TYPE r) { return a + (b - a) * r; }
static FP_TYPE mix_angle(FP_TYPE a, FP_TYPE b, FP_TYPE r) {          /* blend on the unit circle */
  const double x = (1.0 - r) * cos(a) + r * cos(b), y = (1.0 - r) * sin(a) + r * sin(b);
  return (FP_TYPE)atan2(y, x);
}

/* dst (a deep copy of frame A) <- blend of A and B at ratio r in [0, 1) */
static void blend_frame(llsm_container* dst, llsm_container* b, FP_TYPE r) {
  FP_TYPE fa = *(FP_TYPE*)llsm_container_get(dst, LLSM_FRAME_F0), fb = *(FP_TYPE*)llsm_container_get(b, LLSM_FRAME_F0);
  llsm_nmframe* na = (llsm_nmframe*)llsm_container_get(dst, LLSM_FRAME_NM);
  llsm_nmframe* nb = (llsm_nmframe*)llsm_container_get(b, LLSM_FRAME_NM);
  for(int j = 0; j < na -> npsd; j ++) na -> psd[j] = mixf(na -> psd[j], nb -> psd[j], r);
  for(int c = 0; c < na -> nchannel; c ++) na -> edc[c] = mixf(na -> edc[c], nb -> edc[c], r);
  if(r >= 0.5f)                                     /* envelope harmonics: those of the nearer frame */
    for(int c = 0; c < na -> nchannel; c ++) {
      llsm_delete_hmframe(na -> eenv[c]);
      na -> eenv[c] = llsm_copy_hmframe(nb -> eenv[c]);
    }
  if(fa > 0 && fb > 0) {                            /* both voiced: blend the layer-1 members */
    FP_TYPE* rda = (FP_TYPE*)llsm_container_get(dst, LLSM_FRAME_RD); FP_TYPE* rdb = (FP_TYPE*)llsm_container_get(b, LLSM_FRAME_RD);
    FP_TYPE* vta = (FP_TYPE*)llsm_container_get(dst, LLSM_FRAME_VTMAGN); FP_TYPE* vtb = (FP_TYPE*)llsm_container_get(b, LLSM_FRAME_VTMAGN);
    FP_TYPE* vsa = (FP_TYPE*)llsm_container_get(dst, LLSM_FRAME_VSPHSE); FP_TYPE* vsb = (FP_TYPE*)llsm_container_get(b, LLSM_FRAME_VSPHSE);
    const int ns = llsm_fparray_length(vta), nha = llsm_fparray_length(vsa), nhb = llsm_fparray_length(vsb);
    const int nh = nha > nhb ? nha : nhb, nmin = nha < nhb ? nha : nhb;
    FP_TYPE* vt = llsm_create_fparray(ns); FP_TYPE* vs = llsm_create_fparray(nh);
    for(int k = 0; k < ns; k ++) vt[k] = mixf(vta[k], vtb[k], r);
    for(int k = 0; k < nh; k ++) vs[k] = k < nmin ? mix_angle(vsa[k], vsb[k], r) : (nha > nhb ? vsa[k] : vsb[k]);
    const FP_TYPE f0 = mixf(fa, fb, r), rd = mixf(*rda, *rdb, r);
    llsm_container_attach(dst, LLSM_FRAME_F0, llsm_create_fp(f0), llsm_delete_fp, llsm_copy_fp);
    llsm_container_attach(dst, LLSM_FRAME_RD, llsm_create_fp(rd), llsm_delete_fp, llsm_copy_fp);
    llsm_container_attach(dst, LLSM_FRAME_VTMAGN, vt, llsm_delete_fparray, llsm_copy_fparray);
    llsm_container_attach(dst, LLSM_FRAME_VSPHSE, vs, llsm_delete_fparray, llsm_copy_fparray);
  }
  /* voiced / unvoiced boundary: the copy of A stands (no blend across a voicing change) */
}

/* power in `nb` log-spaced bands between 100 Hz and 8 kHz: Hann-windowed single-frequency probes, 16 per band
 * (a harmonic spectrum needs several probes per band to be caught), averaged over 2048-sample segments */
static void band_profile(const FP_TYPE* y, int n, double fs, double* out, int nb) {
  for(int b = 0; b < nb; b ++) {
    double acc = 0;
    for(int q = 0; q < 16; q ++) {
      const double f = 100.0 * pow(80.0, (b + (q + 0.5) / 16.0) / nb);
      for(int seg = 0; seg + 2048 <= n; seg += 2048) {
        double re = 0, im = 0;
        for(int t = 0; t < 2048; t ++) {
          const double w = 0.5 - 0.5 * cos(2 * PI * t / 2047.0), ph = 2 * PI * f * t / fs;
          re += w * y[seg + t] * cos(ph); im -= w * y[seg + t] * sin(ph);
        }
        acc += re * re + im * im;
      }
    }
    out[b] = 10.0 * log10(acc / (n / 2048) + 1e-20);
  }
}

int main(void) {
  if(llsm_gpu_device_count() == 0) { printf("stretch: no device\n"); return 2; }
  const FP_TYPE fs = 22050.0f;
  const int nhop = 128, nx = 30000, nfrm = nx / nhop;
  FP_TYPE* x = (FP_TYPE*)calloc((size_t)nx, sizeof(FP_TYPE));
  FP_TYPE* f0 = (FP_TYPE*)calloc((size_t)nfrm, sizeof(FP_TYPE));
  /* a vowel-like glide 140 -> 180 Hz with a formant-ish harmonic roll-off, an unvoiced gap, a little noise */
  double ph = 0; unsigned s = 12345u;
  for(int t = 0; t < nx; t ++) {
    const double u = (double)t / nx, f = 140.0 + 40.0 * u;
    const int voiced = !(t > 13000 && t < 15500);
    ph += 2 * PI * f / fs;
    double v = 0;
    if(voiced) for(int k = 1; k <= 30; k ++) v += 0.25 / k * (1.0 + 0.8 * exp(-pow((k * f - 900.0) / 400.0, 2))) * cos(k * ph + 0.3 * k);
    s = s * 1664525u + 1013904223u;
    x[t] = (FP_TYPE)(v + 0.004 * ((double)(s >> 8) / 8388608.0 - 1.0));
  }
  for(int i = 0; i < nfrm; i ++) { const int t = i * nhop; f0[i] = (t > 13000 && t < 15500) ? 0.0f : (FP_TYPE)(140.0 + 40.0 * t / nx); }

  llsm_aoptions* oa = llsm_create_aoptions();
  oa -> thop = (FP_TYPE)nhop / fs; oa -> f0_refine = 0;
  llsm_soptions* os = llsm_create_soptions(fs);
  llsm_chunk* chunk = llsm_analyze(oa, x, nx, fs, f0, nfrm, NULL);
  CHECK(chunk != NULL);
  llsm_output* plain = llsm_synthesize(os, chunk);
  CHECK(plain != NULL);

  llsm_chunk_tolayer1(chunk, 2048);
  llsm_chunk_phasepropagate(chunk, -1);
  const int nnew = 2 * nfrm;
  llsm_container* conf2 = llsm_copy_container(chunk -> conf);
  llsm_container_attach(conf2, LLSM_CONF_NFRM, llsm_create_int(nnew), llsm_delete_int, llsm_copy_int);
  llsm_chunk* longer = llsm_create_chunk(conf2, 0);
  llsm_delete_container(conf2);
  CHECK(longer != NULL);
  for(int i = 0; i < nnew; i ++) {
    const FP_TYPE pos = (FP_TYPE)i * nfrm / nnew;
    int a = (int)pos; const FP_TYPE r = pos - a;
    if(a > nfrm - 2) a = nfrm - 2;
    longer -> frames[i] = llsm_copy_container(chunk -> frames[a]);
    blend_frame(longer -> frames[i], chunk -> frames[a + 1], r);
    /* the copy carries A's harmonic model: the stretched frame gets its own from layer 1 */
    llsm_container_attach(longer -> frames[i], LLSM_FRAME_HM, NULL, NULL, NULL);
  }
  llsm_chunk_tolayer0(longer);
  llsm_chunk_phasepropagate(longer, 1);
  llsm_output* out = llsm_synthesize(os, longer);
  CHECK(out != NULL);

  /* ---- checks ---- */
  CHECK(abs(out -> ny - 2 * plain -> ny) <= 2 * nhop + 2);
  double e1 = 0, e2 = 0; int bad = 0;
  for(int t = 0; t < plain -> ny; t ++) e1 += (double)plain -> y[t] * plain -> y[t];
  for(int t = 0; t < out -> ny; t ++) { e2 += (double)out -> y[t] * out -> y[t]; if(!isfinite(out -> y[t])) bad ++; }
  const double lvl = 10.0 * log10((e2 / out -> ny) / (e1 / plain -> ny));
  enum { NB = 12 };
  double p1[NB], p2[NB], worst = 0;
  band_profile(plain -> y, plain -> ny, fs, p1, NB); band_profile(out -> y, out -> ny, fs, p2, NB);
  for(int b = 0; b < NB; b ++) if(fabs(p1[b] - p2[b]) > worst) worst = fabs(p1[b] - p2[b]);
  int nf0 = 0;
  FP_TYPE* f0new = llsm_chunk_getf0(longer, & nf0);
  CHECK(f0new != NULL && nf0 == nnew);
  double f0err = 0;
  for(int i = 0; i < nnew; i ++) { const double want = f0[(i / 2 < nfrm ? i / 2 : nfrm - 1)]; if(want > 0 && f0new[i] > 0 && fabs(f0new[i] - want) > f0err) f0err = fabs(f0new[i] - want); }
  printf("stretch: %d -> %d samples, level %+.2f dB, worst band difference %.2f dB, F0 deviation %.2f Hz, non-finite %d\n",
    plain -> ny, out -> ny, lvl, worst, f0err, bad);
  CHECK(bad == 0 && fabs(lvl) < 1.0 && worst < 3.0 && f0err < 1.0);
  free(f0new);
  llsm_delete_output(out); llsm_delete_output(plain);
  llsm_delete_chunk(longer); llsm_delete_chunk(chunk);
  llsm_delete_aoptions(oa); llsm_delete_soptions(os);
  free(x); free(f0);
  printf("stretch ok\n");
  return 0;
}
