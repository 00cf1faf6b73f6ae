/* C99 host linked against libllsm2_amd.so through the installed headers only (llsm.h, llsmrt.h, dsputils.h,
 * llsmutils.h, buffer.h, llsm_gpu.h), the way a user of the reference's libllsm2.a is.
 *   host_main cpu   no device needed: data-model calls work, compute entry points fail loudly (NULL + message)
 *   host_main gpu   test/test-harmonic.c:32-48 (ICZT == sinusoid bank), llsm_harmonic_czt on a known frame,
 *                   llsm_analyze -> llsm_synthesize on a synthetic vowel, llsmrt feed / fetch, all through C
 * Built and run by tests/test_c_host.py with gcc -std=c99 -Wall -Wextra. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "llsm.h"
#include "llsmrt.h"
#include "dsputils.h"
#include "llsmutils.h"
#include "buffer.h"
#include "llsm_gpu.h"

#define CHECK(c) do { if(!(c)) { fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #c, __LINE__, llsm_gpu_last_error()); return 1; } } while(0)
static const double PI = 3.14159265358979323846;

static int run_cpu(void) {
  llsm_aoptions* ao = llsm_create_aoptions();
  llsm_container* conf = llsm_aoptions_toconf(ao, 22050.0f);
  CHECK(conf != NULL && llsm_conf_checklayer0(conf) && ! llsm_conf_checklayer1(conf));
  llsm_container* fr = llsm_create_frame(10, 4, 4, 256);
  CHECK(llsm_frame_checklayer0(fr) && ! llsm_frame_checklayer1(fr));
  llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(fr, LLSM_FRAME_HM);
  hm -> phse[2] = 1.0f;
  llsm_frame_phaseshift(fr, 0.5f);
  CHECK(fabs(hm -> phse[2] - 2.5f) < 1e-6);
  llsm_ringbuffer* rb = llsm_create_ringbuffer(16);
  llsm_ringbuffer_append(rb, 3.0f);
  CHECK(llsm_ringbuffer_read(rb, -1) == 3.0f);
  llsm_delete_ringbuffer(rb);
  /* host-side helpers need no device */
  FP_TYPE f0v[3] = {0, 120, 200};
  CHECK(llsm_get_fftsize(f0v, 3, 44100.0f, 4.0f) == 2048);
  lfmodel lf = llsm_lfmodel_from_rd(1.0f, 1.0f / 200.0f, 1.0f);
  llsm_gfm g = llsm_lfmodel_to_gfm(lf);
  lfmodel lf2 = llsm_gfm_to_lfmodel(g);
  CHECK(fabs(lf2.te - lf.te) < 1e-5 && fabs(lf2.ta - lf.ta) < 1e-6);
  if(llsm_gpu_device_count() == 0) {
    FP_TYPE x[64] = {0}, f0[2] = {0, 0};
    llsm_chunk* ch = llsm_analyze(ao, x, 64, 44100.0f, f0, 2, NULL);
    CHECK(ch == NULL && strstr(llsm_gpu_last_error(), "no CPU fallback") != NULL);
    printf("cpu: data model ok; llsm_analyze without a device -> NULL: \"%s\"\n", llsm_gpu_last_error());
  } else printf("cpu: data model ok (a device is present)\n");
  llsm_delete_container(fr); llsm_delete_container(conf); llsm_delete_aoptions(ao);
  return 0;
}

static int run_gpu(void) {
  /* ---- test/test-harmonic.c:32-48: the two frame synthesisers agree ---- */
  enum { NH = 100, NX = 1024 };
  FP_TYPE ampl[NH], phse[NH];
  srand(3);
  for(int i = 0; i < NH; i ++) { ampl[i] = (FP_TYPE)rand() / RAND_MAX - 0.5f; phse[i] = 100.0f * ((FP_TYPE)rand() / RAND_MAX - 0.5f); }
  FP_TYPE* y1 = llsm_synthesize_harmonic_frame_iczt(ampl, phse, NH, 0.01f, NX);
  FP_TYPE* y2 = llsm_synthesize_harmonic_frame(ampl, phse, NH, 0.01f, NX);
  double err = 0, en = 0, ref_err = 0;
  for(int t = 0; t < NX; t ++) {
    double r = 0;
    for(int k = 0; k < NH; k ++) r += ampl[k] * cos(2.0 * PI * (double)0.01f * (k + 1) * (t - NX / 2) + phse[k]);
    err += (y1[t] - y2[t]) * (y1[t] - y2[t]); en += r * r; ref_err += (y2[t] - r) * (y2[t] - r);
  }
  CHECK(en > 1.0 && err <= 1e-12 * en && ref_err <= 1e-10 * en);
  printf("gpu: ICZT vs sinusoid bank %.1f dB, vs closed form %.1f dB\n", 10 * log10(err / en + 1e-30), 10 * log10(ref_err / en));
  free(y1); free(y2);
  /* ---- llsm_harmonic_czt on a windowed three-harmonic frame ---- */
  enum { NW = 1470 };
  const double fs = 44100.0, f0 = 120.0;
  FP_TYPE x[NW], a3[3], p3[3];
  const double at[3] = {0.7, 0.2, 0.05}, pt[3] = {0.3, -1.1, 2.0};
  for(int t = 0; t < NW; t ++) {
    x[t] = 0;
    for(int k = 0; k < 3; k ++) x[t] += (FP_TYPE)(at[k] * cos(2.0 * PI * f0 * (k + 1) / fs * (t - NW / 2) + pt[k]));
  }
  llsm_harmonic_czt(x, NW, (FP_TYPE)f0, (FP_TYPE)fs, 3, a3, p3);
  for(int k = 0; k < 3; k ++) CHECK(fabs(a3[k] - at[k]) < 2e-3 && fabs(p3[k] - pt[k]) < 2e-3);
  printf("gpu: llsm_harmonic_czt %.4f %.4f %.4f / %.3f %.3f %.3f\n", a3[0], a3[1], a3[2], p3[0], p3[1], p3[2]);
  /* ---- llsm_analyze -> llsm_synthesize on 0.5 s of a synthetic vowel ---- */
  const int nx = 22050, nfrm = 100;
  FP_TYPE* sig = calloc(nx, sizeof(FP_TYPE)); FP_TYPE* f0s = calloc(nfrm, sizeof(FP_TYPE));
  for(int t = 0; t < nx; t ++)
    for(int k = 1; k <= 20; k ++) sig[t] += (FP_TYPE)(0.3 / k * cos(2.0 * PI * 150.0 * k * t / fs + 0.37 * k * k));
  for(int i = 0; i < nfrm; i ++) f0s[i] = 150.0f;
  llsm_aoptions* ao = llsm_create_aoptions();
  ao -> f0_refine = 0;
  llsm_soptions* so = llsm_create_soptions((FP_TYPE)fs);
  FP_TYPE* xap = NULL;
  llsm_chunk* ch = llsm_analyze(ao, sig, nx, (FP_TYPE)fs, f0s, nfrm, & xap);
  CHECK(ch != NULL && xap != NULL);
  llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(ch -> frames[50], LLSM_FRAME_HM);
  CHECK(hm != NULL && hm -> nhar == 100 && fabs(hm -> ampl[0] - 0.3) < 3e-3 && fabs(hm -> ampl[3] - 0.075) < 2e-3);
  llsm_output* out = llsm_synthesize(so, ch);
  CHECK(out != NULL && out -> ny == 22271);
  double d = 0, e = 0;
  for(int t = 2000; t < nx - 2000; t ++) { d += (out -> y_sin[t] - sig[t]) * (out -> y_sin[t] - sig[t]); e += sig[t] * sig[t]; }
  printf("gpu: analyze -> synthesize, harmonic part vs input %.1f dB\n", 10 * log10(d / e));
  CHECK(d < 1e-3 * e);
  /* ---- llsmrt: one producer / consumer loop ---- */
  llsm_rtsynth_buffer* rt = llsm_create_rtsynth_buffer(so, ch -> conf, 4096);
  CHECK(rt != NULL);
  const int lat = llsm_rtsynth_buffer_getlatency(rt);
  int got = 0; FP_TYPE smp; double e2 = 0, d2 = 0;
  for(int i = 0; i < nfrm; i ++) {
    llsm_rtsynth_buffer_feed(rt, ch -> frames[i]);
    while(llsm_rtsynth_buffer_fetch(rt, & smp)) {
      const int t = got - lat;
      if(t >= 3000 && t < nx - 3000) { d2 += (smp - out -> y[t]) * (smp - out -> y[t]); e2 += out -> y[t] * out -> y[t]; }
      got ++;
    }
  }
  printf("gpu: llsmrt %d samples, latency %d, vs offline %.1f dB\n", got, lat, 10 * log10(d2 / e2));
  CHECK(got > nx - 1000 && d2 < 0.05 * e2);              /* same harmonic part, another noise realisation */
  llsm_delete_rtsynth_buffer(rt);
  llsm_delete_output(out); llsm_delete_chunk(ch); free(xap);
  llsm_delete_aoptions(ao); llsm_delete_soptions(so); free(sig); free(f0s);
  return 0;
}

int main(int argc, char** argv) {
  if(argc > 1 && ! strcmp(argv[1], "gpu")) return run_gpu();
  return run_cpu();
}
