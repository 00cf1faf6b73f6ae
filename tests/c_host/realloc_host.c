/* C99 host: the reference's ownership rule on the DROP-IN entry point (SURVEY 8(b) "Ownership": every pointer is its own
 * heap block).  A host may hand a member ARRAY of an analysed frame to realloc / free itself -- the reference's
 * demo-stretch.c:28-29 does it to eenv->ampl -- so the frames llsm_analyze returns must be ordinary heap objects by
 * default.  (The additive llsm_analyze_batch carves a chunk's frames out of one slab; that is its documented contract.)
 * Run by tests/test_c_host.py under MALLOC_CHECK_=3 / MALLOC_PERTURB_, where glibc aborts on a realloc / free of a
 * pointer that is not the start of a heap block. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "llsm.h"
#include "llsm_gpu.h"

#define CHECK(c) do { if(!(c)) { fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #c, __LINE__, llsm_gpu_last_error()); return 1; } } while(0)

int main(void) {
  enum { NX = 22050, NF = 100 };
  const double PI = 3.14159265358979323846;
  FP_TYPE* x = (FP_TYPE*)calloc(NX, sizeof(FP_TYPE));
  FP_TYPE f0[NF];
  for(int n = 0; n < NX; n ++)
    for(int k = 1; k <= 20; k ++) x[n] += (FP_TYPE)(0.2 / k * cos(2.0 * PI * k * 150.0 * n / 44100.0 + k));
  for(int i = 0; i < NF; i ++) f0[i] = (i % 17 == 3) ? 0.0f : 150.0f;
  llsm_aoptions* ao = llsm_create_aoptions();
  ao -> f0_refine = 0;
  llsm_chunk* ch = llsm_analyze(ao, x, NX, 44100.0f, f0, NF, NULL);
  CHECK(ch != NULL);
  long long live = -1; llsm_slab_stats(&live, NULL, NULL);
  CHECK(live == 0);                                   /* no slab behind a drop-in chunk */
  int grown = 0;
  for(int i = 0; i < NF; i ++) {
    llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(ch -> frames[i], LLSM_FRAME_HM);
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(ch -> frames[i], LLSM_FRAME_NM);
    CHECK(hm != NULL && nm != NULL);
    /* grow the harmonic arrays in place, the way a host that adds harmonics does */
    const int n2 = hm -> nhar + 7;
    hm -> ampl = (FP_TYPE*)realloc(hm -> ampl, sizeof(FP_TYPE) * (size_t)n2);
    hm -> phse = (FP_TYPE*)realloc(hm -> phse, sizeof(FP_TYPE) * (size_t)n2);
    CHECK(hm -> ampl != NULL && hm -> phse != NULL);
    for(int k = hm -> nhar; k < n2; k ++) { hm -> ampl[k] = 0; hm -> phse[k] = 0; }
    hm -> nhar = n2;
    /* replace the PSD by a block of the host's own */
    FP_TYPE* psd = (FP_TYPE*)malloc(sizeof(FP_TYPE) * (size_t)nm -> npsd);
    memcpy(psd, nm -> psd, sizeof(FP_TYPE) * (size_t)nm -> npsd);
    free(nm -> psd); nm -> psd = psd;
    /* demo-stretch.c:28-29: the envelope arrays */
    for(int c = 0; c < nm -> nchannel; c ++) {
      llsm_hmframe* e = nm -> eenv[c];
      e -> ampl = (FP_TYPE*)realloc(e -> ampl, sizeof(FP_TYPE) * (size_t)(e -> nhar + 1));
      e -> phse = (FP_TYPE*)realloc(e -> phse, sizeof(FP_TYPE) * (size_t)(e -> nhar + 1));
      CHECK(e -> ampl != NULL && e -> phse != NULL);
    }
    grown ++;
  }
  /* the edited chunk still synthesises, and every block goes back through the reference's destructors */
  llsm_soptions* so = llsm_create_soptions(44100.0f);
  llsm_output* out = llsm_synthesize(so, ch);
  CHECK(out != NULL && out -> ny > 0);
  double e = 0; for(int n = 0; n < out -> ny; n ++) e += (double)out -> y[n] * out -> y[n];
  CHECK(e > 0 && e == e);
  llsm_delete_output(out); llsm_delete_soptions(so);
  llsm_delete_chunk(ch); llsm_delete_aoptions(ao); free(x);
  printf("realloc_host ok: %d analysed frames had their member arrays realloc'd / freed by the host\n", grown);
  return 0;
}
