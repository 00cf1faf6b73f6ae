/* Frame slabs (libllsm2_amd/csrc/model.cpp) under the container API of container.c / frame.c: a chunk's frames built by
 * llsm_frames_from_flat are used the way a libllsm2 host uses frames -- copied, edited in place, members replaced and
 * grown, deleted one by one and as a chunk -- and every slab must be gone at the end.  tests/test_c_host.py compiles
 * this file with model.cpp under -fsanitize=address (no device code involved). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "llsm.h"
#include "llsm_gpu.h"

void llsm_frames_from_flat(const llsm_flat_params* src, int frm_off, llsm_chunk* dst, int nfrm);

#define CHECK(c) do { if(!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while(0)

static long long live(void) { long long n = 0; llsm_slab_stats(&n, NULL, NULL); return n; }

int main(void) {
  enum { F = 7, MH = 12, ME = 4, NCH = 3, NPSD = 9 };
  float f0[F], ampl[F * MH], phse[F * MH], psd[F * NPSD], psdres[F * NPSD], edc[F * NCH], ea[F * NCH * ME], ep[F * NCH * ME];
  int nhar[F], nhe[F], has[F];
  for(int i = 0; i < F; i ++) {
    f0[i] = (i == 2) ? 0.0f : 100.0f + i; nhar[i] = (i * 5) % (MH + 1); nhe[i] = i % (ME + 1); has[i] = i != 4;
    for(int k = 0; k < MH; k ++) { ampl[i * MH + k] = i + 0.01f * k; phse[i * MH + k] = -i - 0.01f * k; }
    for(int k = 0; k < NPSD; k ++) { psd[i * NPSD + k] = -50.0f - i - k; psdres[i * NPSD + k] = 0.5f * i + k; }
    for(int c = 0; c < NCH; c ++) {
      edc[i * NCH + c] = 1e-3f * (i + c + 1);
      for(int k = 0; k < ME; k ++) { ea[(i * NCH + c) * ME + k] = i + c + k; ep[(i * NCH + c) * ME + k] = 0.1f * (i + c + k); }
    }
  }
  llsm_flat_params v; memset(&v, 0, sizeof(v));
  v.maxnhar = MH; v.maxnhar_e = ME; v.npsd = NPSD; v.nchannel = NCH;
  v.f0 = f0; v.nhar = nhar; v.ampl = ampl; v.phse = phse; v.psd = psd; v.psdres = psdres; v.has_psdres = has;
  v.edc = edc; v.nhar_e = nhe; v.eenv_ampl = ea; v.eenv_phse = ep;

  llsm_aoptions* ao = llsm_create_aoptions();
  ao -> npsd = NPSD; ao -> nchannel = NCH; ao -> maxnhar = MH; ao -> maxnhar_e = ME;
  free(ao -> chanfreq); ao -> chanfreq = (FP_TYPE*)calloc(NCH - 1, sizeof(FP_TYPE)); ao -> chanfreq[0] = 2000; ao -> chanfreq[1] = 6000;
  llsm_container* conf = llsm_aoptions_toconf(ao, 22050.0f);
  llsm_container_attach(conf, LLSM_CONF_NFRM, llsm_create_int(F), llsm_delete_int, llsm_copy_int);
  llsm_chunk* ch = llsm_create_chunk(conf, 0);
  CHECK(ch != NULL && live() == 0);
  llsm_frames_from_flat(&v, 0, ch, F);
  CHECK(live() == 1);
  /* contents */
  for(int i = 0; i < F; i ++) {
    llsm_container* fr = ch -> frames[i];
    FP_TYPE* pf = (FP_TYPE*)llsm_container_get(fr, LLSM_FRAME_F0);
    llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(fr, LLSM_FRAME_HM);
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(fr, LLSM_FRAME_NM);
    FP_TYPE* r = (FP_TYPE*)llsm_container_get(fr, LLSM_FRAME_PSDRES);
    CHECK(pf && *pf == f0[i] && hm && nm);
    CHECK(hm -> nhar == (f0[i] != 0 ? nhar[i] : 0));
    for(int k = 0; k < hm -> nhar; k ++) CHECK(hm -> ampl[k] == ampl[i * MH + k] && hm -> phse[k] == phse[i * MH + k]);
    CHECK(nm -> npsd == NPSD && nm -> nchannel == NCH);
    for(int k = 0; k < NPSD; k ++) CHECK(nm -> psd[k] == psd[i * NPSD + k]);
    for(int c = 0; c < NCH; c ++) {
      CHECK(nm -> edc[c] == edc[i * NCH + c] && nm -> eenv[c] -> nhar == (f0[i] != 0 ? nhe[i] : 0));
      for(int k = 0; k < nm -> eenv[c] -> nhar; k ++) CHECK(nm -> eenv[c] -> ampl[k] == ea[(i * NCH + c) * ME + k]);
    }
    CHECK((r != NULL) == (has[i] != 0));
    if(r) { CHECK(llsm_fparray_length(r) == NPSD); for(int k = 0; k < NPSD; k ++) CHECK(r[k] == psdres[i * NPSD + k]); }
  }
  /* a deep copy is made of ordinary heap objects and survives the chunk */
  llsm_container* keep = llsm_copy_container(ch -> frames[5]);
  llsm_chunk* ch2 = llsm_copy_chunk(ch);
  CHECK(live() == 1);
  /* in-place edits: grow a harmonic frame beyond its slab arrays, shrink and grow a noise frame */
  {
    llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(ch -> frames[1], LLSM_FRAME_HM);
    llsm_hmframe* big = llsm_create_hmframe(40);
    for(int k = 0; k < 40; k ++) { big -> ampl[k] = k; big -> phse[k] = -k; }
    llsm_copy_hmframe_inplace(hm, big);
    CHECK(hm -> nhar == 40 && hm -> ampl[39] == 39 && hm -> phse[39] == -39);
    llsm_hmframe* small = llsm_create_hmframe(2);
    llsm_copy_hmframe_inplace(hm, small);
    llsm_copy_hmframe_inplace(hm, big);                   /* grows the (now heap) arrays again */
    CHECK(hm -> nhar == 40 && hm -> ampl[17] == 17);
    llsm_delete_hmframe(big); llsm_delete_hmframe(small);
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(ch -> frames[3], LLSM_FRAME_NM);
    llsm_nmframe* wide = llsm_create_nmframe(NCH + 2, 6, NPSD + 5);
    llsm_nmframe* narrow = llsm_create_nmframe(1, 1, 3);
    llsm_copy_nmframe_inplace(nm, wide);
    CHECK(nm -> nchannel == NCH + 2 && nm -> npsd == NPSD + 5 && nm -> eenv[NCH + 1] -> nhar == 6);
    llsm_copy_nmframe_inplace(nm, narrow);
    CHECK(nm -> nchannel == 1 && nm -> npsd == 3);
    llsm_copy_nmframe_inplace(nm, wide);
    CHECK(nm -> nchannel == NCH + 2 && nm -> eenv[NCH] -> nhar == 6);
    llsm_delete_nmframe(wide); llsm_delete_nmframe(narrow);
  }
  /* members replaced, removed and attached beyond the container's size (the member arrays move to the heap) */
  llsm_container_attach(ch -> frames[0], LLSM_FRAME_HM, llsm_create_hmframe(3), llsm_delete_hmframe, llsm_copy_hmframe);
  llsm_container_remove(ch -> frames[0], LLSM_FRAME_PSDRES);
  llsm_container_attach(ch -> frames[2], LLSM_FRAME_PBPSYN, llsm_create_int(1), llsm_delete_int, llsm_copy_int);
  CHECK(*(int*)llsm_container_get(ch -> frames[2], LLSM_FRAME_PBPSYN) == 1);
  CHECK(*(FP_TYPE*)llsm_container_get(ch -> frames[2], LLSM_FRAME_F0) == 0.0f);
  llsm_copy_container_inplace(ch -> frames[6], ch -> frames[2]);     /* every member of 6 deleted, copies of 2's attached */
  CHECK(*(int*)llsm_container_get(ch -> frames[6], LLSM_FRAME_PBPSYN) == 1);
  llsm_frame_phaseshift(ch -> frames[4], 0.3f);
  /* a frame taken out of the chunk and deleted on its own; another one replaced by a heap frame */
  llsm_delete_container(ch -> frames[4]);
  ch -> frames[4] = llsm_create_frame(2, NCH, 1, NPSD);
  CHECK(live() == 1);
  llsm_delete_chunk(ch);
  CHECK(live() == 0);
  /* the copies are intact */
  {
    llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(keep, LLSM_FRAME_HM);
    CHECK(hm -> nhar == nhar[5] && hm -> ampl[1] == ampl[5 * MH + 1]);
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(ch2 -> frames[6], LLSM_FRAME_NM);
    CHECK(nm -> psd[2] == psd[6 * NPSD + 2]);
  }
  llsm_delete_container(keep);
  llsm_delete_chunk(ch2);
  /* a second chunk takes the pooled block; trimming hands it back */
  ch = llsm_create_chunk(conf, 0);
  llsm_frames_from_flat(&v, 0, ch, F);
  CHECK(live() == 1);
  llsm_delete_chunk(ch);
  long long pooled = 0; llsm_slab_stats(NULL, NULL, &pooled);
  {
    const char* e = getenv("LLSM_SLAB_POOL_MB");
    CHECK(live() == 0 && (pooled > 0 || (e && e[0] == '0')));     /* LLSM_SLAB_POOL_MB=0: nothing is kept */
  }
  llsm_slab_trim();
  llsm_slab_stats(NULL, NULL, &pooled);
  CHECK(pooled == 0);
  /* llsm_delete_chunks: 40 chunks (slab frames, one with a heap frame put in, one entry NULL) released at once, on
   * several threads; the entries are cleared */
  {
    enum { NC = 40 };
    llsm_chunk* many[NC];
    for(int u = 0; u < NC; u ++) { many[u] = llsm_create_chunk(conf, 0); llsm_frames_from_flat(&v, 0, many[u], F); }
    llsm_delete_container(many[7] -> frames[3]); many[7] -> frames[3] = llsm_create_frame(2, NCH, 1, NPSD);
    llsm_delete_chunk(many[9]); many[9] = NULL;
    CHECK(live() == NC - 1);
    llsm_delete_chunks(many, NC);
    CHECK(live() == 0);
    for(int u = 0; u < NC; u ++) CHECK(many[u] == NULL);
    llsm_slab_trim();
  }
  /* a chunk SPLIT by its host: the last three frame pointers move into a second chunk and NFRM is lowered --
   * legal in the reference, whose frames are independent heap objects.  Deleting the first chunk must not release the
   * slab under the second one (AddressSanitizer sees the use-after-free if it does), in either order of deletion. */
  for(int order = 0; order < 2; order ++) {
    llsm_chunk* a = llsm_create_chunk(conf, 0);
    llsm_frames_from_flat(&v, 0, a, F);
    llsm_container* confb = llsm_copy_container(conf);
    llsm_container_attach(confb, LLSM_CONF_NFRM, llsm_create_int(3), llsm_delete_int, llsm_copy_int);
    llsm_chunk* b = llsm_create_chunk(confb, 0);
    llsm_delete_container(confb);
    for(int i = 0; i < 3; i ++) { b -> frames[i] = a -> frames[F - 3 + i]; a -> frames[F - 3 + i] = NULL; }
    *(int*)llsm_container_get(a -> conf, LLSM_CONF_NFRM) = F - 3;
    CHECK(live() == 1);
    llsm_chunk* first = order ? b : a; llsm_chunk* second = order ? a : b;
    const int nsecond = order ? F - 3 : 3, isecond = order ? 0 : F - 3;
    llsm_delete_chunk(first);
    CHECK(live() == 1);                               /* the other chunk's frames keep the slab alive */
    for(int i = 0; i < nsecond; i ++) {
      llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(second -> frames[i], LLSM_FRAME_NM);
      CHECK(nm && nm -> psd[1] == psd[(isecond + i) * NPSD + 1]);
    }
    llsm_delete_chunk(second);
    CHECK(live() == 0);
  }
  /* ... and a chunk whose host swapped two frames in place is still released whole (by the per-frame walk) */
  {
    llsm_chunk* a = llsm_create_chunk(conf, 0);
    llsm_frames_from_flat(&v, 0, a, F);
    llsm_container* t = a -> frames[1]; a -> frames[1] = a -> frames[5]; a -> frames[5] = t;
    llsm_delete_chunk(a);
    CHECK(live() == 0);
  }
  llsm_slab_trim();
  llsm_delete_container(conf);
  llsm_delete_aoptions(ao);
  printf("slab_host ok\n");
  return 0;
}
