// Frames laid over packed per-frame records (libllsm2_amd/csrc/model.cpp llsm_frames_packed_begin / _finish,
// llsm_chunk_packed_view; record layout csrc/packed.h) -- host code only, no device library in the process: the records
// a device would have written are written here by hand.  tests/test_c_host.py compiles this file with model.cpp under
// -fsanitize=address,undefined.  Checked: every member the reference's accessors return reads the record's values; the
// view is granted while the frames lie untouched (values edited in place included) and withdrawn after any change of
// structure (member replaced, arrays regrown, frame replaced, envelope counts made unequal); copies are heap objects
// that outlive the chunk; deletion frame by frame, by chunk and through the pristine shortcut leaves no slab behind.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "llsm.h"
#include "llsm_gpu.h"
#include "packed.h"
#include "model_internal.h"

namespace {
int g_bad = 0;
#define CHECK(c) do { if(! (c)) { std::printf("line %d: %s\n", __LINE__, #c); g_bad ++; } } while(0)

float val(int frame, int piece, int k) { return (float)(frame * 1000 + piece * 100) + 0.25f * k; }

// what k_pack_frames writes for frame i
void write_record(float* rec, const LlsmPackedLayout& L, int i, int nhar, int nhe, int has) {
  int* ri = (int*)rec;
  std::memset(rec, 0, sizeof(float) * L.words);
  rec[0] = i % 7 == 3 ? 0.0f : 100.0f + i; ri[1] = nhar; ri[2] = nhe; ri[3] = has;
  for(int k = 0; k < L.maxnhar; k ++) { rec[L.o_ampl + k] = val(i, 1, k); rec[L.o_phse + k] = val(i, 2, k); }
  for(int k = 0; k < L.npsd; k ++) { rec[L.o_psd + k] = val(i, 3, k); rec[L.o_psdres + k] = val(i, 4, k); }
  ri[L.o_reshdr + 3] = L.npsd;
  for(int c = 0; c < L.nch; c ++) {
    rec[L.o_edc + c] = val(i, 5, c);
    for(int k = 0; k < L.me; k ++) { rec[L.o_eamp + c * L.me + k] = val(i, 6, c * 10 + k); rec[L.o_ephs + c * L.me + k] = val(i, 7, c * 10 + k); }
  }
}

llsm_chunk* build(llsm_container* conf, const LlsmPackedLayout& L, int F, std::vector<float>& f0_out, const void** recs) {
  llsm_chunk* ch = llsm_create_chunk(conf, 0);
  void* token = nullptr;
  float* rec = (float*)llsm_frames_packed_begin(F, & L, & token, 0);
  CHECK(rec && token);
  for(int i = 0; i < F; i ++)
    write_record(rec + (size_t)i * L.words, L, i, (i * 5) % (L.maxnhar + 1), i % (L.maxnhar_e + 1), i % 5 != 4);
  f0_out.assign(F, -1.0f);
  for(int i = 0; i < F; i ++) { llsm_delete_container(ch -> frames[i]); ch -> frames[i] = nullptr; }
  llsm_frames_packed_finish(token, & L, ch, F, f0_out.data());
  *recs = rec;
  return ch;
}

void run(int MH, int ME, int NPSD, int NCH, int F) {
  const LlsmPackedLayout L = llsm_packed_layout(MH, ME, NPSD, NCH);
  CHECK(L.o_ampl % 4 == 0 && L.o_phse % 4 == 0 && L.o_psd % 4 == 0 && L.o_psdres % 4 == 0 && L.o_edc % 4 == 0 && L.o_eamp % 4 == 0 &&
        L.o_ephs % 4 == 0 && L.words % 4 == 0 && L.o_psdres == L.o_reshdr + 4);
  llsm_aoptions* ao = llsm_create_aoptions();
  ao -> npsd = NPSD; ao -> maxnhar = MH; ao -> maxnhar_e = ME;
  if(NCH != ao -> nchannel) {
    free(ao -> chanfreq); ao -> nchannel = NCH; ao -> chanfreq = (FP_TYPE*)calloc(NCH > 1 ? NCH - 1 : 1, sizeof(FP_TYPE));
    for(int c = 0; c + 1 < NCH; c ++) ao -> chanfreq[c] = 1000.0f * (c + 1);
  }
  llsm_container* conf = llsm_aoptions_toconf(ao, 22050.0f);
  llsm_container_attach(conf, LLSM_CONF_NFRM, llsm_create_int(F), (llsm_fdestructor)llsm_delete_int, (llsm_fcopy)llsm_copy_int);
  std::vector<float> f0;
  const void* recs = nullptr;
  llsm_chunk* ch = build(conf, L, F, f0, & recs);

  // 1. the reference's accessors read the record's values
  for(int i = 0; i < F; i ++) {
    llsm_container* fr = ch -> frames[i];
    const float want_f0 = i % 7 == 3 ? 0.0f : 100.0f + i;
    const bool voiced = want_f0 != 0;
    CHECK(f0[i] == want_f0 && *(FP_TYPE*)llsm_container_get(fr, LLSM_FRAME_F0) == want_f0);
    llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(fr, LLSM_FRAME_HM);
    const int nhar = voiced ? (i * 5) % (MH + 1) : 0;
    CHECK(hm && hm -> nhar == nhar);
    for(int k = 0; k < nhar; k ++) CHECK(hm -> ampl[k] == val(i, 1, k) && hm -> phse[k] == val(i, 2, k));
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(fr, LLSM_FRAME_NM);
    CHECK(nm && nm -> npsd == NPSD && nm -> nchannel == NCH);
    for(int k = 0; k < NPSD; k ++) CHECK(nm -> psd[k] == val(i, 3, k));
    for(int c = 0; c < NCH; c ++) {
      CHECK(nm -> edc[c] == val(i, 5, c) && nm -> eenv[c] -> nhar == (voiced ? i % (ME + 1) : 0));
      for(int k = 0; k < nm -> eenv[c] -> nhar; k ++)
        CHECK(nm -> eenv[c] -> ampl[k] == val(i, 6, c * 10 + k) && nm -> eenv[c] -> phse[k] == val(i, 7, c * 10 + k));
    }
    FP_TYPE* res = (FP_TYPE*)llsm_container_get(fr, LLSM_FRAME_PSDRES);
    CHECK((res != nullptr) == (i % 5 != 4));
    if(res) { CHECK(llsm_fparray_length(res) == NPSD); for(int k = 0; k < NPSD; k ++) CHECK(res[k] == val(i, 4, k)); }
  }

  // 2. the view: granted while untouched, values edited in place are what it shows, counts come from the structs
  LlsmPackedLayout V; const void* vrec = nullptr;
  CHECK(llsm_chunk_packed_view(ch, F, & V, & vrec) == 1 && vrec == recs && V.words == L.words);
  CHECK(llsm_chunk_packed_view(ch, F - 1, & V, & vrec) == 0);                       // another frame count: not this chunk's records
  llsm_hmframe* hm5 = (llsm_hmframe*)llsm_container_get(ch -> frames[4], LLSM_FRAME_HM);
  hm5 -> ampl[0] = 7.5f; const int nh5 = hm5 -> nhar; hm5 -> nhar = nh5 > 1 ? nh5 - 1 : nh5;   // a host trims a harmonic in place
  CHECK(llsm_chunk_packed_view(ch, F, & V, & vrec) == 1);
  CHECK(((const float*)vrec)[4 * (size_t)L.words + L.o_ampl] == 7.5f && ((const int*)vrec)[4 * (size_t)L.words + 1] == hm5 -> nhar);
  llsm_container_remove(ch -> frames[1], LLSM_FRAME_PSDRES);                         // a member removed: the record's flag follows
  CHECK(llsm_chunk_packed_view(ch, F, & V, & vrec) == 1 && ((const int*)vrec)[1 * (size_t)L.words + 3] == 0);

  // 3. copies are heap objects
  llsm_container* copy2 = llsm_copy_container(ch -> frames[2]);
  llsm_chunk* whole = llsm_copy_chunk(ch);
  CHECK(llsm_chunk_packed_view(whole, F, & V, & vrec) == 0);                         // a copy is ordinary frames

  // 4. changes of structure withdraw the view
  if(NCH > 1 && ME > 0) {
    llsm_nmframe* nm6 = (llsm_nmframe*)llsm_container_get(ch -> frames[6], LLSM_FRAME_NM);
    const int keep = nm6 -> eenv[1] -> nhar;
    nm6 -> eenv[1] -> nhar = keep ? keep - 1 : 1;                                     // unequal envelope counts: one count per record
    CHECK(llsm_chunk_packed_view(ch, F, & V, & vrec) == 0);
    nm6 -> eenv[1] -> nhar = keep;
    CHECK(llsm_chunk_packed_view(ch, F, & V, & vrec) == 1);
  }
  llsm_hmframe* big = llsm_create_hmframe(MH + 40);
  for(int k = 0; k < MH + 40; k ++) { big -> ampl[k] = 0.001f * k; big -> phse[k] = -0.5f; }
  llsm_copy_hmframe_inplace((llsm_hmframe*)llsm_container_get(ch -> frames[8], LLSM_FRAME_HM), big);   // regrown beyond the record
  llsm_delete_hmframe(big);
  CHECK(llsm_chunk_packed_view(ch, F, & V, & vrec) == 0);
  CHECK(((llsm_hmframe*)llsm_container_get(ch -> frames[8], LLSM_FRAME_HM)) -> nhar == MH + 40);
  llsm_container_attach(ch -> frames[0], LLSM_FRAME_PBPSYN, llsm_create_int(1), (llsm_fdestructor)llsm_delete_int, (llsm_fcopy)llsm_copy_int);
  llsm_container_attach(ch -> frames[9], LLSM_FRAME_HM, llsm_create_hmframe(3), (llsm_fdestructor)llsm_delete_hmframe, (llsm_fcopy)llsm_copy_hmframe);
  llsm_delete_container(ch -> frames[F - 1]); ch -> frames[F - 1] = llsm_create_frame(2, NCH, 1, NPSD);
  CHECK(llsm_chunk_packed_view(ch, F, & V, & vrec) == 0);

  // 5. deletion: the edited chunk frame by frame through llsm_delete_chunk, the copies, then untouched chunks through the
  // pristine shortcut (llsm_delete_chunk and llsm_delete_chunks)
  llsm_delete_chunk(ch);
  llsm_hmframe* h2 = (llsm_hmframe*)llsm_container_get(copy2, LLSM_FRAME_HM);
  CHECK(h2 -> nhar == (2 * 5) % (MH + 1));
  for(int k = 0; k < h2 -> nhar; k ++) CHECK(h2 -> ampl[k] == val(2, 1, k));
  llsm_delete_container(copy2);
  llsm_delete_chunk(whole);
  llsm_chunk* many[3];
  for(int j = 0; j < 3; j ++) many[j] = build(conf, L, F, f0, & recs);
  llsm_delete_chunk(many[0]);
  llsm_chunk* rest[2] = {many[1], many[2]};
  llsm_delete_chunks(rest, 2);
  // an abandoned begin (the device call failed)
  void* token = nullptr;
  CHECK(llsm_frames_packed_begin(F, & L, & token, 0) != nullptr);
  llsm_frames_packed_abort(token);
  CHECK(llsm_frames_packed_begin(0, & L, & token, 0) == nullptr && token == nullptr);
  // page-locked records without hooks: refused, the caller takes the staged path
  CHECK(llsm_frames_packed_begin(F, & L, & token, 1) == nullptr && token == nullptr);

  llsm_delete_container(conf); llsm_delete_aoptions(ao);
}
}

int main() {
  run(24, 4, 33, 4, 40);
  run(7, 0, 16, 1, 12);          // no envelope harmonics (row width 1), one channel
  run(1, 1, 5, 2, 11);
  long long live = -1; llsm_slab_stats(& live, nullptr, nullptr);
  llsm_slab_trim();
  long long pooled = -1; llsm_slab_stats(nullptr, nullptr, & pooled);
  if(live != 0 || pooled != 0 || g_bad) { std::printf("FAILED: %lld slabs left, %lld bytes pooled after the trim, %d checks\n", live, pooled, g_bad); return 1; }
  std::printf("packed_host ok\n");
  return 0;
}
