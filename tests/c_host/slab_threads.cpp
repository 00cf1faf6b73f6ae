// Frame slabs under concurrency (libllsm2_amd/csrc/model.cpp): eight threads build chunks from flat rows, copy frames out
// of them, grow members in place and delete frames / chunks -- some of them chunks that ANOTHER thread built -- while the
// slab registry, the per-thread lookup cache and the pool are shared.  tests/test_c_host.py compiles this file with
// model.cpp under -fsanitize=thread (and once under -fsanitize=address).
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "llsm.h"
#include "llsm_gpu.h"

extern "C" void llsm_frames_from_flat(const llsm_flat_params* src, int frm_off, llsm_chunk* dst, int nfrm);

namespace {
enum { F = 40, MH = 24, ME = 4, NCH = 4, NPSD = 33 };
struct Rows {
  std::vector<float> f0, ampl, phse, psd, psdres, edc, ea, ep; std::vector<int> nhar, nhe, has;
  llsm_flat_params v;
  Rows() : f0(F), ampl(F * MH), phse(F * MH), psd(F * NPSD), psdres(F * NPSD), edc(F * NCH), ea(F * NCH * ME), ep(F * NCH * ME),
    nhar(F), nhe(F), has(F) {
    for(int i = 0; i < F; i ++) {
      f0[i] = i % 7 == 3 ? 0.0f : 100.0f + i; nhar[i] = (i * 5) % (MH + 1); nhe[i] = i % (ME + 1); has[i] = i % 5 != 4;
      for(int k = 0; k < MH; k ++) { ampl[i * MH + k] = i + 0.01f * k; phse[i * MH + k] = -i; }
      for(int k = 0; k < NPSD; k ++) { psd[i * NPSD + k] = -50.0f - k; psdres[i * NPSD + k] = 0.5f * k; }
      for(int c = 0; c < NCH; c ++) edc[i * NCH + c] = 1e-3f * (c + 1);
    }
    std::memset(& v, 0, sizeof(v));
    v.maxnhar = MH; v.maxnhar_e = ME; v.npsd = NPSD; v.nchannel = NCH;
    v.f0 = f0.data(); v.nhar = nhar.data(); v.ampl = ampl.data(); v.phse = phse.data(); v.psd = psd.data();
    v.psdres = psdres.data(); v.has_psdres = has.data(); v.edc = edc.data(); v.nhar_e = nhe.data();
    v.eenv_ampl = ea.data(); v.eenv_phse = ep.data();
  }
};
std::mutex g_mx; std::vector<llsm_chunk*> g_handoff;     // chunks built by one thread, deleted by another
std::atomic<int> g_bad{0};
}

int main() {
  Rows rows;
  llsm_aoptions* ao = llsm_create_aoptions();
  ao -> npsd = NPSD; ao -> nchannel = NCH; ao -> maxnhar = MH; ao -> maxnhar_e = ME;
  llsm_container* conf = llsm_aoptions_toconf(ao, 22050.0f);
  llsm_container_attach(conf, LLSM_CONF_NFRM, llsm_create_int(F), (llsm_fdestructor)llsm_delete_int, (llsm_fcopy)llsm_copy_int);
  auto work = [&](int id) {
    std::vector<llsm_container*> kept;
    for(int it = 0; it < 200; it ++) {
      llsm_chunk* ch = llsm_create_chunk(conf, 0);
      llsm_frames_from_flat(& rows.v, 0, ch, F);
      llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(ch -> frames[(it + id) % F], LLSM_FRAME_HM);
      if(hm -> nhar > 1 && hm -> ampl[1] != rows.ampl[((it + id) % F) * MH + 1]) g_bad ++;
      kept.push_back(llsm_copy_container(ch -> frames[it % F]));                 // heap copy outlives the chunk
      llsm_hmframe* big = llsm_create_hmframe(64);
      llsm_copy_hmframe_inplace(hm, big);                                          // slab arrays -> heap arrays
      llsm_delete_hmframe(big);
      llsm_container_attach(ch -> frames[0], LLSM_FRAME_PBPSYN, llsm_create_int(1), (llsm_fdestructor)llsm_delete_int, (llsm_fcopy)llsm_copy_int);
      llsm_delete_container(ch -> frames[F - 1]); ch -> frames[F - 1] = llsm_create_frame(2, NCH, 1, NPSD);
      llsm_chunk* other = nullptr;
      {
        std::lock_guard<std::mutex> lock(g_mx);
        if(it % 3 == 0) { g_handoff.push_back(ch); ch = nullptr; }
        if(! g_handoff.empty() && it % 3 == 1) { other = g_handoff.back(); g_handoff.pop_back(); }
      }
      if(other) llsm_delete_chunk(other);
      if(ch) llsm_delete_chunk(ch);
      if(kept.size() > 16) { llsm_delete_container(kept.front()); kept.erase(kept.begin()); }
    }
    for(llsm_container* c : kept) llsm_delete_container(c);
  };
  std::vector<std::thread> th;
  for(int i = 0; i < 8; i ++) th.emplace_back(work, i);
  for(auto& t : th) t.join();
  for(llsm_chunk* c : g_handoff) llsm_delete_chunk(c);
  long long live = -1, pooled = -1; llsm_slab_stats(& live, nullptr, & pooled);
  llsm_slab_trim();
  llsm_delete_container(conf); llsm_delete_aoptions(ao);
  if(live != 0 || g_bad.load() != 0) { std::printf("FAILED: %lld slabs left, %d wrong values\n", live, g_bad.load()); return 1; }
  std::printf("slab_threads ok (pooled %lld bytes before the trim)\n", pooled);
  return 0;
}
