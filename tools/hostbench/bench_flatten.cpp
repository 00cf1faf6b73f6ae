// Host-only timing of llsm_chunk_to_flat (the walk llsm_synthesize_batch does over the container trees) on chunks built by
// llsm_frames_from_flat: config-2 shaped utterances (200 frames, 100 harmonics, 256 PSD points, 4 channels x 4 envelope
// harmonics), T threads each flattening its own block of utterances.  No device needed.
//   g++ -O2 -std=c++17 -I include -I libllsm2_amd/csrc tools/hostbench/bench_flatten.cpp -L libllsm2_amd -l:libllsm2_amd.so -Wl,-rpath,$PWD/libllsm2_amd -pthread -o /tmp/bench_flatten
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "llsm.h"
#include "llsm_gpu.h"
#include "model_internal.h"

struct Rows {
  int F, mh, me, npsd, nch;
  std::vector<float> f0, ampl, phse, psd, psdres, edc, ea, ep; std::vector<int> nhar, nhe, has;
  Rows(int F_, int mh_, int me_, int npsd_, int nch_) : F(F_), mh(mh_), me(me_), npsd(npsd_), nch(nch_),
    f0(F_), ampl((size_t)F_ * mh_), phse((size_t)F_ * mh_), psd((size_t)F_ * npsd_), psdres((size_t)F_ * npsd_),
    edc((size_t)F_ * nch_), ea((size_t)F_ * nch_ * me_), ep((size_t)F_ * nch_ * me_), nhar(F_), nhe(F_), has(F_) {}
  llsm_flat_params view() {
    llsm_flat_params v; v.maxnhar = mh; v.maxnhar_e = me; v.npsd = npsd; v.nchannel = nch;
    v.f0 = f0.data(); v.nhar = nhar.data(); v.ampl = ampl.data(); v.phse = phse.data(); v.psd = psd.data();
    v.psdres = psdres.data(); v.has_psdres = has.data(); v.edc = edc.data(); v.nhar_e = nhe.data();
    v.eenv_ampl = ea.data(); v.eenv_phse = ep.data(); return v;
  }
};

int main(int argc, char** argv) {
  const int n_utt = argc > 1 ? std::atoi(argv[1]) : 256, nf = 200, reps = argc > 2 ? std::atoi(argv[2]) : 5;
  const int slabs = argc > 3 ? std::atoi(argv[3]) : 1;
  Rows src(nf, 100, 4, 256, 4);
  for(int i = 0; i < nf; i ++) { src.f0[i] = 120.0f; src.nhar[i] = 100; src.nhe[i] = 4; src.has[i] = 1; }
  for(auto* v : {& src.ampl, & src.phse, & src.psd, & src.psdres, & src.edc, & src.ea, & src.ep})
    for(auto& x : *v) x = (float)std::rand() / RAND_MAX;
  llsm_aoptions* ao = llsm_create_aoptions();
  std::vector<llsm_chunk*> chunks(n_utt);
  llsm_flat_params sv = src.view();
  for(int u = 0; u < n_utt; u ++) {
    llsm_container* conf = llsm_aoptions_toconf(ao, 22050.0f);
    *(int*)llsm_container_get(conf, LLSM_CONF_NFRM) = nf;
    chunks[u] = llsm_create_chunk(conf, 0); llsm_delete_container(conf);
    llsm_frames_from_flat_ex(& sv, 0, chunks[u], nf, slabs);
  }
  for(int T : {1, 2, 4, 8, 16}) {
    if(T > n_utt) break;
    std::vector<Rows*> dst(T);
    for(int t = 0; t < T; t ++) dst[t] = new Rows(nf * ((n_utt + T - 1) / T), 100, 4, 256, 4);
    double best = 1e30;
    for(int r = 0; r < reps; r ++) {
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for(int t = 0; t < T; t ++) th.emplace_back([&, t] {
        llsm_flat_params v = dst[t] -> view();
        const int per = (n_utt + T - 1) / T, u0 = t * per, u1 = std::min(n_utt, u0 + per);
        for(int u = u0; u < u1; u ++) llsm_chunk_to_flat(chunks[u], & v, (u - u0) * nf);
      });
      for(auto& x : th) x.join();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if(ms < best) best = ms;
    }
    std::printf("threads %2d: %8.2f ms for %d frames = %.2f M frames/s (%.1f GB/s of rows)\n", T, best, n_utt * nf,
      n_utt * nf / best / 1e3, n_utt * nf * 3009.0 / best / 1e6);
    for(auto* d : dst) delete d;
  }
  return 0;
}
