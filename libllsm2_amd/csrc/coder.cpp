// coder.cpp -- the frame coder of the reference (llsm.h:346-362, coder.c:44-292) on the device: frames <->
// fixed-dimensional vectors [voicing, f0, Rd, order_spec spectrum points, order_bap band aperiodicities].
// llsm_coder_encode / llsm_coder_decode_layer{0,1} work on one frame (a batch of one); the additive
// llsm_coder_encode_chunk / llsm_coder_decode_chunk move a whole chunk through one launch each.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "scratch.h"

namespace {
struct Coder {
  int order_spec, order_bap, nspec, nchannel, nhar_e, npsd;
  float fnyq, liprad, mel_floor, mel_ceil;
  std::vector<float> melaxis; float* d_melaxis = nullptr;
  int dim() const { return order_spec + order_bap + 3; }
};
double freq2mel(double f) { return 1127.01048 * std::log(1.0 + f / 700.0); }
double mel2freq(double m) { return 700.0 * (std::exp(m / 1127.01048) - 1.0); }
const float* mel_table(Coder* c) {
  if(c -> d_melaxis) return c -> d_melaxis;
  if(hipMalloc((void**)& c -> d_melaxis, c -> melaxis.size() * 4) != hipSuccess ||
     hipMemcpy(c -> d_melaxis, c -> melaxis.data(), c -> melaxis.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    llsm_set_error("llsm_coder: mel axis upload failed"); return nullptr;
  }
  return c -> d_melaxis;
}
}  // namespace

extern "C" {

llsm_coder* llsm_create_coder(llsm_container* conf, int order_spec, int order_bap) {
  FP_TYPE* fnyq = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_FNYQ);
  int* nchannel = (int*)llsm_container_get(conf, LLSM_CONF_NCHANNEL);
  int* nhar_e = (int*)llsm_container_get(conf, LLSM_CONF_MAXNHAR_E);
  int* npsd = (int*)llsm_container_get(conf, LLSM_CONF_NPSD);
  int* nspec = (int*)llsm_container_get(conf, LLSM_CONF_NSPEC);
  FP_TYPE* liprad = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_LIPRADIUS);
  if(! fnyq || ! nchannel || ! nhar_e || ! npsd || ! nspec || ! liprad || order_spec < 1 || order_bap < 1 ||
     order_spec > *nspec - 1 || *nspec < 33 || ((*nspec - 1) & (*nspec - 2))) {
    llsm_set_error("llsm_create_coder: conf without layer-1 members, or orders out of range"); return NULL;
  }
  Coder* c = new Coder();
  c -> order_spec = order_spec; c -> order_bap = order_bap; c -> nspec = *nspec; c -> nchannel = *nchannel;
  c -> nhar_e = *nhar_e; c -> npsd = *npsd; c -> fnyq = *fnyq; c -> liprad = *liprad;
  const double ceil_ = freq2mel(*fnyq), floor_ = freq2mel(50);          // coder.c:67-72
  c -> mel_floor = (float)floor_; c -> mel_ceil = (float)ceil_;
  c -> melaxis.resize(*nspec);
  for(int i = 0; i < *nspec; i ++) c -> melaxis[i] = (float)mel2freq(floor_ + (ceil_ - floor_) * i / *nspec);
  return (llsm_coder*)c;
}

void llsm_delete_coder(llsm_coder* dst) {
  Coder* c = (Coder*)dst;
  if(! c) return;
  if(c -> d_melaxis) (void)hipFree(c -> d_melaxis);
  delete c;
}

int llsm_coder_dimension(llsm_coder* c) { return c ? ((Coder*)c) -> dim() : 0; }

// frames[0 .. n) -> dst[n][dim]
int llsm_coder_encode_frames(llsm_coder* c_, llsm_container** frames, int n, FP_TYPE* dst) {
  Coder* c = (Coder*)c_;
  if(! c || n < 0) { llsm_set_error("llsm_coder_encode: no coder"); return -1; }
  const int dim = c -> dim(), ns = c -> nspec, npsd = c -> npsd;
  std::memset(dst, 0, sizeof(FP_TYPE) * (size_t)n * dim);
  if(n == 0) return 0;
  std::vector<float> f0(n, 0), rd(n, 0), psd((size_t)n * npsd, -120.0f), vt((size_t)n * ns, 0); std::vector<int> has(n, 0);
  for(int i = 0; i < n; i ++) {
    FP_TYPE* f = (FP_TYPE*)llsm_container_get(frames[i], LLSM_FRAME_F0);
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(frames[i], LLSM_FRAME_NM);
    FP_TYPE* r = (FP_TYPE*)llsm_container_get(frames[i], LLSM_FRAME_RD);
    FP_TYPE* v = (FP_TYPE*)llsm_container_get(frames[i], LLSM_FRAME_VTMAGN);
    if(! f || ! nm) { llsm_set_error("llsm_coder_encode: frame without F0 / NM"); return -1; }
    f0[i] = *f;
    for(int j = 0; j < npsd && j < nm -> npsd; j ++) psd[(size_t)i * npsd + j] = nm -> psd[j];
    if(*f > 0) {
      if(! r || ! v || llsm_fparray_length(v) < ns) { llsm_set_error("llsm_coder_encode: voiced frame without RD / VTMAGN"); return -1; }
      rd[i] = *r; has[i] = 1;
      std::memcpy(vt.data() + (size_t)i * ns, v, sizeof(float) * (size_t)ns);
    }
  }
  Scratch s; if(! s.open()) return -1;
  const float* mel = mel_table(c);
  if(! mel) return -1;
  float* df0 = s.up(f0.data(), n); float* drd = s.up(rd.data(), n); float* dpsd = s.up(psd.data(), psd.size());
  float* dvt = s.up(vt.data(), vt.size()); int* dhas = s.up(has.data(), n); float* denc = s.alloc<float>((size_t)n * dim);
  if(s.bad) return -1;
  if(! s.run(launch_coder_encode(s.P, c -> order_spec, c -> order_bap, ns, npsd, c -> fnyq, c -> liprad, mel, n, df0, drd, dpsd,
       dvt, dhas, denc), "llsm_coder_encode")) return -1;
  s.down(dst, denc, (size_t)n * dim);
  return s.sync() ? 0 : -1;
}

// src[n][dim] -> n new frames (llsm_create_frame + RD, and HM or VTMAGN / VSPHSE), caller-owned
int llsm_coder_decode_frames(llsm_coder* c_, const FP_TYPE* src, int n, int use_layer1, llsm_container** out) {
  Coder* c = (Coder*)c_;
  for(int i = 0; i < n; i ++) out[i] = NULL;
  if(! c || n < 0) { llsm_set_error("llsm_coder_decode: no coder"); return -1; }
  if(n == 0) return 0;
  const int dim = c -> dim(), ns = c -> nspec, npsd = c -> npsd;
  int mh = 1;
  for(int i = 0; i < n; i ++) {                         // nhar = fnyq / max(20, f0) for voiced vectors (coder.c:176-178)
    const float f0 = std::max(20.0f, src[(size_t)i * dim + 1]);
    if(src[(size_t)i * dim] > 0.5f) mh = std::max(mh, (int)(c -> fnyq / f0));
  }
  Scratch s; if(! s.open()) return -1;
  const float* mel = mel_table(c);
  if(! mel) return -1;
  int tw_nmax = 0; const float2* tw = llsm_engine_twiddles(s.ctx, & tw_nmax);
  float* denc = s.up(src, (size_t)n * dim);
  float* df0 = s.alloc<float>(n); float* drd = s.alloc<float>(n); int* dnh = s.alloc<int>(n);
  float* da = s.alloc<float>((size_t)n * mh); float* dp = s.alloc<float>((size_t)n * mh); float* dpsd = s.alloc<float>((size_t)n * npsd);
  float* dvt = s.alloc<float>((size_t)n * ns); float* dvs = s.alloc<float>((size_t)n * mh); int* dnvs = s.alloc<int>(n); int* dhm = s.alloc<int>(n);
  if(s.bad) return -1;
  if(! s.run(launch_coder_decode(s.P, c -> order_spec, c -> order_bap, ns, npsd, mh, c -> fnyq, c -> liprad, mel, c -> mel_floor,
       c -> mel_ceil, n, denc, use_layer1, tw, tw_nmax, df0, drd, dnh, da, dp, dpsd, dvt, dvs, dnvs, dhm), "llsm_coder_decode")) return -1;
  std::vector<float> f0(n), rd(n), a((size_t)n * mh), p((size_t)n * mh), psd((size_t)n * npsd), vt((size_t)n * ns), vs((size_t)n * mh);
  std::vector<int> nh(n), nvs(n);
  s.down(f0.data(), df0, n); s.down(rd.data(), drd, n); s.down(nh.data(), dnh, n); s.down(psd.data(), dpsd, psd.size());
  s.down(nvs.data(), dnvs, n);
  if(use_layer1) { s.down(vt.data(), dvt, vt.size()); s.down(vs.data(), dvs, vs.size()); }
  else { s.down(a.data(), da, a.size()); s.down(p.data(), dp, p.size()); }
  if(! s.sync()) return -1;
  for(int i = 0; i < n; i ++) {
    const int nhar = use_layer1 ? nvs[i] : nh[i];
    llsm_container* fr = llsm_create_frame(nhar, c -> nchannel, c -> nhar_e, npsd);
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(fr, LLSM_FRAME_NM);
    llsm_container_attach_(fr, LLSM_FRAME_RD, llsm_create_fp(rd[i]), (llsm_fdestructor)llsm_delete_fp, (llsm_fcopy)llsm_copy_fp);
    *(FP_TYPE*)llsm_container_get(fr, LLSM_FRAME_F0) = f0[i];
    std::memcpy(nm -> psd, psd.data() + (size_t)i * npsd, sizeof(float) * (size_t)npsd);
    if(nhar > 0 && use_layer1) {
      llsm_container_remove(fr, LLSM_FRAME_HM);
      FP_TYPE* v = llsm_create_fparray(ns); FP_TYPE* ph = llsm_create_fparray(nhar);
      std::memcpy(v, vt.data() + (size_t)i * ns, sizeof(float) * (size_t)ns);
      std::memcpy(ph, vs.data() + (size_t)i * mh, sizeof(float) * (size_t)nhar);
      llsm_container_attach_(fr, LLSM_FRAME_VTMAGN, v, (llsm_fdestructor)llsm_delete_fparray, (llsm_fcopy)llsm_copy_fparray);
      llsm_container_attach_(fr, LLSM_FRAME_VSPHSE, ph, (llsm_fdestructor)llsm_delete_fparray, (llsm_fcopy)llsm_copy_fparray);
    } else if(nhar > 0) {
      llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(fr, LLSM_FRAME_HM);
      std::memcpy(hm -> ampl, a.data() + (size_t)i * mh, sizeof(float) * (size_t)nhar);
      std::memcpy(hm -> phse, p.data() + (size_t)i * mh, sizeof(float) * (size_t)nhar);
    }
    out[i] = fr;
  }
  return 0;
}

FP_TYPE* llsm_coder_encode(llsm_coder* c, llsm_container* src) {
  const int dim = llsm_coder_dimension(c);
  FP_TYPE* enc = (FP_TYPE*)std::calloc((size_t)std::max(dim, 1), sizeof(FP_TYPE));
  if(c) llsm_coder_encode_frames(c, & src, 1, enc);
  return enc;
}
llsm_container* llsm_coder_decode_layer1(llsm_coder* c, FP_TYPE* src) {
  llsm_container* out = NULL; llsm_coder_decode_frames(c, src, 1, 1, & out); return out;
}
llsm_container* llsm_coder_decode_layer0(llsm_coder* c, FP_TYPE* src) {
  llsm_container* out = NULL; llsm_coder_decode_frames(c, src, 1, 0, & out); return out;
}

}  // extern "C"
