// rt.cpp -- llsmrt real-time synthesis buffer on the GPU (harmonic-model path).
// Replaces llsmrt.c:32-602: same producer/consumer contract, same hop
// bookkeeping (float32 cycle / curr_nhop / next_nhop, llsmrt.c:110-129), same
// ring semantics (buffer.h).  The per-hop DSP of one feed() -- harmonic frame,
// noise-envelope frames, excitation mix, FFT noise filter, mix -- is six kernel
// launches on the context stream; the two output rings live in host memory so
// fetch() never touches the device.
//
// Internally a buffer is a "group" of S lock-stepped streams (S = 1 through the
// reference API); every kernel is written for S streams per launch.
#include <hip/hip_runtime.h>

#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "engine.h"
#include "kernels.h"
#include "llsmrt.h"
#include "llsm_gpu.h"
#include "plan.h"

extern const float2* llsm_engine_twiddles(llsm_gpu_context* c, int* nmax);
namespace lp = llsm_plan;

namespace {

struct HostRing {                       // buffer.h:32-138, host side (output rings only)
  std::vector<float> data; int cap = 0, curr = 0;
  void init(int c) { cap = c; curr = 0; data.assign(c, 0.0f); }
  float read(int idx) const { return data[(curr + idx + cap) % cap]; }
  void appendchunk(int n, const float* src) {
    curr = (curr + n) % cap;
    int base = curr + cap;
    for(int i = 0; i < n; i ++) data[(base - n + i) % cap] = src[i];
  }
};

template <class T> struct Dev {
  T* p = nullptr;
  bool alloc(size_t n) { return hipMalloc(& p, (n ? n : 1) * sizeof(T)) == hipSuccess; }
  ~Dev() { if(p) (void)hipFree(p); }
};

struct WinEntry { Dev<float> w; float inv_wsqr; };

template <class T> struct Ptr { T* p = nullptr; };

struct RtBuffer {
  llsm_gpu_context* ctx = nullptr;
  int S = 1;
  // configuration (llsmrt.c:157-206)
  int nchannel = 0, nch_active = 0, ntemplate = 0, ninternal = 0, npsd = 0, maxnhar = 0, me = 0, capacity = 0;
  llsm_soptions opt; llsm_container* conf = nullptr;
  std::vector<float> chanfreq;
  float fs = 0, thop = 0, fnyq = 0;
  // state (llsmrt.c:40-54)
  float cycle = 0;
  int curr_nhop = 0, next_nhop = 0, exc_cycle = 0, sin_pos = 0, nfft = 0;
  std::vector<int> nout;                 // per stream
  int mod_curr = 0, sin_curr = 0, noise_curr = 0, exc_curr = 0;
  std::vector<int> has_prev; std::vector<float> prev_psd;   // per stream [S], [S][npsd]
  unsigned long long seed = 0;
  // host output rings + synchronisation (llsmrt.c:56-57, 74-77)
  std::vector<HostRing> out_p, out_ap;   // per stream
  std::mutex mtx; std::condition_variable cv;
  // device state
  Dev<float> tpl, mod, excr, noiser, sinr, exc_frame, envf, frames_sin, nframes, out;
  Dev<int> live;
  // per-feed parameter rows: ONE pinned host block mirrored by ONE device block, so a hop costs a
  // single host-to-device copy (11 separate copies were half of the feed latency); h_* / d_* are
  // views into the two blocks
  unsigned char* h_params = nullptr; Dev<unsigned char> d_params; size_t params_bytes = 0;
  Ptr<float> d_f0, d_ampl, d_phse, d_edc, d_eamp, d_ephs, d_psd, d_cyc;
  Ptr<float> h_f0, h_ampl, h_phse, h_edc, h_eamp, h_ephs, h_psd, h_cyc;
  Ptr<int> d_nhar, d_nhar_e, d_has_nm, h_nhar, h_nhar_e, h_has_nm;
  Dev<float> d_psdres;
  Dev<int> d_zero, d_frm_utt, d_frm_off;
  float* h_out = nullptr;
  std::map<int, WinEntry*> wins;        // Hann(2 * nhop) by nhop
  int max_hop = 0;

  ~RtBuffer() {
    for(auto& kv : wins) delete kv.second;
    if(h_out) (void)hipHostFree(h_out);
    if(h_params) (void)hipHostFree(h_params);
  }
};

int ilog2(int n) { int l = 0; while((1 << l) < n) l ++; return l; }

WinEntry* get_window(RtBuffer* b, int nhop) {
  auto it = b -> wins.find(nhop);
  if(it != b -> wins.end()) return it -> second;
  const int n = 2 * nhop;
  std::vector<float> w(n);
  double s = 0;
  for(int i = 0; i < n; i ++) {          // hanning_2 -> the (symmetric) Hann of DESIGN.md
    w[i] = n == 1 ? 1.0f : (float)(0.5 - 0.5 * std::cos(2.0 * 3.14159265358979323846 * i / (n - 1)));
    s += (double)w[i] * w[i];
  }
  WinEntry* e = new WinEntry();
  e -> w.alloc(n);
  (void)hipMemcpy(e -> w.p, w.data(), n * sizeof(float), hipMemcpyHostToDevice);
  e -> inv_wsqr = (float)(1.0 / s);
  b -> wins[nhop] = e;
  return e;
}

// llsm_update_cycle, llsmrt.c:110-129 (float32 arithmetic, left to right)
void update_cycle(RtBuffer* b) {
  const int prev_nhop = b -> curr_nhop;
  const float c = lp::fadd(b -> cycle, b -> thop);
  b -> curr_nhop = (int)std::floor((double)lp::fmul(c, b -> fs));
  b -> cycle = lp::fadd(c, -lp::fdiv((float)prev_nhop, b -> fs));
  b -> next_nhop = (int)std::floor((double)lp::fmul(lp::fadd(b -> cycle, b -> thop), b -> fs));
  // appendblank of the modulation / sinusoid / noise rings: heads move, zeroing happens on the device
  b -> mod_curr = (b -> mod_curr + b -> curr_nhop) % b -> ninternal;
  b -> sin_curr = (b -> sin_curr + b -> curr_nhop) % b -> ninternal;
  b -> noise_curr = (b -> noise_curr + b -> curr_nhop) % b -> ninternal;
}

bool fail(const char* msg) { llsm_set_error(msg); return false; }

// llsm_create_rtsynth_buffer / llsm_rtsynth_buffer_clear device-side initial state
bool reset_state(RtBuffer* b) {
  LaunchCtx* P = llsm_engine_launch_ctx(b -> ctx);
  const int S = b -> S, cap = b -> ninternal, nch = b -> nchannel;
  b -> cycle = 0; b -> exc_cycle = 0;
  b -> nout.assign(S, 0); b -> has_prev.assign(S, 0);
  b -> out_p.resize(S); b -> out_ap.resize(S);
  for(int s2 = 0; s2 < S; s2 ++) { b -> out_p[s2].init(b -> capacity); b -> out_ap[s2].init(b -> capacity); }
  b -> mod_curr = b -> sin_curr = b -> noise_curr = b -> exc_curr = 0;
  (void)hipMemsetAsync(b -> mod.p, 0, sizeof(float) * S * nch * cap, P -> stream);
  (void)hipMemsetAsync(b -> sinr.p, 0, sizeof(float) * S * cap, P -> stream);
  (void)hipMemsetAsync(b -> noiser.p, 0, sizeof(float) * S * cap, P -> stream);
  (void)hipMemsetAsync(b -> excr.p, 0, sizeof(float) * S * cap, P -> stream);
  // llsmrt.c:213-217: curr_nhop = 1; update_cycle; cycle = 0; sin_pos
  b -> curr_nhop = 1;
  update_cycle(b);
  b -> cycle = 0;
  b -> sin_pos = -b -> curr_nhop * 2 - b -> nfft / 2;
  // llsm_fill_excitation_buffers, llsmrt.c:149-155: ninternal-1 appends of 1e-5 ...
  std::vector<float> fill((size_t)S * nch * cap, 1e-5f);
  const int hole = (b -> mod_curr + cap - 1) % cap;       // the one slot the appends do not reach
  for(int r = 0; r < S * nch; r ++) fill[(size_t)r * cap + hole] = 0.0f;
  if(hipMemcpyAsync(b -> mod.p, fill.data(), fill.size() * sizeof(float), hipMemcpyHostToDevice, P -> stream) != hipSuccess)
    return fail("llsmrt: modulation ring upload failed");
  (void)hipStreamSynchronize(P -> stream);
  b -> mod_curr = hole;
  // ... then five runs of the excitation mixer over ninternal/5 samples
  for(int i = 0; i < 5; i ++) {
    const int nx = b -> ninternal / 5;
    b -> exc_curr = (b -> exc_curr + nx) % cap;
    if(launch_rt_excite(P, S, b -> mod.p, b -> tpl.p, b -> excr.p, cap, nch, b -> ntemplate, b -> mod_curr,
         b -> exc_curr, b -> exc_cycle, b -> curr_nhop, nx, 0, nullptr)) return fail("llsmrt: k_rt_excite launch failed");
    b -> exc_cycle = (b -> exc_cycle + nx) % b -> ntemplate;
  }
  return hipStreamSynchronize(P -> stream) == hipSuccess;
}

}  // namespace

extern "C" {

static RtBuffer* create_group(llsm_soptions* options, llsm_container* conf, int capacity_samples,
  int n_streams) {
  int* nchannel = (int*)llsm_container_get(conf, LLSM_CONF_NCHANNEL);
  FP_TYPE* thop = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_THOP);
  FP_TYPE* chanfreq = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_CHANFREQ);
  if(nchannel == NULL || thop == NULL || chanfreq == NULL) return NULL;          // llsmrt.c:163
  int* npsd = (int*)llsm_container_get(conf, LLSM_CONF_NPSD);
  FP_TYPE* fnyq = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_FNYQ);
  int* maxnhar = (int*)llsm_container_get(conf, LLSM_CONF_MAXNHAR);
  int* maxnhar_e = (int*)llsm_container_get(conf, LLSM_CONF_MAXNHAR_E);
  if(npsd == NULL || fnyq == NULL) return NULL;
  if(options -> use_l1) {
    llsm_set_error("llsmrt: use_l1 (pulse-by-pulse synthesis) is outside this library's path");
    return NULL;
  }
  llsm_gpu_context* ctx = llsm_default_context();
  if(! ctx) return NULL;
  (void)hipSetDevice(llsm_engine_device(ctx));
  RtBuffer* b = new RtBuffer();
  b -> ctx = ctx; b -> S = n_streams;
  b -> nchannel = *nchannel; b -> npsd = *npsd; b -> fnyq = *fnyq;
  b -> maxnhar = maxnhar ? (*maxnhar > 2048 ? 2048 : *maxnhar) : 2048;
  b -> maxnhar = b -> maxnhar < 1 ? 1 : b -> maxnhar;
  b -> me = maxnhar_e ? *maxnhar_e : 8;
  if(b -> me > 8) b -> me = 8;
  b -> opt = *options; b -> conf = llsm_copy_container(conf);
  b -> chanfreq.assign(chanfreq, chanfreq + (*nchannel - 1));
  b -> fs = options -> fs; b -> thop = *thop;
  b -> ntemplate = (int)options -> fs;
  b -> ninternal = (int)(options -> fs * 0.2);
  b -> capacity = capacity_samples;
  b -> nfft = lp::nextpow2((double)lp::fmul(b -> thop, b -> fs) * 2.2 + 32);   // llsmrt.c:181
  b -> seed = llsm_next_seed();
  b -> prev_psd.assign((size_t)n_streams * b -> npsd, -200.0f);
  b -> max_hop = (int)(b -> thop * b -> fs) + 2;
  int tw_nmax = 0; llsm_engine_twiddles(ctx, & tw_nmax);
  const int S = b -> S, nch = b -> nchannel, cap = b -> ninternal, me = b -> me > 0 ? b -> me : 1;
  const int maxwin = 2 * b -> max_hop;
  bool ok = b -> nfft >= 64 && b -> nfft <= tw_nmax && nch >= 1 && nch <= 8 && capacity_samples > b -> max_hop &&
    n_streams >= 1 && n_streams <= 4096;
  if(! ok) { llsm_set_error("llsmrt: unsupported configuration (FFT size / channels / capacity / streams)"); llsm_delete_rtsynth_buffer(b); return NULL; }
  ok = b -> tpl.alloc((size_t)S * nch * b -> ntemplate) && b -> mod.alloc((size_t)S * nch * cap) &&
    b -> excr.alloc((size_t)S * cap) && b -> noiser.alloc((size_t)S * cap) && b -> sinr.alloc((size_t)S * cap) &&
    b -> exc_frame.alloc((size_t)S * maxwin) && b -> envf.alloc((size_t)S * nch * maxwin) &&
    b -> frames_sin.alloc((size_t)S * maxwin) && b -> nframes.alloc((size_t)S * b -> nfft) &&
    b -> out.alloc((size_t)S * 2 * b -> max_hop) && b -> live.alloc(S) &&
    b -> d_psdres.alloc((size_t)S * b -> npsd) && b -> d_zero.alloc(S) &&
    b -> d_frm_utt.alloc(S) && b -> d_frm_off.alloc(S) &&
    hipHostMalloc((void**)& b -> h_out, sizeof(float) * S * 2 * b -> max_hop) == hipSuccess;
  if(ok) {
    // layout of the per-hop parameter block (16-byte aligned sub-arrays)
    size_t at = 0;
    auto place = [&](size_t count) { size_t o = at; at += (count * 4 + 15) & ~(size_t)15; return o; };
    const size_t o_f0 = place(S), o_cyc = place(S), o_nhar = place(S), o_nhe = place(S), o_nm = place(S);
    const size_t o_ampl = place((size_t)S * b -> maxnhar), o_phse = place((size_t)S * b -> maxnhar);
    const size_t o_edc = place((size_t)S * nch), o_eamp = place((size_t)S * nch * me), o_ephs = place((size_t)S * nch * me);
    const size_t o_psd = place((size_t)S * b -> npsd);
    b -> params_bytes = at;
    ok = b -> d_params.alloc(at) && hipHostMalloc((void**)& b -> h_params, at) == hipSuccess;
    if(ok) {
      unsigned char *hb = b -> h_params, *db = b -> d_params.p;
#define VIEW(name, off, T) b -> h_##name.p = (T*)(hb + off); b -> d_##name.p = (T*)(db + off);
      VIEW(f0, o_f0, float) VIEW(cyc, o_cyc, float) VIEW(nhar, o_nhar, int) VIEW(nhar_e, o_nhe, int)
      VIEW(has_nm, o_nm, int) VIEW(ampl, o_ampl, float) VIEW(phse, o_phse, float) VIEW(edc, o_edc, float)
      VIEW(eamp, o_eamp, float) VIEW(ephs, o_ephs, float) VIEW(psd, o_psd, float)
#undef VIEW
    }
  }
  if(! ok) { llsm_set_error("llsmrt: device allocation failed"); llsm_delete_rtsynth_buffer(b); return NULL; }
  std::vector<int> ids(S);
  for(int s = 0; s < S; s ++) ids[s] = s;
  (void)hipMemcpy(b -> d_frm_utt.p, ids.data(), S * sizeof(int), hipMemcpyHostToDevice);
  (void)hipMemcpy(b -> d_frm_off.p, ids.data(), S * sizeof(int), hipMemcpyHostToDevice);
  (void)hipMemset(b -> d_zero.p, 0, S * sizeof(int));
  (void)hipMemset(b -> d_psdres.p, 0, sizeof(float) * S * b -> npsd);
  // llsm_make_exc_template (llsmrt.c:93-107) with the offline synthesis kernels
  llsm_aoptions ao; std::memset(& ao, 0, sizeof(ao));
  ao.thop = b -> thop; ao.maxnhar = 1; ao.maxnhar_e = 0; ao.npsd = b -> npsd; ao.nchannel = nch;
  ao.chanfreq = b -> chanfreq.data(); ao.rel_winsize = 4; ao.hm_method = LLSM_AOPTION_HMCZT;
  llsm_gpu_batch* tb = llsm_engine_template_batch(ctx, & ao, b -> fs, S, b -> ntemplate, b -> seed);
  if(! tb) { llsm_delete_rtsynth_buffer(b); return NULL; }
  b -> nch_active = llsm_engine_batch_nch_active(tb);
  llsm_gpu_layout L; llsm_gpu_batch_layout(tb, & L);
  int rc = launch_rt_template(llsm_engine_launch_ctx(ctx), llsm_engine_batch_colored(tb), L.ntemplate_ext, nch,
    b -> nch_active, b -> ntemplate, S, b -> tpl.p);
  llsm_gpu_synchronize(ctx);
  llsm_gpu_delete_batch(tb);
  if(rc || ! reset_state(b)) { llsm_delete_rtsynth_buffer(b); return NULL; }
  return b;
}

llsm_rtsynth_buffer* llsm_create_rtsynth_buffer(llsm_soptions* options, llsm_container* conf,
  int capacity_samples) {
  return (llsm_rtsynth_buffer*)create_group(options, conf, capacity_samples, 1);
}

void llsm_delete_rtsynth_buffer(llsm_rtsynth_buffer* dst) {
  if(dst == NULL) return;
  RtBuffer* b = (RtBuffer*)dst;
  (void)hipSetDevice(llsm_engine_device(b -> ctx));
  llsm_gpu_synchronize(b -> ctx);
  if(b -> conf) llsm_delete_container(b -> conf);
  delete b;
}

int llsm_rtsynth_buffer_getlatency(llsm_rtsynth_buffer* src) {          // llsmrt.c:568-571
  RtBuffer* b = (RtBuffer*)src;
  return -b -> sin_pos - b -> curr_nhop;
}

int llsm_rtsynth_buffer_numoutput(llsm_rtsynth_buffer* src) { return ((RtBuffer*)src) -> nout[0]; }

// One hop for every stream of the group: frames[s] is the frame of stream s (llsmrt.c:505-521).
static void feed_group(RtBuffer* b, llsm_container** frames) {
  (void)hipSetDevice(llsm_engine_device(b -> ctx));
  LaunchCtx* P = llsm_engine_launch_ctx(b -> ctx);
  update_cycle(b);
  const int S = b -> S, nch = b -> nchannel, cap = b -> ninternal, me = b -> me > 0 ? b -> me : 1;
  const int nhop = b -> curr_nhop, nwin = 2 * nhop, npsd = b -> npsd, mh = b -> maxnhar;
  if(nhop > b -> max_hop || b -> next_nhop > b -> max_hop) { llsm_set_error("llsmrt: hop exceeds buffer"); return; }
  WinEntry* we = get_window(b, nhop);
  // ---- frames -> parameter rows (llsmrt.c:255-291), written straight into the pinned block
  float *f0v = b -> h_f0.p, *cyc = b -> h_cyc.p, *ampl = b -> h_ampl.p, *phse = b -> h_phse.p;
  float *edc = b -> h_edc.p, *eamp = b -> h_eamp.p, *ephs = b -> h_ephs.p, *psd = b -> h_psd.p;
  int *nharv = b -> h_nhar.p, *nhev = b -> h_nhar_e.p, *hasnm = b -> h_has_nm.p;
  std::memset(b -> h_params, 0, b -> params_bytes);
  for(size_t k = 0; k < (size_t)S * nch; k ++) edc[k] = 1e-5f;
  for(size_t k = 0; k < (size_t)S * npsd; k ++) psd[k] = -200.0f;
  for(int s2 = 0; s2 < S; s2 ++) {
    llsm_container* frame = frames[s2];
    FP_TYPE* f0p = (FP_TYPE*)llsm_container_get(frame, LLSM_FRAME_F0);
    llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(frame, LLSM_FRAME_HM);
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(frame, LLSM_FRAME_NM);
    f0v[s2] = f0p ? *f0p : 0.0f;
    cyc[s2] = b -> cycle;
    int nhar = hm ? hm -> nhar : -1;
    if(nhar > mh) nhar = mh;
    if(nhar > b -> nfft) nhar = b -> nfft;                         // llsmrt.c:280
    for(int k = 0; k < nhar; k ++) { ampl[(size_t)s2 * mh + k] = hm -> ampl[k]; phse[(size_t)s2 * mh + k] = hm -> phse[k]; }
    nharv[s2] = nhar;
    int nhe = 0;
    hasnm[s2] = nm != NULL;
    if(nm)
      for(int c = 0; c < nch && c < nm -> nchannel; c ++) {
        edc[(size_t)s2 * nch + c] = nm -> edc[c];
        int n = nm -> eenv[c] ? nm -> eenv[c] -> nhar : 0;
        if(n > b -> me) n = b -> me;
        if(n > nhe) nhe = n;
        for(int k = 0; k < n; k ++) {
          eamp[((size_t)s2 * nch + c) * me + k] = nm -> eenv[c] -> ampl[k];
          ephs[((size_t)s2 * nch + c) * me + k] = nm -> eenv[c] -> phse[k];
        }
      }
    nhev[s2] = nhe;
    if(b -> has_prev[s2])
      std::memcpy(psd + (size_t)s2 * npsd, b -> prev_psd.data() + (size_t)s2 * npsd, sizeof(float) * npsd);
  }
  hipStream_t st = P -> stream;
  // one copy; the kernels below are ordered after it on the stream, and the pinned block is not
  // touched again before the synchronisation at the end of this call
  (void)hipMemcpyAsync(b -> d_params.p, b -> h_params, b -> params_bytes, hipMemcpyHostToDevice, st);
  BatchDev d; std::memset(& d, 0, sizeof(d));
  d.n_utt = S; d.nframes = S; d.maxnhar = mh; d.maxnhar_e = b -> me; d.npsd = npsd;
  d.nchannel = nch; d.thop = b -> thop; d.fs = b -> fs; d.rel_winsize = 4;
  d.frm_utt = b -> d_frm_utt.p; d.frm_off = b -> d_frm_off.p;
  d.f0 = b -> d_f0.p; d.nhar = b -> d_nhar.p; d.ampl = b -> d_ampl.p; d.phse = b -> d_phse.p;
  d.psd = b -> d_psd.p; d.psdres = b -> d_psdres.p; d.has_psdres = b -> d_zero.p;
  d.edc = b -> d_edc.p; d.nhar_e = b -> d_nhar_e.p; d.eenv_ampl = b -> d_eamp.p; d.eenv_phse = b -> d_ephs.p;
  int tw_nmax = 0; const float2* tw = llsm_engine_twiddles(b -> ctx, & tw_nmax);
  int rc = 0;
  // feed_deterministic: envelope frames + harmonic frame, then the ring adds
  rc |= launch_env_frames(P, d, b -> fs, nwin, we -> w.p, b -> envf.p);
  rc |= launch_synth_frames(P, d, nwin, we -> w.p, b -> d_cyc.p, b -> frames_sin.p, mh);
  rc |= launch_rt_rings(P, S, b -> mod.p, b -> sinr.p, b -> noiser.p, cap, nch, b -> mod_curr, b -> sin_curr,
    b -> noise_curr, nhop, nwin, b -> envf.p, b -> frames_sin.p, b -> d_f0.p, b -> d_has_nm.p, b -> d_nhar.p);
  // run_excitation_buffers(curr_nhop)
  b -> exc_curr = (b -> exc_curr + nhop) % cap;
  rc |= launch_rt_excite(P, S, b -> mod.p, b -> tpl.p, b -> excr.p, cap, nch, b -> ntemplate, b -> mod_curr,
    b -> exc_curr, b -> exc_cycle, nhop, nhop, nwin, b -> exc_frame.p);
  b -> exc_cycle = (b -> exc_cycle + nhop) % b -> ntemplate;
  // feed_filter on the previous frame's noise model (rows at -200 dB are skipped: no prev_nm yet)
  rc |= launch_noise_filter(P, d, b -> exc_frame.p, nullptr, nullptr, b -> fnyq, b -> fs, nwin, we -> w.p,
    we -> inv_wsqr, b -> nfft, ilog2(b -> nfft), tw, tw_nmax, b -> nframes.p, b -> live.p, 1);
  // feed_mix
  rc |= launch_rt_mix(P, S, b -> noiser.p, b -> sinr.p, cap, b -> noise_curr, b -> sin_curr, b -> sin_pos,
    b -> nfft, b -> nframes.p, b -> live.p, b -> next_nhop, b -> max_hop, b -> out.p);
  (void)hipMemcpyAsync(b -> h_out, b -> out.p, sizeof(float) * S * 2 * b -> max_hop, hipMemcpyDeviceToHost, st);
  if(rc || hipStreamSynchronize(st) != hipSuccess) { llsm_set_error("llsmrt: feed failed on the device"); return; }
  {
    std::unique_lock<std::mutex> lock(b -> mtx);
    b -> cv.wait(lock, [&] {                                       // llsmrt.c:489-493, for every stream
      for(int s2 = 0; s2 < S; s2 ++) if(b -> nout[s2] > b -> capacity - b -> next_nhop) return false;
      return true;
    });
    for(int s2 = 0; s2 < S; s2 ++) {
      b -> out_p[s2].appendchunk(b -> next_nhop, b -> h_out + ((size_t)s2 * 2 + 0) * b -> max_hop);
      b -> out_ap[s2].appendchunk(b -> next_nhop, b -> h_out + ((size_t)s2 * 2 + 1) * b -> max_hop);
      b -> nout[s2] += b -> next_nhop;
    }
  }
  b -> cv.notify_all();
  // prev_nm with PSDRES folded in (llsmrt.c:513-520)
  for(int s2 = 0; s2 < S; s2 ++) {
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(frames[s2], LLSM_FRAME_NM);
    FP_TYPE* resvec = (FP_TYPE*)llsm_container_get(frames[s2], LLSM_FRAME_PSDRES);
    b -> has_prev[s2] = nm != NULL;
    if(nm)
      for(int j = 0; j < npsd; j ++) {
        float v = j < nm -> npsd ? nm -> psd[j] : -120.0f;
        if(resvec && j < llsm_fparray_length(resvec)) v += resvec[j] - (float)(0.375 / 2.3025851 * 10.0);
        b -> prev_psd[(size_t)s2 * npsd + j] = v;
      }
  }
}

void llsm_rtsynth_buffer_feed(llsm_rtsynth_buffer* dst, llsm_container* frame) {
  feed_group((RtBuffer*)dst, & frame);
}

// bulk pull of up to `max_samples` samples of one stream (non-blocking)
static int fetch_bulk(RtBuffer* b, int stream, FP_TYPE* dst_p, FP_TYPE* dst_ap, int max_samples) {
  int got = 0;
  {
    std::lock_guard<std::mutex> lock(b -> mtx);
    while(got < max_samples && b -> nout[stream] > 0) {
      const float p = b -> out_p[stream].read(-b -> nout[stream]);
      const float ap = b -> out_ap[stream].read(-b -> nout[stream]);
      if(dst_p) dst_p[got] = p;
      if(dst_ap) dst_ap[got] = ap;
      b -> nout[stream] --;
      got ++;
    }
  }
  if(got) b -> cv.notify_all();
  return got;
}

int llsm_rtsynth_buffer_fetch_decomposed(llsm_rtsynth_buffer* src, FP_TYPE* dst_p, FP_TYPE* dst_ap) {
  return fetch_bulk((RtBuffer*)src, 0, dst_p, dst_ap, 1);          // llsmrt.c:545-566
}

int llsm_rtsynth_buffer_fetch(llsm_rtsynth_buffer* src, FP_TYPE* dst) {  // llsmrt.c:523-543
  FP_TYPE p = 0, ap = 0;
  if(! llsm_rtsynth_buffer_fetch_decomposed(src, & p, & ap)) return 0;
  *dst = p + ap;
  return 1;
}

void llsm_rtsynth_buffer_clear(llsm_rtsynth_buffer* dst) {               // llsmrt.c:578-602
  RtBuffer* b = (RtBuffer*)dst;
  (void)hipSetDevice(llsm_engine_device(b -> ctx));
  reset_state(b);
}

// ---- stream groups (llsm_gpu.h): S lock-stepped streams per launch sequence ----
llsm_rtsynth_group* llsm_create_rtsynth_group(llsm_soptions* options, llsm_container* conf,
  int capacity_samples, int n_streams) {
  int* nchannel = (int*)llsm_container_get(conf, LLSM_CONF_NCHANNEL);
  FP_TYPE* thop = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_THOP);
  FP_TYPE* chanfreq = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_CHANFREQ);
  if(nchannel == NULL || thop == NULL || chanfreq == NULL) return NULL;
  return (llsm_rtsynth_group*)create_group(options, conf, capacity_samples, n_streams);
}
void llsm_delete_rtsynth_group(llsm_rtsynth_group* g) { llsm_delete_rtsynth_buffer((llsm_rtsynth_buffer*)g); }
int llsm_rtsynth_group_getlatency(llsm_rtsynth_group* g) { return llsm_rtsynth_buffer_getlatency((llsm_rtsynth_buffer*)g); }
int llsm_rtsynth_group_numoutput(llsm_rtsynth_group* g, int stream) { return ((RtBuffer*)g) -> nout[stream]; }
void llsm_rtsynth_group_feed(llsm_rtsynth_group* g, llsm_container** frames) { feed_group((RtBuffer*)g, frames); }
int llsm_rtsynth_group_fetch(llsm_rtsynth_group* g, int stream, FP_TYPE* dst_p, FP_TYPE* dst_ap, int max_samples) {
  RtBuffer* b = (RtBuffer*)g;
  if(stream < 0 || stream >= b -> S) return 0;
  return fetch_bulk(b, stream, dst_p, dst_ap, max_samples);
}

}  // extern "C"
