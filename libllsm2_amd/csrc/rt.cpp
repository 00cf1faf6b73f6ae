// rt.cpp -- llsmrt real-time synthesis buffer on the GPU: the harmonic-model path and, with
// options.use_l1 = 1, the pulse-by-pulse path (llsmrt.c:295-420: host-ordered pulse tracker and
// llsm_fgfm callbacks, pulses / dual buffer / hand-over on the device).
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

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "engine.h"
#include "kernels.h"

#ifndef LLSM_RT_GRAPH_DEFAULT
#define LLSM_RT_GRAPH_DEFAULT 0
#endif
#include <chrono>
#include <cstdio>
#include "llsmrt.h"
#include "llsm_gpu.h"
#include "lfmodel.h"
#include "plan.h"

extern const float2* llsm_engine_twiddles(llsm_gpu_context* c, int* nmax);
extern "C" void llsm_rt2_timing_fetch(unsigned long long* out) __attribute__((weak));
namespace lp = llsm_plan;
namespace lf = llsm_lf;
double llsm_l1_pulse_projection(double rd, double f0, double vsphse0, double fs, double origin, lf::Model* model_out,
  double* source_p0_cached, bool cache_valid);

namespace {

struct HostRing {                       // buffer.h:32-138, host side (output rings only)
  std::vector<float> data; int cap = 0, curr = 0;
  void init(int c) { cap = c; curr = 0; data.assign(c, 0.0f); }
  float read(int idx) const { return data[(curr + idx + cap) % cap]; }
  void appendchunk(int n, const float* src) {        // (n <= cap) two straight copies, no index arithmetic per sample
    const int start = curr;                            // first written slot: (new curr - n) mod cap
    curr = (curr + n) % cap;
    const int first = std::min(n, cap - start);
    std::memcpy(data.data() + start, src, sizeof(float) * (size_t)first);
    if(n > first) std::memcpy(data.data(), src + first, sizeof(float) * (size_t)(n - first));
  }
  // dst[0 .. n) = read(idx), read(idx + 1), ...   (idx < 0: samples behind the cursor; n <= cap)
  void readchunk(int idx, int n, float* dst) const {
    const int start = ((curr + idx) % cap + cap) % cap;
    const int first = std::min(n, cap - start);
    std::memcpy(dst, data.data() + start, sizeof(float) * (size_t)first);
    if(n > first) std::memcpy(dst + first, data.data(), sizeof(float) * (size_t)(n - first));
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
  std::vector<int> has_prev;             // per stream
  // prev_nm's level rows [S][npsd] (-200 dB where a stream has none), pinned and double-buffered: a feed writes the
  // rows the NEXT hop filters with into the buffer the device is not reading, and the next hop's kernel takes them
  // from there (or the copy path moves them into the parameter block)
  float* h_psd2[2] = {nullptr, nullptr}; int psd_cur = 0; std::vector<char> psd_blank[2];
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
  // page-locked parameter block: TWO copies (pipelined feeds pack hop h + 1 while the device still reads hop h's);
  // h_params points at the copy of the current hop, the h_* row views are re-based with it at the start of a feed
  unsigned char* h_params0 = nullptr; unsigned char* h_params = nullptr; Dev<unsigned char> d_params; size_t params_bytes = 0;
  std::vector<std::pair<void**, size_t>> h_views;       // (address of a view's pointer, its offset in the block)
  int blk = 0;                                          // copy in use by the hop being fed
  hipEvent_t hop_done[2] = {nullptr, nullptr};          // recorded behind a hop's last device operation
  Ptr<float> d_f0, d_ampl, d_phse, d_edc, d_eamp, d_ephs, d_psd, d_cyc;
  Ptr<float> h_f0, h_ampl, h_phse, h_edc, h_eamp, h_ephs, h_psd, h_cyc;
  Ptr<int> d_nhar, d_nhar_e, d_has_nm, h_nhar, h_nhar_e, h_has_nm;
  Dev<float> d_psdres;
  Dev<int> d_zero, d_frm_utt, d_frm_off;
  float* h_out0 = nullptr; float* h_out = nullptr; size_t out_elems = 0;   // output block: two copies as well
  std::map<int, WinEntry*> wins;        // Hann(2 * nhop) by nhop
  int max_hop = 0;
  // ---- pulse-by-pulse path (options.use_l1; llsmrt.c:49, 58-59, 67)
  bool l1 = false; int nspec = 0, maxnhar_conf = -1, pulse_max = 0, dual_curr = 0; float lip_radius = 1.5f;
  std::vector<double> pulse; std::vector<int> pbp_offset, pbp_state;     // per stream
  std::vector<double> lf_p0; std::vector<float> lf_rd, lf_f0; std::vector<char> lf_valid;   // per stream: LF phase at F0 of the last (Rd, F0)
  // pipelined feeds (llsm_gpu_rt_pipeline): the hop whose device work is still in flight; its samples reach the rings when
  // the next feed starts, when a consumer finds the rings empty, or on clear / delete
  std::atomic<bool> pending{false}; int pending_ostride = 0, pending_blk = 0, pending_nhop = 0; std::mutex pend_mtx;
  Dev<float> dual_f, dual_b, pulse_out;
  Ptr<float> d_rd, d_vtmagn, d_vsphse, d_f0sin, h_rd, h_vtmagn, h_vsphse, h_f0sin;
  Ptr<int> d_nvs, d_sel, d_hashm, h_nvs, h_sel, h_hashm;
  Ptr<PbpJob> d_jobs, h_jobs; Ptr<PbpPulse> d_pulses, h_pulses; Ptr<RtPbpOp> d_ops, h_ops;
  int pulse_pool = 0, npulses_hop = 0, njobs_hop = 0;   // glottal pulses the hop's parameter block can carry / has placed
  size_t params_fixed = 0;                              // bytes of the block in front of the pulse pool
  size_t zero_rng[3][2] = {{0, 0}, {0, 0}, {0, 0}};     // what a feed clears of it: all but the harmonic rows (read up to a
                                                        // frame's own count only) and the level rows (written whole)
  std::vector<float> hm_back;           // rebuilt HM rows coming back for the callers' frames
  // one hop as a replayed graph: the enqueue sequence is stream-captured every hop (no device work), the
  // executable graph is updated in place from it (kernel arguments, grid sizes) and launched once
  hipGraphExec_t gexec = nullptr;
  int graph_hops = 0, graph_rebuilds = 0;

  ~RtBuffer() {
    if(gexec) (void)hipGraphExecDestroy(gexec);
    for(auto& kv : wins) delete kv.second;
    if(h_out0) (void)hipHostFree(h_out0);
    if(h_params0) (void)hipHostFree(h_params0);
    for(int k = 0; k < 2; k ++) if(hop_done[k]) (void)hipEventDestroy(hop_done[k]);
    for(int k = 0; k < 2; k ++) if(h_psd2[k]) (void)hipHostFree(h_psd2[k]);
  }
};

int ilog2(int n) { int l = 0; while((1 << l) < n) l ++; return l; }

// hop-as-a-graph switch (llsm_gpu.h llsm_gpu_rt_graph): default from $LLSM_RT_GRAPH
std::atomic<int> g_rt_graph([] { const char* e = std::getenv("LLSM_RT_GRAPH"); return e ? std::atoi(e) : LLSM_RT_GRAPH_DEFAULT; }());
std::atomic<long long> g_rt_graph_hops(0);
// launches per hop (llsm_gpu.h llsm_gpu_rt_fused): 0 five, 1 two, 2 one (k_rt_hop), 3 one with the hop's temporaries on chip
// (k_rt_hop2 where it fits, else as 2).  Default from $LLSM_RT_FUSED, else 3
int rt_fused_mode(int v) { return v <= 0 ? 0 : (v >= 3 ? 3 : v); }
std::atomic<int> g_rt_fused([] { const char* e = std::getenv("LLSM_RT_FUSED"); return e ? rt_fused_mode(std::atoi(e)) : 3; }());
// the hop's kernels read the pinned parameter block and write the pinned sample block themselves (llsm_gpu.h
// llsm_gpu_rt_direct): default from $LLSM_RT_DIRECT, else on
// feeds return before the device has finished the hop (llsm_gpu.h llsm_gpu_rt_pipeline): default from $LLSM_RT_PIPELINE, else off
std::atomic<int> g_rt_pipeline([] { const char* e = std::getenv("LLSM_RT_PIPELINE"); return e ? (std::atoi(e) > 0 ? 1 : 0) : 0; }());
std::atomic<int> g_rt_direct([] { const char* e = std::getenv("LLSM_RT_DIRECT"); return e ? (std::atoi(e) > 0 ? 1 : 0) : 1; }());

bool fail(const char* msg) { llsm_set_error(msg); return false; }

WinEntry* get_window(RtBuffer* b, int nhop) {
  auto it = b -> wins.find(nhop);
  if(it != b -> wins.end()) return it -> second;
  const int n = 2 * nhop;
  std::vector<float> w(n);
  double s = 0;
  for(int i = 0; i < n; i ++) {          // hanning_2 -> the (symmetric) Hann of DESIGN.md
    w[i] = n == 1 ? 1.0f : (float)(0.5 - 0.5 * std::cos(2.0 * 3.14159265358979323846 * i / (llsm_conv_hann_periodic() ? n : n - 1)));
    s += (double)w[i] * w[i];
  }
  WinEntry* e = new WinEntry();
  if(! e -> w.alloc(n) || hipMemcpy(e -> w.p, w.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    delete e; llsm_set_error("llsmrt: window upload failed"); return nullptr;
  }
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
  for(int s = 0; s < b -> S && b -> l1; s ++) {                    // llsmrt.c:116-118
    b -> pulse[s] -= prev_nhop;
    if(b -> pbp_state[s] && b -> pbp_offset[s] > b -> sin_pos + b -> curr_nhop) b -> pbp_offset[s] -= prev_nhop;
  }
  // appendblank of the modulation / sinusoid / noise rings: heads move, zeroing happens on the device
  b -> mod_curr = (b -> mod_curr + b -> curr_nhop) % b -> ninternal;
  b -> sin_curr = (b -> sin_curr + b -> curr_nhop) % b -> ninternal;
  b -> noise_curr = (b -> noise_curr + b -> curr_nhop) % b -> ninternal;
}

// zero the curr_nhop samples before the ring heads (what llsm_ringbuffer_appendblank leaves behind when no
// feed follows): the ring kernel of a hop with no voiced / noise-model stream does exactly that
bool blank_heads(RtBuffer* b, LaunchCtx* P) {
  return launch_rt_rings(P, b -> S, b -> mod.p, b -> sinr.p, b -> noiser.p, b -> ninternal, b -> nchannel, b -> mod_curr,
    b -> sin_curr, b -> noise_curr, b -> curr_nhop, 2 * b -> curr_nhop, b -> envf.p, b -> frames_sin.p, b -> d_psdres.p,
    b -> d_zero.p, b -> d_zero.p) == 0;
}

// device-side state of llsm_create_rtsynth_buffer (llsmrt.c:186-221); `create` = 0: llsm_rtsynth_buffer_clear
// (llsmrt.c:578-602), which keeps the modulation rings, the previous noise frame and the cycle remainder
bool reset_state(RtBuffer* b, bool create) {
  LaunchCtx* P = llsm_engine_launch_ctx(b -> ctx);
  hipStream_t st = P -> stream;
  const int S = b -> S, cap = b -> ninternal, nch = b -> nchannel;
  b -> nout.assign(S, 0);
  b -> out_p.resize(S); b -> out_ap.resize(S);
  for(int s2 = 0; s2 < S; s2 ++) { b -> out_p[s2].init(b -> capacity); b -> out_ap[s2].init(b -> capacity); }
  bool ok = hipMemsetAsync(b -> sinr.p, 0, sizeof(float) * S * cap, st) == hipSuccess &&
    hipMemsetAsync(b -> noiser.p, 0, sizeof(float) * S * cap, st) == hipSuccess &&
    hipMemsetAsync(b -> excr.p, 0, sizeof(float) * S * cap, st) == hipSuccess;
  if(b -> l1) ok = ok && hipMemsetAsync(b -> dual_f.p, 0, sizeof(float) * S * cap, st) == hipSuccess &&
    hipMemsetAsync(b -> dual_b.p, 0, sizeof(float) * S * cap, st) == hipSuccess;
  if(! ok) return fail("llsmrt: ring reset failed");
  b -> sin_curr = b -> noise_curr = b -> exc_curr = 0; b -> dual_curr = 0;
  b -> pbp_offset.assign(S, 0); b -> pbp_state.assign(S, 0);
  b -> lf_p0.assign(S, 0.0); b -> lf_rd.assign(S, 0.0f); b -> lf_f0.assign(S, 0.0f); b -> lf_valid.assign(S, 0);
  if(create) {
    b -> cycle = 0; b -> exc_cycle = 0; b -> mod_curr = 0;
    b -> has_prev.assign(S, 0);
    for(int k = 0; k < 2; k ++) {
      std::fill(b -> h_psd2[k], b -> h_psd2[k] + (size_t)S * b -> npsd, -200.0f);
      b -> psd_blank[k].assign(S, 1);
    }
    b -> pulse.assign(S, 0.0);
    if(hipMemsetAsync(b -> mod.p, 0, sizeof(float) * S * nch * cap, st) != hipSuccess) return fail("llsmrt: ring reset failed");
  }
  // llsmrt.c:213-217 / 595-601: curr_nhop = 1; update_cycle; cycle = 0; pulse = 0; exc_cycle = 0; sin_pos
  b -> curr_nhop = 1;
  update_cycle(b);
  if(b -> l1) b -> dual_curr = (b -> dual_curr + b -> curr_nhop) % cap;       // llsm_dualbuffer_forward on an empty buffer
  b -> cycle = 0;
  b -> pulse.assign(S, 0.0);
  b -> exc_cycle = 0;
  b -> sin_pos = -b -> curr_nhop * 2 - b -> nfft / 2;
  if(! create) {
    if(! blank_heads(b, P)) return fail("llsmrt: ring reset failed");
    return hipStreamSynchronize(st) == hipSuccess;
  }
  // llsm_fill_excitation_buffers, llsmrt.c:149-155: ninternal-1 appends of 1e-5 ...
  std::vector<float> fill((size_t)S * nch * cap, 1e-5f);
  const int hole = (b -> mod_curr + cap - 1) % cap;       // the one slot the appends do not reach
  for(int r = 0; r < S * nch; r ++) fill[(size_t)r * cap + hole] = 0.0f;
  if(hipMemcpyAsync(b -> mod.p, fill.data(), fill.size() * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess ||
     hipStreamSynchronize(st) != hipSuccess) return fail("llsmrt: modulation ring upload failed");
  b -> mod_curr = hole;
  // ... then five runs of the excitation mixer over ninternal/5 samples
  for(int i = 0; i < 5; i ++) {
    const int nx = b -> ninternal / 5;
    b -> exc_curr = (b -> exc_curr + nx) % cap;
    if(launch_rt_excite(P, S, b -> mod.p, b -> tpl.p, b -> excr.p, cap, nch, b -> ntemplate, b -> mod_curr,
         b -> exc_curr, b -> exc_cycle, b -> curr_nhop, nx, 0, nullptr)) return fail("llsmrt: k_rt_excite launch failed");
    b -> exc_cycle = (b -> exc_cycle + nx) % b -> ntemplate;
  }
  return hipStreamSynchronize(st) == hipSuccess;
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
  int* maxnhar_e = (int*)llsm_container_get(conf, LLSM_CONF_MAXNHAR_E);
  if(npsd == NULL || fnyq == NULL) return NULL;
  int* nspec = (int*)llsm_container_get(conf, LLSM_CONF_NSPEC);
  FP_TYPE* liprad = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_LIPRADIUS);
  if(options -> use_l1 && (nspec == NULL || liprad == NULL || *nspec < 33 || ((*nspec - 1) & (*nspec - 2)))) {
    llsm_set_error("llsmrt: use_l1 needs LLSM_CONF_NSPEC (2^k + 1) and LLSM_CONF_LIPRADIUS (llsm_chunk_tolayer1)");
    return NULL;
  }
  llsm_gpu_context* ctx = llsm_default_context();
  if(! ctx) return NULL;
  if(hipSetDevice(llsm_engine_device(ctx)) != hipSuccess) { llsm_set_error("llsmrt: hipSetDevice failed"); return NULL; }
  RtBuffer* b = new RtBuffer();
  b -> ctx = ctx; b -> S = n_streams;
  b -> nchannel = *nchannel; b -> npsd = *npsd; b -> fnyq = *fnyq;
  b -> opt = *options; b -> conf = llsm_copy_container(conf);
  b -> chanfreq.assign(chanfreq, chanfreq + (*nchannel - 1));
  b -> fs = options -> fs; b -> thop = *thop;
  b -> ntemplate = (int)options -> fs;
  b -> ninternal = (int)(options -> fs * 0.2);
  b -> capacity = capacity_samples;
  b -> nfft = lp::nextpow2((double)lp::fmul(b -> thop, b -> fs) * 2.2 + 32);   // llsmrt.c:181
  // harmonic rows: the reference clamps a frame's harmonics to nfft only (llsmrt.c:280)
  b -> maxnhar = b -> nfft < 2048 ? b -> nfft : 2048;
  if(b -> maxnhar < 1) b -> maxnhar = 1;
  b -> me = maxnhar_e ? *maxnhar_e : 8;
  if(b -> me > 8) b -> me = 8;                          // kernel limit; larger envelope models are truncated (error text set)
  b -> seed = llsm_next_seed();
  // llsmrt.c:110-116 subtracts the PREVIOUS hop from the cycle counter: the hop length is a marginally stable two-step
  // recursion that float32 rounding keeps exciting, so for most fractional hops it swings around thop fs (350 .. 356
  // samples for 352.8, more for unlucky fractions).  Rows and windows are provisioned for twice the nominal hop; a hop
  // beyond that is reported and replaced by silence of the reference's length.
  b -> max_hop = 2 * (int)(b -> thop * b -> fs) + 16;
  b -> l1 = options -> use_l1 != 0;
  if(b -> l1) {
    b -> nspec = *nspec; b -> lip_radius = *liprad;
    int* mc = (int*)llsm_container_get(conf, LLSM_CONF_MAXNHAR);
    b -> maxnhar_conf = mc ? *mc : -1;
    // A hop places floor(hop / period) + 1 pulses per stream, the hop on which pulse-by-pulse synthesis ends
    // ceil(-pbp_offset / period) more (llsmrt.c:380-384; pbp_offset settles near sin_pos + hop): the pool holds that
    // worst case for every stream at F0 = 1 kHz, pulses are placed one after another (a stream takes what its
    // hop needs), and only the used part travels to the device.
    const int nfft_rt = lp::nextpow2((double)(b -> thop * b -> fs) * 2.2 + 32);
    const double reach = 2.0 * b -> max_hop + nfft_rt / 2 + 2.0 * b -> max_hop;
    b -> pulse_pool = n_streams * ((int)std::ceil(reach * 1000.0 / b -> fs) + 4);
  }
  int tw_nmax = 0; llsm_engine_twiddles(ctx, & tw_nmax);
  b -> pulse_max = tw_nmax;
  const int S = b -> S, nch = b -> nchannel, cap = b -> ninternal, me = b -> me > 0 ? b -> me : 1;
  const int maxwin = 2 * b -> max_hop, mh = b -> maxnhar;
  bool ok = b -> nfft >= 64 && b -> nfft <= tw_nmax && nch >= 1 && nch <= 8 && capacity_samples > b -> max_hop &&
    n_streams >= 1 && n_streams <= 4096 && (! b -> l1 || (b -> nspec - 1) * 2 <= tw_nmax);
  if(! ok) { llsm_set_error("llsmrt: unsupported configuration (FFT size / channels / capacity / streams)"); llsm_delete_rtsynth_buffer(b); return NULL; }
  ok = b -> tpl.alloc((size_t)S * nch * b -> ntemplate) && b -> mod.alloc((size_t)S * nch * cap) &&
    b -> excr.alloc((size_t)S * cap) && b -> noiser.alloc((size_t)S * cap) && b -> sinr.alloc((size_t)S * cap) &&
    b -> exc_frame.alloc((size_t)S * maxwin) && b -> envf.alloc((size_t)S * nch * maxwin) &&
    b -> frames_sin.alloc((size_t)S * maxwin) && b -> nframes.alloc((size_t)S * b -> nfft) &&
    b -> out.alloc((size_t)S * 2 * (b -> max_hop + 16)) && b -> live.alloc(S) &&
    b -> d_psdres.alloc((size_t)S * b -> npsd) && b -> d_zero.alloc(S) &&
    b -> d_frm_utt.alloc(S) && b -> d_frm_off.alloc(S) &&
    hipHostMalloc((void**)& b -> h_out0, 2 * sizeof(float) * S * 2 * (b -> max_hop + 16)) == hipSuccess &&
    hipEventCreateWithFlags(& b -> hop_done[0], hipEventDisableTiming) == hipSuccess &&
    hipEventCreateWithFlags(& b -> hop_done[1], hipEventDisableTiming) == hipSuccess &&
    hipHostMalloc((void**)& b -> h_psd2[0], sizeof(float) * (size_t)S * b -> npsd) == hipSuccess &&
    hipHostMalloc((void**)& b -> h_psd2[1], sizeof(float) * (size_t)S * b -> npsd) == hipSuccess;
  if(ok) { b -> out_elems = (size_t)S * 2 * (b -> max_hop + 16); b -> h_out = b -> h_out0; }
  if(ok && b -> l1)
    ok = b -> dual_f.alloc((size_t)S * cap) && b -> dual_b.alloc((size_t)S * cap) && b -> pulse_out.alloc((size_t)S * b -> pulse_max);
  if(ok) {
    // layout of the per-hop parameter block (16-byte aligned sub-arrays)
    size_t at = 0;
    auto place = [&](size_t bytes) { size_t o = at; at += (bytes + 15) & ~(size_t)15; return o; };
    // harmonic-model buffers: counts | harmonic rows | envelope rows | level rows.  Pulse-by-pulse buffers: counts |
    // envelope rows | level rows | layer-1 rows, jobs, ops | pulse pool | harmonic rows -- a hop on which no stream
    // asks for sinusoids (the steady state of pulse-by-pulse synthesis) copies up to the pulses it placed and leaves
    // the harmonic rows (nfft slots per stream) behind
    const size_t o_f0 = place(4 * S), o_cyc = place(4 * S), o_nhar = place(4 * S), o_nhe = place(4 * S), o_nm = place(4 * S);
    size_t o_ampl = 0, o_phse = 0;
    if(! b -> l1) { o_ampl = place(4 * (size_t)S * mh); o_phse = place(4 * (size_t)S * mh); }
    const size_t o_edc = place(4 * (size_t)S * nch), o_eamp = place(4 * (size_t)S * nch * me), o_ephs = place(4 * (size_t)S * nch * me);
    const size_t o_psd = place(4 * (size_t)S * b -> npsd);
    size_t o_rd = 0, o_vt = 0, o_vs = 0, o_f0sin = 0, o_nvs = 0, o_sel = 0, o_hashm = 0, o_jobs = 0, o_pulses = 0, o_ops = 0;
    if(b -> l1) {
      o_rd = place(4 * S); o_vt = place(4 * (size_t)S * b -> nspec); o_vs = place(4 * (size_t)S * mh);
      o_f0sin = place(4 * S); o_nvs = place(4 * S); o_sel = place(4 * S); o_hashm = place(4 * S);
      o_jobs = place(sizeof(PbpJob) * S); o_ops = place(sizeof(RtPbpOp) * S);
      o_pulses = place(sizeof(PbpPulse) * (size_t)b -> pulse_pool);   // a hop copies only the pulses it placed ...
      o_ampl = place(4 * (size_t)S * mh); o_phse = place(4 * (size_t)S * mh);   // ... unless it needs these rows as well
    }
    b -> params_bytes = at;
    b -> params_fixed = b -> l1 ? o_pulses : at;
    // what a feed clears: the counts and the small rows.  Harmonic, vocal-tract and source-phase rows are read up to a
    // frame's own count only, level rows are written whole.
    if(b -> l1) {
      b -> zero_rng[0][0] = 0; b -> zero_rng[0][1] = o_psd;
      b -> zero_rng[1][0] = o_rd; b -> zero_rng[1][1] = o_vt;
      b -> zero_rng[2][0] = o_f0sin; b -> zero_rng[2][1] = o_pulses;
    } else {
      b -> zero_rng[0][0] = 0; b -> zero_rng[0][1] = o_ampl;
      b -> zero_rng[1][0] = o_edc; b -> zero_rng[1][1] = o_psd;
      b -> zero_rng[2][0] = at; b -> zero_rng[2][1] = at;
    }
    ok = b -> d_params.alloc(at) && hipHostMalloc((void**)& b -> h_params0, 2 * at) == hipSuccess;
    if(ok) {
      std::memset(b -> h_params0, 0, 2 * at);
      b -> h_params = b -> h_params0;
      unsigned char *hb = b -> h_params, *db = b -> d_params.p;
#define VIEW(name, off, T) b -> h_##name.p = (T*)(hb + off); b -> d_##name.p = (T*)(db + off); \
      b -> h_views.push_back(std::make_pair((void**)& b -> h_##name.p, (size_t)(off)));
      VIEW(f0, o_f0, float) VIEW(cyc, o_cyc, float) VIEW(nhar, o_nhar, int) VIEW(nhar_e, o_nhe, int)
      VIEW(has_nm, o_nm, int) VIEW(ampl, o_ampl, float) VIEW(phse, o_phse, float) VIEW(edc, o_edc, float)
      VIEW(eamp, o_eamp, float) VIEW(ephs, o_ephs, float) VIEW(psd, o_psd, float)
      if(b -> l1) {
        VIEW(rd, o_rd, float) VIEW(vtmagn, o_vt, float) VIEW(vsphse, o_vs, float) VIEW(f0sin, o_f0sin, float)
        VIEW(nvs, o_nvs, int) VIEW(sel, o_sel, int) VIEW(hashm, o_hashm, int)
        VIEW(jobs, o_jobs, PbpJob) VIEW(pulses, o_pulses, PbpPulse) VIEW(ops, o_ops, RtPbpOp)
        b -> hm_back.resize((size_t)S * mh * 2 + S);
      }
#undef VIEW
    }
  }
  if(! ok) { llsm_set_error("llsmrt: device allocation failed"); llsm_delete_rtsynth_buffer(b); return NULL; }
  std::vector<int> ids(S);
  for(int s = 0; s < S; s ++) ids[s] = s;
  if(hipMemcpy(b -> d_frm_utt.p, ids.data(), S * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
     hipMemcpy(b -> d_frm_off.p, ids.data(), S * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
     hipMemset(b -> d_zero.p, 0, S * sizeof(int)) != hipSuccess ||
     hipMemset(b -> d_psdres.p, 0, sizeof(float) * S * b -> npsd) != hipSuccess) {
    llsm_set_error("llsmrt: device initialisation failed"); llsm_delete_rtsynth_buffer(b); return NULL;
  }
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
  rc |= llsm_gpu_synchronize(ctx);
  llsm_gpu_delete_batch(tb);
  if(rc || ! reset_state(b, true)) { llsm_delete_rtsynth_buffer(b); return NULL; }
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
  { std::lock_guard<std::mutex> lock(b -> pend_mtx); b -> pending = false; }   // (a hop in flight dies with the buffer)
  if(b -> conf) llsm_delete_container(b -> conf);
  delete b;
}

int llsm_rtsynth_buffer_getlatency(llsm_rtsynth_buffer* src) {          // llsmrt.c:568-571
  RtBuffer* b = (RtBuffer*)src;
  return -b -> sin_pos - b -> curr_nhop;
}

static void complete_pending(RtBuffer* b, bool consumer = false);
// Samples ready to be fetched.  With pipelined feeds the hop in flight is not counted -- unless the ring is dry: a consumer
// that drains with `while(numoutput > 0) fetch` must not stop one hop short of the end, so a count of zero waits for the hop
// (the rule of the fetch calls).
static int rt_numoutput(RtBuffer* b, int stream) {
  if(b -> pending && b -> nout[stream] <= 0) complete_pending(b, true);
  return b -> nout[stream];
}
int llsm_rtsynth_buffer_numoutput(llsm_rtsynth_buffer* src) { return rt_numoutput((RtBuffer*)src, 0); }

// the output stage of one hop (llsmrt.c:480-503): block while any stream's ring is full, then append
// may_block = false (a consumer completing the hop in flight): never waits for room -- returns false with nothing appended
// when any stream's ring is too full, and the hop stays pending for the producer or a later call
static bool append_outputs(RtBuffer* b, const float* out /* [S][2][stride] or NULL: zeros */, int stride = 0, int nhop_out = -1,
                           bool may_block = true) {
  if(stride <= 0) stride = b -> max_hop;
  const int S = b -> S;
  const int next_nhop = nhop_out >= 0 ? nhop_out : b -> next_nhop;   // (a pipelined hop is appended after the next feed moved on)
  static const std::vector<float> zeros(1 << 16, 0.0f);
  {
    std::unique_lock<std::mutex> lock(b -> mtx);
    auto room = [&] {
      for(int s2 = 0; s2 < S; s2 ++) if(b -> nout[s2] > b -> capacity - next_nhop) return false;
      return true;
    };
    if(may_block) b -> cv.wait(lock, room);
    else if(! room()) return false;
    for(int s2 = 0; s2 < S; s2 ++) {
      b -> out_p[s2].appendchunk(next_nhop, out ? out + ((size_t)s2 * 2 + 0) * stride : zeros.data());
      b -> out_ap[s2].appendchunk(next_nhop, out ? out + ((size_t)s2 * 2 + 1) * stride : zeros.data());
      b -> nout[s2] += next_nhop;
    }
  }
  b -> cv.notify_all();
  return true;
}

// The hop a pipelined feed left in flight: wait for the device, then its samples into the rings (as the tail of a
// synchronous feed does).  Called by the next feed before it touches the pinned blocks, by a consumer that finds the
// rings empty, and by clear / delete.
// consumer: called from a fetch / numoutput that found its ring dry.  A consumer NEVER blocks here: if the
// producer is completing the hop at this moment it does not queue up behind it, and if it gets the lock itself it
// appends only when every stream's ring has room -- a group drained unevenly by one thread (stream 0 dry, stream 1
// near capacity) would otherwise wait, holding pend_mtx, for room that only it can make while the producer's next feed
// waits for pend_mtx.  Without room the hop stays pending: the producer's next feed (which may block, as the
// reference's feed does on a full ring, llsmrt.c:489-493) or a later consumer call completes it.
static void complete_pending(RtBuffer* b, bool consumer) {
  std::unique_lock<std::mutex> lock(b -> pend_mtx, std::defer_lock);
  if(consumer) { if(! lock.try_lock()) return; } else lock.lock();
  if(! b -> pending) return;
  (void)hipSetDevice(llsm_engine_device(b -> ctx));
  // (the hop's own event, not the stream: the next hop may already be enqueued behind it)
  bool done;
  if(hipEventSynchronize(b -> hop_done[b -> pending_blk]) != hipSuccess) {
    llsm_set_error("llsmrt: feed failed on the device");
    done = append_outputs(b, nullptr, 0, b -> pending_nhop, ! consumer);
  } else done = append_outputs(b, b -> h_out0 + (size_t)b -> pending_blk * b -> out_elems, b -> pending_ostride, b -> pending_nhop,
                               ! consumer);
  if(done) b -> pending = false;
}

// Pulse tracker of one stream for this hop (llsmrt.c:305-379, 396-419): host state machine and effect
// callbacks; fills the stream's job / pulse / op slots.  Returns false on an unsupported pulse size.
static bool schedule_pbp(RtBuffer* b, int s2, llsm_container* frame, float f0, int nhop) {
  RtPbpOp& op = b -> h_ops.p[s2];
  PbpJob& job = b -> h_jobs.p[b -> njobs_hop];          // compact list: only streams with a pulse group this hop
  const double fs = b -> fs;
  int* pbpsyn = (int*)llsm_container_get(frame, LLSM_FRAME_PBPSYN);
  llsm_pbpeffect* pbpeff = (llsm_pbpeffect*)llsm_container_get(frame, LLSM_FRAME_PBPEFF);
  const bool pbp_on = pbpsyn != NULL && pbpsyn[0] == 1;
  const bool has_hm = llsm_container_get(frame, LLSM_FRAME_HM) != NULL;
  double len_period = fs / (double)f0;
  lf::Model source_model;
  // (the LF solve behind the projection is most of a hop's host time; a stream whose Rd and F0 did not move since its
  // last hop -- bit for bit -- reuses the model phase: same numbers)
  const float rd_now = b -> h_rd.p[s2];
  const bool same = b -> lf_valid[s2] && __builtin_memcmp(& rd_now, & b -> lf_rd[s2], 4) == 0 && __builtin_memcmp(& f0, & b -> lf_f0[s2], 4) == 0;
  const double pulse_projected = llsm_l1_pulse_projection((double)rd_now, (double)f0,
    (double)b -> h_vsphse.p[(size_t)s2 * b -> maxnhar], fs, 0.0, & source_model, & b -> lf_p0[s2], same);
  b -> lf_valid[s2] = 1; b -> lf_rd[s2] = rd_now; b -> lf_f0[s2] = f0;
  const int len_reset = (int)(std::max(len_period, (double)nhop) * 2);
  if(pulse_projected - b -> pulse[s2] > len_reset) b -> pulse[s2] = pulse_projected - len_reset;
  int num_periods = (int)std::round((pulse_projected - b -> pulse[s2]) / len_period);
  if(num_periods > 0) len_period = (pulse_projected - b -> pulse[s2]) / num_periods;
  const int pulse_size = lp::nextpow2(std::max(len_period * 2, (double)b -> nspec));
  bool onset = false, termination = false, sinusoids = false;
  if(pbp_on && ! b -> pbp_state[s2]) {
    onset = true; b -> pbp_state[s2] = 1; b -> pbp_offset[s2] = -nhop;
    sinusoids = true;
  }
  if(! pbp_on && b -> pbp_state[s2]) {
    termination = true; b -> pbp_state[s2] = 0;
    num_periods += (int)std::ceil((double)(-b -> pbp_offset[s2]) / len_period);
  }
  bool ok = true;
  if(b -> pbp_state[s2] || termination) {
    const int period_begin = onset ? -2 : 0, num_pulses = num_periods - period_begin;
    const int pre_rotate = (int)std::min(len_period, (double)(nhop * 2));
    if(num_pulses > 0) {
      if(pulse_size > b -> pulse_max || pulse_size >= b -> ninternal) {
        llsm_set_error("llsmrt: pulse group outside the supported size: 2^ceil(log2(max(2 periods, NSPEC))) must stay below the 0.2 s "
          "of the internal buffers (llsmrt.c:169; the reference writes past its dual buffer there) and below 8192"); ok = false;
      } else if(num_pulses > b -> pulse_pool - b -> npulses_hop) {
        llsm_set_error("llsmrt: stream " + std::to_string(s2) + " needs " + std::to_string(num_pulses) + " glottal pulses this hop, " +
          std::to_string(b -> pulse_pool - b -> npulses_hop) + " of the pool of " + std::to_string(b -> pulse_pool) +
          " are left (F0 " + std::to_string(f0) + " Hz): its pulse group is dropped"); ok = false;
      }
      std::vector<double> offsets(num_pulses);
      PbpPulse* pl = b -> h_pulses.p + b -> npulses_hop;
      for(int i = 0; i < num_pulses; i ++) {
        double delta_t = 0; lf::Model src = source_model;
        if(pbpeff != NULL && pbpeff -> modifier != NULL) {
          llsm_gfm gm;
          gm.Fa = (FP_TYPE)(1.0 / (source_model.ta * source_model.T0));
          gm.Rk = (FP_TYPE)((source_model.te - source_model.tp) / source_model.tp);
          gm.Rg = (FP_TYPE)(0.5 / source_model.tp); gm.T0 = (FP_TYPE)source_model.T0; gm.Ee = (FP_TYPE)source_model.Ee;
          FP_TYPE dt = 0;
          pbpeff -> modifier(& gm, & dt, pbpeff -> info, frame);
          delta_t = dt;
          src.ta = 1.0 / (double)gm.Fa / (double)gm.T0; src.tp = 0.5 / (double)gm.Rg;
          src.te = src.tp + src.tp * (double)gm.Rk; src.T0 = gm.T0; src.Ee = gm.Ee;
        }
        offsets[i] = b -> pulse[s2] + (i + period_begin) * len_period + delta_t * fs;
        if(ok) { pl[i].T0 = src.T0; pl[i].te = src.te; pl[i].tp = src.tp; pl[i].ta = src.ta; pl[i].Ee = src.Ee; pl[i].pad = 0; }
      }
      const int pulse_base = (int)offsets[0];
      if(ok) {
        for(int i = 0; i < num_pulses; i ++) pl[i].offset = (float)(offsets[i] - pulse_base);
        job.frame = s2; job.first = b -> npulses_hop; job.npulse = num_pulses; job.size = pulse_size;
        b -> npulses_hop += num_pulses;
        job.pre_rotate = pre_rotate; job.out_off = s2 * b -> pulse_max; job.start = 0; job.zero_extra = -1;
        op.add_off = pulse_base - pre_rotate - nhop; op.add_size = pulse_size;
        b -> njobs_hop ++;
      }
    }
  }
  if(! b -> pbp_state[s2]) sinusoids = true;
  b -> pulse[s2] = pulse_projected;
  if(b -> pbp_state[s2] && b -> pbp_offset[s2] <= b -> sin_pos + nhop) { op.rd_on = 1; op.rd_off = b -> pbp_offset[s2]; }
  if(termination) {
    const int size = -nhop - b -> pbp_offset[s2];
    if(size > 0) { op.term_off = b -> pbp_offset[s2]; op.term_size = size; }
  }
  if(sinusoids) {
    b -> h_f0sin.p[s2] = f0;
    if(! has_hm) b -> h_sel.p[s2] = 1;                  // llsm_frame_tolayer0 first (llsmrt.c:343-344, 389-390)
  }
  return ok;
}

// Packing the frames of MANY streams (llsmrt.c:255-291 per stream: a walk over the caller's frame objects and seven row
// copies) is what bounds a feed beyond ~128 streams (round 5: 0.35 us per stream, 356 us per hop at 1 024 streams against
// a 25 us kernel).  The rows of different streams are disjoint, so groups of >= LLSM_RT_PACK_MIN (128) streams split the
// loop over a few helper threads that spin between hops (a sleeping thread would cost more to wake than a hop lasts) and
// go to sleep 5 ms after the last one.  $LLSM_RT_PACK_THREADS: helpers (default 3, 0 = off).  Harmonic-model buffers only:
// pulse-by-pulse packing runs the callers' llsm_fgfm callbacks, which must stay in stream order on the feeding thread.
namespace {
struct RtPackPool {
  int nhelp = 0; bool started = false;
  std::vector<std::thread> th;
  std::atomic<unsigned> gen{0}; std::atomic<int> next{0}, left{0}; std::atomic<bool> quit{false};
  std::atomic<int> asleep{0};
  const std::function<void(int, int)>* fn = nullptr; int n = 0, chunk = 1;
  std::mutex mx; std::condition_variable cv;
  void worker(unsigned seen) {                          // seen: the generation at start() -- NOT read here: a helper that first
                                                        // runs after run() has raised it would wait for a change that never comes
    auto last = std::chrono::steady_clock::now();
    for(;;) {
      unsigned g = gen.load(std::memory_order_acquire);
      if(g == seen) {
        if(quit.load()) return;
        if(std::chrono::steady_clock::now() - last > std::chrono::milliseconds(5)) {
          std::unique_lock<std::mutex> lock(mx);
          asleep ++;
          cv.wait(lock, [&] { return gen.load(std::memory_order_acquire) != seen || quit.load(); });
          asleep --;
          last = std::chrono::steady_clock::now();
        } else __builtin_ia32_pause();
        continue;
      }
      seen = g;
      for(;;) { const int lo = next.fetch_add(chunk); if(lo >= n) break; (*fn)(lo, std::min(n, lo + chunk)); }
      left.fetch_sub(1, std::memory_order_acq_rel);
      last = std::chrono::steady_clock::now();
    }
  }
  void start() {
    started = true;
    const char* e = std::getenv("LLSM_RT_PACK_THREADS");
    int want = 3;
    if(e && *e) want = std::atoi(e);
    const int hw = (int)std::thread::hardware_concurrency();
    if(want > hw - 1) want = hw - 1;
    if(want < 0) want = 0;
    nhelp = want;
    const unsigned g0 = gen.load(std::memory_order_acquire);
    for(int i = 0; i < nhelp; i ++) th.emplace_back([this, g0] { worker(g0); });
  }
  // f(lo, hi) over [0, count) in chunks, on the calling thread and the helpers; returns when every chunk is done
  void run(int count, const std::function<void(int, int)>& f) {
    if(! started) start();
    if(nhelp == 0) { f(0, count); return; }
    fn = & f; n = count; chunk = std::max(8, count / (4 * (nhelp + 1)));
    next.store(0); left.store(nhelp);
    gen.fetch_add(1, std::memory_order_release);
    if(asleep.load() > 0) { std::lock_guard<std::mutex> lock(mx); cv.notify_all(); }
    for(;;) { const int lo = next.fetch_add(chunk); if(lo >= n) break; f(lo, std::min(n, lo + chunk)); }
    while(left.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
  }
  ~RtPackPool() {
    quit.store(true);
    { std::lock_guard<std::mutex> lock(mx); cv.notify_all(); }
    for(auto& t : th) t.join();
  }
};
RtPackPool g_pack_pool;
std::mutex g_pack_pool_mx;                              // one feed at a time uses the helpers (others pack on their own thread)
const int kPackMin = [] { const char* e = std::getenv("LLSM_RT_PACK_MIN"); const int v = (e && *e) ? std::atoi(e) : 128; return v > 1 ? v : 2; }();
}

// One hop for every stream of the group: frames[s] is the frame of stream s (llsmrt.c:505-521).
static void feed_group(RtBuffer* b, llsm_container** frames, bool force_pipe = false) {
  // LLSM_TIMING=1: phase times of a feed (packing the frames | enqueue | device + completion | rings and prev_nm),
  // averaged over 200 hops, on stderr
  static const bool timing = std::getenv("LLSM_TIMING") != nullptr;
  static thread_local double acc[4] = {0, 0, 0, 0}; static thread_local int nacc = 0;
  static thread_local double pack_sched = 0, pack_copy = 0;          // inside `pack`: pulse trackers | layer-1 row copies
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) {
    return std::chrono::duration<double, std::micro>(c - a).count(); };
  // Pipelined feeds (llsm_gpu_rt_pipeline): the previous hop may still be on the device.  It used the OTHER copy of the
  // pinned blocks, so this hop is packed and enqueued behind it first and the previous hop's samples are appended
  // after that (complete_pending below) -- the device goes from one hop's kernel to the next without waiting for the host.
  const bool pipe = force_pipe || g_rt_pipeline.load() > 0;   // (force_pipe: the inner hops of llsm_rtsynth_group_feed_many)
  if(! pipe) complete_pending(b);
  const auto t_0 = now();
  (void)hipSetDevice(llsm_engine_device(b -> ctx));
  LaunchCtx* P = llsm_engine_launch_ctx(b -> ctx);
  update_cycle(b);
  b -> h_params = b -> h_params0 + (size_t)b -> blk * b -> params_bytes;
  for(auto& v : b -> h_views) {                         // (memcpy: the views are float* / int* / struct pointers behind void**)
    void* q = (void*)(b -> h_params + v.second);
    std::memcpy((void*)v.first, & q, sizeof(q));
  }
  b -> h_out = b -> h_out0 + (size_t)b -> blk * b -> out_elems;
  const int S = b -> S, nch = b -> nchannel, cap = b -> ninternal, me = b -> me > 0 ? b -> me : 1;
  const int nhop = b -> curr_nhop, nwin = 2 * nhop, npsd = b -> npsd, mh = b -> maxnhar;
  WinEntry* we = (nhop > b -> max_hop || b -> next_nhop > b -> max_hop || nhop < 1) ? nullptr : get_window(b, nhop);
  if(! we) {                                            // keep the output length consistent: one hop of silence
    complete_pending(b);
    if(nhop > b -> max_hop || b -> next_nhop > b -> max_hop) llsm_set_error("llsmrt: hop exceeds buffer");
    if(b -> next_nhop > 0 && b -> next_nhop < (1 << 16)) append_outputs(b, nullptr);
    return;
  }
  // ---- frames -> parameter rows (llsmrt.c:255-291), written straight into the pinned block
  float *f0v = b -> h_f0.p, *cyc = b -> h_cyc.p, *ampl = b -> h_ampl.p, *phse = b -> h_phse.p;
  float *edc = b -> h_edc.p, *eamp = b -> h_eamp.p, *ephs = b -> h_ephs.p, *psd = b -> h_psd.p;
  int *nharv = b -> h_nhar.p, *nhev = b -> h_nhar_e.p, *hasnm = b -> h_has_nm.p;
  for(int k = 0; k < 3; k ++)                           // (the pulse pool behind it is written where it is used)
    std::memset(b -> h_params + b -> zero_rng[k][0], 0, b -> zero_rng[k][1] - b -> zero_rng[k][0]);
  for(size_t k = 0; k < (size_t)S * nch; k ++) edc[k] = 1e-5f;
  bool truncated = false, any_sel = false, any_sin = false;
  int size_max = 64;
  b -> njobs_hop = 0; b -> npulses_hop = 0;
  std::atomic<bool> trunc_any{false};
  // the rows every buffer carries (harmonic model, envelope harmonics, band energies) of streams [lo, hi)
  auto pack_rows = [&](int lo, int hi) {
    bool tr = false;
    for(int s2 = lo; s2 < hi; s2 ++) {
      llsm_container* frame = frames[s2];
      FP_TYPE* f0p = (FP_TYPE*)llsm_container_get(frame, LLSM_FRAME_F0);
      llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(frame, LLSM_FRAME_HM);
      llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(frame, LLSM_FRAME_NM);
      f0v[s2] = f0p ? *f0p : 0.0f;
      cyc[s2] = b -> cycle;
      int nhar = hm ? hm -> nhar : -1;
      if(nhar > b -> nfft) nhar = b -> nfft;                         // llsmrt.c:280
      if(nhar > mh) { nhar = mh; tr = true; }
      if(nhar > 0) {
        std::memcpy(ampl + (size_t)s2 * mh, hm -> ampl, sizeof(float) * (size_t)nhar);
        std::memcpy(phse + (size_t)s2 * mh, hm -> phse, sizeof(float) * (size_t)nhar);
      }
      nharv[s2] = nhar;
      int nhe = 0;
      hasnm[s2] = nm != NULL;
      if(nm)
        for(int c = 0; c < nch && c < nm -> nchannel; c ++) {
          edc[(size_t)s2 * nch + c] = nm -> edc[c];
          int n = nm -> eenv[c] ? nm -> eenv[c] -> nhar : 0;
          if(n > b -> me) { n = b -> me; tr = true; }
          if(n > nhe) nhe = n;
          for(int k = 0; k < n; k ++) {
            eamp[((size_t)s2 * nch + c) * me + k] = nm -> eenv[c] -> ampl[k];
            ephs[((size_t)s2 * nch + c) * me + k] = nm -> eenv[c] -> phse[k];
          }
        }
      nhev[s2] = nhe;
    }
    if(tr) trunc_any.store(true, std::memory_order_relaxed);
  };
  {
    bool pooled = false;
    if(S >= kPackMin) {
      std::unique_lock<std::mutex> pl(g_pack_pool_mx, std::try_to_lock);
      if(pl.owns_lock()) { g_pack_pool.run(S, pack_rows); pooled = true; }
    }
    if(! pooled) pack_rows(0, S);
  }
  truncated = trunc_any.load();
  for(int s2 = 0; b -> l1 && s2 < S; s2 ++) {
    llsm_container* frame = frames[s2];
    llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(frame, LLSM_FRAME_HM);
    {
      // llsmrt.c:295-304: without VSPHSE / RD, or unvoiced, the deterministic part of this hop is empty
      FP_TYPE* vs = (FP_TYPE*)llsm_container_get(frame, LLSM_FRAME_VSPHSE);
      FP_TYPE* vt = (FP_TYPE*)llsm_container_get(frame, LLSM_FRAME_VTMAGN);
      FP_TYPE* rd = (FP_TYPE*)llsm_container_get(frame, LLSM_FRAME_RD);
      b -> h_hashm.p[s2] = hm != NULL;
      if(vs && rd && vt && f0v[s2] != 0 && llsm_fparray_length(vs) > 0 && llsm_fparray_length(vt) >= b -> nspec) {
        int n = llsm_fparray_length(vs); if(n > mh) { n = mh; truncated = true; }
        b -> h_rd.p[s2] = *rd; b -> h_nvs.p[s2] = n;
        b -> h_vsphse.p[(size_t)s2 * mh] = vs[0];         // (the pulse tracker's phase reference)
        // a stream whose pulse group cannot be placed (error text set) keeps an empty op and no job: its hop carries
        // no pulses, the other streams of the group are not touched
        const int jobs_before = b -> njobs_hop;
        const auto ts0 = timing ? now() : t_0;
        (void)schedule_pbp(b, s2, frame, f0v[s2], nhop);
        const auto ts1 = timing ? now() : t_0;
        // the source-phase and vocal-tract rows travel only on the hops whose kernels read them: a pulse group placed
        // (k_pbp_pulse) or harmonic rows to rebuild (k_l1_to_l0)
        if(b -> njobs_hop > jobs_before || b -> h_sel.p[s2]) {
          std::memcpy(b -> h_vsphse.p + (size_t)s2 * mh, vs, sizeof(float) * (size_t)n);
          std::memcpy(b -> h_vtmagn.p + (size_t)s2 * b -> nspec, vt, sizeof(float) * (size_t)b -> nspec);
        }
        if(timing) { pack_sched += us(ts0, ts1); pack_copy += us(ts1, now()); }
        any_sel |= b -> h_sel.p[s2] != 0; any_sin |= b -> h_f0sin.p[s2] > 0;
        if(b -> h_ops.p[s2].add_size > 0) size_max = std::max(size_max, b -> h_ops.p[s2].add_size);
      }
    }
  }
  if(truncated) llsm_set_error("llsmrt: frame carries more harmonics than the stream rows hold (truncated)");
  const auto t_1 = now();
  hipStream_t st = P -> stream;
  // LLSM_RT_GRAPH=1: the whole hop (copy in, launches, copy out) goes to the device as ONE graph launch.
  // Hops that hand rebuilt harmonic models back into pageable host memory keep the plain enqueue.
  bool capturing = g_rt_graph.load() > 0 && st != nullptr && ! P -> prof_begin && !(b -> l1 && any_sel) &&
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
  const int fuse_mode = g_rt_fused.load();
  const bool fused = fuse_mode > 0, one_launch = fuse_mode > 1 && b -> nfft <= 2048;   // (k_rt_hop's LDS)
  // direct: no copy in and no copy out.  The first kernel of the hop moves the rows it and the second need (a few
  // hundred of a row's nfft harmonic slots) from the pinned block into the device rows, the second writes the samples
  // into the pinned output block; each copy was a dependent blit launch of 8 - 15 us around kernels of 14 us
  // (tools/ubench/host_io.hip).  Pulse-by-pulse buffers: the pulse kernel reads its groups, pulses and layer-1 rows from
  // the pinned block as well; only a hop that rebuilds harmonic rows on the device (a frame without a harmonic model
  // at the onset / end of pulse-by-pulse synthesis) keeps the copy in.
  const bool direct_out = fused && ! capturing && g_rt_direct.load() > 0;      // samples straight into the pinned block
  const bool direct = direct_out && !(b -> l1 && any_sel);
  // one copy; the kernels below are ordered after it on the stream, and the pinned block is not
  // touched again before the synchronisation at the end of this call
  if(! direct) std::memcpy(psd, b -> h_psd2[b -> psd_cur], sizeof(float) * (size_t)S * npsd);
  size_t copy_bytes = b -> params_fixed + sizeof(PbpPulse) * (size_t)b -> npulses_hop;
  if(b -> l1 && (any_sin || any_sel)) copy_bytes = b -> params_bytes;       // the harmonic rows behind the pulse pool as well
  int rc = direct ? 0 : hipMemcpyAsync(b -> d_params.p, b -> h_params, copy_bytes, hipMemcpyHostToDevice, st) != hipSuccess;
  BatchDev d; std::memset(& d, 0, sizeof(d));
  d.n_utt = S; d.nframes = S; d.maxnhar = mh; d.maxnhar_e = b -> me; d.npsd = npsd;
  d.nchannel = nch; d.thop = b -> thop; d.fs = b -> fs; d.rel_winsize = 4;
  d.frm_utt = b -> d_frm_utt.p; d.frm_off = b -> d_frm_off.p;
  d.f0 = b -> d_f0.p; d.nhar = b -> d_nhar.p; d.ampl = b -> d_ampl.p; d.phse = b -> d_phse.p;
  d.psd = b -> d_psd.p; d.psdres = b -> d_psdres.p; d.has_psdres = b -> d_zero.p;
  d.edc = b -> d_edc.p; d.nhar_e = b -> d_nhar_e.p; d.eenv_ampl = b -> d_eamp.p; d.eenv_phse = b -> d_ephs.p;
  RtRows host; std::memset(& host, 0, sizeof(host));
  if(direct) {                                          // (pinned host memory is mapped into the device's address space)
    host.f0 = b -> h_f0.p; host.cyc = b -> h_cyc.p; host.nhar = b -> h_nhar.p; host.nhar_e = b -> h_nhar_e.p;
    host.has_nm = b -> h_has_nm.p; host.ampl = b -> h_ampl.p; host.phse = b -> h_phse.p; host.edc = b -> h_edc.p;
    host.eamp = b -> h_eamp.p; host.ephs = b -> h_ephs.p; host.psd = b -> h_psd2[b -> psd_cur];
    if(b -> l1) { host.f0sin = b -> h_f0sin.p; host.ops = b -> h_ops.p; }
  }
  RtPbpArgs pbp; std::memset(& pbp, 0, sizeof(pbp));
  if(b -> l1) {
    pbp.ops = b -> d_ops.p; pbp.frwd = b -> dual_f.p; pbp.bkwd = b -> dual_b.p; pbp.dual_curr = b -> dual_curr;
    pbp.pulse_out = b -> pulse_out.p; pbp.pulse_stride = b -> pulse_max;
  }
  int tw_nmax = 0; const float2* tw = llsm_engine_twiddles(b -> ctx, & tw_nmax);
  // feed_deterministic: envelope frames + harmonic frame, then the ring adds.  g_rt_fused (default): the hop is two
  // launches -- k_rt_front (envelope frames beside the harmonic frame, ring adds, excitation) and k_rt_back (noise filter
  // on four wavefronts per pair of streams, noise ring, output samples); 0: the five single-purpose launches
  if(! fused) rc |= launch_env_frames(P, d, b -> fs, nwin, we -> w.p, b -> envf.p);
  const float* f0_sin = d.f0;
  if(b -> l1) {
    L1Dev ld; std::memset(& ld, 0, sizeof(ld));
    ld.nframes = S; ld.maxnhar = mh; ld.nspec = b -> nspec; ld.fnyq = b -> fnyq; ld.lip_radius = b -> lip_radius;
    ld.f0 = b -> d_f0.p; ld.nhar = b -> d_nhar.p; ld.ampl = b -> d_ampl.p; ld.phse = b -> d_phse.p;
    ld.rd = b -> d_rd.p; ld.vtmagn = b -> d_vtmagn.p; ld.vsphse = b -> d_vsphse.p; ld.nvsphse = b -> d_nvs.p;
    ld.has_hm = b -> d_hashm.p;
    if(any_sel) rc |= launch_l1_to_l0(P, ld, b -> maxnhar_conf, 1, b -> d_sel.p, tw, tw_nmax);
    if(b -> njobs_hop > 0) {
      if(direct) {                                      // (no l1_to_l0 on this hop: nothing on the device is newer than the pinned rows)
        ld.f0 = b -> h_f0.p; ld.rd = b -> h_rd.p; ld.vtmagn = b -> h_vtmagn.p; ld.vsphse = b -> h_vsphse.p; ld.nvsphse = b -> h_nvs.p;
      }
      rc |= launch_pbp_pulse(P, ld, direct ? b -> h_jobs.p : b -> d_jobs.p, b -> njobs_hop, direct ? b -> h_pulses.p : b -> d_pulses.p,
        size_max, b -> fs, tw, tw_nmax, b -> pulse_out.p);
    }
    f0_sin = b -> d_f0sin.p;                            // sinusoids only where the state machine asks for them
  }
  b -> exc_curr = (b -> exc_curr + nhop) % cap;
  const int exc_cycle_hop = b -> exc_cycle;
  if(one_launch) { }                                    // (below, with the second half of the hop)
  else if(fused)
    rc |= launch_rt_front(P, d, nwin, we -> w.p, f0_sin, b -> d_cyc.p, b -> envf.p, b -> frames_sin.p, mh, b -> mod.p,
      b -> sinr.p, b -> noiser.p, cap, b -> mod_curr, b -> sin_curr, b -> noise_curr, nhop, b -> d_has_nm.p, b -> tpl.p,
      b -> excr.p, b -> ntemplate, b -> exc_curr, b -> exc_cycle, b -> exc_frame.p, direct ? & host : nullptr,
      b -> l1 ? & pbp : nullptr);
  else {
    BatchDev ds = d; ds.f0 = (float*)f0_sin;
    rc |= launch_synth_frames(P, ds, nwin, we -> w.p, b -> d_cyc.p, b -> frames_sin.p, mh);
    // ring adds + run_excitation_buffers(curr_nhop) in one launch (the excitation reads only its own stream's envelope
    // ring); the pulse-by-pulse adds into the sinusoid ring follow
    rc |= launch_rt_rings_excite(P, S, b -> mod.p, b -> sinr.p, b -> noiser.p, cap, nch, b -> mod_curr, b -> sin_curr,
      b -> noise_curr, nhop, nwin, b -> envf.p, b -> frames_sin.p, f0_sin, b -> d_has_nm.p, b -> d_nhar.p,
      b -> tpl.p, b -> excr.p, b -> ntemplate, b -> exc_curr, b -> exc_cycle, b -> exc_frame.p);
  }
  b -> exc_cycle = (b -> exc_cycle + nhop) % b -> ntemplate;
  if(b -> l1) {
    if(! fused)                                         // (k_rt_front / k_rt_hop carry it themselves)
      rc |= launch_rt_pbp(P, S, b -> d_ops.p, b -> dual_f.p, b -> dual_b.p, cap, b -> dual_curr, b -> sinr.p, b -> sin_curr,
        nhop, we -> w.p, b -> pulse_out.p, b -> pulse_max);
    b -> dual_curr = (b -> dual_curr + nhop) % cap;
  }
  // feed_filter on the previous frame's noise model (rows at -200 dB are skipped: no prev_nm yet), then feed_mix.
  // Output rows are packed at the hop's own length (rounded to 16 samples), not at the buffer's maximum: the copy back
  // is half the bytes at the nominal hop
  const int ostride = (b -> next_nhop + 15) & ~15;
  if(one_launch)
    rc |= launch_rt_hop(P, d, nwin, we -> w.p, f0_sin, b -> d_cyc.p, b -> envf.p, b -> frames_sin.p, mh, b -> mod.p,
      b -> sinr.p, b -> noiser.p, cap, b -> mod_curr, b -> sin_curr, b -> noise_curr, nhop, b -> d_has_nm.p, b -> tpl.p,
      b -> excr.p, b -> ntemplate, b -> exc_curr, exc_cycle_hop, b -> exc_frame.p, direct ? & host : nullptr,
      b -> fnyq, we -> inv_wsqr, b -> nfft, ilog2(b -> nfft), tw, tw_nmax, b -> nframes.p, b -> live.p, b -> sin_pos,
      b -> next_nhop, ostride, direct_out ? b -> h_out : b -> out.p, b -> l1 ? & pbp : nullptr, fuse_mode > 2);
  else if(fused)
    rc |= launch_rt_back(P, d, b -> exc_frame.p, b -> fnyq, b -> fs, nwin, we -> w.p, we -> inv_wsqr, b -> nfft,
      ilog2(b -> nfft), tw, tw_nmax, b -> nframes.p, b -> live.p, b -> noiser.p, b -> sinr.p, cap, b -> noise_curr,
      b -> sin_curr, b -> sin_pos, b -> next_nhop, ostride, direct_out ? b -> h_out : b -> out.p);
  else {
    rc |= launch_noise_filter(P, d, b -> exc_frame.p, nullptr, nullptr, b -> fnyq, b -> fs, nwin, we -> w.p,
      we -> inv_wsqr, b -> nfft, ilog2(b -> nfft), tw, tw_nmax, b -> nframes.p, b -> live.p, 1);
    rc |= launch_rt_mix(P, S, b -> noiser.p, b -> sinr.p, cap, b -> noise_curr, b -> sin_curr, b -> sin_pos,
      b -> nfft, b -> nframes.p, b -> live.p, b -> next_nhop, ostride, b -> out.p);
  }
  if(! direct_out)
    rc |= hipMemcpyAsync(b -> h_out, b -> out.p, sizeof(float) * S * 2 * ostride, hipMemcpyDeviceToHost, st) != hipSuccess;
  if(capturing) {
    hipGraph_t g = nullptr;
    rc |= hipStreamEndCapture(st, & g) != hipSuccess;
    if(! rc && g) {
      if(b -> gexec) {
        hipGraphExecUpdateResult res; hipGraphNode_t bad = nullptr;
        if(hipGraphExecUpdate(b -> gexec, g, & bad, & res) != hipSuccess) {   // topology changed: build anew
          (void)hipGraphExecDestroy(b -> gexec); b -> gexec = nullptr; (void)hipGetLastError();
        }
      }
      if(! b -> gexec) { rc |= hipGraphInstantiate(& b -> gexec, g, nullptr, nullptr, 0) != hipSuccess; b -> graph_rebuilds ++; }
      if(! rc) { rc |= hipGraphLaunch(b -> gexec, st) != hipSuccess; b -> graph_hops ++; g_rt_graph_hops ++; }
    }
    if(g) (void)hipGraphDestroy(g);
  }
  if(b -> l1 && any_sel) {                              // HM rows rebuilt from layer 1 go back onto the callers' frames
    float* hb = b -> hm_back.data();
    rc |= hipMemcpyAsync(hb, b -> d_ampl.p, sizeof(float) * (size_t)S * mh, hipMemcpyDeviceToHost, st) != hipSuccess;
    rc |= hipMemcpyAsync(hb + (size_t)S * mh, b -> d_phse.p, sizeof(float) * (size_t)S * mh, hipMemcpyDeviceToHost, st) != hipSuccess;
    rc |= hipMemcpyAsync(hb + (size_t)S * mh * 2, b -> d_nhar.p, sizeof(int) * S, hipMemcpyDeviceToHost, st) != hipSuccess;
  }
  // behind this hop's last device operation: what complete_pending waits for when the hop is left in flight
  rc |= hipEventRecord(b -> hop_done[b -> blk], st) != hipSuccess;
  // the hop before this one (pipelined feeds): both are on the device now, so its samples can be collected -- and the
  // level rows it read (h_psd2[psd_cur ^ 1], written below for the hop AFTER this one) are free again once it is done
  complete_pending(b);
  // prev_nm with PSDRES folded in (llsmrt.c:513-520): the NEXT hop's filter target.  It depends on the callers' frames only,
  // so it is formed here, while the device works on this hop, not after the synchronisation
  {
    const int nxt = b -> psd_cur ^ 1;
    for(int s2 = 0; s2 < S; s2 ++) {
      llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(frames[s2], LLSM_FRAME_NM);
      FP_TYPE* resvec = (FP_TYPE*)llsm_container_get(frames[s2], LLSM_FRAME_PSDRES);
      b -> has_prev[s2] = nm != NULL;
      float* pp = b -> h_psd2[nxt] + (size_t)s2 * npsd;
      if(nm) {
        const int np = std::min(npsd, nm -> npsd), nr = resvec ? std::min(npsd, llsm_fparray_length(resvec)) : 0;
        const float bias = (float)(0.375 / 2.3025851 * 10.0);
        std::memcpy(pp, nm -> psd, sizeof(float) * (size_t)np);
        for(int j = np; j < npsd; j ++) pp[j] = -120.0f;
        for(int j = 0; j < nr; j ++) pp[j] += resvec[j] - bias;
        b -> psd_blank[nxt][s2] = 0;
      } else if(! b -> psd_blank[nxt][s2]) {
        std::fill(pp, pp + npsd, -200.0f);
        b -> psd_blank[nxt][s2] = 1;
      }
    }
    b -> psd_cur = nxt;
  }
  const auto t_2 = now();
  // Pipelined: return now; the hop's samples are appended when the next feed starts (or a consumer runs dry).  The host
  // side of the next hop -- the caller's pulls, packing its frames -- then runs beside this hop's kernels instead of
  // after them.  Hops that hand rebuilt harmonic models back onto the caller's frames stay synchronous (the frame is the
  // caller's and may be gone by the next feed).
  if(! rc && pipe && !(b -> l1 && any_sel)) {
    std::lock_guard<std::mutex> lock(b -> pend_mtx);
    b -> pending_ostride = ostride; b -> pending_blk = b -> blk; b -> pending_nhop = b -> next_nhop;
    b -> pending = true;
    b -> blk ^= 1;
    return;
  }
  b -> blk ^= 1;
  const bool dev_failed = rc || hipStreamSynchronize(st) != hipSuccess;
  const auto t_3 = now();
  if(dev_failed) {
    llsm_set_error("llsmrt: feed failed on the device");
    append_outputs(b, nullptr);                         // the consumer still gets next_nhop (silent) samples
  } else {
    append_outputs(b, b -> h_out, ostride);
    if(b -> l1 && any_sel) {
      const float* hb = b -> hm_back.data(); const int* nh = (const int*)(hb + (size_t)S * mh * 2);
      for(int s2 = 0; s2 < S; s2 ++) {
        if(! b -> h_sel.p[s2]) continue;
        llsm_hmframe* hm = llsm_create_hmframe(nh[s2]);
        std::memcpy(hm -> ampl, hb + (size_t)s2 * mh, sizeof(float) * (size_t)nh[s2]);
        std::memcpy(hm -> phse, hb + (size_t)S * mh + (size_t)s2 * mh, sizeof(float) * (size_t)nh[s2]);
        llsm_container_attach_(frames[s2], LLSM_FRAME_HM, hm, (llsm_fdestructor)llsm_delete_hmframe, (llsm_fcopy)llsm_copy_hmframe);
      }
    }
  }
  if(timing) {
    acc[0] += us(t_0, t_1); acc[1] += us(t_1, t_2); acc[2] += us(t_2, t_3); acc[3] += us(t_3, now());
    if(++ nacc == 200) {
      if(llsm_rt2_timing_fetch) {                           // (-DRT2_TIMING builds only)
        unsigned long long ts[16]; llsm_rt2_timing_fetch(ts);
        std::fprintf(stderr, "[noise filter] levels + frame into X %.2f, forward transform %.2f, power spectrum %.2f, filter %.2f, inverse transform %.2f us\n",
          (ts[9] - ts[6]) * 0.01, (ts[10] - ts[9]) * 0.01, (ts[11] - ts[10]) * 0.01, (ts[12] - ts[11]) * 0.01, (ts[7] - ts[12]) * 0.01);
        std::fprintf(stderr, "[k_rt_hop2, workgroup 0] requests %.2f, rows over the link %.2f, write early samples .. frames in LDS %.2f, barrier %.2f, "
          "rings + excitation %.2f, pulses %.2f, noise filter %.2f, noise ring + samples %.2f us\n",
          (ts[1] - ts[0]) * 0.01, (ts[2] - ts[1]) * 0.01, (ts[3] - ts[2]) * 0.01, (ts[4] - ts[3]) * 0.01, (ts[5] - ts[4]) * 0.01,
          (ts[6] - ts[5]) * 0.01, (ts[7] - ts[6]) * 0.01, (ts[8] - ts[7]) * 0.01);
      }
      if(pack_sched > 0) std::fprintf(stderr, "[llsmrt pack] pulse trackers %.1f us, vocal-tract / source-phase rows %.1f us per hop\n", pack_sched / 200, pack_copy / 200);
      pack_sched = pack_copy = 0;
      std::fprintf(stderr, "[llsmrt feed, %d streams] pack %.1f us, enqueue %.1f us, device + completion %.1f us, rings + prev_nm %.1f us\n",
        b -> S, acc[0] / nacc, acc[1] / nacc, acc[2] / nacc, acc[3] / nacc);
      acc[0] = acc[1] = acc[2] = acc[3] = 0; nacc = 0;
    }
  }
}

void llsm_rtsynth_buffer_feed(llsm_rtsynth_buffer* dst, llsm_container* frame) {
  feed_group((RtBuffer*)dst, & frame);
}

// bulk pull of up to `max_samples` samples of one stream (non-blocking)
static int fetch_bulk(RtBuffer* b, int stream, FP_TYPE* dst_p, FP_TYPE* dst_ap, int max_samples) {
  int got = 0;
  if(b -> pending && b -> nout[stream] <= 0) complete_pending(b, true);   // a consumer that ran dry waits for the hop in flight
  {
    std::lock_guard<std::mutex> lock(b -> mtx);
    got = std::min(max_samples, b -> nout[stream]);
    if(got > 0) {
      if(dst_p) b -> out_p[stream].readchunk(-b -> nout[stream], got, dst_p);
      if(dst_ap) b -> out_ap[stream].readchunk(-b -> nout[stream], got, dst_ap);
      b -> nout[stream] -= got;
    } else got = 0;
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
  complete_pending(b);
  (void)hipSetDevice(llsm_engine_device(b -> ctx));
  std::lock_guard<std::mutex> lock(b -> mtx);
  reset_state(b, false);
}

int llsm_gpu_rt_graph(int on) { return on < 0 ? g_rt_graph.load() : g_rt_graph.exchange(on > 0 ? 1 : 0); }
long long llsm_gpu_rt_graph_hops(void) { return g_rt_graph_hops.load(); }
int llsm_gpu_rt_fused(int on) { return on < 0 ? g_rt_fused.load() : g_rt_fused.exchange(rt_fused_mode(on)); }
int llsm_gpu_rt_direct(int on) { return on < 0 ? g_rt_direct.load() : g_rt_direct.exchange(on > 0 ? 1 : 0); }
int llsm_gpu_rt_pipeline(int on) { return on < 0 ? g_rt_pipeline.load() : g_rt_pipeline.exchange(on > 0 ? 1 : 0); }

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
int llsm_rtsynth_group_numoutput(llsm_rtsynth_group* g, int stream) { return rt_numoutput((RtBuffer*)g, stream); }
void llsm_rtsynth_group_feed(llsm_rtsynth_group* g, llsm_container** frames) { feed_group((RtBuffer*)g, frames); }
// K hops in one call: frames[k * n_streams + s] is the frame of stream s for hop k.  The reference's producer runs ahead
// of its consumer whenever the output ring has room (feed only blocks on a FULL ring, llsmrt.c:489-493); fed one hop per
// call it pays a launch and a wait for the device per hop.  Here hop k + 1 is packed and enqueued while hop k is on the
// device (the machinery of the pipelined feeds, rt.cpp feed_group), every hop's samples are appended in order, and the
// call returns -- unless llsm_gpu_rt_pipeline(1) is set -- with the samples of ALL K hops in the rings: same kernels on
// the same numbers as K single feeds, bit-identical samples, no extra latency visible to the caller.  Blocks, as feed
// does, while a ring is full.
void llsm_rtsynth_group_feed_many(llsm_rtsynth_group* g, llsm_container** frames, int n_hops) {
  RtBuffer* b = (RtBuffer*)g;
  if(! b || ! frames || n_hops <= 0) return;
  for(int k = 0; k < n_hops; k ++) feed_group(b, frames + (size_t)k * b -> S, k + 1 < n_hops);
}
int llsm_rtsynth_group_fetch(llsm_rtsynth_group* g, int stream, FP_TYPE* dst_p, FP_TYPE* dst_ap, int max_samples) {
  RtBuffer* b = (RtBuffer*)g;
  if(stream < 0 || stream >= b -> S) return 0;
  return fetch_bulk(b, stream, dst_p, dst_ap, max_samples);
}

// every stream of the group at once: row s of dst_p / dst_ap ([n_streams][max_samples], either may be NULL) receives
// up to max_samples samples of stream s; counts (n_streams values, may be NULL) the samples per stream.  Returns the
// smallest count (the streams of a group advance in lockstep, so normally every count).
int llsm_rtsynth_group_fetch_all(llsm_rtsynth_group* g, FP_TYPE* dst_p, FP_TYPE* dst_ap, int max_samples, int* counts) {
  RtBuffer* b = (RtBuffer*)g;
  if(! b || max_samples <= 0) return 0;
  int least = max_samples;
  for(int s = 0; s < b -> S; s ++) {
    const int got = fetch_bulk(b, s, dst_p ? dst_p + (size_t)s * max_samples : nullptr,
      dst_ap ? dst_ap + (size_t)s * max_samples : nullptr, max_samples);
    if(counts) counts[s] = got;
    least = std::min(least, got);
  }
  return least;
}

}  // extern "C"
