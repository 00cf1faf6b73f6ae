// engine.cpp -- device context, batch memory layout in HBM, and the kernel
// sequences that implement llsm_analyze (layer0.c:478-511) and
// llsm_synthesize (layer0.c:636-664) for a whole batch of utterances.
// C-ABI entry points of include/llsm_gpu.h live at the bottom.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "cheby.h"
#include "engine.h"
#include "kernels.h"
#include "llsm_gpu.h"
#include "plan.h"
#include "batch.h"

namespace lp = llsm_plan;

// ------------------------------------------------------------- conventions
// Process-wide; read when a context / batch / llsmrt buffer is created.
namespace {
struct HostConventions { int hann_periodic = 0, mavg_half = 3, filtfilt_pad = 15, interp1u_excl = 0, kalman_init = 0, lobe_1e6 = 133979, lf_rd_clamp = 0; } g_hconv;
}
int llsm_conv_hann_periodic(void) { return g_hconv.hann_periodic; }
int llsm_conv_filtfilt_pad(void) { return g_hconv.filtfilt_pad; }
int llsm_conv_lf_rd_clamp(void) { return g_hconv.lf_rd_clamp; }
static int push_conventions(void) {
  DevConventions d; d.mavg_half = g_hconv.mavg_half; d.interp1u_excl = g_hconv.interp1u_excl;
  d.kalman_init = g_hconv.kalman_init; d.lf_rd_clamp = g_hconv.lf_rd_clamp;
  // units of 1e-6; 133979 stands for the calibrated constant itself
  d.lobe_bias = g_hconv.lobe_1e6 == 133979 ? 0.13397922601295542f : (float)(g_hconv.lobe_1e6 * 1e-6);
  return llsm_kernels_set_conventions(d) | llsm_l1_kernels_set_conventions(d);
}

// ------------------------------------------------------------------ errors
static thread_local std::string g_last_error;
void llsm_set_error(const std::string& msg) { g_last_error = msg; }


static hipEvent_t prof_event(llsm_gpu_context* c) {
  if(! c -> pool.empty()) { hipEvent_t e = c -> pool.back(); c -> pool.pop_back(); return e; }
  hipEvent_t e; hipEventCreate(& e); return e;
}
// Events are recorded on the stream of the launch, so the second stream of the analysis stays in use while profiling
// (round 6: rounds 3 - 5 profiled on one stream and switched the overlap off for it -- bench.py's timed region, which
// keeps the per-kernel events on, measured a serialised variant of the product).  A kernel that runs beside another
// reads longer than alone; `only` (llsm_gpu_set_profiling(ctx, 2) + llsm_gpu_profile_only) keeps the events to ONE
// kernel name, for a timed region that should carry the least of the instrument.
static void prof_begin_cb(void* user, const char* name, hipStream_t st) {
  llsm_gpu_context* c = (llsm_gpu_context*)user;
  c -> prof_skip = ! c -> prof_only.empty() && c -> prof_only != name;
  if(c -> prof_skip) return;
  ProfPending p; p.name = name; p.a = prof_event(c); p.b = prof_event(c);
  hipEventRecord(p.a, st);
  c -> pending.push_back(p);
}
static void prof_end_cb(void* user, hipStream_t st) {
  llsm_gpu_context* c = (llsm_gpu_context*)user;
  if(c -> prof_skip) return;
  hipEventRecord(c -> pending.back().b, st);
}
static void prof_drain(llsm_gpu_context* c) {
  if(c -> pending.empty()) return;
  hipStreamSynchronize(c -> stream);
  if(c -> aux) hipStreamSynchronize(c -> aux);
  for(auto& p : c -> pending) {
    float ms = 0; hipEventElapsedTime(& ms, p.a, p.b);
    ProfEntry& e = c -> prof[p.name];
    e.ms += ms; e.launches ++;
    c -> pool.push_back(p.a); c -> pool.push_back(p.b);
  }
  c -> pending.clear();
}

// Counts the changes of process-wide settings that a batch bakes into its tables when it is created (window conventions,
// filter padding, unit plans): batches kept between calls (capi.cpp worker_batch) are reused only within one epoch.
static std::atomic<unsigned long> g_config_epoch{1};
unsigned long llsm_engine_config_epoch(void) { return g_config_epoch.load(); }

extern "C" int llsm_gpu_set_convention(const char* name, int value) {
  const std::string n = name ? name : "";
  g_config_epoch.fetch_add(1);
  if(n == "hann_periodic" && (value == 0 || value == 1)) g_hconv.hann_periodic = value;
  else if(n == "moving_avg_half" && (value == 1 || value == 3)) g_hconv.mavg_half = value;
  else if(n == "filtfilt_pad" && value >= 1 && value <= 15) g_hconv.filtfilt_pad = value;
  else if(n == "interp1u_exclusive" && (value == 0 || value == 1)) g_hconv.interp1u_excl = value;
  else if(n == "kalman_init" && (value == 0 || value == 1)) g_hconv.kalman_init = value;
  else if(n == "spec2env_lobe_1e6" && value >= 0 && value <= 1000000) g_hconv.lobe_1e6 = value;
  else if(n == "lf_rd_clamp" && (value == 0 || value == 1)) g_hconv.lf_rd_clamp = value;
  else { llsm_set_error("llsm_gpu_set_convention: unknown name or value out of range"); return -1; }
  int ndev = 0;
  if(hipGetDeviceCount(& ndev) != hipSuccess || ndev <= 0) return 0;       // picked up when a context is created
  int cur = 0; hipGetDevice(& cur);
  int rc = 0;
  for(int d = 0; d < ndev; d ++) { if(hipSetDevice(d) == hipSuccess) rc |= push_conventions(); }
  hipSetDevice(cur);
  if(rc) llsm_set_error("llsm_gpu_set_convention: device update failed");
  return rc;
}
extern "C" int llsm_gpu_get_convention(const char* name) {
  const std::string n = name ? name : "";
  if(n == "hann_periodic") return g_hconv.hann_periodic;
  if(n == "moving_avg_half") return g_hconv.mavg_half;
  if(n == "filtfilt_pad") return g_hconv.filtfilt_pad;
  if(n == "interp1u_exclusive") return g_hconv.interp1u_excl;
  if(n == "kalman_init") return g_hconv.kalman_init;
  if(n == "spec2env_lobe_1e6") return g_hconv.lobe_1e6;
  if(n == "lf_rd_clamp") return g_hconv.lf_rd_clamp;
  return -1;
}

// units of the fused overlap-add kernels per batch (about one per resident wavefront: 8 / SIMD for the harmonic frames,
// 2 / SIMD for the noise frame pairs); macros so that tools/kbench.py can sweep them
#ifndef SYN_UNIT_DIV
#define SYN_UNIT_DIV 8192
#endif
#ifndef NF_UNIT_DIV
#define NF_UNIT_DIV 2048
#endif
static int virtual_devices(void);
static std::atomic<int> g_overlap([] { const char* e = std::getenv("LLSM_GPU_OVERLAP"); return (e && e[0] == '0') ? 0 : 1; }());
extern "C" int llsm_gpu_analysis_overlap(int on) { return on < 0 ? g_overlap.load() : g_overlap.exchange(on > 0 ? 1 : 0); }
extern "C" int llsm_gpu_device_count(void) {
  int n = 0;
  if(hipGetDeviceCount(& n) != hipSuccess) return 0;
  return n > 0 && virtual_devices() > 0 ? virtual_devices() : n;
}
extern "C" const char* llsm_gpu_last_error(void) { return g_last_error.c_str(); }

// $LLSM_GPU_VIRTUAL_DEVICES = N (test hook): the library reports N devices and places logical device d on physical
// device d mod (physical count) -- the multi-device branch of the in-process fan-out (capi.cpp fanout_run,
// LLSM_GPU_DEVICES=all) then runs with one context, stream and worker pool PER logical device on a one-GPU box.
static int virtual_devices(void) {
  const char* e = std::getenv("LLSM_GPU_VIRTUAL_DEVICES");
  const int v = e ? std::atoi(e) : 0;
  if(v > 0) {                                          // a test hook in a production library: say so, once, loudly
    static std::atomic<bool> warned{false};
    if(! warned.exchange(true))
      std::fprintf(stderr, "libllsm2_amd: WARNING: LLSM_GPU_VIRTUAL_DEVICES=%d is set -- a TEST hook: the library reports %d "
        "devices and places them on the physical devices round-robin (several contexts per GPU).  Unset it outside tests.\n", v, v);
  }
  return v > 0 ? v : 0;
}

extern "C" llsm_gpu_context* llsm_gpu_create_context(int device, void* stream) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(& n);
  if(e != hipSuccess || n <= 0) {
    llsm_set_error("no HIP device available (libllsm2_amd has no CPU fallback)");
    return nullptr;
  }
  const int nlogical = virtual_devices() > 0 ? virtual_devices() : n;
  if(device < 0 || device >= nlogical) { llsm_set_error("device index out of range"); return nullptr; }
  device %= n;
  if(hipSetDevice(device) != hipSuccess) { llsm_set_error("hipSetDevice failed"); return nullptr; }
  llsm_gpu_context* c = new llsm_gpu_context();
  c -> device = device;
  if(stream) { c -> stream = (hipStream_t)stream; c -> own_stream = false; }
  else {
    if(hipStreamCreateWithFlags(& c -> stream, hipStreamNonBlocking) != hipSuccess) {
      llsm_set_error("hipStreamCreate failed"); delete c; return nullptr;
    }
    c -> own_stream = true;
  }
  // twiddle table e^{-2 pi i m / 8192}, m < 4096, rounded from float64
  c -> tw_nmax = 8192;
  std::vector<float2> tw(c -> tw_nmax / 2);
  for(int m = 0; m < c -> tw_nmax / 2; m ++) {
    double a = 2.0 * 3.14159265358979323846 * m / c -> tw_nmax;
    tw[m] = make_float2((float)std::cos(a), (float)-std::sin(a));
  }
  if(hipMalloc(& c -> tw, tw.size() * sizeof(float2)) != hipSuccess ||
     hipMemcpy(c -> tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
    llsm_set_error("twiddle table allocation failed"); delete c; return nullptr;
  }
  if(push_conventions()) { llsm_set_error("convention upload failed"); delete c; return nullptr; }
  c -> lc.stream = c -> stream;
  c -> lc.prof_begin = nullptr; c -> lc.prof_end = nullptr; c -> lc.prof_user = c;
  return c;
}

extern "C" void llsm_gpu_delete_context(llsm_gpu_context* c) {
  if(! c) return;
  hipSetDevice(c -> device);
  hipStreamSynchronize(c -> stream);
  prof_drain(c);
  for(auto e : c -> pool) hipEventDestroy(e);
  hipFree(c -> tw);
  if(c -> lc.tw_big) hipFree((void*)c -> lc.tw_big);
  if(c -> lc.big_scratch) hipFree(c -> lc.big_scratch);
  if(c -> sections) hipFree(c -> sections);
  if(c -> aux) { hipStreamSynchronize(c -> aux); hipStreamDestroy(c -> aux); hipEventDestroy(c -> ev_fork); hipEventDestroy(c -> ev_join); }
  if(c -> own_stream) hipStreamDestroy(c -> stream);
  delete c;
}
extern "C" void* llsm_gpu_context_stream(llsm_gpu_context* c) { return c ? (void*)c -> stream : nullptr; }
extern "C" int llsm_gpu_synchronize(llsm_gpu_context* c) {
  if(! c) { llsm_set_error("llsm_gpu_synchronize: no context"); return -1; }
  hipSetDevice(c -> device);
  HIP_OK(hipStreamSynchronize(c -> stream));
  return 0;
}
extern "C" int llsm_gpu_fft_selftest(llsm_gpu_context* c, int logn, int count, int inverse,
  const float* in, float* out) {
  if(! c || logn < 8 || logn > 12 || count <= 0) { llsm_set_error("fft_selftest: bad arguments"); return -1; }
  hipSetDevice(c -> device);
  const size_t bytes = sizeof(float2) * ((size_t)count << logn);
  float2 *din = nullptr, *dout = nullptr;
  HIP_OK(hipMalloc(& din, bytes));
  if(hipMalloc(& dout, bytes) != hipSuccess) { hipFree(din); llsm_set_error("fft_selftest: out of memory"); return -1; }
  int rc = 0;
  if(hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, c -> stream) != hipSuccess) rc = -1;
  if(rc == 0 && launch_wf_selftest(& c -> lc, logn, din, dout, count, inverse) != 0) rc = -1;
  if(rc == 0 && hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c -> stream) != hipSuccess) rc = -1;
  if(hipStreamSynchronize(c -> stream) != hipSuccess) rc = -1;
  hipFree(din); hipFree(dout);
  if(rc) llsm_set_error("fft_selftest: HIP failure");
  return rc;
}
extern "C" int llsm_gpu_set_profiling(llsm_gpu_context* c, int enabled) {
  prof_drain(c);
  c -> profiling = enabled != 0;
  c -> lc.prof_begin = enabled ? prof_begin_cb : nullptr;
  c -> lc.prof_end = enabled ? prof_end_cb : nullptr;
  if(enabled != 2) c -> prof_only.clear();              // 2: keep the single-kernel filter of llsm_gpu_profile_only
  return 0;
}
extern "C" int llsm_gpu_profile_only(llsm_gpu_context* c, const char* kernel_name) {
  prof_drain(c);
  c -> prof_only = kernel_name ? kernel_name : "";
  return 0;
}
extern "C" int llsm_gpu_reset_profile(llsm_gpu_context* c) {
  prof_drain(c); c -> prof.clear(); return 0;
}
extern "C" int llsm_gpu_get_profile(llsm_gpu_context* c, int cap, const char** names,
  double* total_ms, int* launches) {
  prof_drain(c);
  c -> prof_names.clear();
  for(auto& kv : c -> prof) c -> prof_names.push_back(kv.first);
  int i = 0;
  for(auto& kv : c -> prof) {
    if(i < cap) {
      if(names) names[i] = c -> prof_names[i].c_str();
      if(total_ms) total_ms[i] = kv.second.ms;
      if(launches) launches[i] = kv.second.launches;
    }
    i ++;
  }
  return (int)c -> prof.size();
}

// ------------------------------------------------------- device allocation cache
// The drop-in entry points (llsm_analyze / llsm_synthesize) create and destroy a batch per call:
// ~30 hipMalloc + hipFree pairs, i.e. 2-3 ms of a 6 ms call.  Freed blocks are therefore kept,
// per device and bucketed by size, and handed back to the next batch of the same shape.  A block
// only enters the cache after the stream that used it has been synchronised (delete_batch does;
// the regrow path of DevBuf synchronises the device first), so reuse by any stream is safe.
// LLSM_GPU_POOL_MB bounds the cached bytes per device (default 8192, 0 disables the cache).
namespace {
struct DevPool {
  std::mutex m;
  std::multimap<size_t, void*> free_;
  std::map<void*, size_t> size_;                      // live and cached blocks -> bucket size
  size_t cached = 0;
};
DevPool g_pool[64];
size_t pool_cap() {
  static size_t cap = [] {
    const char* e = std::getenv("LLSM_GPU_POOL_MB");
    return (size_t)(e ? std::strtoull(e, nullptr, 10) : 8192) << 20;
  }();
  return cap;
}
size_t pool_bucket(size_t bytes) {
  if(bytes <= 4096) return 4096;
  if(bytes < ((size_t)1 << 21)) { size_t b = 4096; while(b < bytes) b <<= 1; return b; }
  return (bytes + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1);   // multiples of 2 MiB
}
DevPool& pool_here() { int d = 0; hipGetDevice(& d); return g_pool[(d < 0 || d >= 64) ? 0 : d]; }
}  // namespace

hipError_t llsm_dev_malloc(void** p, size_t bytes) {
  DevPool& P = pool_here();
  const size_t b = pool_bucket(bytes);
  {
    std::lock_guard<std::mutex> lock(P.m);
    auto it = P.free_.find(b);
    if(it != P.free_.end()) { *p = it -> second; P.free_.erase(it); P.cached -= b; return hipSuccess; }
  }
  hipError_t e = hipMalloc(p, b);
  if(e != hipSuccess) {                               // out of memory: drop the cache and retry once
    llsm_gpu_release_cached_memory();
    e = hipMalloc(p, b);
    if(e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lock(P.m);
  P.size_[*p] = b;
  return hipSuccess;
}
void llsm_dev_free(void* p) {
  if(! p) return;
  DevPool& P = pool_here();
  {
    std::lock_guard<std::mutex> lock(P.m);
    auto it = P.size_.find(p);
    if(it != P.size_.end() && P.cached + it -> second <= pool_cap()) {
      P.free_.emplace(it -> second, p); P.cached += it -> second; return;
    }
    if(it != P.size_.end()) P.size_.erase(it);
  }
  hipFree(p);
}
extern "C" void llsm_gpu_release_cached_memory(void) {
  DevPool& P = pool_here();
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> lock(P.m);
    for(auto& kv : P.free_) { drop.push_back(kv.second); P.size_.erase(kv.second); }
    P.free_.clear(); P.cached = 0;
  }
  for(void* q : drop) hipFree(q);
}

static int ilog2(int n) { int l = 0; while((1 << l) < n) l ++; return l; }

// overlap-add Hann windows: symmetric (denominator n - 1) unless the hann_periodic convention is set
// Hann window, symmetric (den = n - 1) or periodic (den = n) by convention; w[i] == w[den - i] EXACTLY (the second half
// mirrors the first: cos(2 pi i / den) and cos(2 pi (den - i) / den) may round apart in the last place), which lets a
// kernel keep half the table (k_noise_filter_ola)
static int hann_sym(int n) { return n <= 1 ? 0 : (g_hconv.hann_periodic ? n : n - 1); }
static std::vector<float> make_hann(int n) {
  std::vector<float> w(n);
  const int den = g_hconv.hann_periodic ? n : n - 1;
  for(int i = 0; i < n; i ++)
    w[i] = n == 1 ? 1.0f : (float)(0.5 - 0.5 * std::cos(2.0 * 3.14159265358979323846 * (i <= den - i ? i : den - i) / den));
  return w;
}
static std::vector<float> make_blackman(int n) {
  std::vector<float> w(n);
  for(int i = 0; i < n; i ++) {
    double t = 2.0 * 3.14159265358979323846 * i / (n - 1);
    w[i] = n == 1 ? 1.0f : (float)(0.42 - 0.5 * std::cos(t) + 0.08 * std::cos(2.0 * t));
  }
  return w;
}
// shared-F0 tile kernels (llsm_gpu.h llsm_gpu_shared_f0_tiles): default from $LLSM_GPU_F0_TILES, else on
static std::atomic<int> g_f0_tiles([] { const char* e = std::getenv("LLSM_GPU_F0_TILES"); return e ? (std::atoi(e) > 0 ? 1 : 0) : 1; }());
extern "C" int llsm_gpu_shared_f0_tiles(int on) { if(on >= 0) g_config_epoch.fetch_add(1); return on < 0 ? g_f0_tiles.load() : g_f0_tiles.exchange(on > 0 ? 1 : 0); }
// shared phasor tables of the harmonic resynthesis (k_synth_ola4; llsm_gpu.h llsm_gpu_synth_tables): default from
// $LLSM_GPU_SYNTH_TABLES, else on.  Results are bit-identical either way.
static std::atomic<int> g_synth_tables([] { const char* e = std::getenv("LLSM_GPU_SYNTH_TABLES"); return e ? (std::atoi(e) > 0 ? 1 : 0) : 1; }());
extern "C" int llsm_gpu_synth_tables(int on) { if(on >= 0) g_config_epoch.fetch_add(1); return on < 0 ? g_synth_tables.load() : g_synth_tables.exchange(on > 0 ? 1 : 0); }

static BatchDev batch_dev(llsm_gpu_batch* b, float fs) {
  BatchDev d;
  d.n_utt = b -> lay.n_utt; d.nframes = b -> lay.total_frames;
  d.maxnhar = b -> lay.maxnhar; d.maxnhar_e = b -> lay.maxnhar_e;
  d.npsd = b -> lay.npsd; d.nchannel = b -> lay.nchannel;
  d.thop = b -> opt.thop; d.fs = fs; d.rel_winsize = b -> opt.rel_winsize;
  d.x_off = b -> d_x_off.p; d.nx = b -> d_nx.p; d.frm_off = b -> d_frm_off.p; d.nfrm = b -> d_nfrm.p;
  d.frm_utt = b -> d_frm_utt.p;
  d.f0 = (float*)b -> arr[LLSM_GPU_F0]; d.nhar = (int*)b -> arr[LLSM_GPU_NHAR];
  d.ampl = (float*)b -> arr[LLSM_GPU_AMPL]; d.phse = (float*)b -> arr[LLSM_GPU_PHSE];
  d.psd = (float*)b -> arr[LLSM_GPU_PSD]; d.psdres = (float*)b -> arr[LLSM_GPU_PSDRES];
  d.has_psdres = (int*)b -> arr[LLSM_GPU_HAS_PSDRES];
  d.edc = (float*)b -> arr[LLSM_GPU_EDC]; d.nhar_e = (int*)b -> arr[LLSM_GPU_NHAR_E];
  d.eenv_ampl = (float*)b -> arr[LLSM_GPU_EENV_AMPL]; d.eenv_phse = (float*)b -> arr[LLSM_GPU_EENV_PHSE];
  d.x = (const float*)b -> arr[LLSM_GPU_X];
  d.pairs = b -> npairs > 0 ? b -> d_pairs.p : nullptr; d.npairs = b -> npairs;
  const bool tiles = g_f0_tiles.load() > 0;
  d.hblocks = tiles && b -> nhblocks > 0 ? b -> d_hblocks.p : nullptr; d.nhblocks = tiles ? b -> nhblocks : 0;
  d.synth_tables = g_synth_tables.load() > 0 ? 1 : 0;
  return d;
}

// Band plan of channel c (layer0.c:434-436, 541-543) -> filter chain of
// chebyfilt (dsputils.c:51-70): returns number of sections (1 or 2).
static int channel_chain(const llsm_gpu_batch* b, float fs, int c, bool* hp0, float* cut0,
  bool* hp1, float* cut1, bool* from_x) {
  int nch = b -> lay.nchannel;
  float fmin = c == 0 ? 0.0f : b -> chanfreq[c - 1];
  float fmax = c == nch - 1 ? (float)(fs / 2.0) : b -> chanfreq[c];
  *from_x = fmin > 6000.0;
  float c1 = fmin / fs, c2 = fmax / fs;
  if(c1 < 0) c1 = 0;
  if(c2 > 0.5f) c2 = 0.5f;
  if(c1 != 0 && c2 < 0.5f) { *hp0 = true; *cut0 = c1; *hp1 = false; *cut1 = c2; return 2; }
  if(c1 == 0) { *hp0 = false; *cut0 = c2; return 1; }
  *hp0 = true; *cut0 = c1; return 1;
}
// the parameter rows that travel between host and device as one block (batch.h pblock), in block order
static const int kParamIds[11] = {LLSM_GPU_F0, LLSM_GPU_NHAR, LLSM_GPU_AMPL, LLSM_GPU_PHSE, LLSM_GPU_PSD, LLSM_GPU_PSDRES, LLSM_GPU_HAS_PSDRES,
                                  LLSM_GPU_EDC, LLSM_GPU_NHAR_E, LLSM_GPU_EENV_AMPL, LLSM_GPU_EENV_PHSE};


extern "C" llsm_gpu_batch* llsm_gpu_create_batch(llsm_gpu_context* ctx,
  const llsm_aoptions* options, FP_TYPE fs, int n_utt, const int* nx, const int* nfrm) {
  if(! ctx || ! options || n_utt < 0 || n_utt > 65535) {
    llsm_set_error("llsm_gpu_create_batch: bad arguments (n_utt must be 0..65535)"); return nullptr;
  }
  if(options -> nchannel < 1 || options -> nchannel > 8 || options -> maxnhar_e > 8 ||
     options -> maxnhar_e < 0 || options -> maxnhar < 1 || options -> npsd < 2) {
    llsm_set_error("unsupported options: need 1<=nchannel<=8, 0<=maxnhar_e<=8, maxnhar>=1, npsd>=2");
    return nullptr;
  }
  if(n_utt > 0 && (! nx || ! nfrm)) { llsm_set_error("llsm_gpu_create_batch: nx / nfrm missing"); return nullptr; }
  for(int u = 0; u < n_utt; u ++) {
    if(nx[u] < 0 || nfrm[u] < 0) {
      llsm_set_error("llsm_gpu_create_batch: negative sample or frame count"); return nullptr;
    }
    if(nx[u] >= (1 << 29)) {                          // 32-bit byte offsets of the range-checked buffer loads
      llsm_set_error("llsm_gpu_create_batch: utterance longer than 2^29 samples"); return nullptr;
    }
  }
  if(options -> nchannel > 1 && ! options -> chanfreq) {
    llsm_set_error("llsm_gpu_create_batch: chanfreq missing"); return nullptr;
  }
  if(! (fs > 0) || ! (options -> thop > 0)) {
    llsm_set_error("llsm_gpu_create_batch: fs and thop must be positive"); return nullptr;
  }
  hipSetDevice(ctx -> device);
  llsm_gpu_batch* b = new llsm_gpu_batch();
  b -> ctx = ctx; b -> fs = fs; b -> fnyq = (float)(fs / 2.0); b -> opt = *options;
  if(options -> nchannel > 1)
    b -> chanfreq.assign(options -> chanfreq, options -> chanfreq + (options -> nchannel - 1));
  b -> opt.chanfreq = b -> chanfreq.data();
  std::memset(b -> arr, 0, sizeof(b -> arr)); std::memset(b -> arr_bytes, 0, sizeof(b -> arr_bytes));
  const float thop = options -> thop;
  long long X = 0, F = 0, Y = 0;
  for(int u = 0; u < n_utt; u ++) {
    b -> nx.push_back(nx[u]); b -> nfrm.push_back(nfrm[u]);
    int ny = lp::ny(nfrm[u], thop, fs);
    if(ny < 0 || ny >= (1 << 29)) { llsm_set_error("llsm_gpu_create_batch: utterance longer than 2^29 samples"); delete b; return nullptr; }
    b -> ny.push_back(ny);
    b -> x_off.push_back((int)X); b -> frm_off.push_back((int)F); b -> y_off.push_back((int)Y);
    X += nx[u]; F += nfrm[u]; Y += ny;
    b -> max_nx = std::max(b -> max_nx, nx[u]); b -> max_ny = std::max(b -> max_ny, ny);
  }
  b -> x_off.push_back((int)X); b -> frm_off.push_back((int)F); b -> y_off.push_back((int)Y);
  if(X > 0x7fffffffLL || Y > 0x7fffffffLL || F * (long long)std::max(options -> maxnhar, 1) > 0x7fffffffffLL) {
    llsm_set_error("batch too large for 32-bit sample offsets"); delete b; return nullptr;
  }
  llsm_gpu_layout& L = b -> lay;
  L.n_utt = n_utt; L.total_samples = (int)X; L.total_frames = (int)F; L.total_out = (int)Y;
  L.maxnhar = options -> maxnhar; L.maxnhar_e = options -> maxnhar_e;
  L.npsd = options -> npsd; L.nchannel = options -> nchannel;
  L.ntemplate_ext = 20000 + 128;
  b -> nwin_sin = lp::nwin_sin(thop, fs);
  b -> nwin_psd = lp::nwin_psd(thop, fs);
  b -> nfft_psd = lp::nextpow2(b -> nwin_psd);
  b -> nfft_spgm = lp::nextpow2(0.03 * fs);
  b -> nspec = b -> nfft_psd / 2 + 1;
  // the PSD transform (4 hops) may exceed the LDS kernels: global-scratch path up to 2^17 points (a 25 ms hop at 96 kHz
  // needs 16384); the spectrogram transform (30 ms) stays with the LDS / register kernels (8192 points: fs <= 273 kHz)
  if(b -> nfft_spgm > ctx -> tw_nmax || b -> nfft_psd > LLSM_BIG_FFT_MAX || b -> nfft_psd < 64 || b -> nfft_spgm < 64) {
    llsm_set_error("FFT size outside the supported range (spectrogram [64, 8192], PSD [64, 131072])"); delete b; return nullptr;
  }
  const size_t Fz = (size_t)F, nch = L.nchannel, me = std::max(L.maxnhar_e, 1);
  size_t sizes[LLSM_GPU_NARRAYS]; std::memset(sizes, 0, sizeof(sizes));   // layer-1 arrays: on demand (l1.cpp)
  sizes[LLSM_GPU_X] = X * sizeof(float); sizes[LLSM_GPU_F0] = Fz * sizeof(float);
  sizes[LLSM_GPU_NHAR] = Fz * sizeof(int);
  sizes[LLSM_GPU_AMPL] = sizes[LLSM_GPU_PHSE] = Fz * L.maxnhar * sizeof(float);
  sizes[LLSM_GPU_PSD] = sizes[LLSM_GPU_PSDRES] = Fz * L.npsd * sizeof(float);
  sizes[LLSM_GPU_EDC] = Fz * nch * sizeof(float); sizes[LLSM_GPU_NHAR_E] = Fz * sizeof(int);
  sizes[LLSM_GPU_EENV_AMPL] = sizes[LLSM_GPU_EENV_PHSE] = Fz * nch * me * sizeof(float);
  sizes[LLSM_GPU_XRES] = X * sizeof(float);
  sizes[LLSM_GPU_Y] = sizes[LLSM_GPU_YSIN] = sizes[LLSM_GPU_YNOISE] = Y * sizeof(float);
  sizes[LLSM_GPU_WHITE] = (size_t)n_utt * nch * L.ntemplate_ext * sizeof(float);
  sizes[LLSM_GPU_HAS_PSDRES] = Fz * sizeof(int);
  // the parameter rows as pieces of one block (256-byte aligned pieces, kParamIds order)
  bool in_block[LLSM_GPU_NARRAYS]; std::memset(in_block, 0, sizeof(in_block));
  {
    size_t at = 0;
    for(int k = 0; k < 11; k ++) { b -> pblock_off[k] = at; at += (sizes[kParamIds[k]] + 255) & ~(size_t)255; in_block[kParamIds[k]] = true; }
    b -> pblock_bytes = at;
    if(at > 0) {
      hipError_t e = llsm_dev_malloc(& b -> pblock, at);
      if(e != hipSuccess || hipMemsetAsync(b -> pblock, 0, at, ctx -> stream) != hipSuccess) {
        llsm_set_error(std::string("hipMalloc(batch parameter block): ") + hipGetErrorString(e));
        llsm_gpu_delete_batch(b); return nullptr;
      }
      for(int k = 0; k < 11; k ++) b -> arr[kParamIds[k]] = sizes[kParamIds[k]] ? (char*)b -> pblock + b -> pblock_off[k] : nullptr;
    }
  }
  for(int a = 0; a < LLSM_GPU_NARRAYS; a ++) {
    b -> arr_bytes[a] = sizes[a];
    if(sizes[a] == 0 || in_block[a]) continue;
    hipError_t e = llsm_dev_malloc(& b -> arr[a], sizes[a]);
    if(e != hipSuccess) {
      llsm_set_error(std::string("hipMalloc(batch array): ") + hipGetErrorString(e));
      llsm_gpu_delete_batch(b); return nullptr;
    }
    if(hipMemsetAsync(b -> arr[a], 0, sizes[a], ctx -> stream) != hipSuccess) {
      llsm_set_error("hipMemsetAsync(batch array) failed"); llsm_gpu_delete_batch(b); return nullptr;
    }
  }
  std::vector<int> frm_utt(Fz);
  for(int u = 0; u < n_utt; u ++)
    for(int i = 0; i < nfrm[u]; i ++) frm_utt[(size_t)b -> frm_off[u] + i] = u;
  int bad = 0;
  bad |= upload_vec(b -> d_nx, b -> nx); bad |= upload_vec(b -> d_nfrm, b -> nfrm);
  bad |= upload_vec(b -> d_ny, b -> ny); bad |= upload_vec(b -> d_x_off, b -> x_off);
  bad |= upload_vec(b -> d_frm_off, b -> frm_off); bad |= upload_vec(b -> d_y_off, b -> y_off);
  bad |= upload_vec(b -> d_frm_utt, frm_utt);
  {
    // frames sharing one complex transform: neighbours of ONE utterance, its odd last frame alone
    std::vector<int2> pairs;
    pairs.reserve(Fz / 2 + n_utt);
    for(int u = 0; u < n_utt; u ++)
      for(int i = 0; i < nfrm[u]; i += 2) {
        const int g = b -> frm_off[u] + i;
        pairs.push_back(make_int2(g, i + 1 < nfrm[u] ? g + 1 : -1));
      }
    b -> npairs = (int)pairs.size();
    if(! pairs.empty()) bad |= upload_vec(b -> d_pairs, pairs);
    // 16-aligned blocks of an utterance's frames: the units of the shared-F0 tile kernel (k_harm_speech_tile)
    std::vector<int2> blocks;
    blocks.reserve(Fz / 16 + 2 * (size_t)n_utt);
    for(int u = 0; u < n_utt; u ++)
      for(int i = 0; i < nfrm[u]; i += 16) blocks.push_back(make_int2(b -> frm_off[u] + i, std::min(16, nfrm[u] - i)));
    b -> nhblocks = (int)blocks.size();
    if(! blocks.empty()) bad |= upload_vec(b -> d_hblocks, blocks);
  }
  // batch-constant windows and normalisers (rounded from float64)
  bad |= upload_vec(b -> win_sin, make_hann(b -> nwin_sin));
  std::vector<float> wb = make_blackman(b -> nwin_psd);
  double wp = 0; for(float v : wb) wp += (double)v * v;
  b -> inv_wpow = (float)(1.0 / wp);
  bad |= upload_vec(b -> win_psd, wb);
  { double s = 0;                                    // spectrogram normaliser: symmetric Hann(1024), like the device window
    for(int i = 0; i < 1024; i ++) s += (double)(float)(0.5 - 0.5 * std::cos(2.0 * 3.14159265358979323846 * i / 1023.0));
    b -> norm_base = (float)(1024.0 / (0.5 * s)); }
  { std::vector<float> h = make_blackman(1024); double s = 0; for(float v : h) s += v;
    b -> norm_base_blackman = (float)(1024.0 / (0.5 * s)); }
  {
    // work units of k_synth_ola: frames [i0, i1) of one utterance; about 8 wavefronts / SIMD over
    // the batch, never below 4 frames; halo as for the noise frames (window of nwin_sin samples)
    const int udiv = std::min(SYN_UNIT_DIV, synth_ola_unit_div(b -> nwin_sin, std::min(L.maxnhar, 2048)));
    int C = std::max(4, (int)((F + udiv - 1) / udiv));
    if(const char* e = std::getenv("LLSM_GPU_SIN_UNIT")) C = std::max(1, std::atoi(e));   // tuning override
    std::vector<int4> units;
    for(int u = 0; u < n_utt; u ++) {
      int nu = (nfrm[u] + C - 1) / C;                // equal units within the utterance (no short remainder unit)
      // whole groups where the utterance is long enough for it (5 units of 40 frames would cost 3 idle wavefronts:
      // 8 units of 25 instead), units never shorter than 4 frames (2 halo frames are recomputed per unit)
      const int G = synth_ola_group_units();
      if(nu % G && nfrm[u] >= 4 * ((nu + G - 1) / G * G)) nu = (nu + G - 1) / G * G;
      const int sz = nu > 0 ? (nfrm[u] + nu - 1) / nu : 1;
      for(int i0 = 0; i0 < nfrm[u]; i0 += sz) units.push_back(make_int4(u, i0, std::min(i0 + sz, nfrm[u]), 0));
      if(nfrm[u] == 0) units.push_back(make_int4(u, 0, 0, 0));   // frameless utterance: x_res = x, y_sin = 0
      // groups of synth_ola_group_units() (four) units never straddle utterances (k_synth_ola4: one workgroup, one phasor table per group);
      // padding units (w = 1) do nothing
      while(units.size() % (size_t)synth_ola_group_units()) units.push_back(make_int4(u, 0, 0, 1));
    }
    b -> n_sin_units = (int)units.size();
    b -> sin_halo = (int)std::floor((b -> nwin_sin + 1) / std::max((double)thop * fs, 1.0));
    bad |= upload_vec(b -> sin_units, units);
  }
  // filter sections: index 2*row + highpass
  std::vector<FiltSectionD> secs(2 * llsm_cheby::kRows);
  for(int r = 0; r < llsm_cheby::kRows; r ++)
    for(int hp = 0; hp < 2; hp ++) {
      llsm_cheby::Section s = llsm_cheby::make_section_row(r, hp != 0);
      FiltSectionD& d = secs[2 * r + hp];
      std::memcpy(d.b, s.b, sizeof(d.b)); std::memcpy(d.a, s.a, sizeof(d.a)); std::memcpy(d.zi, s.zi, sizeof(d.zi));
      llsm_cheby::block_tables(s.a, IIR_SEG, 6, & d.H[0][0], & d.M[0][0]);
    }
  bad |= upload_vec(b -> sections, secs);
  if(bad) { llsm_gpu_delete_batch(b); return nullptr; }
  return b;
}

extern "C" void llsm_gpu_delete_batch(llsm_gpu_batch* b) {
  if(! b) return;
  hipSetDevice(b -> ctx -> device);
  hipStreamSynchronize(b -> ctx -> stream);
  if(b -> ctx -> aux) hipStreamSynchronize(b -> ctx -> aux);       // the analysis' second stream (normally joined already)
  for(int k = 0; k < 11; k ++) if(b -> pblock) b -> arr[kParamIds[k]] = nullptr;        // pieces of pblock, not allocations
  llsm_dev_free(b -> pblock); b -> pblock = nullptr;
  for(int a = 0; a < LLSM_GPU_NARRAYS; a ++) llsm_dev_free(b -> arr[a]);
  b -> d_nx.release(); b -> d_nfrm.release(); b -> d_ny.release(); b -> d_x_off.release();
  b -> d_frm_off.release(); b -> d_y_off.release(); b -> d_frm_utt.release(); b -> d_pairs.release(); b -> d_hblocks.release();
  b -> packed.release(); b -> ce.release(); b -> mid.release(); b -> iir_tmp.release(); b -> iir_edge[0].release(); b -> iir_edge[1].release(); b -> iir_seg[0].release(); b -> iir_seg[1].release();
  b -> env.release(); b -> psd_log.release(); b -> pbuf.release(); b -> spgm_fix.release(); b -> spgm_fix_count.release();
  b -> colored.release(); b -> env_cplx.release(); b -> env_hits.release(); b -> env_over.release(); b -> nf_units.release(); b -> sin_units.release(); b -> yexc.release(); b -> nframes.release();
  b -> live.release(); b -> win_sin.release(); b -> win_psd.release(); b -> win_env.release();
  b -> win_filt.release(); b -> nfft_u.release(); b -> sections.release(); b -> jobs_ana.release(); b -> jobs_syn.release();
  b -> l1_model_power.release(); b -> l1_model_param.release(); b -> l1_model_inv_t.release(); b -> l1_model_cumlog_t.release(); b -> l1_rd_raw.release(); b -> l1_cont.release();
  b -> l1_f0_hm.release(); b -> l1_pulse_buf.release(); b -> l1_mixw.release(); b -> l1_hm_frames.release(); b -> l1_zero.release(); b -> l1_src_ampl.release();
  if(b -> blob_stage) { (void)hipHostFree(b -> blob_stage); b -> blob_stage = nullptr; }
  b -> l1_prev.release(); b -> l1_next.release(); b -> l1_blk_off.release(); b -> l1_select.release();
  b -> l1_jobs.release(); b -> l1_pulses.release(); b -> l1_segs.release(); b -> l1_blk_jobs.release();
  b -> l1_proj.release(); b -> l1_alpha.release(); b -> l1_alpha_key.release();
  delete b;
}

extern "C" int llsm_gpu_batch_set_fnyq(llsm_gpu_batch* b, FP_TYPE fnyq) {
  if(! b || !(fnyq > 0)) { llsm_set_error("llsm_gpu_batch_set_fnyq: bad arguments"); return -1; }
  b -> fnyq = fnyq;
  return 0;
}
// Intermediate planes of the last analysis, for diagnosis (tools/psd_bisect.py --product): which = 0 the log envelope that
// sets the Kalman process variance, 1 the log periodogram of the residual; both [total_frames][nspec_psd] float32.
// dst == NULL: only the size.  Returns the number of floats of the plane, -1 without one.
extern "C" long long llsm_gpu_batch_debug_plane(llsm_gpu_batch* b, int which, float* dst, long long cap) {
  if(! b || (which != 0 && which != 1)) { llsm_set_error("llsm_gpu_batch_debug_plane: bad arguments"); return -1; }
  DevBuf<float>& src = which == 0 ? b -> env : b -> psd_log;
  const long long n = (long long)b -> lay.total_frames * (b -> nfft_psd / 2 + 1);
  if(! src.p || n <= 0) { llsm_set_error("llsm_gpu_batch_debug_plane: no analysis has run on this batch"); return -1; }
  if(dst) {
    if(cap < n) { llsm_set_error("llsm_gpu_batch_debug_plane: destination too small"); return -1; }
    if(hipStreamSynchronize(b -> ctx -> stream) != hipSuccess ||
       hipMemcpy(dst, src.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
      llsm_set_error("llsm_gpu_batch_debug_plane: copy failed"); return -1;
    }
  }
  return n;
}
extern "C" int llsm_gpu_batch_layout(llsm_gpu_batch* b, llsm_gpu_layout* dst) { *dst = b -> lay; return 0; }
extern "C" int llsm_gpu_batch_offsets(llsm_gpu_batch* b, int* x_off, int* frm_off, int* y_off) {
  size_t n = (size_t)b -> lay.n_utt + 1;
  if(x_off) std::memcpy(x_off, b -> x_off.data(), n * sizeof(int));
  if(frm_off) std::memcpy(frm_off, b -> frm_off.data(), n * sizeof(int));
  if(y_off) std::memcpy(y_off, b -> y_off.data(), n * sizeof(int));
  return 0;
}
extern "C" void* llsm_gpu_batch_device_ptr(llsm_gpu_batch* b, int id) {
  // whoever takes the F0 row's device address may rewrite it behind the library's back: the lowest F0 seen by
  // llsm_gpu_batch_upload no longer describes the batch (0 = unknown: every F0-sized LDS provision falls back to its
  // maximum, so that which kernel a frame takes never depends on a stale value -- ADVICE r3)
  if(b && id == LLSM_GPU_F0) { b -> min_f0 = 0; b -> f0_unknown = true; }
  return (id >= 0 && id < LLSM_GPU_NARRAYS) ? b -> arr[id] : nullptr;
}
extern "C" size_t llsm_gpu_batch_array_bytes(llsm_gpu_batch* b, int id) {
  return (id >= 0 && id < LLSM_GPU_NARRAYS) ? b -> arr_bytes[id] : 0;
}
extern "C" int llsm_gpu_batch_upload(llsm_gpu_batch* b, int id, const void* src, size_t bytes) {
  if(id < 0 || id >= LLSM_GPU_NARRAYS || bytes != b -> arr_bytes[id]) {
    llsm_set_error("llsm_gpu_batch_upload: array id / byte count mismatch"); return -1;
  }
  hipSetDevice(b -> ctx -> device);
  if(bytes == 0) return 0;
  if(id == LLSM_GPU_F0) {
    const float* f = (const float*)src; float m = 0;
    for(size_t i = 0; i < bytes / sizeof(float); i ++) if(f[i] > 0 && (m == 0 || f[i] < m)) m = f[i];
    b -> min_f0 = m; b -> f0_unknown = false;          // a full row: the batch's lowest F0 is known again
  }
  HIP_OK(hipMemcpyAsync(b -> arr[id], src, bytes, hipMemcpyHostToDevice, b -> ctx -> stream));
  HIP_OK(hipStreamSynchronize(b -> ctx -> stream));   // src is pageable host memory
  return 0;
}
// Page-locked host memory for upload / download buffers: copies from ordinary (pageable) memory
// are staged by the runtime and reach a fraction of the PCIe rate (measured 11 GB/s for the
// 1.16 GB of results of the bench batch; tools/bench_pcie.py).
extern "C" void* llsm_gpu_alloc_host(size_t bytes) {
  void* p = nullptr;
  if(hipHostMalloc(& p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
    llsm_set_error("llsm_gpu_alloc_host: hipHostMalloc failed"); return nullptr;
  }
  return p;
}
extern "C" void llsm_gpu_free_host(void* p) { if(p) hipHostFree(p); }

// ---- host-side placement: the NUMA node of a device and the threads that feed it ----
// The staging threads of the fan-out (capi.cpp) and whoever fills / drains page-locked blocks move ~1.3 GB per bench batch
// through host memory; on a two-socket box the far socket's memory costs a hop over the socket link on top of the PCIe
// link.  The node comes from sysfs (`numa_node` of the PCI device; -1 = the platform does not say), the CPUs from
// /sys/devices/system/node/node<N>/cpulist, intersected with what the process may use (cgroup / taskset).
static int pci_numa_node(int physical_device) {
  char bdf[64] = {0};
  if(hipDeviceGetPCIBusId(bdf, sizeof(bdf), physical_device) != hipSuccess) return -1;
  for(char* c = bdf; *c; c ++) *c = (char)std::tolower((unsigned char)*c);
  std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
  FILE* f = std::fopen(path.c_str(), "r");
  if(! f) return -1;
  int node = -1;
  if(std::fscanf(f, "%d", & node) != 1) node = -1;
  std::fclose(f);
  return node;
}
extern "C" int llsm_gpu_device_numa_node(int device) {
  int n = 0;
  if(hipGetDeviceCount(& n) != hipSuccess || n <= 0 || device < 0) return -1;
  return pci_numa_node(device % n);
}
static bool node_cpus(int node, cpu_set_t* out) {       // "0-15,64-79" -> set
  CPU_ZERO(out);
  char path[128]; std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = std::fopen(path, "r");
  if(! f) return false;
  char buf[4096] = {0};
  const bool got = std::fgets(buf, sizeof(buf), f) != nullptr;
  std::fclose(f);
  if(! got) return false;
  int any = 0;
  for(char* tok = std::strtok(buf, ",\n"); tok; tok = std::strtok(nullptr, ",\n")) {
    int a = 0, b2 = 0;
    const int k = std::sscanf(tok, "%d-%d", & a, & b2);
    if(k < 1) continue;
    if(k == 1) b2 = a;
    for(int c = a; c <= b2 && c < CPU_SETSIZE; c ++) { CPU_SET(c, out); any ++; }
  }
  return any > 0;
}
// Binds the CALLING thread to the CPUs of the device's NUMA node (those of them the thread may already use).  Returns the
// number of CPUs in the new mask, 0 when nothing was changed (node unknown, no overlap with the allowed CPUs, or
// $LLSM_GPU_NUMA_BIND=0).  Memory the thread touches first -- and page-locked blocks it allocates -- then land on that node.
extern "C" int llsm_gpu_bind_thread_to_device(int device) {
  static const bool off = [] { const char* e = std::getenv("LLSM_GPU_NUMA_BIND"); return e && e[0] == '0'; }();
  if(off) return 0;
  const int node = llsm_gpu_device_numa_node(device);
  if(node < 0) return 0;
  cpu_set_t want, have, both;
  if(! node_cpus(node, & want) || sched_getaffinity(0, sizeof(have), & have) != 0) return 0;
  CPU_AND(& both, & want, & have);
  const int n = CPU_COUNT(& both);
  if(n <= 0 || n == CPU_COUNT(& have)) return n == CPU_COUNT(& have) ? n : 0;   // (already inside the node: nothing to do)
  return sched_setaffinity(0, sizeof(both), & both) == 0 ? n : 0;
}

extern "C" int llsm_gpu_batch_download(llsm_gpu_batch* b, int id, void* dst, size_t bytes) {
  if(id < 0 || id >= LLSM_GPU_NARRAYS || bytes != b -> arr_bytes[id]) {
    llsm_set_error("llsm_gpu_batch_download: array id / byte count mismatch"); return -1;
  }
  hipSetDevice(b -> ctx -> device);
  if(bytes == 0) return 0;
  HIP_OK(hipMemcpyAsync(dst, b -> arr[id], bytes, hipMemcpyDeviceToHost, b -> ctx -> stream));
  HIP_OK(hipStreamSynchronize(b -> ctx -> stream));
  return 0;
}

// The parameter rows as one block (batch.h pblock): its size and the offsets of the eleven pieces, and the block moved in ONE
// copy.  host: a buffer of *total bytes laid out with the same offsets (page-locked for full link speed).
extern "C" int llsm_gpu_batch_params_layout(llsm_gpu_batch* b, size_t* total, size_t* offsets11, int* ids11) {
  if(total) *total = b -> pblock_bytes;
  for(int k = 0; k < 11; k ++) { if(offsets11) offsets11[k] = b -> pblock_off[k]; if(ids11) ids11[k] = kParamIds[k]; }
  return 0;
}
extern "C" int llsm_gpu_batch_transfer_params(llsm_gpu_batch* b, int to_device, void* host) {
  hipSetDevice(b -> ctx -> device);
  if(b -> pblock_bytes == 0) return 0;
  if(to_device) {
    const float* f = (const float*)((const char*)host + b -> pblock_off[0]); float m = 0;   // F0 is piece 0
    for(size_t i = 0; i < b -> arr_bytes[LLSM_GPU_F0] / sizeof(float); i ++) if(f[i] > 0 && (m == 0 || f[i] < m)) m = f[i];
    b -> min_f0 = m; b -> f0_unknown = false;
    HIP_OK(hipMemcpyAsync(b -> pblock, host, b -> pblock_bytes, hipMemcpyHostToDevice, b -> ctx -> stream));
  } else HIP_OK(hipMemcpyAsync(host, b -> pblock, b -> pblock_bytes, hipMemcpyDeviceToHost, b -> ctx -> stream));
  HIP_OK(hipStreamSynchronize(b -> ctx -> stream));
  return 0;
}

// The analysed rows frame-major: one record of llsm_gpu_batch_packed_words() 4-byte words per frame (csrc/packed.h), formed
// on the device (k_pack_frames) and written per UTTERANCE straight into the caller's PAGE-LOCKED blocks -- dst[u] receives
// the nfrm[u] records of utterance u; `dst` itself must be a page-locked table (the kernel reads it) -- so that the host
// can lay the reference's frame objects over the block in place (capi.cpp / model.cpp) instead of re-scattering eleven row
// arrays frame by frame.  No copy engine: a device-to-host copy costs 0.1 - 0.3 ms whatever its size on this stack, so
// one copy per utterance was 10 ms per block of 32 (profiles/r05_h); posted writes from the kernel run at link speed.
// One wait at the end.
extern "C" int llsm_gpu_batch_packed_words(llsm_gpu_batch* b) {
  return llsm_packed_layout(b -> lay.maxnhar, b -> lay.maxnhar_e, b -> lay.npsd, b -> lay.nchannel).words;
}
// ... the same records as ONE block: packed on the device, one copy-engine transfer into `host` (page-locked, F x words x 4
// bytes, utterance u at frm_off[u] x words), one wait.  The copy engines run beside the compute queues, which the kernel
// that writes host memory itself does not (the object path is device-bound: profiles/r05_p_chunk_api_timeline.txt).
extern "C" int llsm_gpu_batch_download_packed_block(llsm_gpu_batch* b, void* host) {
  hipSetDevice(b -> ctx -> device);
  const LlsmPackedLayout PL = llsm_packed_layout(b -> lay.maxnhar, b -> lay.maxnhar_e, b -> lay.npsd, b -> lay.nchannel);
  const size_t F = (size_t)b -> lay.total_frames;
  if(F == 0) return 0;
  if(b -> packed.alloc(F * (size_t)PL.words)) return -1;
  BatchDev d = batch_dev(b, b -> fs);
  if(launch_pack_frames(& b -> ctx -> lc, d, PL, b -> packed.p, nullptr)) { llsm_set_error("k_pack_frames launch failed"); return -1; }
  HIP_OK(hipMemcpyAsync(host, b -> packed.p, F * PL.words * sizeof(float), hipMemcpyDeviceToHost, b -> ctx -> stream));
  HIP_OK(hipStreamSynchronize(b -> ctx -> stream));
  return 0;
}
extern "C" int llsm_gpu_batch_upload_packed_block(llsm_gpu_batch* b, const void* host) {
  hipSetDevice(b -> ctx -> device);
  const LlsmPackedLayout PL = llsm_packed_layout(b -> lay.maxnhar, b -> lay.maxnhar_e, b -> lay.npsd, b -> lay.nchannel);
  const size_t F = (size_t)b -> lay.total_frames;
  if(F == 0) return 0;
  if(b -> packed.alloc(F * (size_t)PL.words)) return -1;
  HIP_OK(hipMemcpyAsync(b -> packed.p, host, F * PL.words * sizeof(float), hipMemcpyHostToDevice, b -> ctx -> stream));
  b -> min_f0 = 0; b -> f0_unknown = true;
  BatchDev d = batch_dev(b, b -> fs);
  if(launch_unpack_frames(& b -> ctx -> lc, d, PL, b -> packed.p, nullptr)) { llsm_set_error("k_unpack_frames launch failed"); return -1; }
  return 0;
}
extern "C" int llsm_gpu_batch_download_packed(llsm_gpu_batch* b, int n_utt, void* const* dst) {
  if(n_utt != b -> lay.n_utt) { llsm_set_error("llsm_gpu_batch_download_packed: utterance count mismatch"); return -1; }
  hipSetDevice(b -> ctx -> device);
  const LlsmPackedLayout PL = llsm_packed_layout(b -> lay.maxnhar, b -> lay.maxnhar_e, b -> lay.npsd, b -> lay.nchannel);
  if(b -> lay.total_frames == 0) return 0;
  BatchDev d = batch_dev(b, b -> fs);
  if(launch_pack_frames(& b -> ctx -> lc, d, PL, nullptr, (float* const*)dst)) { llsm_set_error("k_pack_frames launch failed"); return -1; }
  HIP_OK(hipStreamSynchronize(b -> ctx -> stream));
  return 0;
}
// ... the other direction (llsm_synthesize_batch on chunks whose frames still lie over their records): src[u] = the nfrm[u]
// records of utterance u in page-locked memory (their nhar / nhar_e / has_psdres words refreshed from the structs by the
// caller; `src` a page-locked table), read by k_unpack_frames straight into the parameter rows.  Asynchronous: the
// synthesis launches follow on the same stream; the records must stay valid until the batch is synchronised.
extern "C" int llsm_gpu_batch_upload_packed(llsm_gpu_batch* b, int n_utt, const void* const* src) {
  if(n_utt != b -> lay.n_utt) { llsm_set_error("llsm_gpu_batch_upload_packed: utterance count mismatch"); return -1; }
  hipSetDevice(b -> ctx -> device);
  const LlsmPackedLayout PL = llsm_packed_layout(b -> lay.maxnhar, b -> lay.maxnhar_e, b -> lay.npsd, b -> lay.nchannel);
  if(b -> lay.total_frames == 0) return 0;
  b -> min_f0 = 0; b -> f0_unknown = true;              // (the F0 row did not pass through the host: largest provisions)
  BatchDev d = batch_dev(b, b -> fs);
  if(launch_unpack_frames(& b -> ctx -> lc, d, PL, nullptr, (const float* const*)src)) { llsm_set_error("k_unpack_frames launch failed"); return -1; }
  return 0;
}
// The three waveforms of every utterance straight into the caller's page-locked arrays: tab[3 u + 0 / 1 / 2] = y / y_sin /
// y_noise of utterance u (ny[u] samples each; NULL entries are skipped; `tab` a page-locked table).  One kernel, one wait.
extern "C" int llsm_gpu_batch_download_outputs(llsm_gpu_batch* b, int n_utt, float* const* tab) {
  if(n_utt != b -> lay.n_utt) { llsm_set_error("llsm_gpu_batch_download_outputs: utterance count mismatch"); return -1; }
  hipSetDevice(b -> ctx -> device);
  if(launch_scatter_outputs(& b -> ctx -> lc, n_utt, b -> max_ny, (const float*)b -> arr[LLSM_GPU_Y], (const float*)b -> arr[LLSM_GPU_YSIN],
       (const float*)b -> arr[LLSM_GPU_YNOISE], b -> d_y_off.p, b -> d_ny.p, tab)) { llsm_set_error("k_scatter_outputs launch failed"); return -1; }
  HIP_OK(hipStreamSynchronize(b -> ctx -> stream));
  return 0;
}

// Several arrays of a batch in one go: every copy is enqueued, the stream is waited for ONCE (eleven parameter rows per
// direction and block were eleven round trips to the stream).  The host buffers must stay valid until the call returns;
// page-locked ones (llsm_gpu_alloc_host) make the copies overlap each other on the link.  to_device != 0: upload.
extern "C" int llsm_gpu_batch_transfer_many(llsm_gpu_batch* b, int to_device, int n, const int* ids, void* const* host, const size_t* bytes) {
  hipSetDevice(b -> ctx -> device);
  for(int k = 0; k < n; k ++) {
    const int id = ids[k];
    if(id < 0 || id >= LLSM_GPU_NARRAYS || bytes[k] != b -> arr_bytes[id]) {
      llsm_set_error("llsm_gpu_batch_transfer_many: array id / byte count mismatch"); return -1;
    }
  }
  int any = 0;
  for(int k = 0; k < n; k ++) {
    const int id = ids[k];
    if(bytes[k] == 0) continue;
    if(to_device) {
      if(id == LLSM_GPU_F0) {
        const float* f = (const float*)host[k]; float m = 0;
        for(size_t i = 0; i < bytes[k] / sizeof(float); i ++) if(f[i] > 0 && (m == 0 || f[i] < m)) m = f[i];
        b -> min_f0 = m; b -> f0_unknown = false;
      }
      HIP_OK(hipMemcpyAsync(b -> arr[id], host[k], bytes[k], hipMemcpyHostToDevice, b -> ctx -> stream));
    } else HIP_OK(hipMemcpyAsync(host[k], b -> arr[id], bytes[k], hipMemcpyDeviceToHost, b -> ctx -> stream));
    any = 1;
  }
  if(any) HIP_OK(hipStreamSynchronize(b -> ctx -> stream));
  return 0;
}

#define RUN(call)                                                                      \
  do {                                                                                 \
    int rc_ = (call);                                                                  \
    if(rc_ != 0) {                                                                     \
      llsm_set_error(std::string(#call) + " failed: " +                                \
        (rc_ > 0 ? hipGetErrorString((hipError_t)rc_) : "unsupported configuration")); \
      return -1;                                                                       \
    }                                                                                  \
  } while(0)

// Build the zero-phase filtering jobs of one stage. `which` 0: analysis
// (sub-band energies of x / x_res), 1: synthesis (band-limited templates).
// Samples after which the impulse response of 1 / A(z) stays below tol of its peak: how far a section's start-up
// transient reaches.
static int decay_length(const double* a, double tol) {
  double h[4] = {0, 0, 0, 0}, peak = 1.0; int last = 0;
  double y = 1.0;                                      // h[0] = 1
  for(int t = 1; t < 60000; t ++) {
    const double v = -(a[1] * y + a[2] * h[0] + a[3] * h[1] + a[4] * h[2]);
    h[2] = h[1]; h[1] = h[0]; h[0] = y; y = v;
    if(std::fabs(v) > peak) peak = std::fabs(v);
    if(std::fabs(v) > tol * peak) last = t;
    else if(t - last > 2000) break;
  }
  return last + 1;
}

static int build_jobs(llsm_gpu_batch* b, int which, float fs, const float* xres,
  const float* white) {
  const int U = b -> lay.n_utt, nch = b -> lay.nchannel;
  std::vector<FiltJob> jobs, edge_jobs;
  size_t tmp_off = 0;
  // Few long signals (the drop-in llsm_analyze: ONE utterance, four band signals of 150 000 samples on a chip with 1 024
  // SIMDs; round 6: k_filtfilt 0.69 of the call's 1.23 ms) are cut along TIME: segment s writes its own samples
  // [w0, w1) from a stretch that reaches H samples further on both sides, H = the reach of the slowest pole to 1e-9 + 64
  // -- the rule the two end jobs of a fused band-pass already live by (identical to 1e-13 beyond that reach against
  // scipy); the stretch's own padding and initial state are a transient that has died before w0 / after w1.  Fused
  // band-pass jobs and single-section jobs only (neither uses `mid`); batches with >= 1 024 signals are left alone.
  std::vector<size_t> seg_tmp_off;                       // tmp offsets of the segment jobs inside iir_seg[which], by job index
  std::vector<int> seg_job;                              // ... and which entries of `jobs` they are
  size_t seg_need = 0;
  static const bool seg_ok = [] { const char* e = std::getenv("LLSM_GPU_FILT_SEGMENTS"); return !(e && e[0] == '0'); }();
  static const int seg_force = [] { const char* e = std::getenv("LLSM_GPU_FILT_SEGMENTS"); const int v = e ? std::atoi(e) : 0; return v > 1 ? v : 0; }();   // experiment: this many segments for every job
  auto push_job = [&](const FiltJob& j, int H, int nsignals) {
    const int lo = j.whi > j.wlo ? j.wlo : 0, hi = j.whi > j.wlo ? j.whi : j.n;
    int S = 1;
    if(seg_ok && nsignals < 1024 && H > 0 && !(j.sec1 >= 0 && ! j.fused))
      S = std::min((hi - lo) / std::max(2048, 6 * H), 64);   // (a function of the signal alone: the same cut in every small batch)
    if(seg_force && H > 0 && !(j.sec1 >= 0 && ! j.fused) && (hi - lo) / std::max(2048, 6 * H) >= seg_force) S = seg_force;
    if(S <= 1) { jobs.push_back(j); return; }
    for(int sg = 0; sg < S; sg ++) {
      const int w0 = lo + (int)((long long)(hi - lo) * sg / S), w1 = lo + (int)((long long)(hi - lo) * (sg + 1) / S);
      const int a0 = std::max(0, w0 - H), a1 = std::min(j.n, w1 + H);
      FiltJob e = j;
      e.src = j.src + a0; e.dst = j.dst + a0; e.n = a1 - a0; e.wlo = w0 - a0; e.whi = w1 - a0;
      e.mid = nullptr; e.tmp = nullptr;
      seg_job.push_back((int)jobs.size()); seg_tmp_off.push_back(seg_need); seg_need += (size_t)e.n + 32;
      jobs.push_back(e);
    }
  };
  static const bool fuse_ok = [] { const char* e = std::getenv("LLSM_GPU_FILT_FUSE"); return !(e && e[0] == '0'); }();
  // scratch of the short end jobs of fused band-pass jobs (below): sized in a first pass over the channels
  size_t edge_need = 0;
  std::vector<int> edge_M(nch, 0), edge_Mp(nch, 0);
  for(int c = 0; c < nch && fuse_ok; c ++) {
    bool hp0 = false, hp1 = false, from_x = false; float cut0 = 0, cut1 = 0;
    if(channel_chain(b, fs, c, & hp0, & cut0, & hp1, & cut1, & from_x) != 2) continue;
    const llsm_cheby::Section s0 = llsm_cheby::make_section_row(llsm_cheby::row_of(cut0), hp0);
    const llsm_cheby::Section s1 = llsm_cheby::make_section_row(llsm_cheby::row_of(cut1), hp1);
    const int L = std::max(decay_length(s0.a, 1e-9), decay_length(s1.a, 1e-9));
    edge_M[c] = (L + 31) & ~31; edge_Mp[c] = 2 * edge_M[c] + 64;
    edge_need += (size_t)U * 2 * (2 * (size_t)edge_Mp[c] + 32);
  }
  DevBuf<float>& edge_buf = b -> iir_edge[which];       // one per stage: the job tables keep pointers into it
  if(edge_buf.alloc(edge_need)) return -1;
  size_t edge_off = 0;
  int nact = nch;
  if(which == 1)
    for(int c = 0; c < nch; c ++) {
      float fmin = c == 0 ? 0.0f : b -> chanfreq[c - 1];
      if(fmin >= fs / 2.0) { nact = c; break; }
    }
  for(int c = 0; c < (which == 0 ? nch : nact); c ++) {
    bool hp0 = false, hp1 = false, from_x = false; float cut0 = 0, cut1 = 0;
    int ns = channel_chain(b, fs, c, & hp0, & cut0, & hp1, & cut1, & from_x);
    for(int u = 0; u < U; u ++) {
      FiltJob j;
      if(which == 0) {
        j.n = b -> nx[u];
        const float* base = from_x ? (const float*)b -> arr[LLSM_GPU_X] : xres;
        j.src = base + b -> x_off[u];
        j.dst = b -> ce.p + (size_t)c * b -> lay.total_samples + b -> x_off[u];
        j.mid = b -> mid.p + (size_t)c * b -> lay.total_samples + b -> x_off[u];
        j.square = 1;
      } else {
        j.n = std::min(20000, b -> ny[u]) + 128;
        size_t o = ((size_t)u * nch + c) * b -> lay.ntemplate_ext;
        j.src = white + o; j.dst = b -> colored.p + o; j.mid = b -> mid.p + o;
        j.square = 0;
      }
      j.tmp = b -> iir_tmp.p + tmp_off; tmp_off += (size_t)j.n + 32;
      j.pad = g_hconv.filtfilt_pad;
      j.sec0 = 2 * llsm_cheby::row_of(cut0) + (hp0 ? 1 : 0);
      j.sec1 = ns == 2 ? 2 * llsm_cheby::row_of(cut1) + (hp1 ? 1 : 0) : -1;
      j.fused = 0; j.wlo = 0; j.whi = 0;
      // Band-pass (two sections): both sections per pass over the interior -- half the plane transfers of
      // F_hp B_hp F_lp B_lp --, and the two ends, where the reference's order shows (the second filtfilt pads and
      // initialises on the FIRST one's output), by short jobs in that order: M samples each from a stretch of M' = 2 M + 64,
      // M = the reach of the slowest pole to 1e-9 (k_filtfilt; measured against scipy: identical to 1e-13 beyond M)
      const int M = edge_M[c], Mp = edge_Mp[c];
      int reach = 0;                                      // of this job's slowest pole (0: not a job that is cut into segments)
      if(ns == 1) reach = decay_length(llsm_cheby::make_section_row(llsm_cheby::row_of(cut0), hp0).a, 1e-9);
      if(ns == 2 && M > 0 && j.n >= 4 * Mp) {
        reach = M;
        j.fused = 1; j.wlo = M; j.whi = j.n - M;
        for(int side = 0; side < 2; side ++) {
          FiltJob e = j;
          const int o = side == 0 ? 0 : j.n - Mp;
          e.fused = 0; e.n = Mp; e.src = j.src + o; e.dst = j.dst + o;
          e.mid = edge_buf.p + edge_off; edge_off += (size_t)Mp;
          e.tmp = edge_buf.p + edge_off; edge_off += (size_t)Mp + 32;
          e.wlo = side == 0 ? 0 : Mp - M; e.whi = side == 0 ? M : Mp;
          edge_jobs.push_back(e);
        }
      }
      push_job(j, reach > 0 ? ((reach + 31) & ~31) + 64 : 0, U * (which == 0 ? nch : nact));
    }
  }
  if(tmp_off > b -> iir_tmp.n || edge_off > edge_buf.n) { llsm_set_error("internal: IIR scratch too small"); return -1; }
  if(! seg_job.empty()) {
    DevBuf<float>& seg_buf = b -> iir_seg[which];
    if(seg_buf.alloc(seg_need)) return -1;
    for(size_t k = 0; k < seg_job.size(); k ++) jobs[(size_t)seg_job[k]].tmp = seg_buf.p + seg_tmp_off[k];
  }
  jobs.insert(jobs.end(), edge_jobs.begin(), edge_jobs.end());       // the short ones last: they fill the tail of the launch
  if(which == 0) { b -> njobs_ana = (int)jobs.size(); return upload_vec(b -> jobs_ana, jobs); }
  b -> njobs_syn = (int)jobs.size(); b -> nch_active = nact;
  return upload_vec(b -> jobs_syn, jobs);
}

// HMPP (dsputils.c:196-213, 318-326) transforms every frame at 2^ceil(log2(longest window)) points; the peak-picking
// kernel ends at 8192 (twiddle table, 128 KB of LDS), i.e. F0 >= 21.6 Hz at 44.1 kHz, 47 Hz at 96 kHz.  A batch whose
// lowest voiced F0 is known to need more is refused here, loudly, instead of coming back with rows of nhar = 0.
static bool hmpp_window_too_long(const llsm_gpu_batch* b) {
  if(!(b -> min_f0 > 0)) return false;
  // the same margin as the LDS provision (pp_lds_n): F0 refinement may lower F0 by just under 10 %
  const float fmin = b -> opt.f0_refine ? b -> min_f0 * 0.9f : b -> min_f0;
  const int n = lp::hwin(fmin, b -> fs, b -> opt.rel_winsize);
  if(n <= LLSM_BIG_FFT_MAX) return false;
  llsm_set_error("hm_method = HMPP: the analysis window at F0 = " + std::to_string(fmin) + " Hz" +
    (b -> opt.f0_refine ? " (lowest F0 of the batch less the 10 % F0 refinement may take)" : "") + " is " + std::to_string(n) +
    " samples; peak picking needs a transform beyond the supported " + std::to_string(LLSM_BIG_FFT_MAX) +
    " points (use LLSM_AOPTION_HMCZT, which has no such limit)");
  return true;
}

// Peak picking (HMPP) of `nsig` signals: frames whose transform fits the LDS (<= pp_lds_n points) by k_harm_pp, the
// rest -- F0 below 21.6 Hz at 44.1 kHz, 47 Hz at 96 kHz -- by k_harm_pp_big on global scratch.  pp_nmax: the largest
// transform the batch can need (from its lowest F0; unknown F0: the largest supported).
static int run_harm_pp(llsm_gpu_context* c, LaunchCtx* P, const BatchDev& d, llsm_gpu_batch* b, const float* sig, size_t sig_stride,
  int nsig, int maxnhar, int pp_lds_n, int pp_nmax, int* nhar_out, float* ampl, float* phse) {
  RUN(launch_harm_pp(P, d, sig, sig_stride, nsig, b -> nfft_u.p, maxnhar, b -> norm_base_blackman, c -> tw, c -> tw_nmax,
    pp_lds_n, nhar_out, ampl, phse));
  if(pp_nmax > pp_lds_n) {
    if(llsm_engine_big_fft(c, pp_nmax, (size_t)llsm_big_fft_grid((size_t)pp_nmax + pp_nmax / 2 + 2) * (size_t)(pp_nmax + pp_nmax / 2 + 2))) return -1;
    RUN(launch_harm_pp_big(P, d, sig, sig_stride, nsig, b -> nfft_u.p, maxnhar, b -> norm_base_blackman, pp_lds_n, pp_nmax,
      nhar_out, ampl, phse));
  }
  return 0;
}
// (pp_lds_n, pp_nmax) of a batch whose lowest F0 (less the refinement margin) is fmin
static void harm_pp_sizes(const llsm_gpu_batch* b, float fmin, int* lds_n, int* nmax) {
  int n = 64;
  const int w = lp::hwin(fmin, b -> fs, b -> opt.rel_winsize);
  while(n < LLSM_BIG_FFT_MAX && n < w) n <<= 1;
  *nmax = n;
  *lds_n = std::min(n, LLSM_LDS_FFT_MAX);
}

extern "C" int llsm_gpu_batch_analyze(llsm_gpu_batch* b) {
  if(! b) { llsm_set_error("llsm_gpu_batch_analyze: no batch"); return -1; }
  llsm_gpu_context* c = b -> ctx;
  hipSetDevice(c -> device);
  const bool hmpp = b -> opt.hm_method == LLSM_AOPTION_HMPP;
  if(! hmpp && b -> opt.hm_method != LLSM_AOPTION_HMCZT) {
    llsm_set_error("unknown hm_method"); return -1;
  }
  const llsm_gpu_layout& L = b -> lay;
  if(L.total_frames == 0) {
    // frameless batch: nothing to subtract, x_res = x (layer0.c:498-503 with no voiced frame)
    if(L.total_samples > 0)
      HIP_OK(hipMemcpyAsync(b -> arr[LLSM_GPU_XRES], b -> arr[LLSM_GPU_X], b -> arr_bytes[LLSM_GPU_X],
        hipMemcpyDeviceToDevice, c -> stream));
    return 0;
  }
  if(L.total_samples == 0) {
    // frames over empty signals: every window is zeros, so every row is the constant the path gives on
    // silence -- PSD floor 10 log10(exp(log 1e-10 + EULERGAMMA) 44100 / fs + 1e-12) (layer0.c:358, 383, 401),
    // zero residual, zero band energies, zero-amplitude harmonics on voiced frames
    const size_t F = L.total_frames;
    std::vector<float> f0(F);
    HIP_OK(hipMemcpyAsync(f0.data(), b -> arr[LLSM_GPU_F0], F * sizeof(float), hipMemcpyDeviceToHost, c -> stream));
    HIP_OK(hipStreamSynchronize(c -> stream));
    const float floor_db = (float)(10.0 * std::log10(std::exp(std::log(1e-10) + 0.57721566) * 44100.0 / b -> fs + 1e-12));
    std::vector<float> psd(F * (size_t)L.npsd, floor_db);
    std::vector<int> nhar(F), nhe(F), one(F, 1);
    for(size_t g = 0; g < F; g ++) {
      nhar[g] = f0[g] > 0 ? lp::nhar(f0[g], b -> fs, L.maxnhar) : 0;
      nhe[g] = f0[g] > 0 ? std::min(lp::nhar(f0[g], b -> fs, L.maxnhar_e), L.maxnhar_e) : 0;
    }
    for(int a : {LLSM_GPU_AMPL, LLSM_GPU_PHSE, LLSM_GPU_PSDRES, LLSM_GPU_EDC, LLSM_GPU_EENV_AMPL, LLSM_GPU_EENV_PHSE})
      if(b -> arr_bytes[a]) HIP_OK(hipMemsetAsync(b -> arr[a], 0, b -> arr_bytes[a], c -> stream));
    HIP_OK(hipMemcpyAsync(b -> arr[LLSM_GPU_PSD], psd.data(), psd.size() * sizeof(float), hipMemcpyHostToDevice, c -> stream));
    HIP_OK(hipMemcpyAsync(b -> arr[LLSM_GPU_NHAR], nhar.data(), F * sizeof(int), hipMemcpyHostToDevice, c -> stream));
    HIP_OK(hipMemcpyAsync(b -> arr[LLSM_GPU_NHAR_E], nhe.data(), F * sizeof(int), hipMemcpyHostToDevice, c -> stream));
    HIP_OK(hipMemcpyAsync(b -> arr[LLSM_GPU_HAS_PSDRES], one.data(), F * sizeof(int), hipMemcpyHostToDevice, c -> stream));
    HIP_OK(hipStreamSynchronize(c -> stream));
    return 0;
  }
  const size_t F = L.total_frames, X = L.total_samples, nch = L.nchannel;
  const size_t nspec = b -> nspec;
  if(b -> ce.alloc(nch * X) || b -> mid.alloc(std::max(nch * X,
       (size_t)L.n_utt * nch * L.ntemplate_ext)) ||
     b -> iir_tmp.alloc(std::max(nch * (X + 32 * (size_t)L.n_utt),
       (size_t)L.n_utt * nch * (L.ntemplate_ext + 32))) ||
     b -> env.alloc(F * nspec) || b -> psd_log.alloc(F * nspec) ||
     b -> pbuf.alloc((F / KAL_CHUNK + (size_t)L.n_utt + 1) * 4 * (size_t)L.npsd)) return -1;
  float* xres = (float*)b -> arr[LLSM_GPU_XRES];
  {
    const void* key[3] = {b -> ce.p, b -> mid.p, b -> iir_tmp.p};
    if(b -> njobs_ana == 0 || std::memcmp(key, b -> key_ana, sizeof(key))) {
      if(build_jobs(b, 0, b -> fs, xres, nullptr)) return -1;
      std::memcpy(b -> key_ana, key, sizeof(key));
    }
  }
  BatchDev d = batch_dev(b, b -> fs);
  LaunchCtx* P = & c -> lc;
  if(b -> opt.f0_refine) RUN(launch_refine_f0(P, d));
  // lowest F0 of the batch (sizes the LDS of the peak-picking FFT)
  // (unknown -- the F0 row was written through its device pointer --: the largest provision there is)
  float fmin = b -> min_f0 > 0 ? b -> min_f0 : 1.0f;
  fmin *= 0.9f;                                       // refinement may lower F0 by < 10 %
  int pp_lds_n = 0, pp_nmax = 0;
  if(hmpp) {
    // one FFT size per utterance (llsm_get_fftsize, dsputils.c:318-326), decided on the device
    // after F0 refinement; LDS is provisioned for the largest size the batch can need
    if(hmpp_window_too_long(b)) return -1;
    harm_pp_sizes(b, fmin, & pp_lds_n, & pp_nmax);
    if(b -> nfft_u.alloc(L.n_utt)) return -1;
    RUN(launch_utt_fftsize(P, d, pp_nmax, b -> nfft_u.p));
    if(run_harm_pp(c, P, d, b, d.x, 0, 1, L.maxnhar, pp_lds_n, pp_nmax, d.nhar, d.ampl, d.phse)) return -1;
  } else {
    RUN(launch_harm_speech(P, d, fmin));
  }
  RUN(launch_synth_ola(P, d, b -> sin_units.p, b -> n_sin_units, b -> sin_halo, b -> nwin_sin, b -> win_sin.p,
    std::min(L.maxnhar, 2048), b -> d_x_off.p, b -> d_nx.p, d.x, xres, 0, nullptr));   // x_res = x - harmonic part
  {
    // pairs whose DC / Nyquist spectrogram bin lies ~100 dB under the frame's harmonics: listed by the first launch, redone
    // with exact bins by the second (kernels.hip k_spgm_env_wf)
    const size_t npairs = b -> npairs > 0 ? (size_t)b -> npairs : (F + 1) / 2;
    if(b -> spgm_fix.alloc(npairs) || b -> spgm_fix_count.alloc(1)) return -1;
    HIP_OK(hipMemsetAsync(b -> spgm_fix_count.p, 0, sizeof(int), c -> stream));
  }
  // Second stream (not while profiling -- the per-kernel events assume one stream -- and not with llsm_gpu_analysis_overlap(0)):
  //  * the FIX launch of the spectrogram envelope (a few dozen wavefronts, 0.05 ms of mostly launch and one pair's latency)
  //    runs beside the residual PSD frames, which do not read the envelope rows;
  //  * the smoother needs the two planes and nothing below needs its rows: it runs beside the band filter and the envelope
  //    analysis (measured in one process, tools/ab_overlap.py: 4.11 -> 4.08 ms per analysis step -- the dispatcher
  //    interleaves the launches only at their edges: two Kalman wavefronts hold a SIMD's registers).
  // Once work sits on the second stream, EVERY way out of this function joins it: the early returns of RUN / HIP_OK
  // below would otherwise leave it reading env / psd_log and writing the PSD rows while the caller tears the
  // batch down (llsm_dev_free is a caching pool: no implicit synchronisation) -- ADVICE r3.
  struct AuxJoin {
    llsm_gpu_context* c; bool armed = false, joined = false;
    ~AuxJoin() {
      if(! armed || joined) return;
      // the join event may not have been recorded (the failure was the launch or the record itself): drain the stream
      (void)hipStreamSynchronize(c -> aux);
      (void)hipGetLastError();
    }
  } aux_join{c};
  bool use_aux = g_overlap.load() > 0;
  if(use_aux && ! c -> aux) {
    if(hipStreamCreateWithFlags(& c -> aux, hipStreamNonBlocking) != hipSuccess ||
       hipEventCreateWithFlags(& c -> ev_fork, hipEventDisableTiming) != hipSuccess ||
       hipEventCreateWithFlags(& c -> ev_join, hipEventDisableTiming) != hipSuccess) { c -> aux = nullptr; (void)hipGetLastError(); }
  }
  if(! c -> aux) use_aux = false;
  LaunchCtx Pa = *P; Pa.stream = c -> aux;
  RUN(launch_spgm_env(P, d, b -> nwin_psd, b -> nfft_spgm, ilog2(b -> nfft_spgm), b -> nfft_psd,
    b -> norm_base, c -> tw, c -> tw_nmax, b -> env.p, b -> spgm_fix.p, b -> spgm_fix_count.p, use_aux ? 1 : 3));
  bool fix_on_aux = false;
  if(use_aux) {
    if(hipEventRecord(c -> ev_fork, c -> stream) == hipSuccess && hipStreamWaitEvent(c -> aux, c -> ev_fork, 0) == hipSuccess) {
      aux_join.armed = true;
      RUN(launch_spgm_env(& Pa, d, b -> nwin_psd, b -> nfft_spgm, ilog2(b -> nfft_spgm), b -> nfft_psd,
        b -> norm_base, c -> tw, c -> tw_nmax, b -> env.p, b -> spgm_fix.p, b -> spgm_fix_count.p, 2));
      fix_on_aux = true;
    } else {
      (void)hipGetLastError();
      RUN(launch_spgm_env(P, d, b -> nwin_psd, b -> nfft_spgm, ilog2(b -> nfft_spgm), b -> nfft_psd,
        b -> norm_base, c -> tw, c -> tw_nmax, b -> env.p, b -> spgm_fix.p, b -> spgm_fix_count.p, 2));
      use_aux = false;
    }
  }
  if(b -> nfft_psd > LLSM_LDS_FFT_MAX && llsm_engine_big_fft(c, b -> nfft_psd, (size_t)llsm_big_fft_grid((size_t)b -> nfft_psd) * b -> nfft_psd)) return -1;
  RUN(launch_psd_frames(P, d, xres, b -> nwin_psd, b -> win_psd.p, b -> inv_wpow, b -> nfft_psd,
    ilog2(b -> nfft_psd), c -> tw, c -> tw_nmax, b -> psd_log.p));
  bool forked = false;
  if(use_aux) {
    // (the smoother follows the FIX launch on the second stream and waits there for the PSD frames of the first)
    if(hipEventRecord(c -> ev_fork, c -> stream) == hipSuccess && hipStreamWaitEvent(c -> aux, c -> ev_fork, 0) == hipSuccess) {
      aux_join.armed = true;
      RUN(launch_kalman(& Pa, d, b -> env.p, b -> psd_log.p, b -> pbuf.p, (int)nspec));
      HIP_OK(hipEventRecord(c -> ev_join, c -> aux));
      forked = true;
    } else (void)hipGetLastError();
  }
  if(! forked && fix_on_aux) {                          // the smoother stays on the first stream: it must see the FIX launch's rows
    HIP_OK(hipEventRecord(c -> ev_join, c -> aux));
    HIP_OK(hipStreamWaitEvent(c -> stream, c -> ev_join, 0));
    aux_join.joined = true;
  }
  if(! forked) RUN(launch_kalman(P, d, b -> env.p, b -> psd_log.p, b -> pbuf.p, (int)nspec));
  RUN(launch_filtfilt(P, b -> jobs_ana.p, b -> njobs_ana, b -> sections.p));
  RUN(launch_harm_env(P, d, b -> ce.p, X));          // edc for every frame (+ CZT envelopes)
  if(hmpp && L.maxnhar_e > 0)                         // HMPP: envelopes by peak picking instead
    if(run_harm_pp(c, P, d, b, b -> ce.p, X, L.nchannel, L.maxnhar_e, pp_lds_n, pp_nmax, d.nhar_e, d.eenv_ampl, d.eenv_phse)) return -1;
  if(forked) {
    HIP_OK(hipStreamWaitEvent(c -> stream, c -> ev_join, 0));               // everything later on the stream sees the smoother's rows
    aux_join.joined = true;
  }
  return 0;
}

// Harmonic stage of the analysis alone (llsm_refine_f0 + llsm_harmonic_analysis, dsputils.c:72-94, 175-228):
// outputs F0 (refined), NHAR, AMPL, PHSE.  Used by the per-frame API (frameapi.cpp).
int llsm_engine_batch_harmonics(llsm_gpu_batch* b, int refine_only) {
  llsm_gpu_context* c = b -> ctx;
  hipSetDevice(c -> device);
  const llsm_gpu_layout& L = b -> lay;
  if(L.total_frames == 0 || L.total_samples == 0) return 0;
  const bool hmpp = b -> opt.hm_method == LLSM_AOPTION_HMPP;
  BatchDev d = batch_dev(b, b -> fs);
  LaunchCtx* P = & c -> lc;
  if(b -> opt.f0_refine || refine_only) RUN(launch_refine_f0(P, d));
  if(refine_only) return 0;
  float fmin = b -> min_f0 > 0 ? b -> min_f0 : 1.0f;
  fmin *= 0.9f;
  if(hmpp) {
    int pp_lds_n = 0, pp_nmax = 0;
    if(hmpp_window_too_long(b)) return -1;
    harm_pp_sizes(b, fmin, & pp_lds_n, & pp_nmax);
    if(b -> nfft_u.alloc(L.n_utt)) return -1;
    RUN(launch_utt_fftsize(P, d, pp_nmax, b -> nfft_u.p));
    if(run_harm_pp(c, P, d, b, d.x, 0, 1, L.maxnhar, pp_lds_n, pp_nmax, d.nhar, d.ampl, d.phse)) return -1;
  } else RUN(launch_harm_speech(P, d, fmin));
  return 0;
}

// chebyfilt (dsputils.c:51-70) of one device signal: zero-phase low / high / band-pass, optionally squared.
// The section tables live in the context (built on first use).
int llsm_engine_chebyfilt(llsm_gpu_context* c, const float* d_src, int n, float c1, float c2, int square, float* d_dst) {
  hipSetDevice(c -> device);
  if(n <= 1) { if(n == 1) HIP_OK(hipMemcpyAsync(d_dst, d_src, sizeof(float), hipMemcpyDeviceToDevice, c -> stream)); return 0; }
  if(! c -> sections) {
    std::vector<FiltSectionD> secs(2 * llsm_cheby::kRows);
    for(int r = 0; r < llsm_cheby::kRows; r ++)
      for(int hp = 0; hp < 2; hp ++) {
        llsm_cheby::Section s = llsm_cheby::make_section_row(r, hp != 0);
        FiltSectionD& d = secs[2 * r + hp];
        std::memcpy(d.b, s.b, sizeof(d.b)); std::memcpy(d.a, s.a, sizeof(d.a)); std::memcpy(d.zi, s.zi, sizeof(d.zi));
        llsm_cheby::block_tables(s.a, IIR_SEG, 6, & d.H[0][0], & d.M[0][0]);
      }
    HIP_OK(hipMalloc((void**)& c -> sections, secs.size() * sizeof(FiltSectionD)));
    HIP_OK(hipMemcpy(c -> sections, secs.data(), secs.size() * sizeof(FiltSectionD), hipMemcpyHostToDevice));
  }
  if(c1 < 0) c1 = 0;
  if(c2 > 0.5f) c2 = 0.5f;
  FiltJob j; std::memset(& j, 0, sizeof(j));
  float *mid = nullptr, *tmp = nullptr; FiltJob* dj = nullptr;
  HIP_OK(hipMalloc((void**)& mid, sizeof(float) * (size_t)n));
  if(hipMalloc((void**)& tmp, sizeof(float) * ((size_t)n + 32)) != hipSuccess || hipMalloc((void**)& dj, sizeof(FiltJob)) != hipSuccess) {
    hipFree(mid); hipFree(tmp); llsm_set_error("chebyfilt: out of memory"); return -1;
  }
  j.src = d_src; j.dst = d_dst; j.mid = mid; j.tmp = tmp; j.n = n; j.square = square; j.pad = g_hconv.filtfilt_pad;
  if(c1 != 0 && c2 < 0.5f) { j.sec0 = 2 * llsm_cheby::row_of(c1) + 1; j.sec1 = 2 * llsm_cheby::row_of(c2); }
  else if(c1 == 0) { j.sec0 = 2 * llsm_cheby::row_of(c2); j.sec1 = -1; }
  else { j.sec0 = 2 * llsm_cheby::row_of(c1) + 1; j.sec1 = -1; }
  int rc = hipMemcpy(dj, & j, sizeof(j), hipMemcpyHostToDevice) != hipSuccess;
  if(! rc) rc = launch_filtfilt(& c -> lc, dj, 1, c -> sections);
  if(hipStreamSynchronize(c -> stream) != hipSuccess) rc = -1;
  hipFree(mid); hipFree(tmp); hipFree(dj);
  if(rc) { llsm_set_error("chebyfilt: launch failed"); return -1; }
  return 0;
}

extern "C" int llsm_gpu_batch_synthesize(llsm_gpu_batch* b, const llsm_soptions* so,
  unsigned long long seed, int use_injected_white) {
  if(! b || ! so) { llsm_set_error("llsm_gpu_batch_synthesize: no batch / options"); return -1; }
  llsm_gpu_context* c = b -> ctx;
  hipSetDevice(c -> device);
  if(so -> use_l1 && b -> l1_nspec == 0) {
    llsm_set_error("use_l1: the batch carries no layer-1 members (llsm_gpu_batch_enable_layer1 / _tolayer1)");
    return -1;
  }
  if(so -> fs != b -> fs) {
    llsm_set_error("llsm_soptions.fs must equal the sampling rate the batch was created with");
    return -1;
  }
  const llsm_gpu_layout& L = b -> lay;
  if(L.total_frames == 0 || L.total_out == 0) return 0;
  const float fs = so -> fs, thop = b -> opt.thop;
  const size_t F = L.total_frames, Y = L.total_out, nch = L.nchannel;
  if(b -> syn_fs != fs) {
    b -> nwin_env = lp::nwin_env(thop, fs);
    b -> nwin_filt = lp::nwin_filt(thop, fs);
    b -> nfft_filt = lp::nextpow2(b -> nwin_filt * 1.2 + 32);
    if(b -> nfft_filt > LLSM_BIG_FFT_MAX || b -> nfft_filt < 64) {
      llsm_set_error("noise-filter FFT size outside the supported range [64, 131072]"); return -1;
    }
    if(upload_vec(b -> win_env, make_hann(b -> nwin_env))) return -1;
    {
      // envelope overlap-add plan (layer0.c:307): sample p receives frame i's window sample j
      // when env_ola(i, j) == p.  Independent of the utterance, so one table per batch; built on the device
      // (k_env_plan: the host loop over frames x window samples was half a millisecond of every synthesis call).
      int max_nfrm = 0; for(int n : b -> nfrm) max_nfrm = std::max(max_nfrm, n);
      if(b -> env_hits.alloc((size_t)b -> max_ny * LLSM_EXC_HITS) || b -> env_over.alloc(1)) return -1;
      HIP_OK(hipMemsetAsync(b -> env_over.p, 0, sizeof(int), c -> stream));
      RUN(launch_env_plan(& c -> lc, b -> max_ny, max_nfrm, b -> nwin_env, thop, fs, b -> env_hits.p, b -> env_over.p));
      int over = 0;
      HIP_OK(hipMemcpyAsync(& over, b -> env_over.p, sizeof(int), hipMemcpyDeviceToHost, c -> stream));
      HIP_OK(hipStreamSynchronize(c -> stream));
      if(over) { llsm_set_error("envelope overlap-add plan: more than 3 frames per sample"); return -1; }
    }
    {
      // work units of k_noise_filter_ola: frames [i0, i1) of one utterance, i0 even; unit length so
      // that the batch gives about one unit per resident wavefront (2 / SIMD), never below 4 frames;
      // halo = frames before i0 whose N-sample output window still reaches sample start(i0)
      const long long Ftot = b -> lay.total_frames;
      int C = (int)((Ftot + NF_UNIT_DIV - 1) / NF_UNIT_DIV);
      C = std::max(4, (C + 1) & ~1);
      if(const char* e = std::getenv("LLSM_GPU_NOISE_UNIT")) C = std::max(2, std::atoi(e) & ~1);   // tuning override
      std::vector<int4> units;
      for(int u = 0; u < b -> lay.n_utt; u ++) {
        const int nf = b -> nfrm[u], nu = (nf + C - 1) / C;   // equal (even-sized) units within the utterance
        const int sz = nu > 0 ? (((nf + nu - 1) / nu) + 1) & ~1 : 2;
        for(int i0 = 0; i0 < nf; i0 += sz) units.push_back(make_int4(u, i0, std::min(i0 + sz, nf), 0));
      }
      b -> n_nf_units = (int)units.size();
      const double hop = (double)thop * fs;
      // frame i0 - k reaches sample start(i0) iff center(i0) - center(i0 - k) < N; centres are
      // k hop rounded to integers (+-1), so k <= floor((N + 1) / hop) covers every such frame
      b -> nf_halo = (int)std::floor((b -> nfft_filt + 1) / std::max(hop, 1.0));
      if(upload_vec(b -> nf_units, units)) return -1;
    }
    std::vector<float> wf = make_hann(b -> nwin_filt);
    double s = 0; for(float v : wf) s += (double)v * v;
    b -> inv_wsqr = (float)(1.0 / s);
    if(upload_vec(b -> win_filt, wf)) return -1;
    b -> syn_fs = fs; b -> njobs_syn = 0;
  }
  if(so -> use_l1 && llsm_l1_prefetch_rows(b, so)) return -1;
  const size_t tplsz = (size_t)L.n_utt * nch * L.ntemplate_ext;
  if(b -> colored.alloc(tplsz) || b -> mid.alloc(tplsz) ||
     b -> iir_tmp.alloc((size_t)L.n_utt * nch * (L.ntemplate_ext + 32)) ||
     b -> env_cplx.alloc(F * nch * std::max(L.maxnhar_e, 1)) || b -> yexc.alloc(Y)) return -1;
  float* white = (float*)b -> arr[LLSM_GPU_WHITE];
  {
    const void* key[3] = {b -> colored.p, b -> mid.p, b -> iir_tmp.p};
    if(b -> njobs_syn == 0 || std::memcmp(key, b -> key_syn, sizeof(key))) {
      if(build_jobs(b, 1, fs, nullptr, white)) return -1;
      std::memcpy(b -> key_syn, key, sizeof(key));
    }
  }
  BatchDev d = batch_dev(b, fs);
  LaunchCtx* P = & c -> lc;
  float* ysin = (float*)b -> arr[LLSM_GPU_YSIN];
  if(! use_injected_white) RUN(launch_white(P, d, white, L.ntemplate_ext, b -> d_ny.p, seed));
  RUN(launch_filtfilt(P, b -> jobs_syn.p, b -> njobs_syn, b -> sections.p));
  RUN(launch_env_params(P, d, b -> env_cplx.p));
  RUN(launch_excite_env(P, d, b -> colored.p, L.ntemplate_ext, b -> env_hits.p, b -> env_cplx.p,
    b -> nwin_env, b -> win_env.p, b -> nch_active, b -> d_y_off.p, b -> d_ny.p, b -> max_ny, fs,
    b -> yexc.p));
  // b -> fnyq: the PSD rows' axis (conf FNYQ; analysis fs / 2, layer0.c:481).  Transforms up to 2048 points: filter and overlap-add
  // in one kernel (the shaped frames stay on chip); larger ones: frames to HBM, gathered by the mix.
  float* ynoise = (float*)b -> arr[LLSM_GPU_YNOISE];
  static const bool fused_ok = [] { const char* e = std::getenv("LLSM_GPU_NOISE_OLA"); return !(e && e[0] == '0'); }();
  int fused = -2;
  if(fused_ok)
    fused = launch_noise_filter_ola(P, d, b -> nf_units.p, b -> n_nf_units, b -> nf_halo, b -> yexc.p,
      b -> d_y_off.p, b -> d_ny.p, b -> fnyq, fs, b -> nwin_filt, b -> win_filt.p, hann_sym(b -> nwin_filt), b -> inv_wsqr,
      ilog2(b -> nfft_filt), ynoise);
  if(fused != 0 && fused != -2) {
    llsm_set_error(std::string("launch_noise_filter_ola failed: ") + hipGetErrorString((hipError_t)fused));
    return -1;
  }
  float* yout = (float*)b -> arr[LLSM_GPU_Y];
  if(so -> use_l1) {
    // layer0.c:148-287: the harmonic part is the HM <-> pulse-by-pulse cross-fade state machine (l1.cpp);
    // the noise part first (fallback path: frames to HBM, gathered against a silent harmonic part)
    if(fused == -2) {
      if(b -> nframes.alloc(F * b -> nfft_filt) || b -> live.alloc(F)) return -1;
      if(b -> nfft_filt > LLSM_LDS_FFT_MAX &&
         llsm_engine_big_fft(c, b -> nfft_filt, (size_t)llsm_big_fft_grid((size_t)b -> nfft_filt + b -> nfft_filt / 2 + 1) * (size_t)(b -> nfft_filt + b -> nfft_filt / 2 + 1))) return -1;
      HIP_OK(hipMemsetAsync(ysin, 0, Y * sizeof(float), c -> stream));
      RUN(launch_noise_filter(P, d, b -> yexc.p, b -> d_y_off.p, b -> d_ny.p, b -> fnyq, fs,
        b -> nwin_filt, b -> win_filt.p, b -> inv_wsqr, b -> nfft_filt, ilog2(b -> nfft_filt),
        c -> tw, c -> tw_nmax, b -> nframes.p, b -> live.p, 0));
      RUN(launch_ola_noise_mix(P, d, b -> nframes.p, b -> live.p, b -> nfft_filt,
        b -> d_y_off.p, b -> d_ny.p, b -> max_ny, fs, ysin, ynoise, yout));
    }
    return llsm_l1_synthesize_harmonics(b, so, ynoise, ysin, yout);
  }
  // harmonic part last: its overlap-add flush also writes y = y_sin + y_noise when y_noise is in place
  RUN(launch_synth_ola(P, d, b -> sin_units.p, b -> n_sin_units, b -> sin_halo, b -> nwin_sin, b -> win_sin.p,
    std::min(L.maxnhar, 2048), b -> d_y_off.p, b -> d_ny.p, fused == 0 ? ynoise : nullptr, ysin, 1,
    fused == 0 ? yout : nullptr));
  if(fused == -2) {
    if(b -> nframes.alloc(F * b -> nfft_filt) || b -> live.alloc(F)) return -1;
    if(b -> nfft_filt > LLSM_LDS_FFT_MAX &&
       llsm_engine_big_fft(c, b -> nfft_filt, (size_t)llsm_big_fft_grid((size_t)b -> nfft_filt + b -> nfft_filt / 2 + 1) * (size_t)(b -> nfft_filt + b -> nfft_filt / 2 + 1))) return -1;
    RUN(launch_noise_filter(P, d, b -> yexc.p, b -> d_y_off.p, b -> d_ny.p, b -> fnyq, fs,
      b -> nwin_filt, b -> win_filt.p, b -> inv_wsqr, b -> nfft_filt, ilog2(b -> nfft_filt),
      c -> tw, c -> tw_nmax, b -> nframes.p, b -> live.p, 0));
    RUN(launch_ola_noise_mix(P, d, b -> nframes.p, b -> live.p, b -> nfft_filt,
      b -> d_y_off.p, b -> d_ny.p, b -> max_ny, fs, ysin, ynoise, yout));
  }
  return 0;
}

// ---- internal helpers for rt.cpp (declared in engine.h) --------------------
// A frame-less batch whose only job is to produce the band-limited Gaussian
// templates of S streams (llsm_generate_bandlimited_noise(len, ...),
// dsputils.c:385-394) with the same kernels the offline synthesis uses.
llsm_gpu_batch* llsm_engine_template_batch(llsm_gpu_context* ctx, const llsm_aoptions* opt,
  float fs, int S, int len, unsigned long long seed) {
  std::vector<int> z(S, 0);
  llsm_gpu_batch* b = llsm_gpu_create_batch(ctx, opt, fs, S, z.data(), z.data());
  if(! b) return nullptr;
  for(int u = 0; u < S; u ++) b -> ny[u] = len;
  const llsm_gpu_layout& L = b -> lay;
  const size_t tplsz = (size_t)S * L.nchannel * L.ntemplate_ext;
  bool bad = upload_vec(b -> d_ny, b -> ny) || b -> colored.alloc(tplsz) || b -> mid.alloc(tplsz) ||
    b -> iir_tmp.alloc((size_t)S * L.nchannel * (L.ntemplate_ext + 32));
  float* white = (float*)b -> arr[LLSM_GPU_WHITE];
  if(! bad) bad = build_jobs(b, 1, fs, nullptr, white) != 0;
  if(! bad) {
    BatchDev d = batch_dev(b, fs);
    bad = launch_white(& ctx -> lc, d, white, L.ntemplate_ext, b -> d_ny.p, seed) != 0 ||
          launch_filtfilt(& ctx -> lc, b -> jobs_syn.p, b -> njobs_syn, b -> sections.p) != 0;
  }
  if(bad) { llsm_gpu_delete_batch(b); return nullptr; }
  return b;
}
const float* llsm_engine_batch_colored(llsm_gpu_batch* b) { return b -> colored.p; }
int llsm_engine_batch_nch_active(llsm_gpu_batch* b) { return b -> nch_active; }
LaunchCtx* llsm_engine_launch_ctx(llsm_gpu_context* c) { return & c -> lc; }
const float2* llsm_engine_twiddles(llsm_gpu_context* c, int* nmax) { *nmax = c -> tw_nmax; return c -> tw; }

// Transforms beyond the 8192 points of the LDS kernels (kernels.h LLSM_LDS_FFT_MAX): a twiddle table
// e^{-2 pi i m / 2^17} (rounded from float64, built on first use) and `elems` float2 of global scratch for the
// persistent workgroups of such a launch.  Returns -1 (error text set) when N is beyond even that table.
int llsm_engine_big_fft(llsm_gpu_context* c, int N, size_t elems) {
  if(N > LLSM_BIG_FFT_MAX) {
    llsm_set_error("transform of " + std::to_string(N) + " points: beyond the supported " + std::to_string(LLSM_BIG_FFT_MAX)); return -1;
  }
  LaunchCtx& lc = c -> lc;
  if(! lc.tw_big) {
    const int nmax = LLSM_BIG_FFT_MAX;
    std::vector<float2> tw((size_t)nmax / 2);
    for(int m = 0; m < nmax / 2; m ++) {
      const double a = 2.0 * 3.14159265358979323846 * m / nmax;
      tw[m] = make_float2((float)std::cos(a), (float)-std::sin(a));
    }
    float2* p = nullptr;
    if(hipMalloc(& p, tw.size() * sizeof(float2)) != hipSuccess ||
       hipMemcpy(p, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
      if(p) hipFree(p);
      llsm_set_error("big twiddle table allocation failed"); return -1;
    }
    lc.tw_big = p; lc.tw_big_nmax = nmax;
  }
  if(lc.big_scratch_elems < elems) {
    hipDeviceSynchronize();                            // earlier launches may still use the old block
    if(lc.big_scratch) hipFree(lc.big_scratch);
    lc.big_scratch = nullptr; lc.big_scratch_elems = 0;
    if(hipMalloc(& lc.big_scratch, elems * sizeof(float2)) != hipSuccess) { llsm_set_error("big-transform scratch allocation failed"); return -1; }
    lc.big_scratch_elems = elems;
  }
  return 0;
}
int llsm_engine_device(llsm_gpu_context* c) { return c -> device; }

extern "C" int llsm_gpu_plan_index(int which, int i, int j, FP_TYPE f0, FP_TYPE thop,
  FP_TYPE fs, FP_TYPE rel) {
  switch(which) {
    case 0: return lp::center(i, thop, fs);
    case 1: return lp::nwin_sin(thop, fs);
    case 2: return lp::nwin_env(thop, fs);
    case 3: return lp::nwin_filt(thop, fs);
    case 4: return lp::nwin_psd(thop, fs);
    case 5: return lp::ny(i, thop, fs);
    case 6: return lp::hwin(f0, fs, rel);
    case 7: return lp::nhar(f0, fs, i);
    case 8: return lp::env_ola(i, j, thop, fs);
    case 9: return lp::dcwin(f0, thop, fs);
    case 10: return lp::spgmwin(f0, fs, i);
    case 11: { int b2; float r; return lp::stretch_index(i, j, (int)f0, 128, & b2, & r); }
    case 12: { int b2; float r; lp::stretch_index(i, j, (int)f0, 128, & b2, & r); return b2; }
  }
  return -1;
}
