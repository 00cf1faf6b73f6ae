// l1.cpp -- layer-1 (source-filter) conversion and pulse-by-pulse synthesis: host side.
//
//   llsm_chunk_tolayer1 / llsm_chunk_tolayer0 / llsm_frame_tolayer0 / llsm_conf_checklayer1
//       (layer1.c:129-195, llsm.h:221, 243, 324-327) as batches of the kernels in l1_kernels.hip
//   llsm_gpu_batch_enable_layer1 / _tolayer1 / _tolayer0: the same on a device-resident batch
//   llsm_l1_synthesize_harmonics: llsm_synthesize_harmonics with options->use_l1 = 1 (layer0.c:148-287)
//
// The pulse tracker of the PbP synthesis is a sequential state machine per utterance that calls the
// host's llsm_fgfm effect callbacks in frame / pulse order (layer0.c:208-217); it runs here on the host,
// in float64, over six small per-frame rows downloaded from the batch, and emits work tables for the
// device: pulse groups (k_pbp_pulse), the frames the harmonic model still has to render
// (k_l1_to_l0 where HM is missing, k_synth_frames), and the cross-fade segments (k_l1_mixcurve).
// Every sample of every signal is computed on the device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "batch.h"
#include "lfmodel.h"
#include "plan.h"

namespace lp = llsm_plan;
namespace lf = llsm_lf;

extern const float2* llsm_engine_twiddles(llsm_gpu_context* c, int* nmax);

#define RUN1(call)                                                                     \
  do {                                                                                 \
    int rc_ = (call);                                                                  \
    if(rc_ != 0) {                                                                     \
      llsm_set_error(std::string(#call) + " failed: " +                                \
        (rc_ > 0 ? hipGetErrorString((hipError_t)rc_) : "unsupported configuration")); \
      return -1;                                                                       \
    }                                                                                  \
  } while(0)

// ------------------------------------------------------------------ batch arrays
int l1_pbp_real_ifft(int on);                        // l1_kernels.hip
// Pulse groups through ONE half-size complex inverse transform (real output; k_pbp_pulse<.., true>) instead of the
// full-size complex transform of the Hermitian-completed spectrum.  on = 1 / 0 switches it for the process (default on),
// on < 0 only queries; returns the previous setting.  Same samples to float32 rounding (tests/test_gpu_l1.py).
extern "C" int llsm_gpu_pbp_real_ifft(int on) { return l1_pbp_real_ifft(on); }

extern "C" int llsm_gpu_batch_enable_layer1(llsm_gpu_batch* b, int nfft) {
  if(! b || nfft < 64 || (nfft & (nfft - 1)) || nfft > 8192) {
    llsm_set_error("llsm_gpu_batch_enable_layer1: nfft must be a power of two in [64, 8192]"); return -1;
  }
  const int nspec = nfft / 2 + 1;
  if(b -> l1_nspec == nspec) return 0;
  if(b -> l1_nspec != 0) { llsm_set_error("llsm_gpu_batch_enable_layer1: already enabled with another size"); return -1; }
  hipSetDevice(b -> ctx -> device);
  const size_t F = (size_t)b -> lay.total_frames;
  size_t sizes[LLSM_GPU_NARRAYS]; std::memset(sizes, 0, sizeof(sizes));
  sizes[LLSM_GPU_RD] = F * sizeof(float); sizes[LLSM_GPU_VTMAGN] = F * nspec * sizeof(float);
  sizes[LLSM_GPU_VSPHSE] = F * (size_t)b -> lay.maxnhar * sizeof(float);
  sizes[LLSM_GPU_NVSPHSE] = sizes[LLSM_GPU_PBPSYN] = sizes[LLSM_GPU_HAS_HM] = F * sizeof(int);
  for(int a : {LLSM_GPU_RD, LLSM_GPU_VTMAGN, LLSM_GPU_VSPHSE, LLSM_GPU_NVSPHSE, LLSM_GPU_PBPSYN, LLSM_GPU_HAS_HM}) {
    b -> arr_bytes[a] = sizes[a];
    if(sizes[a] == 0) continue;
    hipError_t e = llsm_dev_malloc(& b -> arr[a], sizes[a]);
    if(e != hipSuccess) { llsm_set_error(std::string("hipMalloc(layer-1 array): ") + hipGetErrorString(e)); return -1; }
    HIP_OK(hipMemsetAsync(b -> arr[a], 0, sizes[a], b -> ctx -> stream));
  }
  if(F) {                                              // HM rows are valid unless the host says otherwise
    std::vector<int> one(F, 1);
    HIP_OK(hipMemcpyAsync(b -> arr[LLSM_GPU_HAS_HM], one.data(), F * sizeof(int), hipMemcpyHostToDevice, b -> ctx -> stream));
    HIP_OK(hipStreamSynchronize(b -> ctx -> stream));
  }
  b -> l1_nspec = nspec;
  b -> effects.assign(F, llsm_gpu_batch::Effect());
  return 0;
}

extern "C" int llsm_gpu_batch_set_maxnhar_conf(llsm_gpu_batch* b, int maxnhar_conf) {
  if(! b) return -1;
  b -> maxnhar_conf = maxnhar_conf;
  return 0;
}

extern "C" int llsm_gpu_batch_set_pbpeffect(llsm_gpu_batch* b, int frame, llsm_fgfm modifier, void* info,
  llsm_container* src_frame) {
  if(! b || b -> l1_nspec == 0 || frame < 0 || frame >= b -> lay.total_frames) {
    llsm_set_error("llsm_gpu_batch_set_pbpeffect: layer 1 not enabled or frame out of range"); return -1;
  }
  b -> effects[frame].modifier = modifier; b -> effects[frame].info = info; b -> effects[frame].frame = src_frame;
  return 0;
}

static L1Dev l1_dev(llsm_gpu_batch* b) {
  L1Dev d;
  d.nframes = b -> lay.total_frames; d.maxnhar = b -> lay.maxnhar; d.nspec = b -> l1_nspec;
  d.fnyq = b -> fnyq; d.lip_radius = b -> opt.lip_radius;
  d.f0 = (const float*)b -> arr[LLSM_GPU_F0]; d.nhar = (int*)b -> arr[LLSM_GPU_NHAR];
  d.ampl = (float*)b -> arr[LLSM_GPU_AMPL]; d.phse = (float*)b -> arr[LLSM_GPU_PHSE];
  d.rd = (float*)b -> arr[LLSM_GPU_RD]; d.vtmagn = (float*)b -> arr[LLSM_GPU_VTMAGN];
  d.vsphse = (float*)b -> arr[LLSM_GPU_VSPHSE]; d.nvsphse = (int*)b -> arr[LLSM_GPU_NVSPHSE];
  d.has_hm = (int*)b -> arr[LLSM_GPU_HAS_HM];
  d.src_ampl = b -> l1_src_ampl.p;                      // NULL unless tolayer1 allocated it
  d.acache.alpha = b -> l1_alpha.p;                     // NULL until l1_alpha_cache(b) (then keyed by Rd, F0 per frame)
  d.acache.rd = b -> l1_alpha_key.p;
  d.acache.f0 = b -> l1_alpha_key.p ? b -> l1_alpha_key.p + (size_t)b -> lay.total_frames : nullptr;
  d.pairs = b -> npairs > 0 ? b -> d_pairs.p : nullptr; d.npairs = b -> npairs;
  return d;
}

// alpha cache of the batch: keys start as NaN (never equal), so every frame is solved once and found afterwards
static int l1_alpha_cache(llsm_gpu_batch* b) {
  const size_t F = (size_t)b -> lay.total_frames;
  if(F == 0 || b -> l1_alpha.p) return 0;
  if(b -> l1_alpha.alloc(9 * F) || b -> l1_alpha_key.alloc(2 * F)) return -1;   // 9 = LF_CACHE_DOUBLES (l1_kernels.hip)
  HIP_OK(hipMemsetAsync(b -> l1_alpha_key.p, 0xff, 2 * F * sizeof(float), b -> ctx -> stream));
  return 0;
}

// llsm_create_cached_glottal_model(linspace(0.02, 3, 64), 64, 80) (layer1.c:54-57, dsputils.c:519-538)
static int glottal_tables(llsm_gpu_batch* b) {
  if(b -> l1_model_power.p) return 0;
  const int nc = 64, nh = 80;
  std::vector<float> power((size_t)nc * nh), param(nc);
  const double f0 = 200.0;
  for(int i = 0; i < nc; i ++) {
    param[i] = (float)(0.02 + (3.0 - 0.02) * i / (nc - 1));
    const lf::Solved s = lf::solve(lf::from_rd((double)param[i], 1.0 / f0, 1.0, llsm_conv_lf_rd_clamp()));
    for(int j = 0; j < nh; j ++) {
      const double m = lf::magnitude(s, f0 * (1.0 + j)) / (j + 1.0);
      power[(size_t)i * nh + j] = (float)(m * m);
    }
  }
  // the same table as reciprocals and as prefix sums of logarithms, float64, candidate index fastest (glottal_fit_tab)
  std::vector<double> inv_t((size_t)nh * nc), cumlog_t((size_t)(nh + 1) * nc);
  for(int i = 0; i < nc; i ++) {
    double acc = 0.0;
    for(int j = 0; j < nh; j ++) {
      const double m = (double)power[(size_t)i * nh + j];         // (the float32 value the per-term form divides by)
      inv_t[(size_t)j * nc + i] = 1.0 / m;
      cumlog_t[(size_t)j * nc + i] = acc;
      acc += std::log(m);
    }
    cumlog_t[(size_t)nh * nc + i] = acc;
  }
  if(upload_vec(b -> l1_model_power, power) || upload_vec(b -> l1_model_param, param) ||
     upload_vec(b -> l1_model_inv_t, inv_t) || upload_vec(b -> l1_model_cumlog_t, cumlog_t)) return -1;
  return 0;
}

extern "C" int llsm_gpu_batch_tolayer1(llsm_gpu_batch* b, int nfft) {
  if(llsm_gpu_batch_enable_layer1(b, nfft)) return -1;
  llsm_gpu_context* c = b -> ctx;
  hipSetDevice(c -> device);
  const size_t F = (size_t)b -> lay.total_frames;
  if(F == 0) return 0;
  if(glottal_tables(b)) return -1;
  if(b -> l1_rd_raw.alloc(F) || b -> l1_cont.alloc(F) || b -> l1_prev.alloc(F) || b -> l1_next.alloc(F) ||
     b -> l1_src_ampl.alloc(F * (size_t)b -> lay.maxnhar)) return -1;
  if(l1_alpha_cache(b)) return -1;
  L1Dev d = l1_dev(b);
  LaunchCtx* P = & c -> lc;
  int tw_nmax = 0; const float2* tw = llsm_engine_twiddles(c, & tw_nmax);
  static const bool rd_tab = [] { const char* e = std::getenv("LLSM_GPU_RD_FIT_TABLES"); return !(e && e[0] == '0'); }();
  RUN1(launch_l1_rd_fit(P, d, b -> l1_model_power.p, b -> l1_model_param.p, rd_tab ? b -> l1_model_inv_t.p : nullptr,
    rd_tab ? b -> l1_model_cumlog_t.p : nullptr, b -> l1_rd_raw.p));
  const int order = (int)std::round(0.02 / (double)b -> opt.thop);
  RUN1(launch_l1_rd_smooth(P, b -> lay.n_utt, b -> d_frm_off.p, b -> d_nfrm.p, order, b -> l1_rd_raw.p, b -> l1_prev.p,
    b -> l1_next.p, b -> l1_cont.p, d.rd));
  RUN1(launch_l1_frame(P, d, nfft, tw, tw_nmax));
  return 0;
}

extern "C" int llsm_gpu_batch_tolayer0(llsm_gpu_batch* b, int only_missing) {
  if(! b || b -> l1_nspec == 0) { llsm_set_error("llsm_gpu_batch_tolayer0: layer 1 not enabled"); return -1; }
  llsm_gpu_context* c = b -> ctx;
  hipSetDevice(c -> device);
  if(b -> lay.total_frames == 0) return 0;
  int tw_nmax = 0; const float2* tw = llsm_engine_twiddles(c, & tw_nmax);
  if(l1_alpha_cache(b)) return -1;
  RUN1(launch_l1_to_l0(& c -> lc, l1_dev(b), b -> maxnhar_conf, only_missing, nullptr, tw, tw_nmax));
  return 0;
}

// ------------------------------------------------------------------ host threads of the pulse scheduler
// The scheduler forks twice per call (per-utterance state machines, then the merge of their tables); creating and joining
// a dozen threads each time was ~0.4 ms of a 2.7 ms host phase that the device waits for.  The threads are created once
// and parked on a condition variable (never destroyed: they touch nothing but the caller's closure).
namespace {
class SchedPool {
  std::vector<std::thread> th_;
  std::mutex m_; std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr; std::atomic<int> next_{0};
  int items_ = 0, active_ = 0, use_ = 0; unsigned long gen_ = 0;
  void worker(int id) {
    unsigned long seen = 0;
    for(;;) {
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&] { return gen_ != seen; });
      seen = gen_;
      const std::function<void(int)>* f = fn_; const int n = items_; const bool mine = id < use_;
      lk.unlock();
      if(mine) for(int u; (u = next_.fetch_add(1)) < n; ) (*f)(u);
      lk.lock();
      if(-- active_ == 0) done_.notify_one();
    }
  }
public:
  explicit SchedPool(int n) { for(int t = 0; t < n; t ++) th_.emplace_back([this, t] { worker(t); }); for(auto& t : th_) t.detach(); }
  int size() const { return (int)th_.size(); }
  void run(int nthr, int n, const std::function<void(int)>& f) {
    std::unique_lock<std::mutex> lk(m_);
    fn_ = & f; items_ = n; next_.store(0); use_ = nthr; active_ = (int)th_.size(); gen_ ++;
    cv_.notify_all();
    done_.wait(lk, [&] { return active_ == 0; });
    fn_ = nullptr;
  }
};
std::mutex g_sched_run;                                // one scheduler at a time uses the pool
}  // namespace
static bool sched_pool_run(int nthr, int n, const std::function<void(int)>& fn) {
  std::unique_lock<std::mutex> lk(g_sched_run, std::try_to_lock);
  if(! lk.owns_lock()) return false;
  static SchedPool* pool = new SchedPool(16);
  pool -> run(std::min(nthr, pool -> size()), n, fn);
  return true;
}

// ------------------------------------------------------------------ PbP scheduler (layer0.c:155-287)
namespace {
typedef llsm_gpu_batch::L1Rows HostRows;

// The rows the pulse scheduler reads, and the next-cycle projection of every frame (k_l1_projection, enqueued here).
int download_rows(llsm_gpu_batch* b, double fs, HostRows& r) {
  const size_t F = (size_t)b -> lay.total_frames;
  hipStream_t st = b -> ctx -> stream;
  if(! r.block.resize(F * 28 + 8)) return -1;
  r.proj = (double*)r.block.data(); r.f0 = (float*)(r.proj + F); r.rd = r.f0 + F;
  r.nvs = (int*)(r.rd + F); r.pbpsyn = r.nvs + F; r.has_hm = r.pbpsyn + F;
  if(F == 0) return 0;
  hipSetDevice(b -> ctx -> device);
  if(b -> l1_proj.alloc(F * 7 / 2 + 1)) return -1;      // F doubles + 5 F four-byte values
  if(l1_alpha_cache(b)) return -1;
  { const int rc = launch_l1_projection(& b -> ctx -> lc, l1_dev(b), fs, b -> l1_proj.p);
    if(rc != 0) { llsm_set_error("launch_l1_projection failed"); return -1; } }
  char* pk = (char*)(b -> l1_proj.p + F);               // the five rows behind the projections, then ONE copy down
  const void* rows[5] = {b -> arr[LLSM_GPU_F0], b -> arr[LLSM_GPU_RD], b -> arr[LLSM_GPU_NVSPHSE], b -> arr[LLSM_GPU_PBPSYN], b -> arr[LLSM_GPU_HAS_HM]};
  for(int k = 0; k < 5; k ++) HIP_OK(hipMemcpyAsync(pk + (size_t)k * F * 4, rows[k], F * 4, hipMemcpyDeviceToDevice, st));
  HIP_OK(hipMemcpyAsync(r.block.data(), b -> l1_proj.p, F * 28, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  return 0;
}

double wrap_pi(double x) { return x - 2.0 * lf::kPi * std::round(x / (2.0 * lf::kPi)); }
}  // namespace

// where the next glottal cycle begins, relative to `origin` (layer0.c:181-191, llsmrt.c:316-326)
// source_p0_cached != NULL: *source_p0_cached is the LF model's phase at F0 for this (rd, f0) when *cache_valid, else it is
// computed (the model's solve: most of this function) and stored there -- llsmrt keeps one per stream, and a stream whose
// Rd and F0 stand still from one hop to the next (held notes, constant voice quality) skips the solve
double llsm_l1_pulse_projection(double rd, double f0, double vsphse0, double fs, double origin,
  lf::Model* model_out, double* source_p0_cached, bool cache_valid) {
  const double len_period = fs / f0;
  const lf::Model sm = lf::from_rd(rd, 1.0 / f0, 1.0, llsm_conv_lf_rd_clamp());
  if(model_out) *model_out = sm;
  double source_p0;
  if(source_p0_cached && cache_valid) source_p0 = *source_p0_cached;
  else {
    // phase of the model at its own fundamental: a function of Rd alone, tabulated (lfmodel.h phase_at_f0; 7e-14 rad
    // from lf::phase(lf::solve(sm), f0), which took 0.3 us per stream and hop)
    source_p0 = lf::phase_at_f0(rd, llsm_conv_lf_rd_clamp()) - 0.5 * lf::kPi;           // flow derivative -> flow
    if(source_p0_cached) *source_p0_cached = source_p0;
  }
  const double p0 = wrap_pi(vsphse0);
  double p0_dist = wrap_pi(source_p0 - p0);                     // phase_diff(source_p0, p0)
  if(p0_dist < 0) p0_dist += 2.0 * lf::kPi;
  return origin + p0_dist / 2.0 / lf::kPi * len_period;
}

// Called by llsm_gpu_batch_synthesize BEFORE it enqueues the noise branch: the projections are computed, the rows
// fetched and the stream drained while nothing else is queued, so that the host scheduler below runs beside the noise
// kernels instead of after them.
int llsm_l1_prefetch_rows(llsm_gpu_batch* b, const llsm_soptions* so) {
  b -> l1_rows.valid = false;
  if(b -> l1_nspec == 0) return 0;
  if(download_rows(b, (double)so -> fs, b -> l1_rows)) return -1;
  b -> l1_rows.valid = true;
  return 0;
}

int llsm_l1_synthesize_harmonics(llsm_gpu_batch* b, const llsm_soptions* so, const float* ynoise,
  float* ysin, float* yout) {
  llsm_gpu_context* c = b -> ctx;
  if(b -> l1_nspec == 0) { llsm_set_error("use_l1: the batch carries no layer-1 members (llsm_gpu_batch_enable_layer1)"); return -1; }
  const llsm_gpu_layout& L = b -> lay;
  const size_t F = (size_t)L.total_frames, Y = (size_t)L.total_out;
  const double fs = so -> fs, thop = b -> opt.thop;
  const float fsf = so -> fs, thopf = b -> opt.thop;
  static const bool timing = std::getenv("LLSM_L1_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
    return std::chrono::duration<double, std::milli>(b2 - a).count(); };
  const auto t_0 = now();
  HostRows& r = b -> l1_rows;
  if(! r.valid && download_rows(b, fs, r)) return -1;
  r.valid = false;                                     // (one use: the rows may change before the next call)
  const auto t_1 = now();
  const int nspec = b -> l1_nspec, nwin = b -> nwin_sin;
  std::vector<float> f0_hm(F, 0.0f);                   // frames the harmonic model renders
  std::vector<int> need_l0(F, 0);                      // ... of which HM has to be built from layer 1 first
  std::vector<int> blk_off(L.n_utt + 1, 0);
  size_t pulse_total = 0; int size_max = 64; bool any_need_l0 = false;
  const double hop = (double)lp::fmul(thopf, fsf);
  // Phase A, the glottal-closure projection of every frame (LF model from Rd, its alpha solved in float64, phase at
  // f0), is a pure function of the frame's rows: it used to be 95 % of the scheduler's time on the host (7 ms per
  // 204 800 frames on 64 threads) and now arrives with the rows (k_l1_projection).  Phase B below is the reference's
  // sequential state machine (and its callbacks) over those projections, in the reference's order.
  const auto t_1b = now();
  // Phase B: the reference's sequential state machine, one utterance at a time.  Utterances do not share state, so
  // without effect callbacks they are scheduled on host threads into per-utterance tables (indices local to the
  // utterance) that are concatenated afterwards; with callbacks anywhere in the batch the utterances run in order
  // on this thread, so that the host sees its llsm_fgfm calls in frame / pulse order across the whole batch.
  typedef llsm_gpu_batch::L1UttPlan UttPlan;
  std::vector<UttPlan>& plans = b -> l1_plans;
  if(plans.size() != (size_t)L.n_utt) plans.resize((size_t)L.n_utt);
  auto schedule_utt = [&](int u) {
    UttPlan& pl = plans[(size_t)u];
    pl.jobs.clear(); pl.pulses.clear(); pl.segs.clear(); pl.blk.clear(); pl.pulse_total = 0; pl.size_max = 64;   // (capacity stays)
    std::vector<PbpJob>& jobs = pl.jobs; std::vector<PbpPulse>& pulses = pl.pulses; std::vector<PbpSeg>& segs = pl.segs;
    std::vector<int2>& blk_jobs = pl.blk; size_t& pulse_total = pl.pulse_total; int& size_max = pl.size_max;
    const int fo = b -> frm_off[u], nf = b -> nfrm[u], ny = b -> ny[u], yo = b -> y_off[u];
    const size_t job0 = 0;
    double pulse_previous = 0, pbp_switch_rate = 0, pbp_switch_state = 0;
    std::vector<double>& offsets = pl.offsets; offsets.clear();
    int pbp_periods = 0, baseidx_prev = 0; const int pbp_periods_thrd = 3;
    for(int i = 0; i < nf; i ++) {
      const size_t g = (size_t)fo + i;
      const double f0 = r.f0[g];
      if(f0 == 0) continue;
      const int baseidx = (int)lp::fmul(lp::fmul((float)i, thopf), fsf);     // int baseidx = i * thop * fs
      if(r.nvs[g] <= 0) continue;                       // no VSPHSE / VTMAGN / RD on this frame
      const bool pbp_on = r.pbpsyn[g] == 1;
      double len_period = fs / f0;
      const double pulse_projected = (double)baseidx + r.proj[g]; // origin + p0_dist / 2 pi * len_period (llsm_l1_pulse_projection)
      const int len_reset = (int)(std::max(len_period, thop * fs) * 2);
      if(pulse_projected - pulse_previous > len_reset) pulse_previous = pulse_projected - len_reset;
      const int num_periods = (int)std::round((pulse_projected - pulse_previous) / len_period);
      len_period = (pulse_projected - pulse_previous) / num_periods;          // inf / nan when num_periods == 0, as the reference
      if((pbp_on || pbp_periods > 0) && num_periods > 0) {
        const int pulse_size = lp::nextpow2(std::max(len_period * 2, (double)nspec));
        PbpJob job; job.frame = (int)g; job.first = (int)pulses.size(); job.npulse = num_periods; job.size = pulse_size;
        job.pre_rotate = (int)len_period;
        offsets.assign((size_t)num_periods, 0.0);
        const lf::Model source_model = lf::from_rd((double)r.rd[g], 1.0 / f0, 1.0, llsm_conv_lf_rd_clamp());
        const llsm_gpu_batch::Effect& ef = b -> effects[g];
        for(int j = 0; j < num_periods; j ++) {
          double delta_t = 0; lf::Model src = source_model;
          if(ef.modifier) {
            llsm_gfm gm;                                 // llsm_lfmodel_to_gfm, llsmutils.c:24-32
            gm.Fa = (FP_TYPE)(1.0 / (source_model.ta * source_model.T0));
            gm.Rk = (FP_TYPE)((source_model.te - source_model.tp) / source_model.tp);
            gm.Rg = (FP_TYPE)(0.5 / source_model.tp); gm.T0 = (FP_TYPE)source_model.T0; gm.Ee = (FP_TYPE)source_model.Ee;
            FP_TYPE dt = 0;
            ef.modifier(& gm, & dt, ef.info, ef.frame);
            delta_t = dt;
            src.ta = 1.0 / (double)gm.Fa / (double)gm.T0; src.tp = 0.5 / (double)gm.Rg;   // llsm_gfm_to_lfmodel
            src.te = src.tp + src.tp * (double)gm.Rk; src.T0 = gm.T0; src.Ee = gm.Ee;
          }
          offsets[j] = pulse_previous + j * len_period + delta_t * fs;
          PbpPulse pu; pu.T0 = src.T0; pu.te = src.te; pu.tp = src.tp; pu.ta = src.ta; pu.Ee = src.Ee; pu.offset = 0; pu.pad = 0;
          pulses.push_back(pu);
        }
        const int pulse_base = (int)offsets[0];
        for(int j = 0; j < num_periods; j ++) pulses[job.first + j].offset = (float)(offsets[j] - pulse_base);
        // idx = (int)(pulse_base + k - len_period): floor for idx >= 1, truncation towards zero below
        const double cl = std::ceil(len_period);
        job.start = pulse_base - (int)cl;
        job.zero_extra = (cl != len_period) ? (-job.start - 1) : -1;
        job.out_off = (int)pulse_total; pulse_total += (size_t)pulse_size;
        size_max = std::max(size_max, pulse_size);
        jobs.push_back(job);
        pbp_periods += pbp_on ? num_periods : -num_periods;
        pbp_periods = std::min(pbp_periods, pbp_periods_thrd);
        pbp_periods = std::max(pbp_periods, 0);
      }
      pulse_previous = pulse_projected;
      pbp_switch_rate = 1.0 / (len_period < thop * fs ? len_period : thop * fs);
      // cross-fade curve over [baseidx_prev, baseidx): the device replays the additions of this segment
      PbpSeg sg; sg.state = pbp_switch_state; sg.rate = pbp_switch_rate; sg.j0 = baseidx_prev; sg.j1 = baseidx;
      sg.len = ny; sg.out_off = yo; sg.pad = 0; sg.dir = 0;
      bool require_hm = false;
      if(pbp_on && pbp_periods == pbp_periods_thrd) sg.dir = 1;
      else if(! pbp_on && pbp_periods == 0) sg.dir = -1;
      // the per-sample additions of layer0.c:245-256, run only while they still change the state (the sums are
      // sequential float64 additions: replayed, not closed-formed, so that the curve matches the reference's)
      if(sg.dir > 0)
        for(int j = baseidx_prev; j < baseidx && pbp_switch_state < 1.0; j ++) { pbp_switch_state += pbp_switch_rate; require_hm = true; }
      else if(sg.dir < 0)
        for(int j = baseidx_prev; j < baseidx && pbp_switch_state > 0; j ++) { pbp_switch_state -= pbp_switch_rate; require_hm = true; }
      if(sg.j1 > sg.j0) segs.push_back(sg);
      baseidx_prev = baseidx;
      if(pbp_on && pbp_periods == pbp_periods_thrd && ! require_hm) continue;
      f0_hm[g] = (float)f0;
      if(! r.has_hm[g]) need_l0[g] = 1;
    }
    (void)hop;
    // job ranges per block of 256 output samples of this utterance
    const int nblk = (b -> max_ny + 255) / 256;
    const int j_lo = (int)job0, j_hi = (int)jobs.size();
    const size_t bj0 = blk_jobs.size();
    blk_jobs.resize(bj0 + (size_t)nblk, make_int2(j_hi, j_lo));
    for(int q = j_lo; q < j_hi; q ++) {                  // every job marks the blocks its samples [s0, s1) touch
      const int s1 = jobs[q].start + jobs[q].size;
      if(s1 <= 0) continue;
      // first block the job reaches; a pulse sample that (int) truncation also lands on output sample 0
      // (zero_extra >= 0) keeps block 0 in the range
      const int k0 = jobs[q].zero_extra >= 0 ? 0 : std::max(jobs[q].start, 0) / 256, k1 = std::min((s1 - 1) / 256, nblk - 1);
      for(int k = k0; k <= k1; k ++) {
        int2& e = blk_jobs[bj0 + (size_t)k];
        e.x = std::min(e.x, q); e.y = std::max(e.y, q + 1);
      }
    }
    for(int k = 0; k < nblk; k ++) { int2& e = blk_jobs[bj0 + (size_t)k]; if(e.x >= e.y) e = make_int2(0, 0); }
  };
  bool any_effect = false;
  for(size_t g = 0; g < F && ! any_effect; g ++) any_effect = b -> effects[g].modifier != nullptr;
  const int hw = (int)std::thread::hardware_concurrency();
  auto for_each_utt = [&](int nthr, const std::function<void(int)>& fn) {
    if(nthr <= 1) { for(int u = 0; u < L.n_utt; u ++) fn(u); return; }
    // the process-wide parked threads (SchedPool); a second host thread scheduling at the same moment starts its own
    if(sched_pool_run(nthr, L.n_utt, fn)) return;
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    for(int t = 0; t < nthr; t ++)
      pool.emplace_back([&] { for(int u; (u = next.fetch_add(1)) < L.n_utt; ) fn(u); });
    for(auto& th : pool) th.join();
  };
  // (at most 12 threads: a container's CPU quota is usually far below hardware_concurrency(), and 32 threads created
  // twice per call on 16 CPUs' worth of quota cost more in creation, joins and throttling than they scheduled)
  static const int nthr_cap = [] { const char* e = std::getenv("LLSM_L1_THREADS"); const int v = e ? std::atoi(e) : 12; return v >= 1 && v <= 16 ? v : 12; }();
  const int nthr_utt = std::max(1, std::min(std::min(std::max(hw / 2, 1), nthr_cap), L.n_utt / 16));
  for_each_utt(any_effect ? 1 : nthr_utt, schedule_utt);
  const auto t_1c = now();
  // concatenation: pulse, sample and job indices become global.  Offsets by prefix sums, then every utterance copies
  // its tables into place (the tables of 1024 utterances are 16 MB: a serial append was 3 of the scheduler's 4 ms);
  // the host arrays stay with the batch, so that a repeated call does not pay for their initialisation again.
  const size_t U = (size_t)L.n_utt;
  std::vector<size_t> job_o(U + 1, 0), pulse_o(U + 1, 0), seg_o(U + 1, 0), blk_o(U + 1, 0), ptot_o(U + 1, 0);
  for(size_t u = 0; u < U; u ++) {
    const UttPlan& pl = plans[u];
    job_o[u + 1] = job_o[u] + pl.jobs.size(); pulse_o[u + 1] = pulse_o[u] + pl.pulses.size();
    seg_o[u + 1] = seg_o[u] + pl.segs.size(); blk_o[u + 1] = blk_o[u] + pl.blk.size();
    ptot_o[u + 1] = ptot_o[u] + pl.pulse_total; size_max = std::max(size_max, pl.size_max);
    blk_off[u] = (int)blk_o[u];
  }
  pulse_total = ptot_o[U];
  if(ptot_o[U] > 0x7fffffffull || job_o[U] > 0x7fffffffull || pulse_o[U] > 0x7fffffffull) {
    llsm_set_error("use_l1 synthesis: pulse tables exceed 2^31 entries; split the batch"); return -1;
  }
  auto& jobs_h = b -> h_jobs; auto& pulses_h = b -> h_pulses;
  auto& segs_h = b -> h_segs; auto& blk_h = b -> h_blk;
  if((jobs_h.size() < job_o[U] && ! jobs_h.resize(job_o[U])) || (pulses_h.size() < pulse_o[U] && ! pulses_h.resize(pulse_o[U])) ||
     (segs_h.size() < seg_o[U] && ! segs_h.resize(seg_o[U])) || (blk_h.size() < blk_o[U] && ! blk_h.resize(blk_o[U]))) return -1;
  for_each_utt(nthr_utt, [&](int u) {
    const UttPlan& pl = plans[(size_t)u];
    const int job_base = (int)job_o[u], pulse_base = (int)pulse_o[u], out_base = (int)ptot_o[u];
    PbpJob* jd = jobs_h.data() + job_o[u];
    for(size_t q = 0; q < pl.jobs.size(); q ++) { PbpJob j = pl.jobs[q]; j.first += pulse_base; j.out_off += out_base; jd[q] = j; }
    if(! pl.pulses.empty()) std::memcpy(pulses_h.data() + pulse_o[u], pl.pulses.data(), pl.pulses.size() * sizeof(PbpPulse));
    if(! pl.segs.empty()) std::memcpy(segs_h.data() + seg_o[u], pl.segs.data(), pl.segs.size() * sizeof(PbpSeg));
    int2* bd = blk_h.data() + blk_o[u];
    for(size_t q = 0; q < pl.blk.size(); q ++) { int2 e = pl.blk[q]; if(e.x < e.y) { e.x += job_base; e.y += job_base; } bd[q] = e; }
  });
  const size_t n_jobs = job_o[U], n_pulses = pulse_o[U], n_segs = seg_o[U], n_blk = blk_o[U];
  for(size_t g = 0; g < F && ! any_need_l0; g ++) any_need_l0 = need_l0[g] != 0;
  blk_off[L.n_utt] = (int)n_blk;
  const auto t_2 = now();
  // ---- device work
  hipSetDevice(c -> device);
  LaunchCtx* P = & c -> lc;
  int tw_nmax = 0; const float2* tw = llsm_engine_twiddles(c, & tw_nmax);
  if(l1_alpha_cache(b)) return -1;
  L1Dev d = l1_dev(b);
  if(upload_vec(b -> l1_f0_hm, f0_hm) || upload_arr(b -> l1_jobs, jobs_h.data(), n_jobs) ||
     upload_arr(b -> l1_pulses, pulses_h.data(), n_pulses) || upload_arr(b -> l1_segs, segs_h.data(), n_segs) ||
     upload_arr(b -> l1_blk_jobs, blk_h.data(), n_blk) || upload_vec(b -> l1_blk_off, blk_off) ||
     b -> l1_pulse_buf.alloc(std::max<size_t>(pulse_total, 1)) || b -> l1_mixw.alloc(std::max<size_t>(Y, 1)) ||
     b -> l1_hm_frames.alloc(std::max<size_t>(F * (size_t)nwin, 1))) return -1;
  if(any_need_l0) {
    if(upload_vec(b -> l1_select, need_l0)) return -1;
    RUN1(launch_l1_to_l0(P, d, b -> maxnhar_conf, 1, b -> l1_select.p, tw, tw_nmax));
  }
  HIP_OK(hipMemsetAsync(b -> l1_mixw.p, 0, std::max<size_t>(Y, 1) * sizeof(float), c -> stream));
  RUN1(launch_l1_mixcurve(P, b -> l1_segs.p, (int)n_segs, b -> l1_mixw.p));
  // harmonic frames of the selected frames (no fractional-hop phase term on this path, layer0.c:268-277)
  {
    BatchDev bd; std::memset(& bd, 0, sizeof(bd));
    bd.n_utt = L.n_utt; bd.nframes = L.total_frames; bd.maxnhar = L.maxnhar; bd.thop = thopf; bd.fs = fsf;
    bd.frm_utt = b -> d_frm_utt.p; bd.frm_off = b -> d_frm_off.p;
    bd.f0 = b -> l1_f0_hm.p; bd.nhar = d.nhar; bd.ampl = d.ampl; bd.phse = d.phse;
    if(b -> l1_zero.alloc(std::max<size_t>(F, 1))) return -1;
    HIP_OK(hipMemsetAsync(b -> l1_zero.p, 0, std::max<size_t>(F, 1) * sizeof(float), c -> stream));
    RUN1(launch_synth_frames(P, bd, nwin, b -> win_sin.p, b -> l1_zero.p, b -> l1_hm_frames.p, std::min(L.maxnhar, 2048)));
  }
  RUN1(launch_pbp_pulse(P, d, b -> l1_jobs.p, (int)n_jobs, b -> l1_pulses.p, size_max, fsf, tw, tw_nmax, b -> l1_pulse_buf.p));
  RUN1(launch_pbp_mix(P, L.n_utt, b -> max_ny, b -> d_y_off.p, b -> d_ny.p, b -> d_frm_off.p, b -> d_nfrm.p, thopf, fsf,
    nwin, b -> l1_hm_frames.p, b -> l1_f0_hm.p, b -> l1_jobs.p, b -> l1_blk_jobs.p, b -> l1_blk_off.p, b -> l1_pulse_buf.p,
    b -> l1_mixw.p, ynoise, ysin, yout));
  if(timing)
    std::fprintf(stderr, "[l1 synth] projections + rows down %.2f ms, schedule %.2f ms + merge %.2f ms (%zu jobs, %zu pulses), upload + launches %.2f ms\n",
      ms(t_0, t_1), ms(t_1b, t_1c), ms(t_1c, t_2), n_jobs, n_pulses, ms(t_2, now()));
  return 0;
}

// ------------------------------------------------------------------ flat layer-1 rows <-> chunk
extern "C" int llsm_chunk_to_flat_l1(llsm_chunk* src, llsm_flat_l1* dst, int frm_off) {
  int* nfrm = (int*)llsm_container_get(src -> conf, LLSM_CONF_NFRM);
  if(! nfrm) return -1;
  for(int i = 0; i < *nfrm; i ++) {
    llsm_container* fr = src -> frames[i];
    const size_t g = (size_t)frm_off + i;
    FP_TYPE* rd = (FP_TYPE*)llsm_container_get(fr, LLSM_FRAME_RD);
    FP_TYPE* vt = (FP_TYPE*)llsm_container_get(fr, LLSM_FRAME_VTMAGN);
    FP_TYPE* vs = (FP_TYPE*)llsm_container_get(fr, LLSM_FRAME_VSPHSE);
    int* pbp = (int*)llsm_container_get(fr, LLSM_FRAME_PBPSYN);
    dst -> rd[g] = rd ? *rd : 0;
    dst -> has_rd[g] = rd != NULL;
    dst -> pbpsyn[g] = pbp ? *pbp : 0;
    dst -> has_hm[g] = llsm_container_get(fr, LLSM_FRAME_HM) != NULL;
    const bool l1 = rd && vt && vs;
    int n = l1 ? llsm_fparray_length(vs) : 0;
    if(n > dst -> maxnhar) n = dst -> maxnhar;
    dst -> nvsphse[g] = (l1 && n > 0) ? n : 0;
    for(int k = 0; k < dst -> maxnhar; k ++) dst -> vsphse[g * dst -> maxnhar + k] = (l1 && k < n) ? vs[k] : 0;
    const int nv = l1 ? llsm_fparray_length(vt) : 0;
    for(int k = 0; k < dst -> nspec; k ++) dst -> vtmagn[g * dst -> nspec + k] = k < nv ? vt[k] : 0;
  }
  return 0;
}

extern "C" int llsm_flat_l1_to_chunk(const llsm_flat_l1* src, int frm_off, llsm_chunk* dst) {
  int* nfrm = (int*)llsm_container_get(dst -> conf, LLSM_CONF_NFRM);
  if(! nfrm) return -1;
  for(int i = 0; i < *nfrm; i ++) {
    llsm_container* fr = dst -> frames[i];
    const size_t g = (size_t)frm_off + i;
    if(src -> has_rd[g])
      llsm_container_attach_(fr, LLSM_FRAME_RD, llsm_create_fp(src -> rd[g]), (llsm_fdestructor)llsm_delete_fp,
        (llsm_fcopy)llsm_copy_fp);
    if(src -> pbpsyn[g])
      llsm_container_attach_(fr, LLSM_FRAME_PBPSYN, llsm_create_int(src -> pbpsyn[g]), (llsm_fdestructor)llsm_delete_int,
        (llsm_fcopy)llsm_copy_int);
    const int n = src -> nvsphse[g];
    if(n <= 0) continue;
    FP_TYPE* vt = llsm_create_fparray(src -> nspec); FP_TYPE* vs = llsm_create_fparray(n);
    std::memcpy(vt, src -> vtmagn + g * src -> nspec, sizeof(FP_TYPE) * (size_t)src -> nspec);
    std::memcpy(vs, src -> vsphse + g * src -> maxnhar, sizeof(FP_TYPE) * (size_t)n);
    llsm_container_attach_(fr, LLSM_FRAME_VTMAGN, vt, (llsm_fdestructor)llsm_delete_fparray, (llsm_fcopy)llsm_copy_fparray);
    llsm_container_attach_(fr, LLSM_FRAME_VSPHSE, vs, (llsm_fdestructor)llsm_delete_fparray, (llsm_fcopy)llsm_copy_fparray);
  }
  return 0;
}

// ------------------------------------------------------------------ chunk API (layer1.c)
namespace {
struct L1Host {
  int F = 0, nspec = 0, maxnhar = 0;
  std::vector<float> rd, vtmagn, vsphse; std::vector<int> nvs, pbpsyn, has_hm, has_rd;
  void resize(int F_, int nspec_, int maxnhar_) {
    F = F_; nspec = nspec_; maxnhar = maxnhar_;
    rd.assign(F, 0); vtmagn.assign((size_t)F * nspec, 0); vsphse.assign((size_t)F * maxnhar, 0);
    nvs.assign(F, 0); pbpsyn.assign(F, 0); has_hm.assign(F, 1); has_rd.assign(F, 0);
  }
  llsm_flat_l1 view() {
    llsm_flat_l1 v; v.nspec = nspec; v.maxnhar = maxnhar; v.rd = rd.data(); v.has_rd = has_rd.data();
    v.vtmagn = vtmagn.data(); v.vsphse = vsphse.data(); v.nvsphse = nvs.data(); v.pbpsyn = pbpsyn.data();
    v.has_hm = has_hm.data();
    return v;
  }
};

// layer-0 rows a layer-1 conversion needs: f0, nhar, ampl, phse
struct HmHost { std::vector<float> f0, ampl, phse; std::vector<int> nhar; };

int chunk_nfrm_(llsm_chunk* c) { int* n = (int*)llsm_container_get(c -> conf, LLSM_CONF_NFRM); return n ? *n : -1; }

// a parameter-only batch of one utterance holding the HM rows of `frames`
llsm_gpu_batch* hm_batch(llsm_container** frames, int nfrm, llsm_container* conf, int maxnhar, float fnyq, float thop,
  float lip_radius, HmHost& h) {
  llsm_gpu_context* ctx = llsm_default_context();
  if(! ctx) return nullptr;
  llsm_aoptions ao; std::memset(& ao, 0, sizeof(ao));
  int* npsd = (int*)llsm_container_get(conf, LLSM_CONF_NPSD);
  ao.thop = thop; ao.maxnhar = std::max(maxnhar, 1); ao.maxnhar_e = 0; ao.npsd = npsd ? std::max(*npsd, 2) : 2; ao.nchannel = 1;
  ao.lip_radius = lip_radius; ao.hm_method = LLSM_AOPTION_HMCZT; ao.rel_winsize = 4;
  int zero = 0;
  llsm_gpu_batch* b = llsm_gpu_create_batch(ctx, & ao, fnyq * 2, 1, & zero, & nfrm);
  if(! b) return nullptr;
  h.f0.assign(nfrm, 0); h.nhar.assign(nfrm, 0);
  h.ampl.assign((size_t)nfrm * ao.maxnhar, 0); h.phse.assign((size_t)nfrm * ao.maxnhar, 0);
  for(int i = 0; i < nfrm; i ++) {
    FP_TYPE* f0 = (FP_TYPE*)llsm_container_get(frames[i], LLSM_FRAME_F0);
    llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(frames[i], LLSM_FRAME_HM);
    h.f0[i] = f0 ? *f0 : 0;
    const int n = hm ? std::min(hm -> nhar, ao.maxnhar) : 0;
    h.nhar[i] = n;
    for(int k = 0; k < n; k ++) { h.ampl[(size_t)i * ao.maxnhar + k] = hm -> ampl[k]; h.phse[(size_t)i * ao.maxnhar + k] = hm -> phse[k]; }
  }
  int rc = llsm_gpu_batch_upload(b, LLSM_GPU_F0, h.f0.data(), h.f0.size() * 4);
  rc |= llsm_gpu_batch_upload(b, LLSM_GPU_NHAR, h.nhar.data(), h.nhar.size() * 4);
  rc |= llsm_gpu_batch_upload(b, LLSM_GPU_AMPL, h.ampl.data(), h.ampl.size() * 4);
  rc |= llsm_gpu_batch_upload(b, LLSM_GPU_PHSE, h.phse.data(), h.phse.size() * 4);
  if(rc) { llsm_gpu_delete_batch(b); return nullptr; }
  return b;
}
}  // namespace

int llsm_l1_upload(llsm_gpu_batch* b, L1Host& h) {
  int rc = 0;
#define UL1(id, vec) rc |= llsm_gpu_batch_upload(b, id, vec.data(), llsm_gpu_batch_array_bytes(b, id))
  UL1(LLSM_GPU_RD, h.rd); UL1(LLSM_GPU_VTMAGN, h.vtmagn); UL1(LLSM_GPU_VSPHSE, h.vsphse);
  UL1(LLSM_GPU_NVSPHSE, h.nvs); UL1(LLSM_GPU_PBPSYN, h.pbpsyn); UL1(LLSM_GPU_HAS_HM, h.has_hm);
#undef UL1
  return rc;
}

extern "C" int llsm_conf_checklayer1(llsm_container* src) {
  // layer1.c:40-46 + the layer-0 members
  if(! llsm_conf_checklayer0(src)) return 0;
  return llsm_container_get(src, LLSM_CONF_LIPRADIUS) != NULL && llsm_container_get(src, LLSM_CONF_NSPEC) != NULL;
}

extern "C" void llsm_chunk_tolayer1(llsm_chunk* dst, int nfft) {
  // layer1.c:26-38: integrity, else silently return
  int* nfrm_p = (int*)llsm_container_get(dst -> conf, LLSM_CONF_NFRM);
  FP_TYPE* thop = (FP_TYPE*)llsm_container_get(dst -> conf, LLSM_CONF_THOP);
  FP_TYPE* fnyq = (FP_TYPE*)llsm_container_get(dst -> conf, LLSM_CONF_FNYQ);
  FP_TYPE* liprad = (FP_TYPE*)llsm_container_get(dst -> conf, LLSM_CONF_LIPRADIUS);
  if(! nfrm_p || ! thop || ! fnyq || ! liprad) return;
  const int nfrm = *nfrm_p;
  for(int i = 0; i < nfrm; i ++) if(! llsm_frame_checklayer0(dst -> frames[i])) return;
  llsm_container_attach_(dst -> conf, LLSM_CONF_NSPEC, llsm_create_int(nfft / 2 + 1), (llsm_fdestructor)llsm_delete_int,
    (llsm_fcopy)llsm_copy_int);
  int maxnhar = 1;
  for(int i = 0; i < nfrm; i ++) {
    llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(dst -> frames[i], LLSM_FRAME_HM);
    if(hm) maxnhar = std::max(maxnhar, hm -> nhar);
  }
  HmHost h;
  llsm_gpu_batch* b = hm_batch(dst -> frames, nfrm, dst -> conf, maxnhar, *fnyq, *thop, *liprad, h);
  if(! b) return;
  L1Host o; const int nspec = nfft / 2 + 1;
  int rc = llsm_gpu_batch_tolayer1(b, nfft);
  if(! rc) {
    o.resize(nfrm, nspec, maxnhar);
    rc |= llsm_gpu_batch_download(b, LLSM_GPU_RD, o.rd.data(), o.rd.size() * 4);
    rc |= llsm_gpu_batch_download(b, LLSM_GPU_VTMAGN, o.vtmagn.data(), o.vtmagn.size() * 4);
    rc |= llsm_gpu_batch_download(b, LLSM_GPU_VSPHSE, o.vsphse.data(), o.vsphse.size() * 4);
    rc |= llsm_gpu_batch_download(b, LLSM_GPU_NVSPHSE, o.nvs.data(), o.nvs.size() * 4);
  }
  llsm_gpu_delete_batch(b);
  if(rc) return;
  std::fill(o.has_rd.begin(), o.has_rd.end(), 1);       // RD goes onto every frame (layer1.c:141-143)
  llsm_flat_l1 v = o.view();
  llsm_flat_l1_to_chunk(& v, 0, dst);
}

// frames[0..n): layer-1 members -> HM (llsm_frame_tolayer0 on each), results attached to the frames
static void frames_tolayer0(llsm_container** frames, int n, llsm_container* conf) {
  FP_TYPE* fnyq = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_FNYQ);
  FP_TYPE* liprad = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_LIPRADIUS);
  int* nspec = (int*)llsm_container_get(conf, LLSM_CONF_NSPEC);
  if(! fnyq || ! liprad || ! nspec) return;             // layer1.c:40-46
  int* maxnhar_c = (int*)llsm_container_get(conf, LLSM_CONF_MAXNHAR);
  FP_TYPE* thop = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_THOP);
  std::vector<int> pick;
  int maxnhar = 1;
  for(int i = 0; i < n; i ++) {
    if(! llsm_frame_checklayer1(frames[i])) continue;
    FP_TYPE* f0 = (FP_TYPE*)llsm_container_get(frames[i], LLSM_FRAME_F0);
    FP_TYPE* vs = (FP_TYPE*)llsm_container_get(frames[i], LLSM_FRAME_VSPHSE);
    FP_TYPE* vt = (FP_TYPE*)llsm_container_get(frames[i], LLSM_FRAME_VTMAGN);
    if(*f0 == 0 || ! vs || ! vt || llsm_fparray_length(vt) < *nspec) continue;
    pick.push_back(i);
    maxnhar = std::max(maxnhar, llsm_fparray_length(vs));
  }
  if(pick.empty()) return;
  std::vector<llsm_container*> sel(pick.size());
  for(size_t k = 0; k < pick.size(); k ++) sel[k] = frames[pick[k]];
  HmHost h;
  llsm_gpu_batch* b = hm_batch(sel.data(), (int)sel.size(), conf, maxnhar, *fnyq, thop ? *thop : 0.005f, *liprad, h);
  if(! b) return;
  const int nfft = (*nspec - 1) * 2;
  llsm_chunk fake; fake.conf = llsm_create_container(1); fake.frames = sel.data();
  llsm_container_attach_(fake.conf, LLSM_CONF_NFRM, llsm_create_int((int)sel.size()), (llsm_fdestructor)llsm_delete_int,
    (llsm_fcopy)llsm_copy_int);
  L1Host o; o.resize((int)sel.size(), *nspec, maxnhar);
  llsm_flat_l1 v = o.view();
  int rc = llsm_gpu_batch_enable_layer1(b, nfft);
  if(! rc) rc = llsm_chunk_to_flat_l1(& fake, & v, 0);
  llsm_delete_container(fake.conf);
  std::fill(o.has_hm.begin(), o.has_hm.end(), 0);
  if(! rc) rc = llsm_l1_upload(b, o);
  if(! rc) rc = llsm_gpu_batch_set_maxnhar_conf(b, maxnhar_c ? *maxnhar_c : -1);
  if(! rc) rc = llsm_gpu_batch_tolayer0(b, 0);
  if(! rc) {
    rc |= llsm_gpu_batch_download(b, LLSM_GPU_NHAR, h.nhar.data(), h.nhar.size() * 4);
    rc |= llsm_gpu_batch_download(b, LLSM_GPU_AMPL, h.ampl.data(), h.ampl.size() * 4);
    rc |= llsm_gpu_batch_download(b, LLSM_GPU_PHSE, h.phse.data(), h.phse.size() * 4);
  }
  llsm_gpu_delete_batch(b);
  if(rc) return;
  const int mh = std::max(maxnhar, 1);
  for(size_t k = 0; k < sel.size(); k ++) {
    llsm_hmframe* hm = llsm_create_hmframe(h.nhar[k]);
    std::memcpy(hm -> ampl, h.ampl.data() + k * mh, sizeof(FP_TYPE) * (size_t)h.nhar[k]);
    std::memcpy(hm -> phse, h.phse.data() + k * mh, sizeof(FP_TYPE) * (size_t)h.nhar[k]);
    llsm_container_attach_(sel[k], LLSM_FRAME_HM, hm, (llsm_fdestructor)llsm_delete_hmframe, (llsm_fcopy)llsm_copy_hmframe);
  }
}

extern "C" void llsm_frame_tolayer0(llsm_container* dst, llsm_container* conf) { frames_tolayer0(& dst, 1, conf); }

extern "C" void llsm_chunk_tolayer0(llsm_chunk* dst) {
  const int nfrm = chunk_nfrm_(dst);
  if(nfrm > 0) frames_tolayer0(dst -> frames, nfrm, dst -> conf);
}

// ---- helpers for llsm_synthesize_batch (capi.cpp): layer-1 members of the chunks into the batch, effects,
// and the HM rows the synthesis built from layer 1 back onto the callers' frames (the reference attaches
// them as a side effect of llsm_synthesize: layer0.c:265-266)
int llsm_l1_prepare_batch(llsm_gpu_batch* b, llsm_chunk** src, int n_utt, const int* fo, int nspec) {
  if(llsm_gpu_batch_enable_layer1(b, (nspec - 1) * 2)) return -1;
  L1Host o; o.resize(b -> lay.total_frames, nspec, b -> lay.maxnhar);
  llsm_flat_l1 v = o.view();
  for(int u = 0; u < n_utt; u ++) {
    if(llsm_chunk_to_flat_l1(src[u], & v, fo[u])) return -1;
    const int nf = chunk_nfrm_(src[u]);
    for(int i = 0; i < nf; i ++) {
      llsm_pbpeffect* ef = (llsm_pbpeffect*)llsm_container_get(src[u] -> frames[i], LLSM_FRAME_PBPEFF);
      if(ef) llsm_gpu_batch_set_pbpeffect(b, fo[u] + i, ef -> modifier, ef -> info, src[u] -> frames[i]);
    }
  }
  int* mc = (int*)llsm_container_get(src[0] -> conf, LLSM_CONF_MAXNHAR);
  llsm_gpu_batch_set_maxnhar_conf(b, mc ? *mc : -1);
  b -> l1_had_hm = o.has_hm;
  return llsm_l1_upload(b, o);
}

int llsm_l1_writeback_hm(llsm_gpu_batch* b, llsm_chunk** src, int n_utt, const int* fo) {
  const size_t F = (size_t)b -> lay.total_frames, mh = (size_t)b -> lay.maxnhar;
  std::vector<int> has(F), nhar(F);
  if(llsm_gpu_batch_download(b, LLSM_GPU_HAS_HM, has.data(), F * 4) || llsm_gpu_batch_download(b, LLSM_GPU_NHAR, nhar.data(), F * 4)) return -1;
  bool any = false;
  for(size_t g = 0; g < F; g ++) if(has[g] && ! b -> l1_had_hm[g]) any = true;
  if(! any) return 0;
  std::vector<float> ampl(F * mh), phse(F * mh);
  if(llsm_gpu_batch_download(b, LLSM_GPU_AMPL, ampl.data(), ampl.size() * 4) || llsm_gpu_batch_download(b, LLSM_GPU_PHSE, phse.data(), phse.size() * 4)) return -1;
  for(int u = 0; u < n_utt; u ++) {
    const int nf = chunk_nfrm_(src[u]);
    for(int i = 0; i < nf; i ++) {
      const size_t g = (size_t)fo[u] + i;
      if(! has[g] || b -> l1_had_hm[g]) continue;
      llsm_hmframe* hm = llsm_create_hmframe(nhar[g]);
      std::memcpy(hm -> ampl, ampl.data() + g * mh, sizeof(FP_TYPE) * (size_t)nhar[g]);
      std::memcpy(hm -> phse, phse.data() + g * mh, sizeof(FP_TYPE) * (size_t)nhar[g]);
      llsm_container_attach_(src[u] -> frames[i], LLSM_FRAME_HM, hm, (llsm_fdestructor)llsm_delete_hmframe, (llsm_fcopy)llsm_copy_hmframe);
    }
  }
  return 0;
}

// ------------------------------------------------------------------ blob -> batch rows (no container tree)
// Many blobs at once: utterances [utt0, utt0 + n) of the batch from blobs[0 .. n).  The rows of a group of blobs are
// gathered (with the row-width conversion) into one page-locked staging area, array after array, and go to the device
// as ONE copy per array and group -- a blob at a time costs eleven small pageable copies and a synchronisation each.
extern "C" int llsm_gpu_batch_upload_blobs(llsm_gpu_batch* b, int utt0, int n, const void* const* blobs, const size_t* bytes) {
  if(! b || utt0 < 0 || n < 0 || utt0 + n > b -> lay.n_utt) { llsm_set_error("llsm_gpu_batch_upload_blobs: utterances out of range"); return -1; }
  if(n == 0) return 0;
  const llsm_gpu_layout& L = b -> lay;
  const int me = std::max(L.maxnhar_e, 1);
  std::vector<llsm_flat_params> V((size_t)n); std::vector<llsm_flat_l1> Q((size_t)n);
  int nspec = 0;
  for(int k = 0; k < n; k ++) {
    int nfrm = 0;
    if(llsm_blob_view(blobs[k], bytes[k], & V[k], & nfrm, nullptr, nullptr) || llsm_blob_view_l1(blobs[k], bytes[k], & Q[k])) return -1;
    const llsm_flat_params& v = V[k];
    if(nfrm != b -> nfrm[utt0 + k] || v.npsd != L.npsd || v.nchannel != L.nchannel || v.maxnhar > L.maxnhar || v.maxnhar_e > L.maxnhar_e) {
      llsm_set_error("llsm_gpu_batch_upload_blobs: blob " + std::to_string(k) + " does not fit the batch (frames / npsd / nchannel / row widths)");
      return -1;
    }
    if(k == 0) nspec = Q[k].nspec;
    else if(Q[k].nspec != nspec) { llsm_set_error("llsm_gpu_batch_upload_blobs: blobs with and without layer-1 rows (or different NSPEC) in one call"); return -1; }
  }
  if(nspec > 0 && llsm_gpu_batch_enable_layer1(b, (nspec - 1) * 2)) return -1;
  if(nspec > 0 && nspec != b -> l1_nspec) { llsm_set_error("llsm_gpu_batch_upload_blobs: NSPEC differs from the batch"); return -1; }
  hipSetDevice(b -> ctx -> device);
  hipStream_t st = b -> ctx -> stream;
  // destination rows (floats / ints of 4 bytes) per frame and array
  struct Col { int id; size_t w; };
  std::vector<Col> cols = {{LLSM_GPU_F0, 1}, {LLSM_GPU_NHAR, 1}, {LLSM_GPU_AMPL, (size_t)L.maxnhar}, {LLSM_GPU_PHSE, (size_t)L.maxnhar},
    {LLSM_GPU_PSD, (size_t)L.npsd}, {LLSM_GPU_PSDRES, (size_t)L.npsd}, {LLSM_GPU_HAS_PSDRES, 1}, {LLSM_GPU_EDC, (size_t)L.nchannel},
    {LLSM_GPU_NHAR_E, 1}, {LLSM_GPU_EENV_AMPL, (size_t)L.nchannel * me}, {LLSM_GPU_EENV_PHSE, (size_t)L.nchannel * me}};
  if(nspec > 0)
    for(Col c : {Col{LLSM_GPU_RD, 1}, Col{LLSM_GPU_VTMAGN, (size_t)nspec}, Col{LLSM_GPU_VSPHSE, (size_t)L.maxnhar}, Col{LLSM_GPU_NVSPHSE, 1},
                 Col{LLSM_GPU_PBPSYN, 1}, Col{LLSM_GPU_HAS_HM, 1}}) cols.push_back(c);
  size_t wsum = 0; for(const Col& c : cols) wsum += c.w;
  const size_t stage_bytes = (size_t)64 << 20;
  if(! b -> blob_stage && hipHostMalloc(& b -> blob_stage, stage_bytes, hipHostMallocDefault) != hipSuccess) {
    b -> blob_stage = nullptr; llsm_set_error("llsm_gpu_batch_upload_blobs: page-locked staging allocation failed"); return -1;
  }
  auto src_of = [&](int k, int id, size_t* w_src) -> const void* {
    const llsm_flat_params& v = V[k]; const llsm_flat_l1& q = Q[k];
    const size_t meb = (size_t)std::max(v.maxnhar_e, 1);
    switch(id) {
      case LLSM_GPU_F0: *w_src = 1; return v.f0;
      case LLSM_GPU_NHAR: *w_src = 1; return v.nhar;
      case LLSM_GPU_AMPL: *w_src = v.maxnhar; return v.ampl;
      case LLSM_GPU_PHSE: *w_src = v.maxnhar; return v.phse;
      case LLSM_GPU_PSD: *w_src = v.npsd; return v.psd;
      case LLSM_GPU_PSDRES: *w_src = v.npsd; return v.psdres;
      case LLSM_GPU_HAS_PSDRES: *w_src = 1; return v.has_psdres;
      case LLSM_GPU_EDC: *w_src = v.nchannel; return v.edc;
      case LLSM_GPU_NHAR_E: *w_src = 1; return v.nhar_e;
      case LLSM_GPU_EENV_AMPL: *w_src = meb; return v.eenv_ampl;      // rows of one (frame, channel)
      case LLSM_GPU_EENV_PHSE: *w_src = meb; return v.eenv_phse;
      case LLSM_GPU_RD: *w_src = 1; return q.rd;
      case LLSM_GPU_VTMAGN: *w_src = q.nspec; return q.vtmagn;
      case LLSM_GPU_VSPHSE: *w_src = q.maxnhar; return q.vsphse;
      case LLSM_GPU_NVSPHSE: *w_src = 1; return q.nvsphse;
      case LLSM_GPU_PBPSYN: *w_src = 1; return q.pbpsyn;
      case LLSM_GPU_HAS_HM: *w_src = 1; return q.has_hm;
    }
    *w_src = 0; return nullptr;
  };
  int rc = 0;
  for(int k0 = 0; k0 < n; ) {
    int k1 = k0; size_t Fg = 0;                           // group of blobs that fits the staging area (at least one)
    while(k1 < n && (k1 == k0 || (Fg + (size_t)b -> nfrm[utt0 + k1]) * wsum * 4 <= stage_bytes)) { Fg += (size_t)b -> nfrm[utt0 + k1]; k1 ++; }
    if(Fg * wsum * 4 > stage_bytes) { llsm_set_error("llsm_gpu_batch_upload_blobs: one utterance exceeds the staging area"); return -1; }
    const size_t fo = (size_t)b -> frm_off[utt0 + k0];
    char* sp = (char*)b -> blob_stage;
    for(const Col& c : cols) {
      char* base = sp;
      for(int k = k0; k < k1; k ++) {
        const size_t F = (size_t)b -> nfrm[utt0 + k];
        size_t ws = 0; const char* src = (const char*)src_of(k, c.id, & ws);
        const bool env = c.id == LLSM_GPU_EENV_AMPL || c.id == LLSM_GPU_EENV_PHSE;
        const size_t rows = env ? F * (size_t)L.nchannel : F, wd = env ? (size_t)me : c.w;     // destination row of `wd` floats
        if(! src || ws == 0) std::memset(sp, 0, rows * wd * 4);
        else if(ws == wd) std::memcpy(sp, src, rows * wd * 4);
        else for(size_t r2 = 0; r2 < rows; r2 ++) {
          std::memcpy(sp + r2 * wd * 4, src + r2 * ws * 4, ws * 4);
          std::memset(sp + (r2 * wd + ws) * 4, 0, (wd - ws) * 4);
        }
        sp += rows * wd * 4;
      }
      rc |= hipMemcpyAsync((char*)b -> arr[c.id] + fo * c.w * 4, base, (size_t)(sp - base), hipMemcpyHostToDevice, st) != hipSuccess;
    }
    if(hipStreamSynchronize(st) != hipSuccess) rc = 1;     // the staging area is reused by the next group
    if(rc) { llsm_set_error("llsm_gpu_batch_upload_blobs: copy failed"); return -1; }
    k0 = k1;
  }
  float m = b -> min_f0;
  for(int k = 0; k < n; k ++)
    for(int i = 0; i < b -> nfrm[utt0 + k]; i ++) { const float f = V[k].f0[i]; if(f > 0 && (m == 0 || f < m)) m = f; }
  if(! b -> f0_unknown) b -> min_f0 = m;                // rows written through the device pointer are not in `m`: stay unknown
  return 0;
}

extern "C" int llsm_gpu_batch_upload_blob(llsm_gpu_batch* b, int utt, const void* blob, size_t bytes) {
  llsm_flat_params v; llsm_flat_l1 q; int nfrm = 0;
  if(! b || utt < 0 || utt >= b -> lay.n_utt) { llsm_set_error("llsm_gpu_batch_upload_blob: utterance out of range"); return -1; }
  if(llsm_blob_view(blob, bytes, & v, & nfrm, nullptr, nullptr) || llsm_blob_view_l1(blob, bytes, & q)) return -1;
  const llsm_gpu_layout& L = b -> lay;
  const int me_b = std::max(v.maxnhar_e, 1), me = std::max(L.maxnhar_e, 1);
  if(nfrm != b -> nfrm[utt] || v.npsd != L.npsd || v.nchannel != L.nchannel || v.maxnhar > L.maxnhar || v.maxnhar_e > L.maxnhar_e) {
    llsm_set_error("llsm_gpu_batch_upload_blob: blob shape does not fit the batch (frames / npsd / nchannel / row widths)"); return -1;
  }
  if(q.nspec > 0 && llsm_gpu_batch_enable_layer1(b, (q.nspec - 1) * 2)) return -1;
  if(q.nspec > 0 && q.nspec != b -> l1_nspec) { llsm_set_error("llsm_gpu_batch_upload_blob: NSPEC differs from the batch"); return -1; }
  hipSetDevice(b -> ctx -> device);
  hipStream_t st = b -> ctx -> stream;
  const size_t fo = (size_t)b -> frm_off[utt], F = (size_t)nfrm;
  if(F == 0) return 0;
  // rows: (array, source, source row floats, destination row floats)
  auto rows = [&](int id, const void* src, size_t w_src, size_t w_dst) -> int {
    if(w_src == 0 || ! src) return 0;
    char* dst = (char*)b -> arr[id] + fo * w_dst * 4;
    return hipMemcpy2DAsync(dst, w_dst * 4, src, w_src * 4, w_src * 4, F, hipMemcpyHostToDevice, st) != hipSuccess;
  };
  int rc = 0;
  if(v.maxnhar < L.maxnhar) {                           // rows narrower than the batch: clear the tails first
    rc |= hipMemsetAsync((float*)b -> arr[LLSM_GPU_AMPL] + fo * L.maxnhar, 0, F * L.maxnhar * 4, st) != hipSuccess;
    rc |= hipMemsetAsync((float*)b -> arr[LLSM_GPU_PHSE] + fo * L.maxnhar, 0, F * L.maxnhar * 4, st) != hipSuccess;
  }
  rc |= rows(LLSM_GPU_F0, v.f0, 1, 1); rc |= rows(LLSM_GPU_NHAR, v.nhar, 1, 1);
  rc |= rows(LLSM_GPU_AMPL, v.ampl, v.maxnhar, L.maxnhar); rc |= rows(LLSM_GPU_PHSE, v.phse, v.maxnhar, L.maxnhar);
  rc |= rows(LLSM_GPU_PSD, v.psd, v.npsd, L.npsd); rc |= rows(LLSM_GPU_PSDRES, v.psdres, v.npsd, L.npsd);
  rc |= rows(LLSM_GPU_HAS_PSDRES, v.has_psdres, 1, 1); rc |= rows(LLSM_GPU_EDC, v.edc, v.nchannel, L.nchannel);
  rc |= rows(LLSM_GPU_NHAR_E, v.nhar_e, 1, 1);
  if(me_b == me) { rc |= rows(LLSM_GPU_EENV_AMPL, v.eenv_ampl, (size_t)v.nchannel * me, (size_t)L.nchannel * me);
                   rc |= rows(LLSM_GPU_EENV_PHSE, v.eenv_phse, (size_t)v.nchannel * me, (size_t)L.nchannel * me); }
  else {                                                // [F][nch][me_b] -> [F][nch][me]: one strided copy per (frame, channel) row
    rc |= hipMemsetAsync((float*)b -> arr[LLSM_GPU_EENV_AMPL] + fo * L.nchannel * me, 0, F * L.nchannel * me * 4, st) != hipSuccess;
    rc |= hipMemsetAsync((float*)b -> arr[LLSM_GPU_EENV_PHSE] + fo * L.nchannel * me, 0, F * L.nchannel * me * 4, st) != hipSuccess;
    rc |= hipMemcpy2DAsync((float*)b -> arr[LLSM_GPU_EENV_AMPL] + fo * L.nchannel * me, (size_t)me * 4, v.eenv_ampl, (size_t)me_b * 4,
      (size_t)me_b * 4, F * L.nchannel, hipMemcpyHostToDevice, st) != hipSuccess;
    rc |= hipMemcpy2DAsync((float*)b -> arr[LLSM_GPU_EENV_PHSE] + fo * L.nchannel * me, (size_t)me * 4, v.eenv_phse, (size_t)me_b * 4,
      (size_t)me_b * 4, F * L.nchannel, hipMemcpyHostToDevice, st) != hipSuccess;
  }
  if(q.nspec > 0) {
    rc |= rows(LLSM_GPU_RD, q.rd, 1, 1); rc |= rows(LLSM_GPU_VTMAGN, q.vtmagn, q.nspec, q.nspec);
    rc |= hipMemsetAsync((float*)b -> arr[LLSM_GPU_VSPHSE] + fo * L.maxnhar, 0, F * L.maxnhar * 4, st) != hipSuccess;
    rc |= rows(LLSM_GPU_VSPHSE, q.vsphse, q.maxnhar, L.maxnhar); rc |= rows(LLSM_GPU_NVSPHSE, q.nvsphse, 1, 1);
    rc |= rows(LLSM_GPU_PBPSYN, q.pbpsyn, 1, 1); rc |= rows(LLSM_GPU_HAS_HM, q.has_hm, 1, 1);
  }
  if(hipStreamSynchronize(st) != hipSuccess) rc = 1;     // the blob is the caller's (pageable) memory
  if(rc) { llsm_set_error("llsm_gpu_batch_upload_blob: copy failed"); return -1; }
  float m = b -> min_f0;
  for(size_t i = 0; i < F; i ++) if(v.f0[i] > 0 && (m == 0 || v.f0[i] < m)) m = v.f0[i];
  if(! b -> f0_unknown) b -> min_f0 = m;                // rows written through the device pointer are not in `m`: stay unknown
  return 0;
}
