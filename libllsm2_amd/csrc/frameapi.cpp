// frameapi.cpp -- the reference's installed per-frame API (include/dsputils.h, include/llsmutils.h) on the
// default device.  Every function stages its (small) arguments, runs the kernels of frame_kernels.hip /
// l1_kernels.hip or the batch engine, and copies the result back; trivially elementwise helpers
// (llsm_fft_to_psd, the frequency-axis helpers, the lip filter, the smoothing filter) are plain host code, like
// the container / frame helpers of model.cpp.  There is no CPU fallback for the transforms: without a device
// the outputs stay zero / NULL and llsm_gpu_last_error() holds the reason.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "batch.h"
#include "dsputils.h"
#include "lfmodel.h"
#include "llsmutils.h"
#include "plan.h"

namespace lp = llsm_plan;
namespace lf = llsm_lf;

int launch_fa_czt(LaunchCtx* P, const float* x, int nx, double turn0, int nhar, float* ampl, float* phse);
int launch_fa_harm_frame(LaunchCtx* P, const float* ampl, const float* phse, int nhar, double f0n, int nx, float* y);
int launch_fa_stft(LaunchCtx* P, const float* x, int nx, const int* center, const int* winsize, int nfrm, int nfft,
  int blackman, int mode, const float* scale, const float2* tw, int tw_nmax, float* spec, float* phse);
int launch_fa_peakpick(LaunchCtx* P, const float* spectrum, const float* phase, int nfft, float fs, int nhar, float f0,
  float* ampl, float* phse);
int launch_fa_dc(LaunchCtx* P, const float* x, int nx, const int* center, const int* winsize, int nfrm, float* dc);
int launch_fa_white(LaunchCtx* P, float* y, int n, unsigned long long seed);
int launch_fa_stretch(LaunchCtx* P, const float* tpl, int nt, int ny, float* y);
int launch_fa_glottal_fit(LaunchCtx* P, const float* ampl, int nhar, const float* model_power, const float* model_param,
  int ncand, int nhm, float* out);
int launch_fa_l1_frame(LaunchCtx* P, const float* ampl, int nhar, double f0d, int nfft, int what, const float2* tw,
  int tw_nmax, float* out);

#include "scratch.h"

namespace {
FP_TYPE* zeros(int n) { return (FP_TYPE*)std::calloc((size_t)std::max(n, 1), sizeof(FP_TYPE)); }
}  // namespace

extern "C" {

// ------------------------------------------------------------------ dsputils.h
void llsm_harmonic_czt(FP_TYPE* x, int nx, FP_TYPE f0, FP_TYPE fs, int nhar, FP_TYPE* dst_ampl, FP_TYPE* dst_phse) {
  for(int i = 0; i < nhar; i ++) { dst_ampl[i] = 0; dst_phse[i] = 0; }
  Scratch s; if(! s.open() || nx <= 0 || nhar <= 0) return;
  float* dx = s.up(x, nx); float* da = s.alloc<float>(nhar); float* dp = s.alloc<float>(nhar);
  if(s.bad) return;
  if(! s.run(launch_fa_czt(s.P, dx, nx, (double)f0 / (double)fs, nhar, da, dp), "llsm_harmonic_czt")) return;
  s.down(dst_ampl, da, nhar); s.down(dst_phse, dp, nhar); s.sync();
}

static FP_TYPE* harm_frame(FP_TYPE* ampl, FP_TYPE* phse, int nhar, FP_TYPE f0, int nx) {
  FP_TYPE* y = zeros(nx);
  Scratch s; if(! s.open() || nx <= 0) return y;
  if(nhar <= 0) return y;
  float* da = s.up(ampl, nhar); float* dp = s.up(phse, nhar); float* dy = s.alloc<float>(nx);
  if(s.bad) return y;
  if(! s.run(launch_fa_harm_frame(s.P, da, dp, nhar, (double)f0, nx, dy), "llsm_synthesize_harmonic_frame")) return y;
  s.down(y, dy, nx); s.sync();
  return y;
}
FP_TYPE* llsm_synthesize_harmonic_frame(FP_TYPE* ampl, FP_TYPE* phse, int nhar, FP_TYPE f0, int nx) {
  return harm_frame(ampl, phse, nhar, f0, nx);
}
FP_TYPE* llsm_synthesize_harmonic_frame_iczt(FP_TYPE* ampl, FP_TYPE* phse, int nhar, FP_TYPE f0, int nx) {
  return harm_frame(ampl, phse, nhar, f0, nx);
}
FP_TYPE* llsm_synthesize_harmonic_frame_auto(llsm_soptions* options, FP_TYPE* ampl, FP_TYPE* phse, int nhar,
  FP_TYPE f0, int nx) {
  (void)options;                                        // llsmutils.c:45-58 only picks the faster of two equal methods
  return harm_frame(ampl, phse, nhar, f0, nx);
}

void llsm_compute_spectrogram(FP_TYPE* x, int nx, int* center, int* winsize, int nfrm, int nfft, char* wintype,
  FP_TYPE** dst_spec, FP_TYPE** dst_phse) {
  const int ns = nfft / 2 + 1;
  for(int i = 0; i < nfrm; i ++) {
    std::memset(dst_spec[i], 0, sizeof(FP_TYPE) * ns);
    if(dst_phse) std::memset(dst_phse[i], 0, sizeof(FP_TYPE) * ns);
  }
  Scratch s; if(! s.open() || nfrm <= 0) return;
  const bool blackman = wintype && ! std::strcmp(wintype, "blackman");
  // dsputils.c:98-114: scale = 1024 / (0.5 sum(window(1024))) / winsize[i]
  double wsum = 0;
  for(int i = 0; i < 1024; i ++) {
    const double t = 2.0 * 3.14159265358979323846 * i / 1023.0;
    wsum += blackman ? (double)(float)(0.42 - 0.5 * std::cos(t) + 0.08 * std::cos(2.0 * t)) : (double)(float)(0.5 - 0.5 * std::cos(t));
  }
  std::vector<float> scale(nfrm);
  for(int i = 0; i < nfrm; i ++) scale[i] = (float)(1024.0 / (0.5 * wsum) / winsize[i]);
  int tw_nmax = 0; const float2* tw = llsm_engine_twiddles(s.ctx, & tw_nmax);
  float* dx = s.up(x, nx); int* dc = s.up(center, nfrm); int* dw = s.up(winsize, nfrm); float* dsc = s.up(scale.data(), nfrm);
  float* dspec = s.alloc<float>((size_t)nfrm * ns); float* dph = dst_phse ? s.alloc<float>((size_t)nfrm * ns) : nullptr;
  if(s.bad) return;
  if(! s.run(launch_fa_stft(s.P, dx, nx, dc, dw, nfrm, nfft, blackman, 0, dsc, tw, tw_nmax, dspec, dph), "llsm_compute_spectrogram")) return;
  std::vector<float> h((size_t)nfrm * ns), hp(dst_phse ? (size_t)nfrm * ns : 0);
  s.down(h.data(), dspec, h.size()); if(dst_phse) s.down(hp.data(), dph, hp.size());
  if(! s.sync()) return;
  for(int i = 0; i < nfrm; i ++) {
    std::memcpy(dst_spec[i], h.data() + (size_t)i * ns, sizeof(FP_TYPE) * ns);
    if(dst_phse) std::memcpy(dst_phse[i], hp.data() + (size_t)i * ns, sizeof(FP_TYPE) * ns);
  }
}

void llsm_estimate_psd(FP_TYPE* x, int nx, int nfft, FP_TYPE* dst_psd) {
  const int ns = nfft / 2 + 1;
  std::memset(dst_psd, 0, sizeof(FP_TYPE) * ns);
  Scratch s; if(! s.open() || nx <= 0) return;
  int tw_nmax = 0; const float2* tw = llsm_engine_twiddles(s.ctx, & tw_nmax);
  float* dx = s.up(x, nx); float* dspec = s.alloc<float>(ns);
  if(s.bad) return;
  if(! s.run(launch_fa_stft(s.P, dx, nx, nullptr, nullptr, 1, nfft, 1, 1, nullptr, tw, tw_nmax, dspec, nullptr), "llsm_estimate_psd")) return;
  s.down(dst_psd, dspec, ns); s.sync();
}

void llsm_compute_dc(FP_TYPE* x, int nx, int* center, int* winsize, int nfrm, FP_TYPE* dst_dc) {
  for(int i = 0; i < nfrm; i ++) dst_dc[i] = 0;
  Scratch s; if(! s.open() || nfrm <= 0) return;
  float* dx = s.up(x, nx); int* dc = s.up(center, nfrm); int* dw = s.up(winsize, nfrm); float* dd = s.alloc<float>(nfrm);
  if(s.bad) return;
  if(! s.run(launch_fa_dc(s.P, dx, nx, dc, dw, nfrm, dd), "llsm_compute_dc")) return;
  s.down(dst_dc, dd, nfrm); s.sync();
}

void llsm_harmonic_peakpicking(FP_TYPE* spectrum, FP_TYPE* phase, int nfft, FP_TYPE fs, int nhar, FP_TYPE f0,
  FP_TYPE* dst_ampl, FP_TYPE* dst_phse) {
  for(int i = 0; i < nhar; i ++) { dst_ampl[i] = 0; dst_phse[i] = 0; }
  Scratch s; if(! s.open() || nhar <= 0) return;
  const int ns = nfft / 2 + 1;
  float* dsp = s.up(spectrum, ns); float* dph = s.up(phase, ns); float* da = s.alloc<float>(nhar); float* dp = s.alloc<float>(nhar);
  if(s.bad) return;
  if(! s.run(launch_fa_peakpick(s.P, dsp, dph, nfft, fs, nhar, f0, da, dp), "llsm_harmonic_peakpicking")) return;
  s.down(dst_ampl, da, nhar); s.down(dst_phse, dp, nhar); s.sync();
}

// one-utterance batch around the harmonic stage of the analysis
static llsm_gpu_batch* harm_batch(llsm_gpu_context* ctx, FP_TYPE* x, int nx, FP_TYPE fs, FP_TYPE* f0, int nfrm, FP_TYPE thop,
  FP_TYPE rel_winsize, int maxnhar, int method, int refine) {
  llsm_aoptions ao; std::memset(& ao, 0, sizeof(ao));
  ao.thop = thop; ao.maxnhar = std::max(maxnhar, 1); ao.maxnhar_e = 0; ao.npsd = 2; ao.nchannel = 1;
  ao.lip_radius = 1.5f; ao.f0_refine = refine; ao.hm_method = method; ao.rel_winsize = rel_winsize;
  llsm_gpu_batch* b = llsm_gpu_create_batch(ctx, & ao, fs, 1, & nx, & nfrm);
  if(! b) return nullptr;
  if(llsm_gpu_batch_upload(b, LLSM_GPU_X, x, sizeof(float) * (size_t)nx) ||
     llsm_gpu_batch_upload(b, LLSM_GPU_F0, f0, sizeof(float) * (size_t)nfrm)) { llsm_gpu_delete_batch(b); return nullptr; }
  return b;
}

void llsm_harmonic_analysis(FP_TYPE* x, int nx, FP_TYPE fs, FP_TYPE* f0, int nfrm, FP_TYPE thop, FP_TYPE rel_winsize,
  int maxnhar, int method, int* dst_nhar, FP_TYPE** dst_ampl, FP_TYPE** dst_phse) {
  for(int i = 0; i < nfrm; i ++) { dst_nhar[i] = 0; dst_ampl[i] = NULL; dst_phse[i] = NULL; }
  llsm_gpu_context* ctx = llsm_default_context();
  if(! ctx || nfrm <= 0) return;
  llsm_gpu_batch* b = harm_batch(ctx, x, nx, fs, f0, nfrm, thop, rel_winsize, maxnhar, method, 0);
  if(! b) return;
  const int mh = std::max(maxnhar, 1);
  std::vector<int> nh(nfrm); std::vector<float> a((size_t)nfrm * mh), p((size_t)nfrm * mh);
  int rc = llsm_engine_batch_harmonics(b, 0);
  rc |= llsm_gpu_batch_download(b, LLSM_GPU_NHAR, nh.data(), nh.size() * 4);
  rc |= llsm_gpu_batch_download(b, LLSM_GPU_AMPL, a.data(), a.size() * 4);
  rc |= llsm_gpu_batch_download(b, LLSM_GPU_PHSE, p.data(), p.size() * 4);
  llsm_gpu_delete_batch(b);
  if(rc) return;
  for(int i = 0; i < nfrm; i ++) {
    if(f0[i] == 0) continue;                            // dsputils.c:185-188: voiced frames only
    dst_nhar[i] = nh[i];
    dst_ampl[i] = zeros(nh[i]); dst_phse[i] = zeros(nh[i]);
    std::memcpy(dst_ampl[i], a.data() + (size_t)i * mh, sizeof(FP_TYPE) * (size_t)nh[i]);
    std::memcpy(dst_phse[i], p.data() + (size_t)i * mh, sizeof(FP_TYPE) * (size_t)nh[i]);
  }
}

void llsm_refine_f0(FP_TYPE* x, int nx, FP_TYPE fs, FP_TYPE* f0, int nfrm, FP_TYPE thop) {
  llsm_gpu_context* ctx = llsm_default_context();
  if(! ctx || nfrm <= 0) return;
  llsm_gpu_batch* b = harm_batch(ctx, x, nx, fs, f0, nfrm, thop, 4.0f, 1, LLSM_AOPTION_HMCZT, 1);
  if(! b) return;
  std::vector<float> r(nfrm);
  int rc = llsm_engine_batch_harmonics(b, 1);
  rc |= llsm_gpu_batch_download(b, LLSM_GPU_F0, r.data(), r.size() * 4);
  llsm_gpu_delete_batch(b);
  if(! rc) std::memcpy(f0, r.data(), sizeof(FP_TYPE) * (size_t)nfrm);
}

FP_TYPE* llsm_subband_energy(FP_TYPE* x, int nx, FP_TYPE fmin, FP_TYPE fmax) {
  FP_TYPE* y = zeros(nx);
  Scratch s; if(! s.open() || nx <= 0) return y;
  float* dx = s.up(x, nx); float* dy = s.alloc<float>(nx);
  if(s.bad) return y;
  if(llsm_engine_chebyfilt(s.ctx, dx, nx, fmin, fmax, 1, dy)) return y;
  s.down(y, dy, nx); s.sync();
  return y;
}

FP_TYPE* llsm_generate_white_noise(int nx) {
  FP_TYPE* y = zeros(nx);
  Scratch s; if(! s.open() || nx <= 0) return y;
  const int nt = std::min(20000, nx);                   // dsputils.c:355-360: 20000 fresh samples, then they repeat
  float* dt = s.alloc<float>(nt); float* dy = s.alloc<float>(nx);
  if(s.bad) return y;
  if(! s.run(launch_fa_white(s.P, dt, nt, llsm_next_seed()), "llsm_generate_white_noise")) return y;
  for(int o = 0; o < nx && ! s.bad; o += nt)
    if(hipMemcpyAsync(dy + o, dt, sizeof(float) * (size_t)std::min(nt, nx - o), hipMemcpyDeviceToDevice, s.st) != hipSuccess) s.bad = true;
  s.down(y, dy, nx); s.sync();
  return y;
}

FP_TYPE* llsm_generate_bandlimited_noise(int nx, FP_TYPE fmin, FP_TYPE fmax) {
  FP_TYPE* y = zeros(nx);
  Scratch s; if(! s.open() || nx <= 0) return y;
  const int nt = std::min(20000, nx), n = nt + 128;     // dsputils.c:385-394
  float* dw = s.alloc<float>(n); float* dcol = s.alloc<float>(n); float* dy = s.alloc<float>(nx);
  if(s.bad) return y;
  if(! s.run(launch_fa_white(s.P, dw, nt, llsm_next_seed()), "llsm_generate_bandlimited_noise")) return y;
  // the 128-sample extension wraps around (llsm_generate_white_noise(n) with n > 20000 repeats)
  if(hipMemcpyAsync(dw + nt, dw, sizeof(float) * (size_t)std::min(128, nt), hipMemcpyDeviceToDevice, s.st) != hipSuccess) return y;
  for(int o = nt + std::min(128, nt); o < n; o ++)
    if(hipMemcpyAsync(dw + o, dw + (o - nt) % nt, sizeof(float), hipMemcpyDeviceToDevice, s.st) != hipSuccess) return y;
  if(llsm_engine_chebyfilt(s.ctx, dw, n, fmin, fmax, 0, dcol)) return y;
  if(! s.run(launch_fa_stretch(s.P, dcol, nt, nx, dy), "llsm_generate_bandlimited_noise")) return y;
  s.down(y, dy, nx); s.sync();
  return y;
}

static FP_TYPE* l1_frame(FP_TYPE* ampl, int nhar, double f0d, int nfft, int what, int nout) {
  FP_TYPE* y = zeros(nout);
  Scratch s; if(! s.open() || nhar <= 0) return y;
  int tw_nmax = 0; const float2* tw = llsm_engine_twiddles(s.ctx, & tw_nmax);
  float* da = s.up(ampl, nhar); float* dy = s.alloc<float>(nout);
  if(s.bad) return y;
  if(! s.run(launch_fa_l1_frame(s.P, da, nhar, f0d, nfft, what, tw, tw_nmax, dy), "layer-1 frame helper")) return y;
  s.down(y, dy, nout); s.sync();
  return y;
}
FP_TYPE* llsm_harmonic_minphase(FP_TYPE* ampl, int nhar) { return l1_frame(ampl, nhar, 0, 0, 0, nhar); }
FP_TYPE* llsm_harmonic_spectrum(FP_TYPE* ampl, int nhar, FP_TYPE f0, int nfft) { return l1_frame(ampl, nhar, f0, nfft, 1, nfft / 2 + 1); }
FP_TYPE* llsm_harmonic_envelope(FP_TYPE* ampl, int nhar, FP_TYPE f0, int nfft) { return l1_frame(ampl, nhar, f0, nfft, 2, nfft / 2 + 1); }

// frame.c:180-213: signal-to-noise ratio (dB) or aperiodicity (linear) of a layer-0 frame on the warped axis of
// LLSM_CONF_NOSWARP.  (2.1 confs carry no NOSWARP -- llsm_aoptions_toconf does not write one --, so this returns NULL
// on them exactly as the reference does; a host that attaches the deprecated member gets the reference's result.)
FP_TYPE* llsm_frame_compute_snr(llsm_container* src, llsm_container* conf, int as_aperiodicity) {
  FP_TYPE* f0 = (FP_TYPE*)llsm_container_get(src, LLSM_FRAME_F0);
  llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(src, LLSM_FRAME_HM);
  llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(src, LLSM_FRAME_NM);
  FP_TYPE* fnyq = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_FNYQ);
  FP_TYPE* noswarp = (FP_TYPE*)llsm_container_get(conf, LLSM_CONF_NOSWARP);
  if(f0 == NULL || hm == NULL || nm == NULL) return NULL;
  if(fnyq == NULL || noswarp == NULL) return NULL;
  if(hm -> nhar < 1) return NULL;
  const int nfft = std::max(64, (int)std::pow(2.0, std::ceil(std::log2((double)hm -> nhar) + 2)));
  FP_TYPE* spec_env = llsm_harmonic_envelope(hm -> ampl, hm -> nhar, *f0 / *fnyq / 2.0f, nfft);
  if(! spec_env) return NULL;
  for(int i = 0; i < nfft / 2 + 1; i ++) {
    spec_env[i] = (FP_TYPE)std::pow(10.0, spec_env[i] / 20.0);   // dB to magnitude
    spec_env[i] *= spec_env[i] * 0.5f;                            // magnitude to variance
  }
  FP_TYPE* warp_axis = llsm_warp_frequency(0, *fnyq, nm -> npsd, *noswarp);
  FP_TYPE* spec_warp = llsm_spectral_mean(spec_env, nfft / 2 + 1, *fnyq, warp_axis, nm -> npsd);
  for(int i = 0; i < nm -> npsd; i ++) {
    if(as_aperiodicity) {
      const FP_TYPE snr = spec_warp[i] / (FP_TYPE)std::pow(10.0, nm -> psd[i] / 10.0);
      spec_warp[i] = 1.0f / (1.0f + snr);
    } else
      spec_warp[i] = 10.0f * (FP_TYPE)std::log10(spec_warp[i]) - nm -> psd[i];
  }
  std::free(warp_axis); std::free(spec_env);
  return spec_warp;
}

// cached LF responses (dsputils.c:512-538): squared, 1/k-weighted magnitudes of the LF spectrum at 200 Hz
struct GlottalCache { int nparam, nhar; std::vector<float> power, param; float* d_power = nullptr; float* d_param = nullptr; };
llsm_cached_glottal_model* llsm_create_cached_glottal_model(FP_TYPE* param, int nparam, int nhar) {
  if(nparam < 1 || nparam > 64 || nhar < 1) { llsm_set_error("llsm_create_cached_glottal_model: 1..64 parameters supported"); return NULL; }
  GlottalCache* g = new GlottalCache();
  g -> nparam = nparam; g -> nhar = nhar;
  g -> power.resize((size_t)nparam * nhar); g -> param.assign(param, param + nparam);
  const double f0 = 200.0;
  for(int i = 0; i < nparam; i ++) {
    const lf::Solved s = lf::solve(lf::from_rd((double)param[i], 1.0 / f0, 1.0, llsm_conv_lf_rd_clamp()));
    for(int j = 0; j < nhar; j ++) {
      const double m = lf::magnitude(s, f0 * (1.0 + j)) / (j + 1.0);
      g -> power[(size_t)i * nhar + j] = (float)(m * m);
    }
  }
  return (llsm_cached_glottal_model*)g;
}
void llsm_delete_cached_glottal_model(llsm_cached_glottal_model* dst) {
  GlottalCache* g = (GlottalCache*)dst;
  if(! g) return;
  if(g -> d_power) (void)hipFree(g -> d_power);
  if(g -> d_param) (void)hipFree(g -> d_param);
  delete g;
}
FP_TYPE llsm_spectral_glottal_fitting(FP_TYPE* ampl, int nhar, llsm_cached_glottal_model* model) {
  GlottalCache* g = (GlottalCache*)model;
  Scratch s; if(! g || ! s.open() || nhar <= 0) return 0;
  if(! g -> d_power) {
    if(hipMalloc((void**)& g -> d_power, g -> power.size() * 4) != hipSuccess || hipMalloc((void**)& g -> d_param, g -> param.size() * 4) != hipSuccess ||
       hipMemcpy(g -> d_power, g -> power.data(), g -> power.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
       hipMemcpy(g -> d_param, g -> param.data(), g -> param.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
      llsm_set_error("llsm_spectral_glottal_fitting: table upload failed"); return 0;
    }
  }
  float* da = s.up(ampl, nhar); float* dout = s.alloc<float>(1);
  if(s.bad) return 0;
  if(! s.run(launch_fa_glottal_fit(s.P, da, nhar, g -> d_power, g -> d_param, g -> nparam, g -> nhar, dout), "llsm_spectral_glottal_fitting")) return 0;
  float r = 0; s.down(& r, dout, 1); s.sync();
  return r;
}

// ---- elementwise / index helpers (host) ----
void llsm_fft_to_psd(FP_TYPE* X_re, FP_TYPE* X_im, int nfft, FP_TYPE wsqr, FP_TYPE* dst_psd) {   // dsputils.c:237-244
  for(int i = 0; i < nfft / 2 + 1; i ++) dst_psd[i] = (X_re[i] * X_re[i] + X_im[i] * X_im[i]) / wsqr;
}
FP_TYPE* llsm_warp_frequency(FP_TYPE fmin, FP_TYPE fmax, int n, FP_TYPE warp_const) {          // dsputils.c:267-276
  FP_TYPE* f = zeros(n);
  const double wmin = 5000.0 * std::log(1.0 + fmin / warp_const), wmax = 5000.0 * std::log(1.0 + fmax / warp_const);
  for(int i = 0; i < n; i ++) f[i] = (FP_TYPE)(warp_const * (std::exp(((double)i / n * (wmax - wmin) + wmin) / 5000.0) - 1.0));
  return f;
}
FP_TYPE* llsm_spectral_mean(FP_TYPE* spectrum, int nspec, FP_TYPE fnyq, FP_TYPE* freq, int nfreq) {   // dsputils.c:278-306
  FP_TYPE* env = zeros(nfreq);
  auto clampi = [&](int v) { return std::min(nspec - 1, std::max(0, v)); };
  for(int i = 0; i < nfreq; i ++) {
    const FP_TYPE fprev = i == 0 ? 0 : freq[i - 1];
    const FP_TYPE fnext = i == nfreq - 1 ? freq[i] * 2 - freq[i - 1] : freq[i + 1];
    const int lo = clampi((int)(fprev / fnyq * nspec)), hi = clampi((int)(fnext / fnyq * nspec + 1));
    if(i > 0 && hi == lo) { env[i] = env[i - 1]; continue; }
    int center = (int)((hi + lo) / 2.0);
    center = std::min(std::max(center, 1), nspec - 2);
    const int width = std::max(1, center - lo);
    const FP_TYPE acc = (FP_TYPE)((width + 2) * width + 1);      // triangular weights, sum = acc
    FP_TYPE v = spectrum[center] * (width + 1) / acc;
    for(int j = 0; j < width; j ++) v += (spectrum[center + j + 1] + spectrum[center - j - 1]) * ((width - j) / acc);
    env[i] = v;
  }
  return env;
}
FP_TYPE* llsm_spectrum_from_envelope(FP_TYPE* freq, FP_TYPE* ampl, int nfreq, int nspec, FP_TYPE fnyq) {   // dsputils.c:308-316
  FP_TYPE* out = zeros(nspec);
  int k = 0;
  for(int i = 0; i < nspec; i ++) {
    const FP_TYPE f = (FP_TYPE)i * fnyq / nspec;
    while(k < nfreq - 2 && freq[k + 1] < f) k ++;
    if(nfreq == 1 || f <= freq[0]) out[i] = ampl[0];
    else if(f >= freq[nfreq - 1]) out[i] = ampl[nfreq - 1];
    else out[i] = ampl[k] + (ampl[k + 1] - ampl[k]) * (f - freq[k]) / (freq[k + 1] - freq[k]);
  }
  return out;
}
int llsm_get_fftsize(FP_TYPE* f0, int nfrm, FP_TYPE fs, FP_TYPE rel_winsize) {                   // dsputils.c:318-326
  FP_TYPE minf0 = 1000;
  for(int i = 0; i < nfrm; i ++) if(f0[i] > 0 && f0[i] < minf0) minf0 = f0[i];
  return lp::nextpow2(lp::hwin(minf0, fs, rel_winsize));
}
static void lip_response(FP_TYPE radius, double omega, double* re, double* im) {                   // dsputils.c:398-403
  const double Rr = 128.0 / 9.0 / lf::kPi / lf::kPi, Lr = 8.0 * radius / 100.0 / 3.0 / lf::kPi / 340.0;
  const double a = omega * Lr * Rr, b = omega * Lr, d = Rr * Rr + b * b;
  *re = a * b / d; *im = a * Rr / d;                   // i a / (Rr + i b)
}
void llsm_lipfilter(FP_TYPE radius, FP_TYPE f0, int nhar, FP_TYPE* dst_ampl, FP_TYPE* dst_phse, int inverse) {
  for(int i = 0; i < nhar; i ++) {
    double re, im; lip_response(radius, (double)f0 * (1.0 + i) * 2.0 * lf::kPi, & re, & im);
    const double mag = std::sqrt(re * re + im * im), arg = std::atan2(im, re);
    if(dst_ampl) dst_ampl[i] = (FP_TYPE)(inverse ? dst_ampl[i] / mag : dst_ampl[i] * mag);
    if(dst_phse) dst_phse[i] = (FP_TYPE)(inverse ? dst_phse[i] - arg : dst_phse[i] + arg);
  }
}
void llsm_lipfilter_reim(FP_TYPE radius, FP_TYPE f0, int nhar, FP_TYPE* dst_re, FP_TYPE* dst_im, int inverse) {
  for(int i = 0; i < nhar; i ++) {
    double re, im; lip_response(radius, (double)f0 * (1.0 + i) * 2.0 * lf::kPi, & re, & im);
    const double xr = dst_re[i], xi = dst_im[i];
    if(inverse) { const double d = re * re + im * im; dst_re[i] = (FP_TYPE)((xr * re + xi * im) / d); dst_im[i] = (FP_TYPE)((xi * re - xr * im) / d); }
    else { dst_re[i] = (FP_TYPE)(xr * re - xi * im); dst_im[i] = (FP_TYPE)(xr * im + xi * re); }
  }
}
FP_TYPE* llsm_smoothing_filter(FP_TYPE* x, int nx, int order) {                                     // dsputils.c:582-608
  FP_TYPE* y = zeros(nx);
  if(nx < order) { std::memcpy(y, x, sizeof(FP_TYPE) * (size_t)nx); return y; }
  auto mean = [&](int lo) { FP_TYPE m = 0; for(int j = 0; j < order; j ++) m += x[lo + j]; return m / order; };
  const FP_TYPE m0 = mean(0), m1 = mean(nx - order);
  for(int i = 0; i < order / 2; i ++) { y[i] = m0; y[nx - i - 1] = m1; }
  for(int i = order / 2; i < nx - order / 2; i ++) {
    const int lo = i - order / 2;
    const FP_TYPE m = mean(lo);
    int above = 0, below = 0; FP_TYPE excess = 0;
    for(int j = lo; j < lo + order; j ++) { above += x[j] >= m; below += x[j] <= m; excess += std::max((FP_TYPE)0, x[j] - m); }
    y[i] = m + (above - below) * excess / order / order;
  }
  return y;
}

// ------------------------------------------------------------------ llsmutils.h
lfmodel llsm_lfmodel_from_rd(FP_TYPE rd, FP_TYPE T0, FP_TYPE Ee) {
  const lf::Model m = lf::from_rd(rd, T0, Ee, llsm_conv_lf_rd_clamp());
  lfmodel r; r.T0 = (FP_TYPE)m.T0; r.te = (FP_TYPE)m.te; r.tp = (FP_TYPE)m.tp; r.ta = (FP_TYPE)m.ta; r.Ee = (FP_TYPE)m.Ee;
  return r;
}
FP_TYPE* llsm_lfmodel_spectrum(lfmodel model, FP_TYPE* freq, int nf, FP_TYPE* dst_phase) {
  // a handful of closed-form evaluations per call (the same host routine the pulse scheduler uses)
  FP_TYPE* magn = zeros(nf);
  lf::Model m; m.T0 = model.T0; m.te = model.te; m.tp = model.tp; m.ta = model.ta; m.Ee = model.Ee;
  const lf::Solved s = lf::solve(m);
  for(int i = 0; i < nf; i ++) {
    double re, im; lf::spectrum(s, (double)freq[i], & re, & im);
    magn[i] = (FP_TYPE)std::sqrt(re * re + im * im);
    if(dst_phase) dst_phase[i] = (FP_TYPE)std::atan2(im, re);
  }
  return magn;
}
llsm_gfm llsm_lfmodel_to_gfm(lfmodel src) {                                                          // llsmutils.c:24-32
  llsm_gfm r;
  r.Fa = (FP_TYPE)(1.0 / (src.ta * src.T0)); r.Rk = (src.te - src.tp) / src.tp; r.Rg = (FP_TYPE)(0.5 / src.tp);
  r.T0 = src.T0; r.Ee = src.Ee;
  return r;
}
lfmodel llsm_gfm_to_lfmodel(llsm_gfm src) {                                                          // llsmutils.c:34-43
  lfmodel r;
  r.ta = (FP_TYPE)(1.0 / src.Fa / src.T0); r.tp = (FP_TYPE)(0.5 / src.Rg); r.te = r.tp + r.tp * src.Rk;
  r.T0 = src.T0; r.Ee = src.Ee;
  return r;
}

FP_TYPE* llsm_make_filtered_pulse(llsm_container* src, lfmodel* sources, FP_TYPE* offsets, int num_pulses, int pre_rotate,
  int size, FP_TYPE fnyq, FP_TYPE lip_radius, FP_TYPE fs) {
  FP_TYPE* y = zeros(size);
  FP_TYPE* f0 = (FP_TYPE*)llsm_container_get(src, LLSM_FRAME_F0);
  FP_TYPE* rd = (FP_TYPE*)llsm_container_get(src, LLSM_FRAME_RD);
  FP_TYPE* vt = (FP_TYPE*)llsm_container_get(src, LLSM_FRAME_VTMAGN);
  FP_TYPE* vs = (FP_TYPE*)llsm_container_get(src, LLSM_FRAME_VSPHSE);
  Scratch s;
  if(! f0 || ! rd || ! vt || ! vs || size < 4 || (size & (size - 1)) || num_pulses < 0 || ! s.open()) return y;
  const int nspec = llsm_fparray_length(vt), nhar = llsm_fparray_length(vs);
  if(nhar <= 0) return y;
  int tw_nmax = 0; const float2* tw = llsm_engine_twiddles(s.ctx, & tw_nmax);
  PbpJob job; job.frame = 0; job.first = 0; job.npulse = num_pulses; job.size = size; job.pre_rotate = pre_rotate;
  job.out_off = 0; job.start = 0; job.zero_extra = -1;
  std::vector<PbpPulse> pl(std::max(num_pulses, 1));
  for(int i = 0; i < num_pulses; i ++) {
    pl[i].T0 = sources[i].T0; pl[i].te = sources[i].te; pl[i].tp = sources[i].tp; pl[i].ta = sources[i].ta; pl[i].Ee = sources[i].Ee;
    pl[i].offset = offsets[i]; pl[i].pad = 0;
  }
  int one_n = nhar;
  L1Dev d; std::memset(& d, 0, sizeof(d));
  d.nframes = 1; d.maxnhar = nhar; d.nspec = nspec; d.fnyq = fnyq; d.lip_radius = lip_radius;
  d.f0 = s.up(f0, 1); d.rd = s.up(rd, 1); d.vtmagn = s.up(vt, nspec); d.vsphse = s.up(vs, nhar); d.nvsphse = s.up(& one_n, 1);
  PbpJob* dj = s.up(& job, 1); PbpPulse* dp = s.up(pl.data(), pl.size()); float* dy = s.alloc<float>(size);
  if(s.bad) return y;
  if(! s.run(launch_pbp_pulse(s.P, d, dj, 1, dp, size, fs, tw, tw_nmax, dy), "llsm_make_filtered_pulse")) return y;
  s.down(y, dy, size); s.sync();
  return y;
}

}  // extern "C"
