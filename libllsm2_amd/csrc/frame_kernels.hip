// frame_kernels.hip -- one-shot kernels behind the installed per-frame API of the reference
// (dsputils.h:26-91, llsmutils.h:30-46; csrc/frameapi.cpp).  These entry points work on ONE frame or
// ONE short signal per call, so the kernels are written for generality (any window length, any power-
// of-two FFT up to 8192), not for throughput; the batch kernels of kernels.hip are the fast path.
//
//   k_fa_czt         llsm_harmonic_czt                      dsputils.c:145-169
//   k_fa_harm_frame  llsm_synthesize_harmonic_frame{,_iczt} dsputils.c:328-351 (bank == ICZT)
//   k_fa_stft        llsm_compute_spectrogram / llsm_estimate_psd   dsputils.c:96-115, 246-265
//   k_fa_peakpick    llsm_harmonic_peakpicking              dsputils.c:126-143
//   k_fa_dc          llsm_compute_dc                        dsputils.c:117-124
//   k_fa_white       llsm_generate_white_noise              dsputils.c:353-361 (counter RNG)
//   k_fa_stretch     stretch_stationary_noise               dsputils.c:363-383
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "plan.h"

#pragma clang fp contract(fast)
#include "dev_common.h"

namespace lp = llsm_plan;
extern __shared__ __attribute__((aligned(16))) unsigned char fa_lds[];

DEV float fa_blackman(int t, int n) {
  if(n == 1) return 1.0f;
  const double u = 2.0 * 3.14159265358979323846 * (double)t / (double)(n - 1);
  return (float)(0.42 - 0.5 * cos(u) + 0.08 * cos(2.0 * u));
}
DEV float fa_hann(int t, int n) {
  if(n == 1) return 1.0f;
  return (float)(0.5 - 0.5 * cos(2.0 * 3.14159265358979323846 * (double)t / (double)(n - 1)));
}

// X_k = sum_t w[t] x[t] e^{-j w0 k (t - nx/2)}, k = 1..nhar; ampl = |X| 2 / sum(w); phse = arg X
__global__ __launch_bounds__(256) void k_fa_czt(const float* __restrict__ x, int nx, double turn0, int nhar,
  float* __restrict__ ampl, float* __restrict__ phse) {
  __shared__ float wsum_s[256];
  const int tid = threadIdx.x, shift = nx / 2;
  float ws = 0;
  for(int t = tid; t < nx; t += 256) ws += fa_blackman(t, nx);
  wsum_s[tid] = ws;
  __syncthreads();
  for(int o = 128; o > 0; o >>= 1) { if(tid < o) wsum_s[tid] += wsum_s[tid + o]; __syncthreads(); }
  const float wsum = wsum_s[0];
  for(int h = tid; h < nhar; h += 256) {
    const double tk = turn0 * (double)(h + 1);           // turns per sample of harmonic h + 1
    float sr = 0, si = 0, c = 1, s = 0, dc, ds;
    cs_turns(tk, & dc, & ds);
    for(int t = 0; t < nx; t ++) {
      if((t & 63) == 0) cs_turns(tk * (double)(t - shift), & c, & s);   // exact re-seed
      const float v = fa_blackman(t, nx) * x[t];
      sr += v * c; si -= v * s;
      const float c2 = c * dc - s * ds, s2 = c * ds + s * dc; c = c2; s = s2;
    }
    ampl[h] = sqrtf(sr * sr + si * si) * 2.0f / wsum;
    phse[h] = atan2f(si, sr);
  }
}

// y[t] = sum_k a_k cos(2 pi f0n (k + 1) (t - nx/2) + phi_k)
__global__ __launch_bounds__(256) void k_fa_harm_frame(const float* __restrict__ ampl, const float* __restrict__ phse,
  int nhar, double f0n, int nx, float* __restrict__ y) {
  float2* A = (float2*)fa_lds;                          // a_k e^{j phi_k}
  for(int k = threadIdx.x; k < nhar; k += 256) {
    float c, s; cs_turns((double)phse[k] * 0.15915494309189533577, & c, & s);
    A[k] = make_float2(ampl[k] * c, ampl[k] * s);
  }
  __syncthreads();
  const int t = blockIdx.x * 256 + threadIdx.x;
  if(t >= nx) return;
  const double th = f0n * (double)(t - nx / 2);          // turns per harmonic unit
  float dc, ds, c = 1, s = 0, acc = 0;
  cs_turns(th, & dc, & ds);
  for(int k = 0; k < nhar; k ++) {
    if((k & 31) == 0) cs_turns(th * (double)(k + 1), & c, & s);
    const float2 a = A[k];
    acc += a.x * c - a.y * s;
    const float c2 = c * dc - s * ds, s2 = c * ds + s * dc; c = c2; s = s2;
  }
  y[t] = acc;
}

// One block (64 lanes) per frame.  mode 0: cig_stft_forward as dsputils.c:96-115 uses it -- window of
// winsize[i] centred on center[i], zero-phase placement, magnitude x scale[i] and phase on nfft/2 + 1 bins.
// mode 1: llsm_estimate_psd -- Blackman(nx) x signal from index 0, |X|^2 / sum(w^2).
__global__ __launch_bounds__(WAVE) void k_fa_stft(const float* __restrict__ x, int nx, const int* __restrict__ center,
  const int* __restrict__ winsize, int nfft, int blackman, int mode, const float* __restrict__ scale,
  const float2* __restrict__ tw_glob, int tw_nmax, float* __restrict__ spec, float* __restrict__ phse) {
  const int i = blockIdx.x, lane = threadIdx.x;
  float2* X = (float2*)fa_lds; float2* TW = X + nfft;
  int logN = 0; while((1 << logN) < nfft) logN ++;
  load_twiddles(TW, tw_glob, nfft, tw_nmax, lane);
  for(int k = lane; k < nfft; k += WAVE) X[k] = make_float2(0.0f, 0.0f);
  __syncthreads();
  const int n = mode == 0 ? winsize[i] : nx, c = mode == 0 ? center[i] : 0;
  float wpow = 0;
  // one lane at a time per LDS slot when the window is longer than the transform (time aliasing): serialise by rounds
  for(int j0 = 0; j0 < n; j0 += nfft) {
    for(int j = j0 + lane; j < n && j < j0 + nfft; j += WAVE) {
      const float w = blackman ? fa_blackman(j, n) : fa_hann(j, n);
      const int src = mode == 0 ? c - n / 2 + j : j;
      const float v = (src >= 0 && src < nx) ? x[src] : 0.0f;
      const int pos = mode == 0 ? (((j - n / 2) % nfft) + nfft) % nfft : j;
      wpow += w * w;
      if(pos < nfft) X[pos].x += v * w;
    }
    __syncthreads();
  }
  wpow = wave_sum(wpow);
  fft_dif(X, TW, 1, nfft, logN, lane);
  const int ns = nfft / 2 + 1;
  for(int k = lane; k < ns; k += WAVE) {
    const float2 z = X[brevN(k, logN)];
    if(mode == 0) {
      spec[(size_t)i * ns + k] = sqrtf(z.x * z.x + z.y * z.y) * scale[i];
      if(phse) phse[(size_t)i * ns + k] = atan2f(z.y, z.x);
    } else spec[(size_t)i * ns + k] = (z.x * z.x + z.y * z.y) / wpow;
  }
}

// llsm_harmonic_peakpicking on a LOG spectrum: thread = harmonic
__global__ void k_fa_peakpick(const float* __restrict__ spectrum, const float* __restrict__ phase, int nfft, float fs,
  int nhar, float f0, float* __restrict__ ampl, float* __restrict__ phse) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
  if(i > nhar) return;
  const float tol = 0.3f;
  int lo = (int)round((double)(f0 * ((float)i - tol) / fs * (float)nfft));
  int hi = (int)round((double)(f0 * ((float)i + tol) / fs * (float)nfft));
  lo = max(1, lo); hi = min(nfft / 2 - 1, hi);
  int peak = lo;
  for(int j = lo; j <= hi; j ++) if(spectrum[j] > spectrum[peak]) peak = j;
  const float a = spectrum[peak - 1], b = spectrum[peak], c = spectrum[peak + 1];
  const float a1 = (a + c) / 2.0f - b, a2 = c - b - a1;
  float xo = a1 == 0.0f ? 0.0f : -a2 / a1 * 0.5f;
  if(xo < -1.0f || xo > 1.0f) xo = 0.0f;
  const float pf = (float)peak + xo;
  ampl[i - 1] = expf(a1 * xo * xo + a2 * xo + b);
  const int k = (int)pf;
  const float r = fmodf(pf, 1.0f);
  phse[i - 1] = phase[k] + (phase[k + 1] - phase[k]) * r;       // no unwrapping (dsputils.c:140-141)
}

// windowed mean: block per frame
__global__ __launch_bounds__(WAVE) void k_fa_dc(const float* __restrict__ x, int nx, const int* __restrict__ center,
  const int* __restrict__ winsize, float* __restrict__ dc) {
  const int i = blockIdx.x, lane = threadIdx.x, n = winsize[i], c = center[i];
  float acc = 0;
  for(int j = lane; j < n; j += WAVE) { const int s = c - n / 2 + j; if(s >= 0 && s < nx) acc += x[s]; }
  acc = wave_sum(acc);
  if(lane == 0) dc[i] = acc / (float)n;
}

__global__ void k_fa_white(float* __restrict__ y, int n, unsigned long long seed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  float u1, u2; lp::rng_uniforms(seed, (unsigned long long)i, & u1, & u2);
  y[i] = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}

// template of nt samples tiled to ny samples with 128-sample power-preserving cross-fades (plan.h stretch_index)
__global__ void k_fa_stretch(const float* __restrict__ tpl, int nt, int ny, float* __restrict__ y) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if(p >= ny) return;
  int b; float r;
  const int a = lp::stretch_index(p, nt, ny, 128, & b, & r);
  float v = tpl[a];
  if(b >= 0) { v *= 1.0f - r; v += tpl[b] * r; v /= sqrtf(2.0f * r * (r - 1.0f) + 1.0f); }
  y[p] = v;
}

#define FA_LAUNCH(kern, grid, block, lds, ...)                                        \
  do {                                                                                \
    hipLaunchKernelGGL(kern, grid, block, lds, P -> stream, __VA_ARGS__);             \
    hipError_t e_ = hipGetLastError();                                                \
    if(e_ != hipSuccess) return (int)e_;                                              \
  } while(0)

int launch_fa_czt(LaunchCtx* P, const float* x, int nx, double turn0, int nhar, float* ampl, float* phse) {
  FA_LAUNCH(k_fa_czt, dim3(1), dim3(256), 0, x, nx, turn0, nhar, ampl, phse); return 0;
}
int launch_fa_harm_frame(LaunchCtx* P, const float* ampl, const float* phse, int nhar, double f0n, int nx, float* y) {
  if(nx <= 0) return 0;
  FA_LAUNCH(k_fa_harm_frame, dim3((nx + 255) / 256), dim3(256), sizeof(float2) * (size_t)(nhar > 0 ? nhar : 1), ampl, phse,
    nhar, f0n, nx, y);
  return 0;
}
int launch_fa_stft(LaunchCtx* P, const float* x, int nx, const int* center, const int* winsize, int nfrm, int nfft,
  int blackman, int mode, const float* scale, const float2* tw, int tw_nmax, float* spec, float* phse) {
  if(nfrm <= 0) return 0;
  if(nfft > tw_nmax || nfft < 4 || (nfft & (nfft - 1))) return -1;
  const size_t lds = sizeof(float2) * ((size_t)nfft + nfft / 2);
  if(lds > 64 * 1024 && hipFuncSetAttribute((const void*)k_fa_stft, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return -1;
  FA_LAUNCH(k_fa_stft, dim3(nfrm), dim3(WAVE), lds, x, nx, center, winsize, nfft, blackman, mode, scale, tw, tw_nmax, spec, phse);
  return 0;
}
int launch_fa_peakpick(LaunchCtx* P, const float* spectrum, const float* phase, int nfft, float fs, int nhar, float f0,
  float* ampl, float* phse) {
  if(nhar <= 0) return 0;
  FA_LAUNCH(k_fa_peakpick, dim3((nhar + 63) / 64), dim3(64), 0, spectrum, phase, nfft, fs, nhar, f0, ampl, phse); return 0;
}
int launch_fa_dc(LaunchCtx* P, const float* x, int nx, const int* center, const int* winsize, int nfrm, float* dc) {
  if(nfrm <= 0) return 0;
  FA_LAUNCH(k_fa_dc, dim3(nfrm), dim3(WAVE), 0, x, nx, center, winsize, dc); return 0;
}
int launch_fa_white(LaunchCtx* P, float* y, int n, unsigned long long seed) {
  if(n <= 0) return 0;
  FA_LAUNCH(k_fa_white, dim3((n + 255) / 256), dim3(256), 0, y, n, seed); return 0;
}
int launch_fa_stretch(LaunchCtx* P, const float* tpl, int nt, int ny, float* y) {
  if(ny <= 0) return 0;
  FA_LAUNCH(k_fa_stretch, dim3((ny + 255) / 256), dim3(256), 0, tpl, nt, ny, y); return 0;
}
