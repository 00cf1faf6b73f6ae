// l1_kernels.hip -- gfx950 kernels of the layer-1 (source-filter) conversion and of the
// pulse-by-pulse (PbP) harmonic synthesis.
//
//   k_l1_rd_fit      llsm_analyze_rd's per-frame fit (layer1.c:59-75, dsputils.c:540-579)
//   k_l1_rd_smooth   interp_in_blank + llsm_smoothing_filter over an utterance (layer1.c:78-80, dsputils.c:582-608)
//   k_l1_frame       llsm_frame_tolayer1 (layer1.c:90-127): LF source removal, lip filter, minimum-phase
//                    vocal tract, spectral envelope -> VSPHSE, VTMAGN
//   k_l1_to_l0       llsm_frame_tolayer0 (layer1.c:151-195)
//   k_pbp_pulse      llsm_make_filtered_pulse (llsmutils.c:60-201), one workgroup (4 wavefronts) per pulse group
//   k_l1_mixcurve    the HM <-> PbP cross-fade curve of layer0.c:240-262 from per-frame segments
//   k_pbp_mix        overlap-add of the pulse groups and of the masked harmonic frames, cross-fade,
//                    y = y_sin + y_noise (layer0.c:224-227, 273-283, 657-659)
//
// One 64-lane wavefront owns one frame / pulse group.  Transforms are the in-place LDS wavefront
// FFT of dev_common.h (sizes 64 ... 8192).  The LF model is evaluated in float64 (lfmodel.h): its
// open and return phases cancel to first order at high frequencies.  alpha (the implicit LF growth
// rate) is found by a wave-parallel search: 64 lanes evaluate the net flow at 64 points of the
// bracket, the sign change picks the next bracket (9 rounds to float64 resolution).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "kernels.h"
#include "lfmodel.h"
#include "plan.h"

#pragma clang fp contract(fast)
#include "dev_common.h"
#include "wave_fft.h"

namespace lf = llsm_lf;
namespace lp = llsm_plan;

extern __shared__ __attribute__((aligned(16))) unsigned char l1_lds[];

#define DB2LOG_F(x) ((x) * (2.3025851f / 20.0f))
// DESIGN.md section 6, cig_spec2env: own calibration, switchable (llsm_gpu_set_convention "spec2env_lobe_1e6")
__device__ DevConventions g_conv_l1 = {3, 0, 0, 0.13397922601295542f, 0};
int llsm_l1_kernels_set_conventions(const DevConventions& c) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_conv_l1), & c, sizeof(c)) == hipSuccess ? 0 : -1;
}
#define LOBE_BIAS (g_conv_l1.lobe_bias)

DEV float wrapf(float x) {                     // (-pi, pi]
  const float t = x * 0.15915494309189535f;
  float r = (t - rintf(t)) * 6.283185307179586f;
  return r;
}
DEV double wave_bcast_d(double v, int src) {
  int lo = __shfl(__double2loint(v), src, WAVE), hi = __shfl(__double2hiint(v), src, WAVE);
  return __hiloint2double(hi, lo);
}

// ---- LF: wave-parallel solve of alpha (same root as lf::solve's bisection) ----
DEV lf::Solved lf_solve_wave(const lf::Model& m, int lane) {
  lf::Solved s = lf::prepare(m);
  const double Ar = lf::return_area(s);
  // grid k / Te, k = -60 ... 60: first sign change between neighbours
  double lo = 0, hi = 0; bool found = false;
  for(int base = -60; base <= 60 && ! found; base += 63) {
    const int k = base + lane;
    const double f = k <= 60 ? lf::open_area(s, (double)k / s.Te) + Ar : 0.0;
    const double fp = wave_bcast_d(f, lane > 0 ? lane - 1 : 0);
    const bool chg = lane > 0 && k <= 60 && ((fp <= 0 && f > 0) || (fp >= 0 && f < 0));
    const unsigned long long mask = __ballot(chg);
    if(mask) {
      const int l = __ffsll((long long)mask) - 1;
      lo = (double)(base + l - 1) / s.Te; hi = (double)(base + l) / s.Te; found = true;
    }
  }
  if(! found) return s;
  double flo = lf::open_area(s, lo) + Ar;
  for(int it = 0; it < 10; it ++) {
    // 64 interior points split [lo, hi] into 65 parts
    const double a = lo + (hi - lo) * (double)(lane + 1) / 65.0;
    const double f = lf::open_area(s, a) + Ar;
    const bool same = (f <= 0) == (flo <= 0);                // still on lo's side
    const unsigned long long mask = __ballot(! same);
    int l = mask ? __ffsll((long long)mask) - 1 : 64;        // first point on the other side
    const double nlo = l == 0 ? lo : lo + (hi - lo) * (double)l / 65.0;
    const double nhi = l == 64 ? hi : lo + (hi - lo) * (double)(l + 1) / 65.0;
    if(l > 0) flo = wave_bcast_d(f, l - 1);
    lo = nlo; hi = nhi;
    if(hi - lo < 1e-15 * fmax(fabs(lo), fabs(hi))) break;
  }
  s.alpha = 0.5 * (lo + hi);
  return s;
}

// The same, through the per-frame cache: a hit needs the (Rd, F0) the cached solution was computed for to be bit-equal to
// the frame's (rows a host may have rewritten miss and are solved again); keys start as NaN.  One block per frame in every
// kernel that calls this, so a frame's entry has one writer per launch.  Since round 4 an entry holds the WHOLE solution
// (LF_CACHE_DOUBLES float64 values: Te, Ta, T0, Ee, wg, eps, alpha, sw, cw), not alpha alone: a hit used to re-run
// lf::prepare -- the Newton iteration for eps with a float64 exp per step and a float64 sin / cos pair, ~800
// instructions per lane, in each of the four kernels that ask per frame.  The Rd key carries the lf_rd_clamp convention
// in its sign (Rd > 0), so entries written under the other setting miss.
#define LF_CACHE_DOUBLES 9
DEV lf::Solved lf_solve_cached(const lf::Model& m, int lane, const AlphaCache& c, int g, float rd, float f0) {
  const float krd = g_conv_l1.lf_rd_clamp ? -rd : rd;
  if(c.alpha && c.rd[g] == krd && c.f0[g] == f0) {
    const double* e = c.alpha + (size_t)g * LF_CACHE_DOUBLES;
    lf::Solved s;
    s.Te = e[0]; s.Ta = e[1]; s.T0 = e[2]; s.Ee = e[3]; s.wg = e[4]; s.eps = e[5]; s.alpha = e[6]; s.sw = e[7]; s.cw = e[8];
    return s;
  }
  const lf::Solved s = lf_solve_wave(m, lane);
  // values first, keys after a release at WORKGROUP scope: the only concurrent readers of a frame's entry are the other
  // wavefronts of its own block (k_pbp_pulse runs several, all writing the same values), which share this CU's L1; an
  // agent-scope __threadfence() here costs a cache write-back per frame and made k_l1_frame 2.5 x slower
  if(c.alpha && lane == 0) {
    double* e = c.alpha + (size_t)g * LF_CACHE_DOUBLES;
    e[0] = s.Te; e[1] = s.Ta; e[2] = s.T0; e[3] = s.Ee; e[4] = s.wg; e[5] = s.eps; e[6] = s.alpha; e[7] = s.sw; e[8] = s.cw;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    c.rd[g] = krd; c.f0[g] = f0;
  }
  return s;
}

// ---- LF spectrum in float32 (round 4).  lfmodel.h evaluates the closed-form transform in float64 with float64 sin / cos /
// exp / atan2 -- a few hundred instructions each on this device, and the pulse and layer-1 kernels asked for them per
// harmonic and per pulse: they were bound by exactly that.  alpha, eps and the two exponentials of a model still come
// from float64 (once per frame / pulse); the per-frequency part runs in float32 from phasors that are either
// cs_turns values of a float64-reduced phase (1.5e-7 absolute) or float32 rotations of such seeds over a few bins.
// Error budget: the open-phase denominator (alpha - jw)^2 + wg^2 is formed as alpha^2 + (wg - w)(wg + w) - j 2 alpha w,
// whose relative error in 1 / den stays below ~10 ulp at every w (|den| >= 2 alpha w at the resonance, ~w^2 beyond);
// the result agrees with the float64 form to ~1e-6 relative (tests/test_gpu_l1.py tolerances unchanged).
#ifndef LF_FAST
#define LF_FAST 1
#endif
struct LfFast { float wg, eps, alpha, sw, ea, ed, k0, kr, pc; double Te, D; };
DEV LfFast lf_fast(const lf::Solved& s) {
  LfFast q; const double D = s.T0 - s.Te;
  q.wg = (float)s.wg; q.eps = (float)s.eps; q.alpha = (float)s.alpha; q.sw = (float)s.sw;
  q.ea = (float)exp(-s.alpha * s.Te); q.ed = (float)exp(-s.eps * D);
  q.k0 = (float)(-s.Ee / s.sw); q.kr = (float)(-(s.Ee / (s.eps * s.Ta)));
  q.pc = (float)(s.alpha * s.sw - s.wg * s.cw);
  q.Te = s.Te; q.D = D;
  return q;
}
// transform at angular frequency w > 0 given e^{-j w Te} = cte + j ste and e^{-j w D} = cd + j sd (lf::spectrum_core)
DEV void lf_spec_fast(const LfFast& s, float w, float cte, float ste, float cd, float sd, float* re, float* im) {
  const float pi_ = -w * s.sw;                                  // Im((alpha - jw) sw - wg cw); the real part is s.pc
  const float nr = cte * s.pc - ste * pi_ + s.wg * s.ea, ni = cte * pi_ + ste * s.pc;
  const float dr = fmaf(s.wg - w, s.wg + w, s.alpha * s.alpha), di = -2.0f * s.alpha * w;
  const float idn = __builtin_amdgcn_rcpf(dr * dr + di * di);
  const float Or = s.k0 * (nr * dr + ni * di) * idn, Oi = s.k0 * (ni * dr - nr * di) * idn;
  const float t1r = 1.0f - s.ed * cd, t1i = -s.ed * sd;
  const float iq = __builtin_amdgcn_rcpf(fmaf(s.eps, s.eps, w * w)), iw = __builtin_amdgcn_rcpf(w);
  const float ur = (t1r * s.eps + t1i * w) * iq, ui = (t1i * s.eps - t1r * w) * iq;
  const float vr = -s.ed * sd * iw, vi = -s.ed * (1.0f - cd) * iw;
  const float br = ur - vr, bi = ui - vi;
  *re = Or + s.kr * (cte * br - ste * bi); *im = Oi + s.kr * (cte * bi + ste * br);
}
// transform at frequency f (Hz) > 0: the phasors from float64-reduced phases
DEV void lf_spec_fast_at(const LfFast& s, double f, float* re, float* im) {
  float cte, ste, cd, sd;
  cs_turns(f * s.Te, & cte, & ste); cs_turns(f * s.D, & cd, & sd);
  lf_spec_fast(s, (float)(6.283185307179586 * f), cte, -ste, cd, -sd, re, im);
}
DEV float lf_mag_fast(const LfFast& s, double f) { float r, i; lf_spec_fast_at(s, f, & r, & i); return sqrtf(r * r + i * i); }
DEV float lf_phase_fast(const LfFast& s, double f) { float r, i; lf_spec_fast_at(s, f, & r, & i); return atan2f(i, r); }

// lip radiation response at angular frequency omega: i omega Lr Rr / (Rr + i omega Lr)  (dsputils.c:396-413)
DEV void lip_resp(float radius, float omega, float* mag, float* arg) {
  const float Rr = (float)(128.0 / 9.0 / 3.14159265358979323846 / 3.14159265358979323846);
  const float Lr = (float)(8.0 * radius / 100.0 / 3.0 / 3.14159265358979323846 / 340.0);
  const float a = omega * Lr * Rr, b = omega * Lr;
  *mag = a / sqrtf(Rr * Rr + b * b);
  *arg = 1.5707963267948966f - atan2f(b, Rr);
}
DEV void lip_resp_reim(float radius, float omega, float* re, float* im) {
  const float Rr = (float)(128.0 / 9.0 / 3.14159265358979323846 / 3.14159265358979323846);
  const float Lr = (float)(8.0 * radius / 100.0 / 3.0 / 3.14159265358979323846 / 340.0);
  const float a = omega * Lr * Rr, b = omega * Lr, d = Rr * Rr + b * b;
  // i a / (Rr + i b) = i a (Rr - i b) / d = (a b + i a Rr) / d
  *re = a * b / d; *im = a * Rr / d;
}

DEV int minphase_fftsize(int nhar) {           // max(64, 2^(ceil(log2 nhar) + 2)), dsputils.c:490
  int l = 0; while((1 << l) < nhar) l ++;
  const int n = 1 << (l + 2);
  return n < 64 ? 64 : n;
}
DEV int ilog2_dev(int n) { int l = 0; while((1 << l) < n) l ++; return l; }

// interp1 on a uniform axis linspace(0, top, n) with clamping
DEV float interp_lin(const float* __restrict__ y, int n, float top, float x) {
  const float pos = x / top * (float)(n - 1);
  int k = (int)floorf(pos);
  if(k < 0) return y[0];
  if(k >= n - 1) return y[n - 1];
  const float r = pos - (float)k;
  return y[k] + (y[k + 1] - y[k]) * r;
}

// llsm_harmonic_minphase (dsputils.c:486-510).  A[0..nhar): linear amplitudes (LDS); out[0..nhar) (LDS).
// X: N float2, TW: N/2 float2 (N = minphase_fftsize(nhar), twiddles loaded by the caller for N).
template <int NT = WAVE>
DEV void harmonic_minphase_dev(const float* A, int nhar, float2* X, const float2* TW, int N, float* out, int lane) {
  const int logN = ilog2_dev(N), ns = N / 2 + 1;
  // har_idx[i] = i / (nhar + 1) * N / 2 (i = 1..nhar), har_ampl[i] = log(A[i-1] + 1e-10), har_ampl[0] = har_ampl[1]
  const float hlast = (float)((double)nhar / ((double)nhar + 1.0) * (double)N / 2.0);
  const float hprev = (float)(((double)nhar - 1.0) / ((double)nhar + 1.0) * (double)N / 2.0);
  const float x1 = hlast * 2.0f - hprev;
  for(int m = lane; m < N; m += NT) {
    const int mm = m <= N / 2 ? m : N - m;                       // symmetric log spectrum
    const float pos = (float)mm / x1 * (float)(nhar + 1);
    int k = (int)floorf(pos);
    float v;
    auto ha = [&](int i) { return logf(A[i > 0 ? i - 1 : 0] + 1e-10f); };
    if(k < 0) v = ha(0);
    else if(k >= nhar) v = ha(nhar);
    else { const float r = pos - (float)k; const float a = ha(k), b = ha(k + 1); v = a + (b - a) * r; }
    X[brevN(m, logN)] = make_float2(v, 0.0f);
  }
  __syncthreads();
  ifft_dit<NT>(X, TW, 1, N, logN, lane);                             // N * cepstrum, natural order
  const float inv = 1.0f / (float)N;
  for(int m = lane; m < N; m += NT) {
    float c = X[m].x * inv;
    if(m > 0 && m < N / 2) c *= 2.0f; else if(m > N / 2) c = 0.0f;
    X[m] = make_float2(c, 0.0f);
  }
  __syncthreads();
  fft_dif<NT>(X, TW, 1, N, logN, lane);                              // log H, bit-reversed
  // har_phse[i] = interp1u_excl(0, ns, sphase, ns, har_idx[i]), i = 0..nhar; then the (sic) shift
  auto hp = [&](int i) {
    const float hx = i == 0 ? 0.0f : (float)(((double)i) / ((double)nhar + 1.0) * (double)N / 2.0);
    const float pos = hx / (float)ns * (float)ns;
    int k = (int)floorf(pos);
    if(k >= ns - 1) return X[brevN(ns - 1, logN)].y;
    const float r = pos - (float)k;
    const float a = X[brevN(k, logN)].y, b = X[brevN(k + 1, logN)].y;
    return a + (b - a) * r;
  };
  // entries 1 .. nhar-1 move down by one; the last one keeps its own value (dsputils.c:505-506, sic)
  for(int k = lane; k < nhar; k += NT) out[k] = k <= nhar - 2 ? hp(k + 1) : hp(nhar - 1);
  __syncthreads();
}

// =====================================================================
// Rd fit: lanes = the 64 cached candidates (layer1.c:53-57: linspace(0.02, 3, 64), 80 harmonics)
// =====================================================================
#define RD_NCAND 64
#define RD_NHAR 80
// P[0..n): power of the frame's harmonics (LDS); lane c < ncand owns candidate c (model_power[c][nhm]).
// Returns (on lane 0) the refined parameter: Itakura-Saito distance, global minimum, parabolic refinement.
DEV float glottal_fit_dev(const float* P, int n, const float* __restrict__ model_power, const float* __restrict__ model_param,
  int ncand, int nhm, int lane) {
  float dist = 3.0e38f;
  if(lane < ncand) {
    const float* mp = model_power + (size_t)lane * nhm;
    const float gain = P[0] / mp[0];
    float is = 0.0f;
    for(int j = 0; j < n; j ++) {
      const float r = P[j] / (mp[j] * gain);
      is += r - logf(r) - 1.0f;
    }
    dist = expf(is / (float)n);
  }
  float best = dist; int bi = lane;                              // global minimum (first occurrence), dsputils.c:569
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, WAVE); const int oi = __shfl_xor(bi, o, WAVE);
    if(ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  const float a = __shfl(dist, bi > 0 ? bi - 1 : 0, WAVE), b = best, c = __shfl(dist, bi < 63 ? bi + 1 : 63, WAVE);
  float refined = model_param[bi];
  if(bi > 0 && bi < ncand - 1) {
    const float den = a - 2.0f * b + c;
    const float d = den == 0.0f ? 0.0f : 0.5f * (a - c) / den;
    const float pos = (float)bi + d;
    const int k = (int)pos;
    refined = model_param[k] + (model_param[k + 1] - model_param[k]) * fmodf(pos, 1.0f);
  }
  return refined;
}

// The same fit from tables precomputed per batch (round 4).  With r_j = P_j / (M_cj g), g = P_0 / M_c0:
//   sum_j (r_j - log r_j - 1) = (1 / g) sum_j P_j / M_cj  -  (sum_j log P_j - sum_j log M_cj - n log g)  -  n,
// i.e. per candidate ONE dot product with the reciprocal table (inv_t[j][c] = 1 / M_cj, lanes coalesced) and a prefix sum
// of log M_cj read from a table (cumlog_t[j][c] = sum_{i < j} log M_ci); sum_j log P_j is the frame's own and is formed
// once by the wavefront.  float64 throughout: the three terms are each ~n log-units large and cancel to a distance of
// ~1e-2, which the per-term float32 form (a division and a logarithm per candidate and harmonic: 2 200 VALU
// instructions per frame, 0.73 ms per 204 800 frames) avoided by construction.  Result: the oracle's float64 sum to
// ~1e-13; k_l1_rd_fit 0.73 -> see DESIGN.md.
DEV float glottal_fit_tab(const float* P, int n, const double* __restrict__ inv_t, const double* __restrict__ cumlog_t,
  const float* __restrict__ model_param, int lane) {
  double slog = 0.0;
  for(int j = lane; j < n; j += WAVE) slog += log((double)P[j]);
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) slog += __shfl_xor(slog, o, WAVE);
  double dot = 0.0;
  for(int j = 0; j < n; j ++) dot = fma((double)P[j], inv_t[(size_t)j * RD_NCAND + lane], dot);
  const double m0 = 1.0 / inv_t[lane], g = (double)P[0] / m0;            // M_c0, gain
  const double is = dot / g - (slog - cumlog_t[(size_t)n * RD_NCAND + lane] - (double)n * log(g)) - (double)n;
  const float dist = expf((float)(is / (double)n));
  float best = dist; int bi = lane;                              // global minimum (first occurrence), dsputils.c:569
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, WAVE); const int oi = __shfl_xor(bi, o, WAVE);
    if(ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  const float a = __shfl(dist, bi > 0 ? bi - 1 : 0, WAVE), b = best, c = __shfl(dist, bi < 63 ? bi + 1 : 63, WAVE);
  float refined = model_param[bi];
  if(bi > 0 && bi < RD_NCAND - 1) {
    const float den = a - 2.0f * b + c;
    const float d = den == 0.0f ? 0.0f : 0.5f * (a - c) / den;
    const float pos = (float)bi + d;
    const int k = (int)pos;
    refined = model_param[k] + (model_param[k + 1] - model_param[k]) * fmodf(pos, 1.0f);
  }
  return refined;
}

__global__ __launch_bounds__(WAVE) void k_l1_rd_fit(
  int nframes, const float* __restrict__ f0, const int* __restrict__ nhar, const float* __restrict__ ampl,
  int maxnhar, float lip_radius, const float* __restrict__ model_power, const float* __restrict__ model_param,
  const double* __restrict__ inv_t, const double* __restrict__ cumlog_t, float* __restrict__ rd_raw) {
  const int g = blockIdx.x, lane = threadIdx.x;
  float* P = (float*)l1_lds;                                     // lip-corrected power of the frame
  const float f = f0[g];
  if(!(f != 0)) { if(lane == 0) rd_raw[g] = 0.0f; return; }
  int n = nhar[g];
  const int lim = (int)round(8000.0 / (double)f);
  if(n > lim) n = lim;
  if(n > RD_NHAR) n = RD_NHAR;
  if(n > maxnhar) n = maxnhar;
  if(n <= 0) { if(lane == 0) rd_raw[g] = model_param[0]; return; }
  for(int k = lane; k < n; k += WAVE) {
    float mag, arg; lip_resp(lip_radius, f * (1.0f + (float)k) * 6.283185307179586f, & mag, & arg);
    const float a = ampl[(size_t)g * maxnhar + k] / mag;
    P[k] = a * a;
  }
  __syncthreads();
  const float r = inv_t ? glottal_fit_tab(P, n, inv_t, cumlog_t, model_param, lane)
                        : glottal_fit_dev(P, n, model_power, model_param, RD_NCAND, RD_NHAR, lane);
  if(lane == 0) rd_raw[g] = r;
}

// llsm_spectral_glottal_fitting on one amplitude vector (dsputils.c:540-579), ncand <= 64 cached responses
__global__ __launch_bounds__(WAVE) void k_fa_glottal_fit(const float* __restrict__ ampl, int nhar,
  const float* __restrict__ model_power, const float* __restrict__ model_param, int ncand, int nhm, float* __restrict__ out) {
  const int lane = threadIdx.x;
  float* P = (float*)l1_lds;
  const int n = nhar < nhm ? nhar : nhm;
  for(int k = lane; k < n; k += WAVE) P[k] = ampl[k] * ampl[k];
  __syncthreads();
  const float r = glottal_fit_dev(P, n, model_power, model_param, ncand, nhm, lane);
  if(lane == 0) out[0] = r;
}

// one block per utterance: blanks (unvoiced frames, rd == 0) filled by linear interpolation, then the
// impulse-insensitive moving average of `order` frames
__global__ __launch_bounds__(256) void k_l1_rd_smooth(
  const int* __restrict__ frm_off, const int* __restrict__ nfrm, int order,
  const float* __restrict__ rd_raw, int* __restrict__ prev_idx, int* __restrict__ next_idx,
  float* __restrict__ cont, float* __restrict__ rd_out) {
  const int u = blockIdx.x, tid = threadIdx.x;
  const int fo = frm_off[u], n = nfrm[u];
  const float* x = rd_raw + fo; float* c = cont + fo; float* y = rd_out + fo;
  int* pv = prev_idx + fo; int* nx = next_idx + fo;
  if(tid == 0) {
    int p = -1;
    for(int i = 0; i < n; i ++) { if(x[i] != 0.0f) p = i; pv[i] = p; }
    p = -1;
    for(int i = n - 1; i >= 0; i --) { if(x[i] != 0.0f) p = i; nx[i] = p; }
  }
  __syncthreads();
  for(int i = tid; i < n; i += 256) {
    const int a = pv[i], b = nx[i];
    float v;
    if(a < 0 && b < 0) v = x[i];
    else if(a < 0) v = x[b];
    else if(b < 0) v = x[a];
    else if(a == b) v = x[i];
    else v = x[a] + (x[b] - x[a]) * (float)(i - a) / (float)(b - a);
    c[i] = v;
  }
  __syncthreads();
  if(n < order) { for(int i = tid; i < n; i += 256) y[i] = c[i]; return; }
  for(int i = tid; i < n; i += 256) {
    float out;
    if(i < order / 2) { float m = 0; for(int j = 0; j < order; j ++) m += c[j]; out = m / (float)order; }
    else if(i >= n - order / 2) { float m = 0; for(int j = 0; j < order; j ++) m += c[n - order + j]; out = m / (float)order; }
    else {
      const int lo = i - order / 2, hi = lo + order;
      float mean = 0; for(int j = lo; j < hi; j ++) mean += c[j];
      mean /= (float)order;
      int npos = 0, nneg = 0; float dt = 0;
      for(int j = lo; j < hi; j ++) { npos += c[j] >= mean; nneg += c[j] <= mean; dt += c[j] - mean > 0 ? c[j] - mean : 0.0f; }
      out = mean + (float)(npos - nneg) * dt / (float)order / (float)order;
    }
    y[i] = out;
  }
}

// llsm_harmonic_spectrum (dsputils.c:433-456) and llsm_harmonic_envelope (dsputils.c:458-484) on nfft bins.
// A[0..n): linear amplitudes (LDS, kept); C[0..n): LDS scratch; X: nfft float2; TW: nfft / 2 float2.
// mode 0: out[k] = 3-period Hann lobes, max over harmonics, x f0 (the "harmonic spectrum" of the amplitudes as
// given); mode 1: out[k] = envelope in dB of the log-compressed amplitudes (cig_spec2env: cepstral sinc lifter +
// lobe constant).
DEV void harmonic_envelope_dev(const float* A, float* C, int n, double f0d, int nfft, float2* X, float2* TW,
  const float2* __restrict__ tw_glob, int tw_nmax, int mode, float* __restrict__ out, int lane) {
  const int nspec = nfft / 2 + 1;
  float peak = 0.0f;
  if(mode == 1) {
    float mx = 0.0f;
    for(int k = lane; k < n; k += WAVE) mx = fmaxf(mx, A[k]);
    mx = wave_max(mx);
    peak = logf(mx);
    __syncthreads();
    for(int k = lane; k < n; k += WAVE) {
      float x = logf(A[k]) - peak;
      if(!(x > -10.0f)) x = (x + 10.0f) / 2.0f - 10.0f;
      C[k] = expf(x);                                            // compressed amplitudes
    }
  } else {
    __syncthreads();
    for(int k = lane; k < n; k += WAVE) C[k] = A[k];
  }
  __syncthreads();
  const float f0n = (float)f0d;
  const int T = (int)(3.0 / f0d);
  const int width = (int)ceil(f0d * nfft * 1.5);
  const int logN = ilog2_dev(nfft);
  load_twiddles(TW, tw_glob, nfft, tw_nmax, lane);
  const double invT = 1.0 / (double)T;
  for(int j = lane; j < nfft; j += WAVE) {
    const int jj = j <= nfft / 2 ? j : nfft - j;
    float best = 0.0f;
    const float sp = f0n * (float)nfft;                          // harmonic spacing in bins
    int ilo = (int)floorf((float)(jj - width) / sp) - 2; if(ilo < 0) ilo = 0;
    int ihi = (int)ceilf((float)(jj + width) / sp) + 1; if(ihi > n - 1) ihi = n - 1;
    for(int i = ilo; i <= ihi; i ++) {
      const double ifreq = f0d * (1.0 + i);
      const int center = (int)round(ifreq * nfft);
      if(jj < center - width || jj > center + width) continue;
      // omega / (2 pi) in turns; numerator sin(T omega / 2) shared (the +-2 pi / T shifts flip its sign)
      const double dt = (double)jj / (double)nfft - ifreq;
      float cn, sn; cs_turns_rel(dt * (double)T * 0.5, & cn, & sn);   // (relative accuracy at its zeros m / T: 0 / 0 with the kernels below)
      auto asinc = [&](double turns_half, float num) {
        float c, sd; cs_turns(turns_half, & c, & sd);
        return fabsf(sd) < 1e-12f ? (float)T : num / sd;
      };
      const float r0 = asinc(dt * 0.5, sn);
      const float r1 = asinc((dt - invT) * 0.5, -sn);
      const float r2 = asinc((dt + invT) * 0.5, -sn);
      const float resp = 0.5f * r0 + 0.25f * r1 + 0.25f * r2;
      best = fmaxf(best, resp * C[i]);
    }
    if(mode == 0) { if(j <= nfft / 2) out[j] = best * f0n; }
    else X[brevN(j, logN)] = make_float2(logf(best * f0n + 1e-10f), 0.0f);
  }
  if(mode == 0) return;
  __syncthreads();
  ifft_dit(X, TW, 1, nfft, logN, lane);
  const float invN = 1.0f / (float)nfft;
  for(int q = lane; q < nfft; q += WAVE) {
    const int qq = q <= nfft / 2 ? q : nfft - q;
    float l = 1.0f;
    if(qq > 0) { const float a = 3.14159265358979323846f * (float)qq * f0n; l = sinf(a) / a; }
    X[q] = make_float2(X[q].x * invN * l, 0.0f);
  }
  __syncthreads();
  fft_dif(X, TW, 1, nfft, logN, lane);
  for(int k = lane; k < nspec; k += WAVE) {
    float e = X[brevN(k, logN)].x + LOBE_BIAS;
    if(!(e > -10.0f)) e = (e + 10.0f) * 2.0f - 10.0f;
    out[k] = (e + peak) / 2.3025851f * 20.0f;
  }
}

// one-frame entry points of dsputils.h: llsm_harmonic_minphase / llsm_harmonic_spectrum / llsm_harmonic_envelope
__global__ __launch_bounds__(WAVE) void k_fa_l1_frame(const float* __restrict__ ampl, int nhar, double f0d, int nfft,
  int what, int nmax, const float2* __restrict__ tw_glob, int tw_nmax, float* __restrict__ out) {
  const int lane = threadIdx.x, nh4 = (nhar + 3) & ~3;
  float* A = (float*)l1_lds; float* C = A + nh4;
  float2* X = (float2*)(C + nh4); float2* TW = X + nmax;
  for(int k = lane; k < nhar; k += WAVE) A[k] = ampl[k];
  __syncthreads();
  if(what == 0) {
    const int Nm = minphase_fftsize(nhar);
    load_twiddles(TW, tw_glob, Nm, tw_nmax, lane);
    __syncthreads();
    harmonic_minphase_dev(A, nhar, X, TW, Nm, C, lane);
    for(int k = lane; k < nhar; k += WAVE) out[k] = C[k];
  } else harmonic_envelope_dev(A, C, nhar, f0d, nfft, X, TW, tw_glob, tw_nmax, what == 2 ? 1 : 0, out, lane);
}

// =====================================================================
// llsm_frame_tolayer1 (layer1.c:90-127)
// LDS: A[nh4] Ph[nh4] VT[nh4] floats | X[NMAX] float2 | TW[NMAX / 2] float2
// =====================================================================
__global__ __launch_bounds__(WAVE) void k_l1_frame(
  int nframes, const float* __restrict__ f0, const int* __restrict__ nhar, const float* __restrict__ ampl,
  const float* __restrict__ phse, int maxnhar, const float* __restrict__ rd, float lip_radius, float fnyq,
  int nfft, int nmax, const float2* __restrict__ tw_glob, int tw_nmax,
  float* __restrict__ vtmagn, float* __restrict__ vsphse, int* __restrict__ nvsphse, float* __restrict__ src_out,
  AlphaCache acache) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int nspec = nfft / 2 + 1;
  const float f = f0[g];
  int n = nhar[g]; if(n > maxnhar) n = maxnhar;
  if(!(f != 0) || n <= 0) { if(lane == 0) nvsphse[g] = 0; return; }
  const int nh4 = (maxnhar + 3) & ~3;
  float* A = (float*)l1_lds; float* Ph = A + nh4; float* VT = Ph + nh4;
  float2* X = (float2*)(VT + nh4); float2* TW = X + nmax;
  // LF source amplitudes at the harmonics, normalised as layer1.c:104-107
  lf::Model m = lf::from_rd((double)rd[g], 1.0 / (double)f, 1.0, g_conv_l1.lf_rd_clamp);
  const lf::Solved s = lf_solve_cached(m, lane, acache, g, rd[g], f);
#if LF_FAST
  const LfFast sf = lf_fast(s);
  const float vs0 = lf_mag_fast(sf, (double)f);
#else
  const double vs0 = lf::magnitude(s, (double)f);
#endif
  for(int k = lane; k < n; k += WAVE) {
    const float fk = (float)((double)f * (k + 1.0));
#if LF_FAST
    const float vs = k == 0 ? 1.0f : lf_mag_fast(sf, (double)fk) / ((1.0f + (float)k) * vs0);
#else
    const float vs = k == 0 ? 1.0f : (float)(lf::magnitude(s, (double)fk) / ((1.0 + k) * vs0));
#endif
    float mag, arg; lip_resp(lip_radius, (float)((double)f * (1.0 + k) * 2.0 * 3.14159265358979323846), & mag, & arg);
    A[k] = ampl[(size_t)g * maxnhar + k] / mag / vs;
    Ph[k] = phse[(size_t)g * maxnhar + k] - arg;
  }
  __syncthreads();
  const int Nm = minphase_fftsize(n);
  load_twiddles(TW, tw_glob, Nm, tw_nmax, lane);
  __syncthreads();
  harmonic_minphase_dev(A, n, X, TW, Nm, VT, lane);
  for(int k = lane; k < n; k += WAVE) vsphse[(size_t)g * maxnhar + k] = Ph[k] - VT[k];
  for(int k = n + lane; k < maxnhar; k += WAVE) vsphse[(size_t)g * maxnhar + k] = 0.0f;
  if(lane == 0) nvsphse[g] = n;
  if(src_out) {                                      // the envelope is k_l1_env_wf's (two frames per transform)
    for(int k = lane; k < n; k += WAVE) src_out[(size_t)g * maxnhar + k] = A[k];
    return;
  }
  harmonic_envelope_dev(A, VT, n, (double)f / (double)fnyq / 2.0, nfft, X, TW, tw_glob, tw_nmax, 1,
    vtmagn + (size_t)g * nspec, lane);
}

// One 3-period Hann lobe of llsm_harmonic_spectrum (dsputils.c:433-456) in closed form.  With u = T dt (dt = j / N - h,
// the distance of bin j from the harmonic in cycles per sample, T the window length) the three Dirichlet kernels
//   resp = sin(pi u) [ 1/2 / sin(pi u / T) - 1/4 / sin(pi (u - 1) / T) - 1/4 / sin(pi (u + 1) / T) ]
// expand with 1 / sin y = 1 / y + y / 6 + 7 y^3 / 360 + ...: the 1 / y terms sum to T / (pi u (1 - u^2)), the y / 6 terms
// cancel exactly, the y^3 terms leave -(7 / 120) u (pi / T)^3, the y^5 terms (31 / 15120)(pi / T)^5 (10 u^3 + 5 u) -- below
// 2.3e-6 of the lobe for |u| <= 4.5 and T >= 64 and dropped:
//   resp = 1/2 [ T f(u) - (7 / 120)(pi / T)^3 u sin(pi u) ],   f(u) = sin(pi u) / (pi u (1 - u^2)).
// f has removable singularities at u = 0, +-1 (the peaks of the three kernels, 0/0 in the sum above): with k = round(u),
// v = u - k, sin(pi u) = (-1)^k pi v sinc(v), the vanishing factor of the denominator IS v and cancels symbolically.  All
// that needs float64 is u itself (a difference of two numbers ~ 1e3 times its size) and v; the rest is float32 --
// 2 float64 operations and 1 reciprocal per lobe where the angle-addition form has 10 and 3, and no sincospi per harmonic.
#define L1_LOBE_FAST_MIN_T 64
DEV float hann_lobe_fast(double ud, float Tf, float c3) {
  const double kd = rint(ud);
  const float u = (float)ud, v = (float)(ud - kd);
  const int k = (int)kd;
  const float w = v * v;
  // sinc(v) = sin(pi v) / (pi v), |v| <= 1/2: Taylor in (pi v)^2 (next term 4e-10)
  float sc = 1.4842879303107100e-04f;                              // pi^12 / 13!
  sc = fmaf(sc, w, -2.3460810354558236e-03f);                      // -pi^10 / 11!
  sc = fmaf(sc, w, 2.6147847817654800e-02f);                       // pi^8 / 9!
  sc = fmaf(sc, w, -1.9075182412208421e-01f);                      // -pi^6 / 7!
  sc = fmaf(sc, w, 8.1174242528335364e-01f);                       // pi^4 / 5!
  sc = fmaf(sc, w, -1.6449340668482264e+00f);                      // -pi^2 / 3!
  sc = fmaf(sc, w, 1.0f);
  const float snpi = (k & 1) ? - v * sc : v * sc;                  // sin(pi u) / pi
  const float om = 1.0f - u, op = 1.0f + u;
  float num = snpi, den = u * om * op;
  if(k == 0) { num = sc; den = om * op; }
  else if(k == 1) { num = sc; den = u * op; }
  else if(k == -1) { num = - sc; den = u * om; }
  return 0.5f * fmaf(Tf * num, __builtin_amdgcn_rcpf(den), - c3 * u * snpi);
}

// =====================================================================
// llsm_harmonic_envelope (dsputils.c:458-484) of the source-removed amplitudes, for PAIRS of frames on the
// register-resident wavefront FFT (wave_fft.h): the harmonic spectrum of dsputils.c:433-456 (3-period Hann lobes,
// maximum over the harmonics) is evaluated straight into the registers that own its bins, both log spectra are real
// and even, so ONE complex inverse transform returns both cepstra and one forward transform both envelopes
// (z = a + j b -> Z = A + j B with A, B real): no unpacking, no LDS round trips besides the FFT exchanges.
// Same quantities as harmonic_envelope_dev (mode 1), which stays for transform sizes without a register plan and for the
// one-frame entry points (it evaluates every lobe by float64 angle addition; here windows of T >= 64 samples take the
// closed form of hann_lobe_fast -- the two agree to 2e-7 of a lobe's peak, tests/test_lobe_model.py).  Lobes: resp = (D(dt) / 2 + D(dt - 1/T) / 4 + D(dt + 1/T) / 4), D the Dirichlet
// kernel sin(pi T x) / sin(pi x), numerator shared (the +-1 / T shifts flip its sign).
// LDS: wave-FFT exchange area (before the first transform: Hs[2][nh4] harmonic phasors, Vb log-lobe values) |
// C[2][nh4] compressed amplitudes.
// =====================================================================
template <int LOGN>
__global__ __launch_bounds__(WAVE, 2) void k_l1_env_wf(int nframes, const float* __restrict__ f0,
  const int* __restrict__ nvsphse, const float* __restrict__ src, int maxnhar, float fnyq,
  float* __restrict__ vtmagn, const int2* __restrict__ pairs, int npair) {
  constexpr int N = 1 << LOGN, P = N / WAVE, H = P / 2, nspec = N / 2 + 1;
  const int lane = threadIdx.x;
  const int nh4 = (maxnhar + 3) & ~3;
  float2* lds = (float2*)l1_lds;
  // per-harmonic phasors of the lobe kernels, (sin, cos)(pi T h) and (sin, cos)(pi h) with h = (1 + k) f0 / fs, float64:
  // they live in the area the transforms exchange through (used before the first transform only)
  double4* Hs = (double4*)l1_lds;
  const size_t fft_bytes = sizeof(float2) * wf_lds_elems<LOGN>(), har_bytes = sizeof(double4) * 2 * (size_t)nh4;
  float* Vb = (float*)((char*)l1_lds + har_bytes);               // [2][H + 1][WAVE] log-lobe values on their way to registers
  const size_t pre_bytes = har_bytes + sizeof(float) * 2 * (H + 1) * WAVE;
  float* Cc = (float*)((char*)l1_lds + (fft_bytes > pre_bytes ? fft_bytes : pre_bytes));
  WfTw<LOGN> tw; wf_init(tw, lane);
  const float invN = 1.0f / (float)N;
  const int per = (npair + gridDim.x - 1) / gridDim.x;
  for(int p = blockIdx.x * per; p < min(npair, (blockIdx.x + 1) * per); p ++) {
    int gg[2];
    if(pairs) { const int2 q = pairs[p]; gg[0] = q.x; gg[1] = q.y < 0 ? nframes : q.y; }
    else { gg[0] = 2 * p; gg[1] = 2 * p + 1; }
    int n[2]; double f0d[2]; float peak[2];
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      n[e] = 0; f0d[e] = 0.01; peak[e] = 0.0f;
      if(gg[e] >= nframes) continue;
      n[e] = min(nvsphse[gg[e]], maxnhar);
      if(n[e] > 0) f0d[e] = (double)f0[gg[e]] / (double)fnyq / 2.0;
    }
    if(n[0] <= 0 && n[1] <= 0) continue;
    __syncthreads();
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      if(n[e] <= 0) continue;
      const float* A = src + (size_t)gg[e] * maxnhar;
      float mx = 0.0f;
      for(int k = lane; k < n[e]; k += WAVE) mx = fmaxf(mx, A[k]);
      mx = wave_max(mx);
      peak[e] = logf(mx);
      for(int k = lane; k < n[e]; k += WAVE) {
        float x = logf(A[k]) - peak[e];
        if(!(x > -10.0f)) x = (x + 10.0f) / 2.0f - 10.0f;
        Cc[e * nh4 + k] = expf(x);                                 // compressed amplitudes
        const double hk = f0d[e] * (1.0 + k);
        const int Te = (int)(3.0 / f0d[e]);
        double4 h = {0, 0, 0, 0};
        if(Te >= L1_LOBE_FAST_MIN_T) h.x = (double)Te * hk;          // closed-form lobes: the harmonic's place in units of 1 / T
        else {
          sincospi((double)Te * hk, & h.x, & h.y);
          sincospi(hk, & h.z, & h.w);
        }
        Hs[e * nh4 + k] = h;
      }
    }
    __syncthreads();
    // (the lane index is made opaque once per pair: every bin-dependent value below is otherwise loop-invariant
    // over the pairs this wavefront walks and gets hoisted into registers that the transforms need)
    int lv = lane; asm volatile("" : "+v"(lv));
    // The lobe evaluation runs as a ROLLED loop over the 17 bins of a lane (unrolled it was 34 copies of the harmonic
    // search: 13 000 instructions, more than the instruction cache holds, 256 registers and spills); the values
    // go through LDS (Vb, beside Hs) into the statically indexed registers the transforms need.
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      float* vb = Vb + e * (H + 1) * WAVE + lane;
      if(n[e] <= 0) continue;
      const double fd = f0d[e];
      const float f0n = (float)fd;
      const double inv_sp = 1.0 / (fd * (double)N);                // 1 / harmonic spacing in bins
      const int T = (int)(3.0 / fd);
      const int width = (int)ceil(fd * N * 1.5);
      const int ne = n[e];
      const float* C = Cc + e * nh4;
      const double4* HH = Hs + e * nh4;
      // Each lobe is a sum of three Dirichlet kernels sin(pi T dt) / sin(pi (dt + {0, -1/T, 1/T})), dt = j / N - h.  Near a
      // kernel's peak numerator and denominator both vanish, so they need ABSOLUTE accuracy far below float32's: every
      // sine is formed in float64 by angle addition from the bin's phasors (pi T j / N and pi j / N: exactly reduced at
      // the lane's first bin, then rotated by 64 bins per step) and the harmonic's (Hs) -- 10 float64 operations per
      // lobe instead of four range-reduced sine evaluations.
      if(T >= L1_LOBE_FAST_MIN_T) {
        // closed-form lobes (hann_lobe_fast): u = T j / N - T h, the first term exact (N is a power of two)
        const float Tf = (float)T;
        const float c3 = (float)(7.0 / 120.0 * 97.40909103400243723644 / ((double)T * (double)T * (double)T));   // (7/120) pi^4 / T^3
        const double invNd = 1.0 / (double)N;
#pragma unroll 1
        for(int m = 0; m <= H; m ++) {
          const int jj = lv + WAVE * m;
          float val = 0.0f;
          if(jj <= N / 2) {
            float best = 0.0f;
            const double tj = (double)(T * jj) * invNd;
            int i_first = (int)ceil(((double)(jj - width) - 0.5) * inv_sp - 1.0); if(i_first < 0) i_first = 0;
            int i_last = (int)ceil(((double)(jj + width) + 0.5) * inv_sp - 1.0) - 1; if(i_last > ne - 1) i_last = ne - 1;
            for(int i0 = i_first; i0 <= i_last; i0 += 4) {
#pragma unroll
              for(int q = 0; q < 4; q ++) {
                const int i = min(i0 + q, i_last);
                best = fmaxf(best, hann_lobe_fast(tj - HH[i].x, Tf, c3) * C[i]);
              }
            }
            val = __logf(best * f0n + 1e-10f);
          }
          vb[m * WAVE] = val;
        }
        continue;
      }
      double sA, cA, sB, cB, sSa, cSa, sSb, cSb, sd, cd;
      sincospi((double)((T * lv) & (2 * N - 1)) / (double)N, & sA, & cA);
      sincospi((double)((T * WAVE) & (2 * N - 1)) / (double)N, & sSa, & cSa);
      sincospi((double)lv / (double)N, & sB, & cB);
      sincospi((double)WAVE / (double)N, & sSb, & cSb);
      sincospi(1.0 / (double)T, & sd, & cd);
#pragma unroll 1
      for(int m = 0; m <= H; m ++) {
        const int jj = lv + WAVE * m;                            // bins <= N / 2 (m = H: lane 0 only)
        float val = 0.0f;
        if(jj <= N / 2) {
          float best = 0.0f;
          // harmonics whose lobe (centre round(sp (1 + i)), half-width `width`) reaches bin jj: centres ascend with i, so
          // they are the run i_first .. i_last, found arithmetically (round(x) >= n <=> x >= n - 1/2) instead of probing
          // the centres in LDS one dependent load at a time; evaluated four at a time with the index clamped to the run
          // (a repeated lobe does not change the maximum), so that the loads of a group are in flight together
          int i_first = (int)ceil(((double)(jj - width) - 0.5) * inv_sp - 1.0); if(i_first < 0) i_first = 0;
          int i_last = (int)ceil(((double)(jj + width) + 0.5) * inv_sp - 1.0) - 1; if(i_last > ne - 1) i_last = ne - 1;
          for(int i0 = i_first; i0 <= i_last; i0 += 4) {
#pragma unroll
            for(int q = 0; q < 4; q ++) {
              const int i = min(i0 + q, i_last);
              const double4 h = HH[i];
              const double sn = sA * h.y - cA * h.x;               // sin(pi T dt)
              const double s0 = sB * h.w - cB * h.z, c0 = cB * h.w + sB * h.z;   // sin, cos(pi dt)
              const double s1 = s0 * cd - c0 * sd, s2 = s0 * cd + c0 * sd;       // sin(pi (dt -+ 1 / T))
              const float snf = (float)sn, s0f = (float)s0, s1f = (float)s1, s2f = (float)s2;
              const float r0 = fabsf(s0f) < 1e-10f ? (float)T : snf * __builtin_amdgcn_rcpf(s0f);
              const float r1 = fabsf(s1f) < 1e-10f ? (float)T : - snf * __builtin_amdgcn_rcpf(s1f);
              const float r2 = fabsf(s2f) < 1e-10f ? (float)T : - snf * __builtin_amdgcn_rcpf(s2f);
              best = fmaxf(best, (0.5f * r0 + 0.25f * r1 + 0.25f * r2) * C[i]);
            }
          }
          val = __logf(best * f0n + 1e-10f);
        }
        vb[m * WAVE] = val;
        { const double t = sA * cSa + cA * sSa; cA = cA * cSa - sA * sSa; sA = t; }   // next bin of this lane: + 64
        { const double t = sB * cSb + cB * sSb; cB = cB * cSb - sB * sSb; sB = t; }
      }
    }
    float xr[P], xi[P];
#pragma unroll
    for(int m = 0; m <= H; m ++) {                                 // (each lane reads back what it wrote)
      xr[m] = n[0] > 0 ? Vb[m * WAVE + lane] : 0.0f;
      xi[m] = n[1] > 0 ? Vb[(H + 1) * WAVE + m * WAVE + lane] : 0.0f;
    }
    __syncthreads();                                               // Hs is dead: the transforms exchange through its area
    wave_reflect<P>(xr, xr, lane);                                 // even: L[N - k] = L[k]
    wave_reflect<P>(xi, xi, lane);
    wave_fft<LOGN>(xi, xr, tw, lds, lane);                         // inverse (x N): both real cepstra
    // lifter sinc(qq f0), qq = min(q, N - q): sin(pi qq f0) by float64 rotation -- up from the lane's first quefrency
    // in the first half (q = lane + 64 m), down from N / 2 - lane in the second (qq = N - q)
    {
      double sa, ca, sb, cb, ssa, csa, ssb, csb;
      sincospi((double)lv * f0d[0], & sa, & ca); sincospi((double)WAVE * f0d[0], & ssa, & csa);
      sincospi((double)lv * f0d[1], & sb, & cb); sincospi((double)WAVE * f0d[1], & ssb, & csb);
      const float pfa = 3.14159265358979323846f * (float)f0d[0], pfb = 3.14159265358979323846f * (float)f0d[1];
#pragma unroll
      for(int m = 0; m < P / 2; m ++) {
        const int qq = lane + WAVE * m;
        float la = invN, lb = invN;
        // (reciprocals, not divisions: a float32 division is ten instructions, and there are 2 x 32 of them per lane here)
        if(qq > 0) { la = invN * (float)sa * __builtin_amdgcn_rcpf(pfa * (float)qq); lb = invN * (float)sb * __builtin_amdgcn_rcpf(pfb * (float)qq); }
        xr[m] *= la; xi[m] *= lb;
        { const double t = sa * csa + ca * ssa; ca = ca * csa - sa * ssa; sa = t; }
        { const double t = sb * csb + cb * ssb; cb = cb * csb - sb * ssb; sb = t; }
      }
      sincospi((double)(N / 2 - lv) * f0d[0], & sa, & ca);
      sincospi((double)(N / 2 - lv) * f0d[1], & sb, & cb);
#pragma unroll
      for(int m = P / 2; m < P; m ++) {
        const int qq = N - (lane + WAVE * m);                      // N / 2 - lane - 64 (m - P / 2) >= 1
        xr[m] *= invN * (float)sa * __builtin_amdgcn_rcpf(pfa * (float)qq); xi[m] *= invN * (float)sb * __builtin_amdgcn_rcpf(pfb * (float)qq);
        { const double t = sa * csa - ca * ssa; ca = ca * csa + sa * ssa; sa = t; }
        { const double t = sb * csb - cb * ssb; cb = cb * csb + sb * ssb; sb = t; }
      }
    }
    wave_fft<LOGN>(xr, xi, tw, lds, lane);                         // forward: envelope of frame a in xr, of b in xi
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      if(n[e] <= 0) continue;
      float* row = vtmagn + (size_t)gg[e] * nspec;
#pragma unroll
      for(int m = 0; m <= H; m ++) {
        const int k = lane + WAVE * m;
        if(k < nspec) {
          float ev = (e == 0 ? xr[m] : xi[m]) + LOBE_BIAS;
          if(!(ev > -10.0f)) ev = (ev + 10.0f) * 2.0f - 10.0f;
          row[k] = (ev + peak[e]) * (20.0f / 2.3025851f);          // nepers to dB
        }
      }
    }
  }
}

// =====================================================================
// Where the next glottal cycle begins relative to the frame's first sample (layer0.c:181-191): the LF model of the
// frame's Rd solved on the wavefront, its phase at f0 against the first source phase.  The pulse scheduler on the
// host needs this value for every layer-1 frame of the batch -- 95 % of its time when it solved the models itself.
// proj[g] = p0_dist / 2 pi * fs / f0 (0 where the frame has no layer-1 members); float64 throughout.
// =====================================================================
__global__ __launch_bounds__(WAVE) void k_l1_projection(int nframes, const float* __restrict__ f0,
  const float* __restrict__ rd, const float* __restrict__ vsphse, const int* __restrict__ nvsphse, int maxnhar,
  double fs, double* __restrict__ proj, AlphaCache acache) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const double f = (double)f0[g];
  if(f == 0 || nvsphse[g] <= 0) { if(lane == 0) proj[g] = 0.0; return; }
  const lf::Model m = lf::from_rd((double)rd[g], 1.0 / f, 1.0, g_conv_l1.lf_rd_clamp);
  const lf::Solved s = lf_solve_cached(m, lane, acache, g, rd[g], f0[g]);
  if(lane != 0) return;
  const double two_pi = 2.0 * 3.14159265358979323846;
  const double source_p0 = lf::phase(s, f) - 0.25 * two_pi;     // flow derivative -> flow
  const double v0 = (double)vsphse[(size_t)g * maxnhar];
  const double p0 = v0 - two_pi * round(v0 / two_pi);
  double d = source_p0 - p0; d -= two_pi * round(d / two_pi);   // phase_diff(source_p0, p0)
  if(d < 0) d += two_pi;
  proj[g] = d / two_pi * (fs / f);
}

// =====================================================================
// llsm_frame_tolayer0 (layer1.c:151-195).  only_missing: skip frames whose has_hm flag is set.
// LDS as k_l1_frame (X sized for the largest minimum-phase transform).
// =====================================================================
__global__ __launch_bounds__(WAVE) void k_l1_to_l0(
  int nframes, const float* __restrict__ f0, int* __restrict__ nhar, float* __restrict__ ampl,
  float* __restrict__ phse, int maxnhar, const float* __restrict__ rd, float lip_radius, float fnyq, int nspec,
  int maxnhar_conf, int only_missing, const int* __restrict__ select, int nmax,
  const float2* __restrict__ tw_glob, int tw_nmax,
  const float* __restrict__ vtmagn, const float* __restrict__ vsphse, const int* __restrict__ nvsphse,
  int* __restrict__ has_hm, AlphaCache acache) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const float f = f0[g];
  if(!(f != 0) || nvsphse[g] <= 0) return;
  if(only_missing && has_hm[g]) return;
  if(select && ! select[g]) return;
  int n = nvsphse[g];
  if(maxnhar_conf >= 0 && n > maxnhar_conf) n = maxnhar_conf;
  const int nq = (int)(fnyq / f); if(n > nq) n = nq;
  if(n > maxnhar) n = maxnhar;
  const int nh4 = (maxnhar + 3) & ~3;
  float* A = (float*)l1_lds; float* Ph = A + nh4; float* VT = Ph + nh4;
  float2* X = (float2*)(VT + nh4); float2* TW = X + nmax;
  if(n <= 0) { if(lane == 0) { nhar[g] = 0; has_hm[g] = 1; } return; }
  lf::Model m = lf::from_rd((double)rd[g], 1.0 / (double)f, 1.0, g_conv_l1.lf_rd_clamp);
  const lf::Solved s = lf_solve_cached(m, lane, acache, g, rd[g], f);
#if LF_FAST
  const LfFast sf = lf_fast(s);
  const float vs0 = lf_mag_fast(sf, (double)f);
#else
  const double vs0 = lf::magnitude(s, (double)f);
#endif
  const float* env = vtmagn + (size_t)g * nspec;
  for(int k = lane; k < n; k += WAVE) {
    const float fk = (float)((double)f * (k + 1.0));
#if LF_FAST
    const float vs = k == 0 ? 1.0f : lf_mag_fast(sf, (double)fk) / ((1.0f + (float)k) * vs0);
#else
    const float vs = k == 0 ? 1.0f : (float)(lf::magnitude(s, (double)fk) / ((1.0 + k) * vs0));
#endif
    A[k] = expf(DB2LOG_F(interp_lin(env, nspec, fnyq, fk)));
    Ph[k] = vs;
  }
  __syncthreads();
  const int Nm = minphase_fftsize(n);
  load_twiddles(TW, tw_glob, Nm, tw_nmax, lane);
  __syncthreads();
  harmonic_minphase_dev(A, n, X, TW, Nm, VT, lane);
  for(int k = lane; k < n; k += WAVE) {
    float mag, arg; lip_resp(lip_radius, (float)((double)f * (1.0 + k) * 2.0 * 3.14159265358979323846), & mag, & arg);
    ampl[(size_t)g * maxnhar + k] = A[k] * Ph[k] * mag;
    phse[(size_t)g * maxnhar + k] = VT[k] + vsphse[(size_t)g * maxnhar + k] + arg;
  }
  for(int k = n + lane; k < maxnhar; k += WAVE) { ampl[(size_t)g * maxnhar + k] = 0.0f; phse[(size_t)g * maxnhar + k] = 0.0f; }
  if(lane == 0) { nhar[g] = n; has_hm[g] = 1; }
}

// =====================================================================
// llsm_make_filtered_pulse (llsmutils.c:132-201) with make_filtered_pulse_spectrum (:60-131).
// One workgroup of NT threads per pulse group (the pulses of one frame, summed in the spectrum): the chain
// minimum phase -> spectrum -> inverse FFT is a latency chain, several wavefronts walk it faster than one.
// LDS: A[nh4] VT[nh4] PC[nh4 + 4] PS[nh4 + 4] floats | X[size_max] float2 | TW[size_max / 2] float2
// =====================================================================
#ifndef PBP_NT
#define PBP_NT 128                                 // threads per pulse group in large launches: 64 -> 2.55, 128 -> 2.03, 256 -> 2.81 ms (l1 bench)
#endif
#ifndef PBP_WIDE_BELOW
#define PBP_WIDE_BELOW 256                         // launches of at most this many pulse groups use 256 threads per group
#endif
// REAL (round 4): the pulse group is REAL, so its spectrum is needed on the bins 0 .. size / 2 only and the inverse
// transform is ONE complex transform of size / 2 points (even samples in the real parts, odd samples in the imaginary
// parts; Z[k] = (S[k] + conj S[M - k]) + j e^{j 2 pi k / size} (S[k] - conj S[M - k]), M = size / 2): half the LDS of the
// full-size complex transform of a Hermitian-completed spectrum (12 instead of 26 KB per group at 2048 points: twice
// the pulse groups resident per CU), half the butterflies, no completion pass.  REAL = false: the round-2 form (kept
// for pulse groups below 32 samples and as the A/B reference: llsm_gpu_pbp_real_ifft(0)).
#ifndef PBP_WPE
#define PBP_WPE 1                                  // wavefronts per SIMD the register budget is cut for (1: the compiler's choice)
#endif
template <int NT, bool REAL>
__global__ __launch_bounds__(NT, PBP_WPE) void k_pbp_pulse(
  const PbpJob* __restrict__ jobs, const PbpPulse* __restrict__ pulses,
  const float* __restrict__ f0, const float* __restrict__ rd, const float* __restrict__ vtmagn, int nspec,
  const float* __restrict__ vsphse, const int* __restrict__ nvsphse, int maxnhar,
  float fnyq, float lip_radius, float fs, int nmax, const float2* __restrict__ tw_glob, int tw_nmax,
  float* __restrict__ out, AlphaCache acache) {
  const int lane = threadIdx.x, wl = threadIdx.x & (WAVE - 1);   // thread of the group, lane of its wavefront
  const PbpJob job = jobs[blockIdx.x];
  const int g = job.frame, size = job.size, halfsize = size / 2 + 1;
  const float f = f0[g];
  const int n = nvsphse[g] < maxnhar ? nvsphse[g] : maxnhar;
  const int nh4 = (maxnhar + 3) & ~3;
  float* A = (float*)l1_lds; float* VT = A + nh4; float* PC = VT + nh4; float* PS = PC + nh4 + 4;
  float2* X = (float2*)(PS + nh4 + 4); float2* TW = X + nmax;
  const float* env = vtmagn + (size_t)g * nspec;
  const float* vsp = vsphse + (size_t)g * maxnhar;
  // vocal-tract phase from the harmonic amplitudes (llsmutils.c:149-160)
  for(int k = lane; k < n; k += NT) A[k] = expf(DB2LOG_F(interp_lin(env, nspec, fnyq, (float)(k + 1) * f)));
  __syncthreads();
  const int Nm = minphase_fftsize(n);
  load_twiddles<NT>(TW, tw_glob, Nm, tw_nmax, lane);
  __syncthreads();
  harmonic_minphase_dev<NT>(A, n, X, TW, Nm, VT, lane);
  // phase delta between the LF model and the stored source phases, per harmonic (llsmutils.c:69-86)
  lf::Model mo = lf::from_rd((double)rd[g], 1.0 / (double)f, 1.0, g_conv_l1.lf_rd_clamp);
  const lf::Solved so = lf_solve_cached(mo, wl, acache, g, rd[g], f);
#if LF_FAST
  const LfFast sof = lf_fast(so);
  const float ph1 = lf_phase_fast(sof, (double)f);
#else
  const float ph1 = (float)lf::phase(so, (double)f);
#endif
  const float vsshift = vsp[0] - (ph1 - 1.5707963267948966f);
  for(int i = lane; i <= n; i += NT) {
    float d = 0.0f;
    if(i >= 1) {
#if LF_FAST
      const float ph = lf_phase_fast(sof, (double)i * (double)f) - 1.5707963267948966f;
#else
      const float ph = (float)lf::phase(so, (double)i * (double)f) - 1.5707963267948966f;
#endif
      d = wrapf(vsp[i - 1] - ph - vsshift * (float)i) + VT[i - 1];
    }
    PC[i] = cosf(d); PS[i] = sinf(d);
  }
#if LF_FAST
  const float lfmagnf0 = lf_mag_fast(sof, (double)f);
#else
  const float lfmagnf0 = (float)lf::magnitude(so, (double)f);
#endif
  __syncthreads();
  // spectrum of the summed pulses (REAL: bins 0 .. size / 2 in natural order; else bit-reversed, all `size` bins)
  const int logN = ilog2_dev(size);
  auto at = [&](int i) { return REAL ? i : brevN(i, logN); };
  if(REAL) {
    load_twiddles<NT>(TW, tw_glob, size / 2, tw_nmax, lane);
    for(int i = lane; i < halfsize; i += NT) X[i] = make_float2(0.0f, 0.0f);
  } else {
    load_twiddles<NT>(TW, tw_glob, size, tw_nmax, lane);
    for(int i = lane; i < size; i += NT) X[brevN(i, logN)] = make_float2(0.0f, 0.0f);
  }
  __syncthreads();
  // a pulse no effect has edited carries the frame's own model (the host's lf::from_rd of the same Rd and F0, equal
  // up to the contraction of a few float64 operations): its solution is `so`
  lf::Solved sp = so; double pte = mo.te, ptp = mo.tp, pta = mo.ta, pT0 = mo.T0, pEe = mo.Ee;
  auto differs = [](double a, double b) { return fabs(a - b) > 1e-13 * fabs(b); };
  for(int p = 0; p < job.npulse; p ++) {
    const PbpPulse pu = pulses[job.first + p];
    if(differs(pu.te, pte) || differs(pu.tp, ptp) || differs(pu.ta, pta) || differs(pu.T0, pT0) || differs(pu.Ee, pEe)) {
      lf::Model mp; mp.T0 = pu.T0; mp.te = pu.te; mp.tp = pu.tp; mp.ta = pu.ta; mp.Ee = pu.Ee;
      sp = lf_solve_wave(mp, wl);
      pte = pu.te; ptp = pu.tp; pta = pu.ta; pT0 = pu.T0; pEe = pu.Ee;
    }
    const float phase_shift = -pu.offset - (float)job.pre_rotate;
    // The LF spectrum on the bins i = 1 + lane + 64 r (float64).  Its two phasors e^{-j w Te}, e^{-j w (T0 - Te)} advance
    // from bin to bin by constant rotations (seeded once per pulse and lane), the spectrum is rotated by the delay /
    // phase-delta term as a complex product -- no atan2 / polar round trip, no trigonometric call per bin.
    const double df = (double)fs / (double)size, tpi = 2.0 * 3.14159265358979323846;
    const float gscale = fnyq / lfmagnf0;
#if LF_FAST
    // float32 per bin (lf_spec_fast): the phasors e^{-j w Te}, e^{-j w D} are seeded at this lane's first bin from
    // float64-reduced phases and rotated by NT bins per step in float32 (at most size / 2 / NT steps: 8 at 2048 points)
    const LfFast spf = lf_fast(sp);
    float zc, zs, yc, ys, zrc, zrs, yrc, yrs;
    { float c, sn;
      cs_turns((double)(1 + lane) * df * spf.Te, & c, & sn); zc = c; zs = -sn;
      cs_turns((double)(1 + lane) * df * spf.D, & c, & sn); yc = c; ys = -sn;
      cs_turns((double)NT * df * spf.Te, & c, & sn); zrc = c; zrs = -sn;
      cs_turns((double)NT * df * spf.D, & c, & sn); yrc = c; yrs = -sn; }
#else
    const double Dd = sp.T0 - sp.Te;
    const double ea = exp(-sp.alpha * sp.Te), ed = exp(-sp.eps * Dd);
    double zc, zs, yc, ys;                             // e^{-j w Te}, e^{-j w D} at this lane's first bin
    { double sn, cs; sincos(tpi * (double)(1 + lane) * df * sp.Te, & sn, & cs); zc = cs; zs = -sn;
      sincos(tpi * (double)(1 + lane) * df * Dd, & sn, & cs); yc = cs; ys = -sn; }
    double zrc, zrs, yrc, yrs;                         // their steps over NT bins
    { double sn, cs; sincos(tpi * (double)NT * df * sp.Te, & sn, & cs); zrc = cs; zrs = -sn;
      sincos(tpi * (double)NT * df * Dd, & sn, & cs); yrc = cs; yrs = -sn; }
#endif
    for(int i = 1 + lane; i < halfsize; i += NT) {
      const double fqd = (double)i * df;
      const float fq = (float)fqd;
      // phase delta interpolated over the harmonics (cos / sin separately, then the direction of the sum)
      const float pos = fq / f;
      int k = (int)floorf(pos);
      float dc, ds;
      if(k >= n) { dc = PC[n]; ds = PS[n]; }
      else { const float r = pos - (float)k; dc = PC[k] + (PC[k + 1] - PC[k]) * r; ds = PS[k] + (PS[k + 1] - PS[k]) * r; }
      const float h2 = dc * dc + ds * ds;
      const float hinv = h2 > 0 ? __frsqrt_rn(h2) : 0.0f;
      const float ux = h2 > 0 ? dc * hinv : 1.0f, uy = ds * hinv;         // e^{j delta} (atan2(0, 0) = 0)
#if LF_FAST
      float re, im; lf_spec_fast(spf, (float)(tpi * fqd), zc, zs, yc, ys, & re, & im);
#else
      double re, im; lf::spectrum_core(sp, tpi * fqd, zc, zs, yc, ys, ea, ed, & re, & im);
#endif
      float ec, es; cs_turns((double)phase_shift * (double)i / (double)size - 0.25, & ec, & es);
      const float er = ec * ux - es * uy, ei = ec * uy + es * ux;
      const float g0 = gscale / fq;
      const float vr = (float)re * g0, vi = (float)im * g0;
      float2 v = X[at(i)];
      v.x += vr * er - vi * ei; v.y += vr * ei + vi * er;
      X[at(i)] = v;
#if LF_FAST
      { const float t = zc * zrc - zs * zrs; zs = zc * zrs + zs * zrc; zc = t; }
      { const float t = yc * yrc - ys * yrs; ys = yc * yrs + ys * yrc; yc = t; }
#else
      double t = zc * zrc - zs * zrs; zs = zc * zrs + zs * zrc; zc = t;
      t = yc * yrc - ys * yrs; ys = yc * yrs + ys * yrc; yc = t;
#endif
    }
    __syncthreads();
  }
  // lip radiation (llsm_lipfilter_reim with f0 = fs / size: bin i sees the response at (i + 1) fs / size),
  // vocal-tract magnitude, Hermitian completion
  for(int i = lane; i < halfsize; i += NT) {
    float lr, li; lip_resp_reim(lip_radius, fs / (float)size * (1.0f + (float)i) * 6.283185307179586f, & lr, & li);
    const float gain = expf(DB2LOG_F(interp_lin(env, nspec, fnyq, (float)i * fs / (float)size)));
    const float2 v = X[at(i)];
    float2 y = make_float2((v.x * lr - v.y * li) * gain, (v.x * li + v.y * lr) * gain);
    if(REAL) X[i] = y;
    else if(i == size / 2) {                                   // x[n/2] untouched by complete_(a)symm: keep both parts
      X[brevN(i, logN)] = y;
    } else {
      X[brevN(i, logN)] = y;
      if(i > 0) X[brevN(size - i, logN)] = make_float2(y.x, -y.y);
    }
  }
  __syncthreads();
  const float inv = 1.0f / (float)size;
  const int fadein = job.pre_rotate < 256 ? job.pre_rotate : 256, fadeout = size < 256 ? size : 256;
  float* dst = out + job.out_off;
  if(REAL) {
    // (only the real parts of bins 0 and size / 2 reach a real output: what the full-size transform's kept real part saw)
    const int M = size / 2, logM = logN - 1, ws = tw_nmax / size;
    for(int k = lane; k <= M / 2; k += NT) {
      const int k2 = M - k;
      float2 a = X[k], b = X[k2];
      if(k == 0) { a.y = 0.0f; b.y = 0.0f; }
      const float2 w = tw_glob[k * ws];                        // e^{-j 2 pi k / size}: conj = the factor wanted
      const float wr = w.x, wi = -w.y;
      const float sr = a.x + b.x, si = a.y - b.y, dr = a.x - b.x, di = a.y + b.y;
      const float p = wr * di + wi * dr, q = wr * dr - wi * di;
      // conj(Z): the inverse transform runs as conj(forward(conj Z))
      X[k] = make_float2(sr - p, -(si + q));
      if(k > 0 && k2 != k) X[k2] = make_float2(sr + p, -(q - si));
    }
    __syncthreads();
    fft_dif<NT>(X, TW, 1, M, logM, lane);                      // bit-reversed out; ends with a barrier
    for(int i = lane; i < size; i += NT) {
      const float2 v = X[brevN(i >> 1, logM)];
      float y = ((i & 1) ? -v.y : v.x) * inv;
      if(i < fadein) y *= (float)i / (float)fadein;
      if(i >= size - fadeout) y *= (float)(size - i) / (float)fadeout;
      dst[i] = y;
    }
    return;
  }
  ifft_dit<NT>(X, TW, 1, size, logN, lane);
  for(int i = lane; i < size; i += NT) {
    float y = X[i].x * inv;
    if(i < fadein) y *= (float)i / (float)fadein;
    if(i >= size - fadeout) y *= (float)(size - i) / (float)fadeout;
    dst[i] = y;
  }
}

// cross-fade curve: one thread per frame segment (layer0.c:240-262); the float64 state is carried by the
// host scheduler from segment to segment, each segment replays its own additions
__global__ void k_l1_mixcurve(const PbpSeg* __restrict__ segs, int nsegs, float* __restrict__ mixw) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if(t >= nsegs) return;
  const PbpSeg s = segs[t];
  double st = s.state;
  float* m = mixw + s.out_off;
  for(int j = s.j0; j < s.j1; j ++) {
    if(s.dir > 0) { if(st < 1.0) st += s.rate; }
    else if(s.dir < 0) { if(st > 0) st -= s.rate; }
    if(j >= 0 && j < s.len) m[j] = (float)st;
  }
}

// y_sin = y_hm (1 - w) + y_pbp w;  y = y_sin + y_noise.  Thread per output sample; the harmonic frames
// (HBM, nwin samples each, centred on trunc(i thop fs)) and the pulse groups are gathered in ascending order.
__global__ __launch_bounds__(256) void k_pbp_mix(
  const int* __restrict__ out_off, const int* __restrict__ out_len, const int* __restrict__ frm_off,
  const int* __restrict__ nfrm, float thop, float fs, int nwin,
  const float* __restrict__ hm_frames, const float* __restrict__ f0_hm,
  const PbpJob* __restrict__ jobs, const int2* __restrict__ blk_jobs, const int* __restrict__ blk_off,
  const float* __restrict__ pulse_buf, const float* __restrict__ mixw,
  const float* __restrict__ ynoise, float* __restrict__ ysin, float* __restrict__ y) {
  const int u = blockIdx.y, len = out_len[u];
  const int p = blockIdx.x * 256 + threadIdx.x;
  if(p >= len) return;
  const size_t oo = (size_t)out_off[u];
  const int fo = frm_off[u], nf = nfrm[u];
  // harmonic frames covering p: trunc(i thop fs) - nwin / 2 <= p < ... + nwin
  float hm = 0.0f;
  {
    const float hop = lp::fmul(thop, fs);
    int i0 = (int)floorf((float)(p - nwin / 2) / hop) - 1; if(i0 < 0) i0 = 0;
    int i1 = (int)floorf((float)(p + nwin / 2) / hop) + 2; if(i1 > nf - 1) i1 = nf - 1;
    for(int i = i0; i <= i1; i ++) {
      if(!(f0_hm[fo + i] > 0)) continue;
      const int base = (int)lp::fmul(lp::fmul((float)i, thop), fs);
      const int j = p - base + nwin / 2;
      if(j >= 0 && j < nwin) hm += hm_frames[(size_t)(fo + i) * nwin + j];
    }
  }
  float pb = 0.0f;
  const int2 jr = blk_jobs[blk_off[u] + blockIdx.x];
  for(int q = jr.x; q < jr.y; q ++) {
    const PbpJob job = jobs[q];
    const int k = p - job.start;
    if(k >= 0 && k < job.size) pb += pulse_buf[job.out_off + k];
    // (int) truncates towards zero: the sample whose index falls in (-1, 0) also lands on 0
    if(p == 0 && job.zero_extra >= 0 && job.zero_extra < job.size) pb += pulse_buf[job.out_off + job.zero_extra];
  }
  const float w = mixw[oo + p];
  const float v = (float)((double)hm * (1.0 - (double)w) + (double)pb * (double)w);
  ysin[oo + p] = v;
  y[oo + p] = v + ynoise[oo + p];
}

// llsmrt pulse-by-pulse bookkeeping of one hop, one block per stream (llsmrt.c:118-128, 380-419):
// dual-buffer forward, the new pulse group added, the windowed read into the sinusoid ring and the
// trapezoid catch-up at termination.  All streams of a group share the ring cursors (lock-step hops).
__global__ __launch_bounds__(256) void k_rt_pbp(
  const RtPbpOp* __restrict__ ops, float* __restrict__ frwd, float* __restrict__ bkwd, int cap, int dual_curr,
  float* __restrict__ sinr, int sin_curr, int nhop, const float* __restrict__ win,
  const float* __restrict__ pulse_out, int pulse_stride) {
  rt_pbp_body(ops, frwd, bkwd, cap, dual_curr, sinr, sin_curr, nhop, win, pulse_out, pulse_stride, blockIdx.x, threadIdx.x, true);
}

// =====================================================================
// Frame coder (coder.c:88-286): one wavefront per frame.
//   k_coder_encode   frame -> [voicing, f0, rd, order_spec mel-cepstral-domain spectrum points, order_bap band aperiodicities]
//   k_coder_decode   the inverse, to layer-0 (AMPL / PHSE) or layer-1 (VTMAGN / VSPHSE) rows
// ddct (Ooura): DCT-II  C[k] = sum_j a[j] cos(pi (j + 1/2) k / n)  and its inverse.
// Only order_spec coefficients of the 1024-point transforms are non-zero / needed, so both are direct
// O(n order_spec) sums with phasor recurrences (re-seeded from exact phases every 64 terms).
// LDS: three float rows of ns, then the minimum-phase scratch (decode to layer 0).
// =====================================================================
struct CoderDev { int order_spec, order_bap, ns, npsd, maxnhar; float fnyq, liprad; const float* melaxis; float mel_floor, mel_ceil; };

DEV float lip_mag(float radius, float omega) { float m, a; lip_resp(radius, omega, & m, & a); return m; }

__global__ __launch_bounds__(WAVE) void k_coder_encode(CoderDev c, int nframes, const float* __restrict__ f0v,
  const float* __restrict__ rdv, const float* __restrict__ psd, const float* __restrict__ vtmagn,
  const int* __restrict__ nvsphse, float* __restrict__ enc) {
  const int g = blockIdx.x, lane = threadIdx.x, ns = c.ns, N = ns - 1, os = c.order_spec;
  const int dim = os + c.order_bap + 3;
  float* SP = (float*)l1_lds; float* EN = SP + ns; float* ML = EN + ns;
  float* out = enc + (size_t)g * dim;
  const float f0 = f0v[g];
  const bool voiced = f0 > 0 && nvsphse[g] > 0;
  for(int j = lane; j < ns; j += WAVE) {
    const float fj = (float)j * c.fnyq / (float)N;
    SP[j] = expf(interp_lin(psd + (size_t)g * c.npsd, c.npsd, c.fnyq, fj) * (2.3025851f / 10.0f));
  }
  if(lane == 0) { out[0] = f0 > 0 ? 1.0f : 0.0f; out[1] = f0; out[2] = 0.0f; }
  __syncthreads();
  if(voiced) {
    const float rd = rdv[g];
    if(lane == 0) out[2] = rd;
    lf::Model m = lf::from_rd((double)rd, 1.0 / (double)f0, 1.0, g_conv_l1.lf_rd_clamp);
    const lf::Solved s = lf_solve_wave(m, lane);
    const float lf0 = (float)lf::magnitude(s, (double)f0);
    const float* vt = vtmagn + (size_t)g * ns;
    for(int j = lane; j < ns; j += WAVE) {
      const int jj = j == 0 ? 1 : j;
      const float fj = (float)jj * c.fnyq / (float)N;
      float e = expf(DB2LOG_F(vt[jj])) * (float)lf::magnitude(s, (double)fj) / lf0 * f0 / fj;
      // llsm_lipfilter(liprad, fnyq / ns, ns, ...): entry j sees the response at (j + 1) fnyq / ns (coder.c:119)
      e *= lip_mag(c.liprad, c.fnyq / (float)ns * (1.0f + (float)j) * 6.283185307179586f);
      if(j >= 1) e *= e * 44100.0f / 4.0f / f0;
      EN[j] = e;
    }
    __syncthreads();
    for(int j = lane; j < ns; j += WAVE) SP[j] += EN[j];
    __syncthreads();
    for(int b = 0; b < c.order_bap; b ++) {
      const int n0 = b * N / c.order_bap, n1 = (b + 1) * N / c.order_bap;
      float acc = 0;
      for(int k = n0 + lane; k < n1; k += WAVE) acc += 1.0f - EN[k] / SP[k];
      acc = wave_sum(acc);
      if(lane == 0) out[3 + os + b] = acc / (float)(n1 - n0);
    }
  } else if(lane < c.order_bap) out[3 + os + lane] = 1.0f;
  __syncthreads();
  for(int j = lane; j < ns; j += WAVE) EN[j] = logf(SP[j]) * 0.5f;
  __syncthreads();
  for(int j = lane; j < ns; j += WAVE) ML[j] = interp_lin(EN, ns, c.fnyq, c.melaxis[j]);
  __syncthreads();
  // DCT-II of ML[0 .. N), coefficients k < order_spec -> SP[k]
  for(int k = lane; k < os; k += WAVE) {
    const double tk = (double)k / (2.0 * (double)N);           // turns per unit of (j + 1/2)
    float cr = 1, ci = 0, dr, di, acc = 0;
    cs_turns(tk, & dr, & di);
    for(int j = 0; j < N; j ++) {
      if((j & 63) == 0) cs_turns(tk * ((double)j + 0.5), & cr, & ci);
      acc += ML[j] * cr;
      const float t = cr * dr - ci * di; ci = cr * di + ci * dr; cr = t;
    }
    SP[k] = k == 0 ? 0.5f * acc : acc;
  }
  __syncthreads();
  // inverse transform of length order_spec: out[m] = sum_k SP[k] cos(pi k (m + 1/2) / os) * 2 / N
  for(int mm = lane; mm < os; mm += WAVE) {
    float acc = 0;
    for(int k = 0; k < os; k ++) { float cr, ci; cs_turns((double)k * ((double)mm + 0.5) / (2.0 * (double)os), & cr, & ci); acc += SP[k] * cr; }
    out[3 + mm] = acc * 2.0f / (float)N;
  }
}

// interp1 on the (non-uniform, increasing) mel axis: position from the closed form, corrected against the table
DEV float interp_mel(const float* Y, const CoderDev& c, float f) {
  const int ns = c.ns;
  const float* ax = c.melaxis;
  if(f <= ax[0]) return Y[0];
  if(f >= ax[ns - 1]) return Y[ns - 1];
  const float mel = 1127.01048f * logf(1.0f + f / 700.0f);
  int k = (int)((mel - c.mel_floor) / (c.mel_ceil - c.mel_floor) * (float)ns);
  if(k < 0) k = 0;
  if(k > ns - 2) k = ns - 2;
  while(k > 0 && ax[k] > f) k --;
  while(k < ns - 2 && ax[k + 1] <= f) k ++;
  const float r = (f - ax[k]) / (ax[k + 1] - ax[k]);
  return Y[k] + (Y[k + 1] - Y[k]) * r;
}

__global__ __launch_bounds__(WAVE) void k_coder_decode(CoderDev c, int nframes, const float* __restrict__ enc, int use_l1,
  int nmax, const float2* __restrict__ tw_glob, int tw_nmax,
  float* __restrict__ f0v, float* __restrict__ rdv, int* __restrict__ nharv, float* __restrict__ ampl,
  float* __restrict__ phse, float* __restrict__ psd, float* __restrict__ vtmagn, float* __restrict__ vsphse,
  int* __restrict__ nvsphse, int* __restrict__ has_hm) {
  const int g = blockIdx.x, lane = threadIdx.x, ns = c.ns, N = ns - 1, os = c.order_spec, mh = c.maxnhar;
  const int dim = os + c.order_bap + 3;
  const float* src = enc + (size_t)g * dim;
  float* MP = (float*)l1_lds; float* FS_ = MP + ns; float* AP = FS_ + ns;       // mel spectrum | full spectrum | aperiodicity
  const int nh4 = (mh + 3) & ~3;
  float* A = AP + ns; float* VT = A + nh4;
  float2* X = (float2*)(VT + nh4); float2* TW = X + nmax;
  const bool voicing = src[0] > 0.5f;
  const float f0 = fmaxf(20.0f, src[1]);
  const float rd = fminf(3.0f, fmaxf(0.02f, src[2]));
  int nhar = voicing ? (int)(c.fnyq / f0) : 0;
  if(nhar > mh) nhar = mh;
  if(lane == 0) { f0v[g] = voicing ? f0 : 0.0f; rdv[g] = rd; nharv[g] = (nhar > 0 && use_l1) ? 0 : nhar;
    nvsphse[g] = (nhar > 0 && use_l1) ? nhar : 0; has_hm[g] = (nhar > 0 && use_l1) ? 0 : 1; }
  // undo the low-order inverse transform: DCT-II of length os
  for(int k = lane; k < os; k += WAVE) {
    float acc = 0;
    for(int j = 0; j < os; j ++) { float cr, ci; cs_turns(((double)j + 0.5) * (double)k / (2.0 * (double)os), & cr, & ci);
      acc += src[3 + j] * 0.5f * (float)N * 2.0f / (float)os * cr; }
    FS_[k] = k == 0 ? 0.5f * acc : acc;
  }
  __syncthreads();
  // full-order inverse transform from the os non-zero coefficients
  for(int j = lane; j < N; j += WAVE) {
    const double tj = ((double)j + 0.5) / (2.0 * (double)N);
    float cr = 1, ci = 0, dr, di, acc = 0;
    cs_turns(tj, & dr, & di);
    for(int k = 0; k < os; k ++) {
      acc += FS_[k] * cr;
      const float t = cr * dr - ci * di; ci = cr * di + ci * dr; cr = t;
    }
    MP[j] = acc * 2.0f / (float)N;
  }
  __syncthreads();
  if(lane == 0) MP[ns - 1] = MP[ns - 2];
  __syncthreads();
  for(int j = lane; j < ns; j += WAVE) {
    const float fj = (float)j * c.fnyq / (float)N;             // faxis[j]
    float p = interp_mel(MP, c, fj);
    // band aperiodicity on linspace(0, fnyq, order_bap + 1), bap_pad[0] = voicing ? 0 : 1
    // (interp1's own form, r = (x - x_k) / (x_k+1 - x_k) on the knots as linspace stores them: next to a knot of
    // aperiodicity 1 the decoder divides by 1 - ap, and `pos - floor(pos)` of the uniform-grid shortcut carries the
    // rounding of pos -- several ulps of r, where this form has one)
    const int ob = c.order_bap;
    int k = (int)floorf(fj / c.fnyq * (float)ob); if(k > ob - 1) k = ob - 1; if(k < 0) k = 0;
    auto knot = [&](int q) { return (float)((double)c.fnyq * (double)q / (double)ob); };
    while(k < ob - 1 && knot(k + 1) <= fj) k ++;
    while(k > 0 && knot(k) > fj) k --;
    const float x0 = knot(k), x1 = knot(k + 1);
    const float b0 = k == 0 ? (voicing ? 0.0f : 1.0f) : src[3 + os + k - 1], b1 = src[3 + os + k];
    float ap = fj >= knot(ob) ? src[3 + os + ob - 1] : fj <= 0.0f ? b0 : b0 + (b1 - b0) * ((fj - x0) / (x1 - x0));
    if(voicing) {
      const float fz = (float)j * c.fnyq / (float)ns;
      if(fz < 500.0f) ap = 1e-3f;
      else if(fz < 2000.0f) ap = 1e-3f + (ap - 1e-3f) * (fz - 500.0f) / 1500.0f;
    }
    const float sum_psd = expf(2.0f * p);
    FS_[j] = sqrtf(sum_psd * (1.0f - ap) * f0 * 4.0f / 44100.0f);
    AP[j] = sum_psd * ap;
  }
  __syncthreads();
  for(int i = lane; i < c.npsd; i += WAVE) {
    const float fq = c.fnyq * (float)i / (float)(c.npsd - 1);
    psd[(size_t)g * c.npsd + i] = logf(interp_lin(AP, ns, c.fnyq, fq)) / 2.3025851f * 10.0f;
  }
  if(nhar <= 0) return;
  lf::Model m = lf::from_rd((double)rd, 1.0 / (double)f0, 1.0, g_conv_l1.lf_rd_clamp);
  const lf::Solved s = lf_solve_wave(m, lane);
  if(use_l1) {
    const float lf0 = (float)lf::magnitude(s, (double)f0);
    for(int j = lane; j < ns; j += WAVE) {
      const int jj = j == 0 ? 1 : j;
      const float fj = (float)jj * c.fnyq / (float)N;
      const float sp = FS_[jj] / lip_mag(c.liprad, c.fnyq / (float)ns * (1.0f + (float)jj) * 6.283185307179586f);
      vtmagn[(size_t)g * ns + j] = logf(sp * fj / f0 * lf0 / (float)lf::magnitude(s, (double)fj)) / 2.3025851f * 20.0f;
    }
    for(int i = lane; i < nhar; i += WAVE)
      vsphse[(size_t)g * mh + i] = (float)lf::phase(s, (double)((float)nhar * f0) * (double)(i + 1) / (double)nhar);
    for(int i = nhar + lane; i < mh; i += WAVE) vsphse[(size_t)g * mh + i] = 0.0f;
    return;
  }
  // layer 0: amplitudes sampled from the full spectrum, minimum-phase vocal tract + LF source phases
  const double m1 = lf::magnitude(s, (double)((float)nhar * f0) * 1.0 / (double)nhar);
  for(int i = lane; i < nhar; i += WAVE) {
    const float fh = (float)((double)((float)nhar * f0) * (double)(i + 1) / (double)nhar);
    const float a = interp_lin(FS_, ns, c.fnyq, fh);
    ampl[(size_t)g * mh + i] = a;
    const float vs = (float)(lf::magnitude(s, (double)fh) / (i + 1.0) / m1);
    A[i] = a / lip_mag(c.liprad, (float)((double)f0 * (1.0 + i) * 2.0 * 3.14159265358979323846)) / vs;
  }
  for(int i = nhar + lane; i < mh; i += WAVE) { ampl[(size_t)g * mh + i] = 0.0f; phse[(size_t)g * mh + i] = 0.0f; }
  __syncthreads();
  const int Nm = minphase_fftsize(nhar);
  load_twiddles(TW, tw_glob, Nm, tw_nmax, lane);
  __syncthreads();
  harmonic_minphase_dev(A, nhar, X, TW, Nm, VT, lane);
  for(int i = lane; i < nhar; i += WAVE) {
    const float fh = (float)((double)((float)nhar * f0) * (double)(i + 1) / (double)nhar);
    float mg, ar; lip_resp(c.liprad, (float)((double)f0 * (1.0 + i) * 2.0 * 3.14159265358979323846), & mg, & ar);
    phse[(size_t)g * mh + i] = VT[i] + ar + (float)lf::phase(s, (double)fh);
  }
}

// ---------------------------------------------------------------- launchers
#define L1_LAUNCH(name, kern, grid, block, lds, ...)                                  \
  do {                                                                                \
    if(P -> prof_begin) P -> prof_begin(P -> prof_user, name, P -> stream);                        \
    hipLaunchKernelGGL(kern, grid, block, lds, P -> stream, __VA_ARGS__);             \
    if(P -> prof_end) P -> prof_end(P -> prof_user, P -> stream);                                  \
    hipError_t e_ = hipGetLastError();                                                \
    if(e_ != hipSuccess) return (int)e_;                                              \
  } while(0)

static size_t l1_lds_bytes(int maxnhar, int nmax, int extra_rows) {
  const int nh4 = (maxnhar + 3) & ~3;
  return sizeof(float) * (size_t)(nh4 * 3 + extra_rows) + sizeof(float2) * ((size_t)nmax + nmax / 2);
}
static int l1_set_lds(const void* fn, size_t bytes) {
  if(bytes <= 64 * 1024) return 0;
  if(bytes > 160 * 1024) return -1;
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}
static int pow2ge(int n) { int p = 1; while(p < n) p <<= 1; return p; }
int l1_minphase_nmax(int maxnhar) { int n = pow2ge(maxnhar) * 4; return n < 64 ? 64 : n; }

int launch_l1_rd_fit(LaunchCtx* P, const L1Dev& d, const float* model_power, const float* model_param,
  const double* inv_t, const double* cumlog_t, float* rd_raw) {
  if(d.nframes == 0) return 0;
  L1_LAUNCH("k_l1_rd_fit", k_l1_rd_fit, dim3(d.nframes), dim3(WAVE), sizeof(float) * RD_NHAR,
    d.nframes, d.f0, d.nhar, d.ampl, d.maxnhar, d.lip_radius, model_power, model_param, inv_t, cumlog_t, rd_raw);
  return 0;
}
int launch_coder_encode(LaunchCtx* P, int order_spec, int order_bap, int ns, int npsd, float fnyq, float liprad,
  const float* melaxis, int nframes, const float* f0, const float* rd, const float* psd, const float* vtmagn,
  const int* nvsphse, float* enc) {
  if(nframes == 0) return 0;
  CoderDev c; c.order_spec = order_spec; c.order_bap = order_bap; c.ns = ns; c.npsd = npsd; c.maxnhar = 0; c.fnyq = fnyq;
  c.liprad = liprad; c.melaxis = melaxis; c.mel_floor = 0; c.mel_ceil = 0;
  const size_t lds = sizeof(float) * 3 * (size_t)ns;
  if(order_spec > ns - 1 || l1_set_lds((const void*)k_coder_encode, lds)) return -1;
  L1_LAUNCH("k_coder_encode", k_coder_encode, dim3(nframes), dim3(WAVE), lds, c, nframes, f0, rd, psd, vtmagn, nvsphse, enc);
  return 0;
}
int launch_coder_decode(LaunchCtx* P, int order_spec, int order_bap, int ns, int npsd, int maxnhar, float fnyq, float liprad,
  const float* melaxis, float mel_floor, float mel_ceil, int nframes, const float* enc, int use_l1, const float2* tw,
  int tw_nmax, float* f0, float* rd, int* nhar, float* ampl, float* phse, float* psd, float* vtmagn, float* vsphse,
  int* nvsphse, int* has_hm) {
  if(nframes == 0) return 0;
  CoderDev c; c.order_spec = order_spec; c.order_bap = order_bap; c.ns = ns; c.npsd = npsd; c.maxnhar = maxnhar; c.fnyq = fnyq;
  c.liprad = liprad; c.melaxis = melaxis; c.mel_floor = mel_floor; c.mel_ceil = mel_ceil;
  const int nmax = l1_minphase_nmax(maxnhar);
  if(nmax > tw_nmax || order_spec > ns - 1) return -1;
  const size_t lds = sizeof(float) * (3 * (size_t)ns + 2 * (size_t)((maxnhar + 3) & ~3)) + sizeof(float2) * ((size_t)nmax + nmax / 2);
  if(l1_set_lds((const void*)k_coder_decode, lds)) return -1;
  L1_LAUNCH("k_coder_decode", k_coder_decode, dim3(nframes), dim3(WAVE), lds, c, nframes, enc, use_l1, nmax, tw, tw_nmax, f0, rd,
    nhar, ampl, phse, psd, vtmagn, vsphse, nvsphse, has_hm);
  return 0;
}
int launch_fa_glottal_fit(LaunchCtx* P, const float* ampl, int nhar, const float* model_power, const float* model_param,
  int ncand, int nhm, float* out) {
  if(ncand < 1 || ncand > WAVE || nhm < 1) return -1;
  L1_LAUNCH("k_fa_glottal_fit", k_fa_glottal_fit, dim3(1), dim3(WAVE), sizeof(float) * (size_t)nhm, ampl, nhar, model_power,
    model_param, ncand, nhm, out);
  return 0;
}
// what: 0 minimum phase (out[nhar]), 1 harmonic spectrum (out[nfft / 2 + 1]), 2 envelope in dB (out[nfft / 2 + 1])
int launch_fa_l1_frame(LaunchCtx* P, const float* ampl, int nhar, double f0d, int nfft, int what, const float2* tw,
  int tw_nmax, float* out) {
  if(nhar <= 0) return -1;
  int nmax = l1_minphase_nmax(nhar); if(what != 0 && nfft > nmax) nmax = nfft;
  if(nmax > tw_nmax || (what != 0 && (nfft < 4 || (nfft & (nfft - 1))))) return -1;
  const size_t lds = sizeof(float) * (size_t)(((nhar + 3) & ~3) * 2) + sizeof(float2) * ((size_t)nmax + nmax / 2);
  if(l1_set_lds((const void*)k_fa_l1_frame, lds)) return -1;
  L1_LAUNCH("k_fa_l1_frame", k_fa_l1_frame, dim3(1), dim3(WAVE), lds, ampl, nhar, f0d, nfft, what, nmax, tw, tw_nmax, out);
  return 0;
}
int launch_l1_rd_smooth(LaunchCtx* P, int n_utt, const int* frm_off, const int* nfrm, int order,
  const float* rd_raw, int* prev_idx, int* next_idx, float* cont, float* rd_out) {
  if(n_utt == 0) return 0;
  L1_LAUNCH("k_l1_rd_smooth", k_l1_rd_smooth, dim3(n_utt), dim3(256), 0, frm_off, nfrm, order, rd_raw, prev_idx,
    next_idx, cont, rd_out);
  return 0;
}
int launch_l1_frame(LaunchCtx* P, const L1Dev& d, int nfft, const float2* tw, int tw_nmax) {
  if(d.nframes == 0) return 0;
  int logn = 0; while((1 << logn) < nfft) logn ++;
  // the envelope on the register-resident FFT, two frames per transform, when there is a plan for nfft and the
  // caller brought the scratch rows; otherwise inside k_l1_frame on the LDS FFT
  const bool split = d.src_ampl && (1 << logn) == nfft && logn >= 10 && logn <= 11;   // 4096: 128 data registers per lane
  if(! split) {
    int nmax = l1_minphase_nmax(d.maxnhar); if(nfft > nmax) nmax = nfft;
    if(nmax > tw_nmax) return -1;
    const size_t lds = l1_lds_bytes(d.maxnhar, nmax, 0);
    if(l1_set_lds((const void*)k_l1_frame, lds)) return -1;
    L1_LAUNCH("k_l1_frame", k_l1_frame, dim3(d.nframes), dim3(WAVE), lds, d.nframes, d.f0, d.nhar, d.ampl, d.phse,
      d.maxnhar, d.rd, d.lip_radius, d.fnyq, nfft, nmax, tw, tw_nmax, d.vtmagn, d.vsphse, d.nvsphse, (float*)nullptr, d.acache);
    return 0;
  }
  {
    const int nmax = l1_minphase_nmax(d.maxnhar);
    if(nmax > tw_nmax) return -1;
    const size_t lds = l1_lds_bytes(d.maxnhar, nmax, 0);
    if(l1_set_lds((const void*)k_l1_frame, lds)) return -1;
    L1_LAUNCH("k_l1_frame", k_l1_frame, dim3(d.nframes), dim3(WAVE), lds, d.nframes, d.f0, d.nhar, d.ampl, d.phse,
      d.maxnhar, d.rd, d.lip_radius, d.fnyq, nfft, nmax, tw, tw_nmax, d.vtmagn, d.vsphse, d.nvsphse, d.src_ampl, d.acache);
  }
  const int npair = d.pairs ? d.npairs : (d.nframes + 1) / 2;
  const int nh4 = (d.maxnhar + 3) & ~3;
  const int grid = npair < 4096 ? npair : 4096;
#define ENV_CASE(LN) \
  if(logn == LN) { \
    const size_t lds = std::max(sizeof(float2) * (size_t)wf_lds_elems<LN>(), \
      sizeof(double4) * 2 * (size_t)nh4 + sizeof(float) * 2 * (((size_t)1 << LN) / WAVE / 2 + 1) * WAVE) + sizeof(float) * 4 * (size_t)nh4; \
    if(l1_set_lds((const void*)k_l1_env_wf<LN>, lds)) return -1; \
    L1_LAUNCH("k_l1_env_wf", (k_l1_env_wf<LN>), dim3(grid), dim3(WAVE), lds, d.nframes, d.f0, d.nvsphse, d.src_ampl, \
      d.maxnhar, d.fnyq, d.vtmagn, d.pairs, npair); \
    return 0; \
  }
  ENV_CASE(10) ENV_CASE(11)
#undef ENV_CASE
  return -1;
}
int launch_l1_to_l0(LaunchCtx* P, const L1Dev& d, int maxnhar_conf, int only_missing, const int* select,
  const float2* tw, int tw_nmax) {
  if(d.nframes == 0) return 0;
  const int nmax = l1_minphase_nmax(d.maxnhar);
  if(nmax > tw_nmax) return -1;
  const size_t lds = l1_lds_bytes(d.maxnhar, nmax, 0);
  if(l1_set_lds((const void*)k_l1_to_l0, lds)) return -1;
  L1_LAUNCH("k_l1_to_l0", k_l1_to_l0, dim3(d.nframes), dim3(WAVE), lds, d.nframes, d.f0, d.nhar, d.ampl, d.phse,
    d.maxnhar, d.rd, d.lip_radius, d.fnyq, d.nspec, maxnhar_conf, only_missing, select, nmax, tw, tw_nmax,
    d.vtmagn, d.vsphse, d.nvsphse, d.has_hm, d.acache);
  return 0;
}
static int g_pbp_real = [] { const char* e = std::getenv("LLSM_GPU_PBP_REAL"); return (e && e[0] == '0') ? 0 : 1; }();   // llsm_gpu_pbp_real_ifft
int l1_pbp_real_ifft(int on) { const int prev = g_pbp_real; if(on >= 0) g_pbp_real = on ? 1 : 0; return prev; }
int launch_pbp_pulse(LaunchCtx* P, const L1Dev& d, const PbpJob* jobs, int njobs, const PbpPulse* pulses,
  int size_max, float fs, const float2* tw, int tw_nmax, float* out) {
  if(njobs == 0) return 0;
  const int nmin = l1_minphase_nmax(d.maxnhar);
  int nmax = nmin; if(size_max > nmax) nmax = size_max;
  if(nmax > tw_nmax) return -1;
  // real-output inverse transform: X holds size / 2 + 1 bins (and the minimum-phase transform before it), the
  // twiddles of size / 2 points.  Pulse groups are powers of two >= NSPEC; anything below 32 samples keeps the full form.
  const bool real = g_pbp_real && size_max >= 32;
  const int xcap = real ? std::max(nmin, (size_max / 2 + 4) & ~3) : nmax;
  const int twcap = real ? std::max(nmin / 2, size_max / 4) : nmax / 2;
  const size_t lds = sizeof(float) * (size_t)(((d.maxnhar + 3) & ~3) * 4 + 8) + sizeof(float2) * ((size_t)xcap + twcap);
#define PBP_GO(NTH, RL) do { \
    if(l1_set_lds((const void*)k_pbp_pulse<NTH, RL>, lds)) return -1; \
    L1_LAUNCH("k_pbp_pulse", (k_pbp_pulse<NTH, RL>), dim3(njobs), dim3(NTH), lds, jobs, pulses, d.f0, d.rd, d.vtmagn, d.nspec, \
      d.vsphse, d.nvsphse, d.maxnhar, d.fnyq, d.lip_radius, fs, xcap, tw, tw_nmax, out, d.acache); } while(0)
  // Fewer pulse groups than the device has compute units (a hop of llsmrt: at most one per stream): the launch waits for
  // the slowest group's chain, so a group gets four wavefronts; a batch of thousands is throughput-bound and keeps fewer.
  if(njobs <= PBP_WIDE_BELOW) { if(real) PBP_GO(256, true); else PBP_GO(256, false); }
  else { if(real) PBP_GO(PBP_NT, true); else PBP_GO(PBP_NT, false); }
#undef PBP_GO
  return 0;
}
int launch_rt_pbp(LaunchCtx* P, int S, const RtPbpOp* ops, float* frwd, float* bkwd, int cap, int dual_curr,
  float* sinr, int sin_curr, int nhop, const float* win, const float* pulse_out, int pulse_stride) {
  L1_LAUNCH("k_rt_pbp", k_rt_pbp, dim3(S), dim3(256), 0, ops, frwd, bkwd, cap, dual_curr, sinr, sin_curr, nhop, win,
    pulse_out, pulse_stride);
  return 0;
}
int launch_l1_projection(LaunchCtx* P, const L1Dev& d, double fs, double* proj) {
  if(d.nframes == 0) return 0;
  L1_LAUNCH("k_l1_projection", k_l1_projection, dim3(d.nframes), dim3(WAVE), 0, d.nframes, d.f0, d.rd, d.vsphse, d.nvsphse,
    d.maxnhar, fs, proj, d.acache);
  return 0;
}
int launch_l1_mixcurve(LaunchCtx* P, const PbpSeg* segs, int nsegs, float* mixw) {
  if(nsegs == 0) return 0;
  L1_LAUNCH("k_l1_mixcurve", k_l1_mixcurve, dim3((nsegs + 63) / 64), dim3(64), 0, segs, nsegs, mixw);
  return 0;
}
int launch_pbp_mix(LaunchCtx* P, int n_utt, int max_len, const int* out_off, const int* out_len, const int* frm_off,
  const int* nfrm, float thop, float fs, int nwin, const float* hm_frames, const float* f0_hm, const PbpJob* jobs,
  const int2* blk_jobs, const int* blk_off, const float* pulse_buf, const float* mixw, const float* ynoise,
  float* ysin, float* y) {
  if(n_utt == 0 || max_len == 0) return 0;
  L1_LAUNCH("k_pbp_mix", k_pbp_mix, dim3((max_len + 255) / 256, n_utt), dim3(256), 0, out_off, out_len, frm_off, nfrm,
    thop, fs, nwin, hm_frames, f0_hm, jobs, blk_jobs, blk_off, pulse_buf, mixw, ynoise, ysin, y);
  return 0;
}
