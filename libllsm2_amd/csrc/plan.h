// plan.h -- the integer index plan of the layer-0 path, shared verbatim by the
// host code and every HIP kernel (at hop 220.5 the
// reference's frame centres and window lengths sit on float32 rounding knife
// edges, so host and device must evaluate the same IEEE float32 expression).
//
// Every product is a single correctly rounded float32 (or, where the
// reference expression contains a double literal, float64) operation: on the
// device through the __f*_rn / __d*_rn intrinsics, which the compiler may not
// contract into an FMA; on the host through volatile temporaries (the library
// is also built with -ffp-contract=off).
#ifndef LLSM_AMD_PLAN_H
#define LLSM_AMD_PLAN_H

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LLSM_HD __host__ __device__ inline
#else
#define LLSM_HD inline
#endif

namespace llsm_plan {

LLSM_HD float fmul(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __fmul_rn(a, b);
#else
  volatile float r = a * b; return r;
#endif
}
LLSM_HD float fadd(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __fadd_rn(a, b);
#else
  volatile float r = a + b; return r;
#endif
}
LLSM_HD float fdiv(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __fdiv_rn(a, b);
#else
  volatile float r = a / b; return r;
#endif
}
LLSM_HD double dmul(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __dmul_rn(a, b);
#else
  volatile double r = a * b; return r;
#endif
}
LLSM_HD double ddiv(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __ddiv_rn(a, b);
#else
  volatile double r = a / b; return r;
#endif
}
// C round(): half away from zero.
LLSM_HD int iround(double x) { return (int)round(x); }

// round(i * thop * fs): dsputils.c:191; layer0.c:332, 429, 588
LLSM_HD int center(int i, float thop, float fs) {
  return iround((double)fmul(fmul((float)i, thop), fs));
}
// rawidx = i*thop*fs, baseidx = round(rawidx); returns rawidx - baseidx (layer0.c:127-129)
LLSM_HD float rawfrac(int i, float thop, float fs, int* baseidx) {
  float raw = fmul(fmul((float)i, thop), fs);
  int base = iround((double)raw);
  *baseidx = base;
  return fadd(raw, -(float)base);
}
LLSM_HD int nwin_sin(float thop, float fs) {            // layer0.c:121
  return iround((double)fmul(thop, fs)) * 2;
}
LLSM_HD int nwin_env(float thop, float fs) {            // layer0.c:293 (2.0 is a double)
  return iround(dmul(dmul((double)thop, 2.0), (double)fs));
}
LLSM_HD int nwin_filt(float thop, float fs) {           // layer0.c:560
  return iround((double)fmul(fmul(thop, fs), 2.0f));
}
LLSM_HD int nwin_psd(float thop, float fs) {            // layer0.c:320
  return iround((double)fmul(fmul(thop, 4.0f), fs));
}
LLSM_HD int ny(int nfrm, float thop, float fs) {        // layer0.c:643
  return iround((double)fmul(fmul((float)(nfrm + 1), thop), fs));
}
LLSM_HD int hwin(float f0, float fs, float rel) {       // dsputils.c:190
  return iround((double)fdiv(fmul(fdiv(fs, f0), rel), 2.0f)) * 2;
}
LLSM_HD int nhar(float f0, float fs, int maxnhar) {     // dsputils.c:171-173, 206, 218
  int n = (int)floor((double)fdiv(fdiv(fs, f0), 2.0f));
  return n < maxnhar ? n : maxnhar;
}
LLSM_HD int env_ola(int i, int j, float thop, float fs) { // layer0.c:307
  return iround((double)fadd(fmul(fmul((float)(i - 1), thop), fs), (float)j));
}
LLSM_HD int dcwin(float f0, float thop, float fs) {     // layer0.c:430 (ternary is double)
  if(f0 == 0) return iround(dmul((double)fmul(thop, 2.0f), (double)fs));
  return iround(dmul(ddiv(2.0, (double)f0), (double)fs));
}
LLSM_HD int spgmwin(float f0, float fs, int nwin_psd_) { // layer0.c:331 (int truncation)
  if(f0 == 0) return nwin_psd_;
  return (int)fmul(fdiv(fs, f0), 3.0f);
}

// pow(2, ceil(log2(x))) -- host only, per-batch constants.
inline int nextpow2(double x) { return (int)pow(2.0, ceil(log2(x))); }

// Counter-based Gaussian generator replacing ciglet randn over libc rand()
// (dsputils.c:353-361): splitmix64 of (seed, idx) -> two 24-bit uniforms ->
// Box-Muller cosine branch.  The integer stage is bit-identical everywhere.
LLSM_HD void rng_uniforms(unsigned long long seed, unsigned long long idx, float* u1, float* u2) {
  unsigned long long z = idx + seed * 0x9E3779B97F4A7C15ULL + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  *u1 = (float)((z >> 40) + 1) * (1.0f / 16777216.0f);
  *u2 = (float)((z >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
}

// Closed form of stretch_stationary_noise (dsputils.c:363-383): output sample p
// of a template of nx samples tiled to ny samples with `ov`-sample cross-fades.
// Returns the template index a; *b >= 0 and *r > 0 mean the sample is the
// cross-fade (x[a]*(1-r) + x[b]*r) / sqrt(2r(r-1)+1).
LLSM_HD int stretch_index(int p, int nx, int ny, int ov, int* b, float* r) {
  *b = -1; *r = 0;
  if(ny <= nx || p < nx - ov) return p;
  int T = nx - ov;
  int q = p - T;
  int m = q / T, ri = q - m * T;
  if(ri >= ov) return ri;
  // boundary m: head = nx + m*T; the fade is applied iff the tiling loop got there
  int head = nx + m * T;
  bool applied = (m == 0) ? (ny > nx) : (ny >= head);
  if(! applied) return T + ri;
  *b = ri;
  *r = (float)ri / (float)ov;
  return T + ri;
}

}  // namespace llsm_plan
#endif
