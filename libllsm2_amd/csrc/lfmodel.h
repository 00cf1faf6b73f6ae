// lfmodel.h -- Liljencrants-Fant glottal flow-derivative model, float64, host + device.
//
// The reference takes `lfmodel`, `lfmodel_from_rd` and `lfmodel_spectrum` from ciglet, which is
// neither vendored nor pinned; the definitions here are this library's own,
// from the published model:
//   Rd -> (Ra, Rk, Rg): Fant 1995 regression, the usual extension outside 0.21 <= Rd <= 2.7
//   waveform          E(t) = E0 e^{alpha t} sin(wg t),                                   0 <= t <= Te
//                     E(t) = -(Ee / (eps Ta)) (e^{-eps (t - Te)} - e^{-eps (T0 - Te)}),  Te < t <= T0
//                     eps Ta = 1 - e^{-eps (T0 - Te)}, alpha from zero net flow, E(Te) = -Ee
//   spectrum          the Fourier transform (e^{-j 2 pi f t}, t = 0 at the glottal opening) of E(t),
//                     in closed form; magnitude and phase
// Used by llsm_chunk_tolayer1 / llsm_frame_tolayer0 (layer1.c:48-195), by the pulse scheduler of the
// pulse-by-pulse synthesis on the host (layer0.c:181-198, llsmrt.c:316-333) and by the pulse kernels.
// te, tp, ta are relative to T0, as llsm_lfmodel_to_gfm / llsm_gfm_to_lfmodel require (llsmutils.c:24-43).
#ifndef LLSM_AMD_LFMODEL_H
#define LLSM_AMD_LFMODEL_H

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LF_HD __host__ __device__ inline
#else
#define LF_HD inline
#endif

namespace llsm_lf {

constexpr double kPi = 3.14159265358979323846;

struct Model { double T0, te, tp, ta, Ee; };                   // ciglet's `lfmodel` fields
struct Solved { double Te, Ta, T0, Ee, wg, eps, alpha, sw, cw; };

LF_HD Model from_rd(double rd, double T0, double Ee) {
  double Rap, Rkp, Rgp;
  if(rd < 0.21) Rap = 1e-6;
  else if(rd <= 2.7) Rap = (4.8 * rd - 1.0) / 100.0;
  else Rap = (32.3 / rd) / 100.0;
  if(rd <= 2.7) {
    Rkp = (22.4 + 11.8 * rd) / 100.0;
    Rgp = Rkp / (4.0 * (0.11 * rd / (0.5 + 1.2 * Rkp) - Rap));
  } else {
    const double OQupp = 1.0 - 1.0 / (2.17 * rd);
    Rgp = 9.3552e-3 + 596e-2 / (7.96 - 2.0 * OQupp);
    Rkp = 2.0 * Rgp * OQupp - 1.04;
  }
  Model m;
  m.T0 = T0; m.Ee = Ee;
  m.tp = 1.0 / (2.0 * Rgp);
  m.te = m.tp * (Rkp + 1.0);
  m.ta = Rap;
  return m;
}

// net flow of the open phase as a function of alpha (E0 eliminated through E(Te) = -Ee)
LF_HD double open_area(const Solved& s, double a) {
  return -s.Ee * (a - s.wg * s.cw / s.sw + s.wg * exp(-a * s.Te) / s.sw) / (a * a + s.wg * s.wg);
}
LF_HD double return_area(const Solved& s) {
  const double D = s.T0 - s.Te;
  return -(s.Ee / (s.eps * s.Ta)) * ((1.0 - exp(-s.eps * D)) / s.eps - D * exp(-s.eps * D));
}

// everything but alpha
LF_HD Solved prepare(const Model& m) {
  Solved s;
  s.T0 = m.T0; s.Te = m.te * m.T0; s.Ta = m.ta * m.T0; s.Ee = m.Ee;
  if(s.Te > 0.999 * s.T0) s.Te = 0.999 * s.T0;
  if(s.Ta < 1e-9 * s.T0) s.Ta = 1e-9 * s.T0;
  s.wg = kPi / (m.tp * m.T0);
  s.sw = sin(s.wg * s.Te); s.cw = cos(s.wg * s.Te);
  const double D = s.T0 - s.Te;
  double e = 1.0 / s.Ta;
  for(int it = 0; it < 100; it ++) {                           // Newton on e Ta - 1 + exp(-e D)
    const double g = e * s.Ta - 1.0 + exp(-e * D), dg = s.Ta - D * exp(-e * D);
    const double step = g / dg;
    e -= step;
    if(fabs(step) < 1e-15 * fabs(e)) break;
  }
  s.eps = e;
  s.alpha = 0;
  return s;
}

// alpha: bracket the sign change of the net flow on a grid in units of 1 / Te, then bisect
LF_HD Solved solve(const Model& m) {
  Solved s = prepare(m);
  const double Ar = return_area(s);
  double lo = 0, hi = 0, flo = 0; bool found = false;
  double prev = open_area(s, -60.0 / s.Te) + Ar;
  for(int k = -59; k <= 60 && ! found; k ++) {
    const double a = k / s.Te, f = open_area(s, a) + Ar;
    if((prev <= 0 && f > 0) || (prev >= 0 && f < 0)) { lo = (k - 1) / s.Te; hi = a; flo = prev; found = true; }
    prev = f;
  }
  if(! found) return s;
  for(int it = 0; it < 200; it ++) {
    const double mid = 0.5 * (lo + hi), f = open_area(s, mid) + Ar;
    if((f <= 0) == (flo <= 0)) { lo = mid; flo = f; } else hi = mid;
    if(hi - lo < 1e-15 * fmax(fabs(lo), fabs(hi))) break;
  }
  s.alpha = 0.5 * (lo + hi);
  return s;
}

// G(f) = open phase + return phase; re / im of the transform at frequency f (Hz)
// Fourier transform of the flow derivative at angular frequency w > 0, given the phasors e^{-j w Te} = cte + j ste and
// e^{-j w (T0 - Te)} = cd + j sd and the two exponentials ea = e^{-alpha Te}, ed = e^{-eps (T0 - Te)} (kernels that
// walk a frequency grid advance the phasors by a constant rotation instead of evaluating four trigonometric functions
// per bin).
// 1 / x: on the device v_rcp_f64 + two Newton steps (full float64 accuracy, a third of the cost of the IEEE division
// sequence; the pulse kernel evaluates six quotients per bin), on the host the plain division
LF_HD double recip(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return fma(fma(-x, r, 1.0), r, r);
#else
  return 1.0 / x;
#endif
}

LF_HD void spectrum_core(const Solved& s, double w, double cte, double ste, double cd, double sd, double ea, double ed,
  double* re, double* im) {
  // open phase: (-Ee / sw) (e^{-s Te} ((alpha - s) sw - wg cw) + wg e^{-alpha Te}) / ((alpha - s)^2 + wg^2),  s = j w
  const double ar = s.alpha, ai = -w;                            // alpha - s
  const double pr = ar * s.sw - s.wg * s.cw, pi_ = ai * s.sw;    // (alpha - s) sw - wg cw
  double nr = cte * pr - ste * pi_ + s.wg * ea, ni = cte * pi_ + ste * pr;
  const double dr = ar * ar - ai * ai + s.wg * s.wg, di = 2.0 * ar * ai;
  const double idn = recip(dr * dr + di * di), k0 = -s.Ee / s.sw;
  const double Or = k0 * (nr * dr + ni * di) * idn, Oi = k0 * (ni * dr - nr * di) * idn;
  // return phase: (1 - e^{-(eps + s) D}) / (eps + s)  -  e^{-eps D} (1 - e^{-s D}) / s,  1 / s = -j / w
  const double kr = -(s.Ee / (s.eps * s.Ta));
  const double t1r = 1.0 - ed * cd, t1i = -ed * sd;
  const double iq = recip(s.eps * s.eps + w * w), iw = recip(w);
  const double ur = (t1r * s.eps + t1i * w) * iq, ui = (t1i * s.eps - t1r * w) * iq;
  const double t2r = 1.0 - cd, t2i = -sd;
  const double vr = ed * t2i * iw, vi = -ed * t2r * iw;
  const double br = ur - vr, bi = ui - vi;
  const double Rr = kr * (cte * br - ste * bi), Ri = kr * (cte * bi + ste * br);
  *re = Or + Rr; *im = Oi + Ri;
}

LF_HD void spectrum(const Solved& s, double f, double* re, double* im) {
  const double w = 2.0 * kPi * f, D = s.T0 - s.Te;
  if(f == 0) {                                                   // net flow (zero by construction of alpha)
    const double ar = s.alpha;
    const double nr = ar * s.sw - s.wg * s.cw + s.wg * exp(-s.alpha * s.Te);
    const double Or = (-s.Ee / s.sw) * nr / (ar * ar + s.wg * s.wg);
    const double kr = -(s.Ee / (s.eps * s.Ta));
    *re = Or + kr * ((1.0 - exp(-s.eps * D)) / s.eps - D * exp(-s.eps * D)); *im = 0;
    return;
  }
  spectrum_core(s, w, cos(w * s.Te), -sin(w * s.Te), cos(w * D), -sin(w * D), exp(-s.alpha * s.Te), exp(-s.eps * D), re, im);
}

LF_HD double magnitude(const Solved& s, double f) { double r, i; spectrum(s, f, & r, & i); return sqrt(r * r + i * i); }
LF_HD double phase(const Solved& s, double f) { double r, i; spectrum(s, f, & r, & i); return atan2(i, r); }

}  // namespace llsm_lf
#endif
