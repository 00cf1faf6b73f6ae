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

// clamp (convention "lf_rd_clamp", llsm_gpu_set_convention): 0 = the regression with the usual extension outside
// 0.21 <= Rd <= 2.7 (default); 1 = Rd limited to the range Fant's regression was fitted on, 0.3 .. 2.7, first --
// the other common reading of "lfmodel_from_rd"; ciglet's own choice cannot be confirmed from the reference tree
LF_HD Model from_rd(double rd, double T0, double Ee, int clamp = 0) {
  if(clamp) rd = rd < 0.3 ? 0.3 : (rd > 2.7 ? 2.7 : rd);
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

// e^{-k}, k = -60 .. 60: the grid below evaluates the net flow at alpha = k / Te, where e^{-alpha Te} is one of these
LF_HD double exp_neg_int(int k) {
  static const double t[121] = {
  1.1420073898156842e+26, 4.2012104037905144e+25, 1.5455389355901039e+25, 5.685719999335932e+24,
  2.0916594960129961e+24, 7.6947852651420175e+23, 2.8307533032746939e+23, 1.0413759433029089e+23,
  3.8310080007165769e+22, 1.4093490824269389e+22, 5.184705528587072e+21, 1.9073465724950998e+21,
  7.0167359120976314e+20, 2.5813128861900675e+20, 9.4961194206024483e+19, 3.4934271057485095e+19,
  1.2851600114359308e+19, 4.7278394682293463e+18, 1.739274941520501e+18, 6.3984349353005491e+17,
  2.3538526683702e+17, 86593400423993744, 31855931757113756, 11719142372802612,
  4311231547115195, 1586013452313430.8, 583461742527454.88, 214643579785916.06,
  78962960182680.688, 29048849665247.426, 10686474581524.463, 3931334297144.042,
  1446257064291.4751, 532048240601.79865, 195729609428.83878, 72004899337.38588,
  26489122129.843472, 9744803446.2489033, 3584912846.1315918, 1318815734.4832146,
  485165195.40979028, 178482300.96318725, 65659969.13733051, 24154952.753575299,
  8886110.5205078721, 3269017.3724721107, 1202604.2841647768, 442413.39200892049,
  162754.79141900392, 59874.141715197817, 22026.465794806718, 8103.0839275753842,
  2980.9579870417283, 1096.6331584284585, 403.42879349273511, 148.4131591025766,
  54.598150033144236, 20.085536923187668, 7.3890560989306504, 2.7182818284590451,
  1, 0.36787944117144233, 0.1353352832366127, 0.049787068367863944,
  0.018315638888734179, 0.006737946999085467, 0.0024787521766663585, 0.00091188196555451624,
  0.00033546262790251185, 0.00012340980408667956, 4.5399929762484854e-05, 1.6701700790245659e-05,
  6.1442123533282098e-06, 2.2603294069810542e-06, 8.3152871910356788e-07, 3.0590232050182579e-07,
  1.1253517471925912e-07, 4.1399377187851668e-08, 1.5229979744712629e-08, 5.6027964375372678e-09,
  2.0611536224385579e-09, 7.5825604279119066e-10, 2.7894680928689246e-10, 1.026187963170189e-10,
  3.7751345442790977e-11, 1.3887943864964021e-11, 5.1090890280633251e-12, 1.8795288165390832e-12,
  6.914400106940203e-13, 2.5436656473769228e-13, 9.3576229688401748e-14, 3.4424771084699768e-14,
  1.2664165549094176e-14, 4.6588861451033977e-15, 1.713908431542013e-15, 6.3051167601469892e-16,
  2.3195228302435696e-16, 8.5330476257440658e-17, 3.1391327920480296e-17, 1.1548224173015786e-17,
  4.2483542552915889e-18, 1.5628821893349888e-18, 5.7495222642935599e-19, 2.1151310375910805e-19,
  7.7811322411337966e-20, 2.8625185805493937e-20, 1.0530617357553812e-20, 3.8739976286871868e-21,
  1.4251640827409352e-21, 5.2428856633634639e-22, 1.9287498479639178e-22, 7.0954741622847037e-23,
  2.6102790696677047e-23, 9.6026800545086756e-24, 3.5326285722008071e-24, 1.2995814250075031e-24,
  4.7808928838854688e-25, 1.7587922024243116e-25, 6.4702349256454599e-26, 2.3802664086944007e-26,
  8.75651076269652e-27
  };
  return t[k + 60];
}
// alpha: bracket the (leftmost) sign change of the net flow on a grid in units of 1 / Te, then refine it inside the
// bracket by Newton steps on the closed-form derivative, bisecting whenever a step would leave the bracket (the plain
// bisection this replaces took ~50 exponentials after a ~65-exponential scan, on the host once per stream and hop of the
// llsmrt pulse scheduler and on the device once per pulse group; the root is the same to ~1e-15 relative)
LF_HD Solved solve(const Model& m) {
  Solved s = prepare(m);
  const double Ar = return_area(s);
  // on the grid only the SIGN of open_area(s, k / Te) + Ar matters: that of -Ee N + Ar D (D = alpha^2 + wg^2 > 0), with
  // e^{-alpha Te} = e^{-k} from the table -- no exponential and no division per grid point
  const double c0 = s.wg * s.cw / s.sw, c1 = s.wg / s.sw, inv_te = 1.0 / s.Te, w2 = s.wg * s.wg;
  auto grid = [&](int k) { const double a = k * inv_te; return -s.Ee * (a - c0 + c1 * exp_neg_int(k)) + Ar * (a * a + w2); };
  double lo = 0, hi = 0, flo = 0; bool found = false;
  double prev = grid(-60);
  for(int k = -59; k <= 60 && ! found; k ++) {
    const double f = grid(k);
    if((prev <= 0 && f > 0) || (prev >= 0 && f < 0)) { lo = (k - 1) / s.Te; hi = k / s.Te; flo = prev; found = true; }
    prev = f;
  }
  if(! found) return s;
  double x = 0.5 * (lo + hi);
  for(int it = 0; it < 100; it ++) {
    const double e = exp(-x * s.Te), N = x - c0 + c1 * e, D = x * x + w2;
    const double f = -s.Ee * N / D + Ar;
    if(f == 0) break;
    if((f <= 0) == (flo <= 0)) { lo = x; flo = f; } else hi = x;
    const double df = -s.Ee * ((1.0 - c1 * s.Te * e) * D - N * 2.0 * x) / (D * D);
    double xn = x - f / df;
    if(!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);              // (also when df = 0 or the step is not finite)
    const double step = fabs(xn - x);
    x = xn;
    if(step <= 1e-15 * fmax(fabs(lo), fabs(hi)) || hi - lo < 1e-15 * fmax(fabs(lo), fabs(hi))) break;
  }
  s.alpha = x;
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
#include <mutex>
#include <vector>
namespace llsm_lf {
// Phase of the Rd-parametrised model at ITS OWN fundamental, phase(solve(from_rd(rd, T0, 1)), 1 / T0).  Every product the
// closed form contains (w Te, alpha Te, eps (T0 - Te), wg Te) is invariant under the time scale, so this is a function
// of Rd alone: the llsmrt pulse tracker (llsmrt.c:316-333) asks for it once per stream and hop, and solving alpha and
// eps for it was 0.3 us each -- 21 of the 25 us a 64-stream pulse-by-pulse hop spent packing.  Tabulated once per
// process on three smooth pieces (from_rd switches formulas at Rd = 0.21 and 2.7, where the phase jumps) and read with
// 4-point Lagrange interpolation: 7e-14 rad from the direct evaluation over [0.01, 8] (tests/c_host/lf_solve_check.cpp);
// outside that range the direct evaluation.  (Host functions.)
inline double phase_at_f0_direct(double rd) { return phase(solve(from_rd(rd, 1.0, 1.0)), 1.0); }
inline double phase_at_f0(double rd, int clamp = 0) {
  if(clamp) rd = rd < 0.3 ? 0.3 : (rd > 2.7 ? 2.7 : rd);            // from_rd's "lf_rd_clamp" convention
  struct Seg { double lo, h; int n; std::vector<double> v; };
  static Seg seg[3]; static std::once_flag once;
  std::call_once(once, [] {
    const double lo[3] = {0.01, 0.21, 2.7}, hi[3] = {0.21, 2.7, 8.0}; const int n[3] = {1024, 8192, 2048};
    for(int k = 0; k < 3; k ++) {
      Seg& s = seg[k]; s.lo = lo[k]; s.n = n[k]; s.h = (hi[k] - lo[k]) / (n[k] - 1); s.v.resize(n[k]);
      for(int i = 0; i < n[k]; i ++) {
        double x = lo[k] + s.h * i;
        if(k == 0 && i == n[k] - 1) x = nextafter(hi[k], lo[k]);     // [0.01, 0.21): the last node just below the switch
        if(k == 2 && i == 0) x = nextafter(lo[k], hi[k]);            // (2.7, 8]: the first node just above it
        s.v[i] = phase_at_f0_direct(x);
      }
    }
  });
  if(!(rd >= 0.01 && rd <= 8.0)) return phase_at_f0_direct(rd);
  const Seg& s = rd < 0.21 ? seg[0] : (rd <= 2.7 ? seg[1] : seg[2]);
  const double t = (rd - s.lo) / s.h;
  int i = (int)t; if(i < 1) i = 1; if(i > s.n - 3) i = s.n - 3;
  const double u = t - i, um1 = u + 1.0, u1 = u - 1.0, u2 = u - 2.0;
  const double* p = & s.v[i - 1];
  return p[0] * (-u * u1 * u2 / 6.0) + p[1] * (um1 * u1 * u2 / 2.0) + p[2] * (-um1 * u * u2 / 2.0) + p[3] * (um1 * u * u1 / 6.0);
}

}  // namespace llsm_lf
#endif
