// lfmodel.h's alpha solve (table-driven sign scan + safeguarded Newton, round 3) against a plain reference written here:
// the same leftmost bracket found with real exponentials, refined by bisection to machine precision.  Rd curve
// (0.02 .. 8), four fundamental periods, and random LF shapes such as the llsm_fgfm effects produce.
#include <cmath>
#include <cstdio>
#include <initializer_list>
#include <random>
#include "lfmodel.h"
using namespace llsm_lf;

static bool reference_alpha(const Model& m, double* alpha) {
  Solved s = prepare(m);
  const double Ar = return_area(s);
  double lo = 0, hi = 0, flo = 0; bool found = false;
  double prev = open_area(s, -60.0 / s.Te) + Ar;
  for(int k = -59; k <= 60 && ! found; k ++) {
    const double a = k / s.Te, f = open_area(s, a) + Ar;
    if((prev <= 0 && f > 0) || (prev >= 0 && f < 0)) { lo = (k - 1) / s.Te; hi = a; flo = prev; found = true; }
    prev = f;
  }
  if(! found) return false;
  for(int it = 0; it < 200; it ++) {
    const double mid = 0.5 * (lo + hi), f = open_area(s, mid) + Ar;
    if((f <= 0) == (flo <= 0)) { lo = mid; flo = f; } else hi = mid;
    if(hi - lo < 1e-15 * fmax(fabs(lo), fabs(hi))) break;
  }
  *alpha = 0.5 * (lo + hi);
  return true;
}

int main() {
  double worst_curve = 0, worst_rand = 0; int n = 0, bad = 0;
  auto check = [&](const Model& m, double* worst) {
    double ref = 0; const bool ok = reference_alpha(m, & ref);
    const Solved s = solve(m);
    if(! ok) { if(s.alpha != 0) bad ++; return; }
    if(! std::isfinite(s.alpha)) { bad ++; return; }
    const double scale = fmax(fabs(ref), 1e-3 / (m.te * m.T0));
    const double d = fabs(s.alpha - ref) / scale;
    if(d > *worst) *worst = d;
    n ++;
  };
  for(double rd = 0.02; rd < 8.0; rd *= 1.002)
    for(double f0 : {50.0, 120.0, 400.0, 1000.0}) check(from_rd(rd, 1.0 / f0, 1.0), & worst_curve);
  std::mt19937_64 g(7); std::uniform_real_distribution<double> U(0, 1);
  for(int i = 0; i < 200000; i ++) {
    Model m; m.T0 = 1.0 / (50 + 950 * U(g)); m.te = 0.2 + 0.79 * U(g); m.tp = m.te * (0.45 + 0.5 * U(g));
    m.ta = pow(10.0, -6 + 5.5 * U(g)); if(m.ta > 0.9 * (1 - m.te)) m.ta = 0.9 * (1 - m.te); m.Ee = 1.0;
    check(m, & worst_rand);
  }
  std::printf("lf solve: %d models, worst relative difference to the bisection reference: Rd curve %.3g, random shapes %.3g, %d bad\n",
    n, worst_curve, worst_rand, bad);
  // phase of the Rd model at its own fundamental, tabulated (phase_at_f0, round 4: the llsmrt pulse tracker reads it once per
  // stream and hop) against the direct evaluation -- at the model's own period AND at other periods (time-scale invariance)
  double worst_tab = 0, worst_scale = 0;
  for(int i = 0; i < 200000; i ++) {
    const double rd = 0.005 + 8.5 * U(g), f0 = 50 + 950 * U(g);
    const double direct = phase(solve(from_rd(rd, 1.0 / f0, 1.0)), f0);
    worst_tab = fmax(worst_tab, fabs(phase_at_f0(rd) - direct));
    worst_scale = fmax(worst_scale, fabs(phase_at_f0_direct(rd) - direct));
  }
  for(double rd : {0.01, 0.2099999, 0.21, 0.2100001, 2.6999999, 2.7, 2.7000001, 8.0})   // both sides of the formula switches
    worst_tab = fmax(worst_tab, fabs(phase_at_f0(rd) - phase_at_f0_direct(rd)));
  std::printf("lf phase at F0: tabulated vs direct %.3g rad, direct at T0 = 1 vs at the frame's period %.3g rad\n", worst_tab, worst_scale);
  if(!(worst_tab < 1e-12 && worst_scale < 1e-12)) bad ++;
  std::printf("%d bad\n", bad);
  // the random shapes include ill-conditioned ones (return phase of 1e-6 of the period: the net flow is flat near its zero)
  return (bad == 0 && worst_curve < 1e-13 && worst_rand < 1e-11) ? 0 : 1;
}
