// cheby.h -- regenerates the reference's band-filter table instead of copying
// it: filter-coef.h holds 48 low-pass + 48 high-pass 4th-order Chebyshev-I
// sections which are exactly cheby1(N = 4, ripple = 0.5 dB, Wn = (i+1)*0.02)
// (SURVEY.md section 2).  Row selection follows dsputils.c:28-49; the
// steady-state initial conditions implement the zero-phase forward-backward
// filter contract documented in DESIGN.md ("filtfilt").
#ifndef LLSM_AMD_CHEBY_H
#define LLSM_AMD_CHEBY_H

// Samples per lane of the wave-parallel block IIR (kernels.hip K5): a tile is 64 lanes x
// IIR_SEG samples (a multiple of 4: 16-byte accesses).  Measured at 3 wavefronts / SIMD: 20 -> 1.28 ms
// per step, 24 -> 1.32 (a 20 128-sample template wastes 7 % of its last 1536-sample tile), 28 -> 1.29,
// 16 -> 1.33, 12 -> 1.46 (the per-tile state scan weighs more); 32 needs 2 wavefronts / SIMD: 1.31.  Measured again after the
// band-pass channels went to two sections per pass (the kernel no longer waits for HBM, tools/kbench.py): 20 -> 1.01 - 1.02,
// 24 -> 1.02, 28 -> 0.96 - 0.98, 32 -> 1.07 at 3 wavefronts / SIMD; 20 -> 1.01 and 24 -> 1.31 (spills) at 4; 28 -> 1.07,
// 32 -> 1.04, 36 -> 1.03 at 2.
#ifndef IIR_SEG
#define IIR_SEG 28
#endif
#include <cmath>
#include <complex>
#include <vector>

namespace llsm_cheby {

const int kOrder = 4;
const int kTaps = 5;
const int kRows = 48;
const double kRipple = 0.5;
const double kStep = 0.02;

struct Section {
  double b[kTaps];
  double a[kTaps];
  double zi[kOrder];   // DF2T steady-state for a unit step input
};

// 4x4 helpers for the block-parallel recursion tables (row-major).
inline void mat_mul(const double* p, const double* q, double* r) {
  double t[16];
  for(int i = 0; i < 4; i ++)
    for(int j = 0; j < 4; j ++) {
      long double acc = 0;
      for(int k = 0; k < 4; k ++) acc += (long double)p[4 * i + k] * q[4 * k + j];
      t[4 * i + j] = (double)acc;
    }
  for(int i = 0; i < 16; i ++) r[i] = t[i];
}
// DF2T state transition z' = A z + B x: A = [[-a1,1,0,0],[-a2,0,1,0],[-a3,0,0,1],[-a4,0,0,0]]
inline void transition(const double* a, double* A) {
  for(int i = 0; i < 16; i ++) A[i] = 0;
  for(int i = 0; i < 4; i ++) { A[4 * i] = -a[i + 1]; if(i < 3) A[4 * i + i + 1] = 1; }
}
// H[i] = first row of A^i, i < seg;  M[d] = (A^seg)^(2^d), d < nd
inline void block_tables(const double* a, int seg, int nd, double* H, double* M) {
  double A[16], P[16];
  transition(a, A);
  for(int i = 0; i < 16; i ++) P[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for(int i = 0; i < seg; i ++) {
    for(int j = 0; j < 4; j ++) H[4 * i + j] = P[j];
    mat_mul(P, A, P);
  }
  for(int i = 0; i < 16; i ++) M[i] = P[i];             // A^seg
  for(int d = 1; d < nd; d ++) mat_mul(M + 16 * (d - 1), M + 16 * (d - 1), M + 16 * d);
}

// Bilinear-transformed Chebyshev-I prototype, digital cutoff `wn` (1 = Nyquist).
inline void design(double wn, bool highpass, double* b, double* a) {
  typedef std::complex<double> cd;
  const double pi = 3.14159265358979323846;
  double eps = std::sqrt(std::pow(10.0, 0.1 * kRipple) - 1.0);
  double mu = std::asinh(1.0 / eps) / kOrder;
  std::vector<cd> pole(kOrder);
  cd gain_num(1, 0);
  for(int i = 0; i < kOrder; i ++) {
    double th = pi * (2 * i + 1 - kOrder) / (2.0 * kOrder);
    pole[i] = -std::sinh(cd(mu, th));
    gain_num *= -pole[i];
  }
  double k = gain_num.real() / std::sqrt(1.0 + eps * eps);   // even order
  const double fs2 = 4.0;
  double warped = fs2 * std::tan(pi * wn / 2.0);
  double zero_at;
  if(! highpass) {
    for(auto& p : pole) p *= warped;
    k *= std::pow(warped, kOrder);
    zero_at = -1.0;
    cd den(1, 0);
    for(auto& p : pole) den *= (fs2 - p);
    k *= (cd(1, 0) / den).real();
  } else {
    cd pn(1, 0);
    for(auto& p : pole) pn *= -p;
    k *= (cd(1, 0) / pn).real();
    for(auto& p : pole) p = warped / p;
    zero_at = 1.0;
    cd den(1, 0);
    for(auto& p : pole) den *= (fs2 - p);
    k *= (cd(std::pow(fs2, kOrder), 0) / den).real();
  }
  for(auto& p : pole) p = (fs2 + p) / (fs2 - p);
  // expand (z - zero)^N and prod (z - pole_i)
  std::vector<cd> pa(1, cd(1, 0)), pb(1, cd(1, 0));
  for(int i = 0; i < kOrder; i ++) {
    pa.push_back(0); pb.push_back(0);
    for(int j = i + 1; j >= 1; j --) {
      pa[j] -= pa[j - 1] * pole[i];
      pb[j] -= pb[j - 1] * zero_at;
    }
  }
  for(int i = 0; i < kTaps; i ++) { a[i] = pa[i].real(); b[i] = pb[i].real() * k; }
}

// dsputils.c:31-32: row = max(0, round(cutoff*2/0.02 - 1)), clamped to 47
inline int row_of(float cutoff) {
  int r = (int)std::round((double)cutoff * 2.0 / (double)0.02f - 1);
  if(r < 0) r = 0;
  if(r >= kRows) r = kRows - 1;
  return r;
}

inline Section make_section_row(int row, bool highpass) {
  double b[kTaps], a[kTaps];
  design((row + 1) * kStep, highpass, b, a);
  Section s;
  for(int i = 0; i < kTaps; i ++) { s.b[i] = b[i]; s.a[i] = a[i]; }
  // steady state of the transposed direct form II for x == 1
  double asum = 0, csum = 0;
  for(int i = 0; i < kTaps; i ++) asum += a[i];
  for(int i = 1; i < kTaps; i ++) csum += b[i] - a[i] * b[0];
  double zi[kOrder];
  zi[0] = csum / asum;
  double acc = 1.0, cs = 0;
  for(int i = 1; i < kOrder; i ++) {
    acc += a[i];
    cs += b[i] - a[i] * b[0];
    zi[i] = acc * zi[0] - cs;
  }
  for(int i = 0; i < kOrder; i ++) s.zi[i] = zi[i];
  return s;
}

inline Section make_section(float cutoff, bool highpass) {
  return make_section_row(row_of(cutoff), highpass);
}

}  // namespace llsm_cheby
#endif
