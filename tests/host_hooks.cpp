// host_hooks.cpp -- TEST-ONLY host emulation of the wave-parallel block IIR
// (kernels.hip K5) using the product's own coefficient / table generator
// (cheby.h).  Lets the CPU test-suite validate the block decomposition
// (zero-state segments + Kogge-Stone state scan + zero-input correction)
// against scipy.signal.filtfilt without a GPU.  Not linked into the product.
#include <cmath>
#include <cstring>
#include <vector>
#include "cheby.h"

#define SEG IIR_SEG                      // the product's segment length
#define LANES 64

static void pass(const llsm_cheby::Section& s, const double* H, const double* M,
  const std::vector<double>& in, double init_scale, std::vector<double>& out) {
  const int ne = (int)in.size();
  out.assign(ne, 0.0);
  double c[4];
  for(int i = 0; i < 4; i ++) c[i] = s.zi[i] * init_scale;
  const int tile = SEG * LANES;
  for(int base = 0; base < ne; base += tile) {
    double v[LANES][SEG], z[LANES][4];
    for(int m = 0; m < LANES; m ++) {
      double z0 = 0, z1 = 0, z2 = 0, z3 = 0;
      for(int i = 0; i < SEG; i ++) {
        int t = base + m * SEG + i;
        double xi = t < ne ? in[t] : 0.0;
        double yi = s.b[0] * xi + z0;
        z0 = s.b[1] * xi + z1 - s.a[1] * yi;
        z1 = s.b[2] * xi + z2 - s.a[2] * yi;
        z2 = s.b[3] * xi + z3 - s.a[3] * yi;
        z3 = s.b[4] * xi - s.a[4] * yi;
        v[m][i] = yi;
      }
      z[m][0] = z0; z[m][1] = z1; z[m][2] = z2; z[m][3] = z3;
    }
    for(int r = 0; r < 4; r ++)
      for(int k = 0; k < 4; k ++) z[0][r] += M[4 * r + k] * c[k];
    for(int d = 0; d < 6; d ++) {
      int off = 1 << d;
      double u[LANES][4];
      std::memcpy(u, z, sizeof(u));
      for(int m = off; m < LANES; m ++)
        for(int r = 0; r < 4; r ++)
          for(int k = 0; k < 4; k ++) z[m][r] += M[16 * d + 4 * r + k] * u[m - off][k];
    }
    for(int m = 0; m < LANES; m ++) {
      const double* si = m == 0 ? c : z[m - 1];
      for(int i = 0; i < SEG; i ++) {
        int t = base + m * SEG + i;
        if(t < ne) out[t] = v[m][i] + H[4 * i] * si[0] + H[4 * i + 1] * si[1] + H[4 * i + 2] * si[2] + H[4 * i + 3] * si[3];
      }
    }
    double nc[4] = {z[LANES - 1][0], z[LANES - 1][1], z[LANES - 1][2], z[LANES - 1][3]};
    std::memcpy(c, nc, sizeof(c));
  }
}

extern "C" void hook_block_filtfilt(int row, int highpass, const double* x, int n, double* y) {
  llsm_cheby::Section s = llsm_cheby::make_section_row(row, highpass != 0);
  double H[SEG * 4], M[6 * 16];
  llsm_cheby::block_tables(s.a, SEG, 6, H, M);
  int pad = n - 1 < 15 ? n - 1 : 15, ne = n + 2 * pad;
  std::vector<double> ext(ne), f, r(ne), b;
  for(int t = 0; t < ne; t ++) {
    if(t < pad) ext[t] = 2.0 * x[0] - x[pad - t];
    else if(t >= pad + n) ext[t] = 2.0 * x[n - 1] - x[n - 2 - (t - pad - n)];
    else ext[t] = x[t - pad];
  }
  pass(s, H, M, ext, ext[0], f);
  for(int t = 0; t < ne; t ++) r[t] = f[ne - 1 - t];
  pass(s, H, M, r, r[0], b);
  for(int t = 0; t < n; t ++) y[t] = b[ne - 1 - (t + pad)];
}

extern "C" void hook_section(int row, int highpass, double* b, double* a, double* zi) {
  llsm_cheby::Section s = llsm_cheby::make_section_row(row, highpass != 0);
  for(int i = 0; i < 5; i ++) { b[i] = s.b[i]; a[i] = s.a[i]; }
  for(int i = 0; i < 4; i ++) zi[i] = s.zi[i];
}
extern "C" int hook_row_of(float cutoff) { return llsm_cheby::row_of(cutoff); }

// ---- lfmodel.h (the product's LF glottal model, host build) for tests/test_oracle_l1.py ----
#include "lfmodel.h"
extern "C" void hook_lf_spectrum(double rd, double T0, const double* freq, int n, double* magn, double* phase,
  double* params /* te tp ta eps alpha */) {
  llsm_lf::Model m = llsm_lf::from_rd(rd, T0, 1.0);
  llsm_lf::Solved s = llsm_lf::solve(m);
  for(int i = 0; i < n; i ++) { magn[i] = llsm_lf::magnitude(s, freq[i]); phase[i] = llsm_lf::phase(s, freq[i]); }
  params[0] = m.te; params[1] = m.tp; params[2] = m.ta; params[3] = s.eps; params[4] = s.alpha;
}
