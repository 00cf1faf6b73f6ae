// synth_frame.h -- the stationary harmonic frame on the f32 MFMA (synth_frame), shared by the offline overlap-add
// kernels (synth_kernels.hip) and the llsmrt hop kernels (kernels.hip).  Include after plan.h / dev_common.h and after
// the translation unit's `#pragma clang fp contract(fast)`.
#pragma once
#include "dev_common.h"
#include "plan.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// (cos, sin) <- (cos, sin) rotated by (dc, ds): angles add
DEV void cs_rot(float& c, float& sn, float dc, float ds) {
  const float t1 = c * dc - sn * ds, t2 = c * ds + sn * dc; c = t1; sn = t2;
}

// XCD-aware work mapping (cdna guide T1): workgroup b is observed to run on XCD b % 8, each
// XCD has a private 4 MiB L2, and neighbouring frames read windows that overlap 4-7x.  Map
// workgroups so that each XCD walks its own contiguous range of frames (bijective for any n);
// a different placement only costs speed.
DEV int xcd_frame(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// frame lookup: global frame g -> (utterance u, local index i)
DEV void frame_owner(const int* __restrict__ frm_utt, const int* __restrict__ frm_off,
  int g, int* u, int* i) {
  *u = frm_utt[g];
  *i = g - frm_off[*u];
}

// =====================================================================
// K3  stationary harmonic frame * Hann window (HOT LOOPS B and D) on the f32 MFMA
// replaces llsm_synthesize_harmonics_l0's per-frame body, layer0.c:124-134,
// with llsm_synthesize_harmonic_frame{,_iczt,_auto} (dsputils.c:328-351,
// llsmutils.c:45-58; the bank and the ICZT compute the same signal, so one
// evaluation serves both):
//   y[t] = sum_h a_h cos(2 pi (h+1) f0/fs (t - nwin/2) + phi_h - corr*(h+1))
// Same two-level factorisation as K1, transposed.  Row a (of 16) covers the L
// samples tau = t - nwin/2 in [rho_a - L/2, rho_a + L/2), rho_a = L (a - 8) + L/2.
// With P[a][h] = A_h e^{j th_h rho_a} (A_h = a_h e^{j phi'_h}, th_h = 2 pi (h+1) f0/fs):
//   y(rho_a + b) = E[a][b] + O[a][b],   y(rho_a - b) = E[a][b] - O[a][b],
//   E[a][b] = sum_h Re P[a][h] cos(th_h b),  O[a][b] = - sum_h Im P[a][h] sin(th_h b),
// i.e. two 16 x K x (L/2 + 1) GEMMs on v_mfma_f32_16x16x4_f32 -- half the
// columns of the plain 16 x 2K x L product.  The A operands (rotated complex
// amplitudes) and the B operands (cos / sin tables) are generated in registers
// by phasor recurrences over the harmonic index (four harmonics per MFMA
// k-step, re-seeded from float64 phases every SYN_RESEED steps).
// The complex amplitudes A_h are staged in LDS.
// Output row g of frames[F][nwin] (k_synth_frames, llsmrt) or straight into the overlap-add ring (k_synth_ola).
// cyc_shift != NULL: llsmrt phase convention (llsmrt.c:279-282), the
// correction is cycle*2*pi*f0 instead of the fractional-hop term.
// =====================================================================
#define SYN_RESEED 32  // k-steps (= 128 harmonics) between float64 re-seeds

// NT column tiles of 16 offsets b per pass; L/2 + 1 = 16 * NT * npass columns in all (host-chosen
// so that 16 L >= nwin).  NT is a template parameter so that the MFMA loop is branch-free.
// One frame: complex amplitudes staged in LDS (A, Kp + 4 float2), then the two GEMMs; every window
// sample t of the frame is handed to sink(t, y[t] * win[t]) exactly once.
// A_h of synth_frame for harmonic k (0-based) of a frame whose llsmrt cycle remainder is `cyc`: a e^{j (phi - corr (k + 1))}
DEV float2 synth_amplitude(float a, float ph, int k, float cyc, float f) {
  const float corr = (float)((double)(cyc * 2.0f) * 3.14159265358979323846 * (double)f);
  const double phd = (double)ph - (double)corr * (k + 1.0);
  float sn, cs; cs_turns(phd * 0.15915494309189533577, & cs, & sn);
  return make_float2(a * cs, a * sn);
}
// Re(A V) and -Im(A V) of one k-step: ONE source expression for the register-recurrence path below and the table path of
// k_synth_ola4 (synth_kernels.hip), whose results must agree bit for bit
// (explicit fma / mul: under `fp contract(fast)` the compiler picks which product of a*b + c*d is fused from the
// surrounding code, and the two paths must not depend on that)
DEV float syn_pr(float2 a, float vr, float vi) { return __fmaf_rn(a.x, vr, -__fmul_rn(a.y, vi)); }
DEV float syn_npi(float2 a, float vr, float vi) { return -__fmaf_rn(a.x, vi, __fmul_rn(a.y, vr)); }
// cs_rot with the same pinned operations, for the phasor recurrences of the resynthesis
DEV void syn_rot(float& c, float& sn, float dc, float ds) {
  const float t1 = __fmaf_rn(c, dc, -__fmul_rn(sn, ds)), t2 = __fmaf_rn(c, ds, __fmul_rn(sn, dc)); c = t1; sn = t2;
}
// fractional-hop phase correction of frame i (layer0.c:127-131), radians per harmonic unit
DEV float syn_corr(int i, float thop, float fs, float f) {
  int baseidx; float frac = lp::rawfrac(i, thop, fs, & baseidx);
  return (float)((double)(frac * 2.0f) * 3.14159265358979323846 / (double)fs * (double)f);
}
// complex amplitudes a_k e^{j (phi_k - corr (k + 1))} of frame row g into A[0 .. Kp), zero beyond K
DEV void syn_stage(float2* A, int lane, int K, int Kp, float corr, const float* __restrict__ arow, const float* __restrict__ prow) {
  for(int k = lane; k < Kp; k += WAVE) {
    float2 v = make_float2(0.0f, 0.0f);
    if(k < K) {
      const double phd = (double)prow[k] - (double)corr * (k + 1.0);
      float sn, cs; cs_turns(phd * 0.15915494309189533577, & cs, & sn);   // radians -> turns
      float a = arow[k];
      v = make_float2(a * cs, a * sn);
    }
    A[k] = v;
  }
}
template <int NT, class Sink, bool WAVE_ONLY = false>
DEV void synth_frame(int g, int i, float f, const int* __restrict__ nhar,
  const float* __restrict__ ampl, const float* __restrict__ phse, int maxnhar,
  float thop, float fs, int nwin, int L, const float* __restrict__ win,
  const float* __restrict__ cyc_shift, float2* A, int lane, Sink sink, int K_ready = -1) {
  // K_ready >= 0: A already holds the (K_ready + 3) & ~3 amplitudes (synth_amplitude below); nhar / ampl / phse / cyc_shift unread
  int K = K_ready >= 0 ? K_ready : nhar[g]; if(K > 2048) K = 2048; if(K > maxnhar) K = maxnhar; if(K < 0) K = 0;
  float corr = 0;
  if(K_ready >= 0) { }
  else if(cyc_shift) {
    corr = (float)((double)(cyc_shift[g] * 2.0f) * 3.14159265358979323846 * (double)f);
  } else {
    corr = syn_corr(i, thop, fs, f);
  }
  const int Kp = (K + 3) & ~3;                       // harmonic slots, multiple of 4
  if(K_ready < 0) syn_stage(A, lane, K, Kp, corr, ampl + (size_t)g * maxnhar, phse + (size_t)g * maxnhar);
  if(WAVE_ONLY) {                                    // caller runs this on ONE wavefront of a larger workgroup (k_rt_front):
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS is in order per wavefront; keep the compiler from moving
    __builtin_amdgcn_wave_barrier();                 // the reads of A above the writes, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  } else __syncthreads();
  const double turn1 = (double)f / (double)fs;
  const int half = nwin / 2;
  const int row = lane & 15, q = lane >> 4;          // A operand: (row a, harmonic 4 ks + q); B: (harmonic, column)
  const int nks = Kp / 4;
  const int rho = L * (row - 8) + L / 2;             // this lane's row centre (A side)
  const double ta = turn1 * (double)rho;             // turns per harmonic unit on the A side
  float u4r, u4i;
  cs_turns(4.0 * ta, & u4r, & u4i);                  // A-side step of four harmonics
  const int ncol = L / 2 + 1;
  for(int cb = 0; cb < ncol; cb += 16 * NT) {
    f32x4 accE[NT], accO[NT];
    double tb[NT];
    float bx[NT], by[NT], s4r[NT], s4i[NT];
#pragma unroll
    for(int ct = 0; ct < NT; ct ++) {
      accE[ct] = (f32x4){0, 0, 0, 0}; accO[ct] = (f32x4){0, 0, 0, 0};
      tb[ct] = turn1 * (double)(cb + 16 * ct + row);          // B operand column = lane & 15
      cs_turns(4.0 * tb[ct], & s4r[ct], & s4i[ct]);
      bx[ct] = 1.0f; by[ct] = 0.0f;
    }
    float vr = 1.0f, vi = 0.0f;                      // A-side phasor e^{j 2 pi ta (h+1)}
    float2 a_nxt = nks > 0 ? A[q] : make_float2(0.0f, 0.0f);   // (the next step's amplitude is requested a step ahead)
    for(int ks = 0; ks < nks; ks ++) {
      const int h = 4 * ks + q;                      // 0-based harmonic of this lane
      if((ks & (SYN_RESEED - 1)) == 0) {
        cs_turns(ta * (double)(h + 1), & vr, & vi);
#pragma unroll
        for(int ct = 0; ct < NT; ct ++) cs_turns(tb[ct] * (double)(h + 1), & bx[ct], & by[ct]);
      }
      const float2 a = a_nxt;
      a_nxt = A[ks + 1 < nks ? h + 4 : h];
      const float pr = syn_pr(a, vr, vi);            // Re(A V)
      const float npi = syn_npi(a, vr, vi);          // -Im(A V)
#pragma unroll
      for(int ct = 0; ct < NT; ct ++) {
        accE[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr, bx[ct], accE[ct], 0, 0, 0);
        accO[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(npi, by[ct], accO[ct], 0, 0, 0);
      }
      // advance both phasors by four harmonics
      syn_rot(vr, vi, u4r, u4i);
#pragma unroll
      for(int ct = 0; ct < NT; ct ++) syn_rot(bx[ct], by[ct], s4r[ct], s4i[ct]);
    }
    // D[row a = 4 q + r][col = lane & 15] of tile ct: offset b = cb + 16 ct + col from the centre of row a
#pragma unroll
    for(int ct = 0; ct < NT; ct ++) {
#pragma unroll
      for(int r = 0; r < 4; r ++) {
        const int b = cb + 16 * ct + row;
        const int tc = L * (4 * q + r - 8) + L / 2 + half;     // window index of the row centre
        const float e = accE[ct][r], o = accO[ct][r];
        const int tp = tc + b, tm = tc - b;
        if(b < L / 2 && tp >= 0 && tp < nwin) sink(tp, (e + o) * win[tp]);
        if(b >= 1 && b <= L / 2 && tm >= 0 && tm < nwin) sink(tm, (e - o) * win[tm]);
      }
    }
  }
}
