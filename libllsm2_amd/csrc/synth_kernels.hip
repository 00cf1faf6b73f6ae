// synth_kernels.hip -- harmonic resynthesis of the offline path: stationary harmonic frames on the f32 MFMA
// (synth_frame.h) with the overlap-add fused in LDS (k_synth_ola*), and the frames-to-HBM form llsmrt and the
// layer-1 path use (k_synth_frames).  Replaces llsm_synthesize_harmonics_l0 (layer0.c:117-146) and the harmonic
// half of the analysis residual (layer0.c:500-501).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <climits>

#include "kernels.h"
#include "plan.h"

namespace lp = llsm_plan;
#pragma clang fp contract(fast)                      // after plan.h: see kernels.hip
#include "dev_common.h"
#include "synth_frame.h"
#include "launch.h"

extern __shared__ __attribute__((aligned(16))) unsigned char g_lds[];

// NT column tiles of 16 offsets b per pass; L/2 + 1 = 16 * NT * npass columns in all (host-chosen
// so that 16 L >= nwin).  NT is a template parameter so that the MFMA loop is branch-free.
// Frames to HBM, one wavefront per frame (llsmrt; the offline path uses k_synth_ola).
template <int NT>
__global__ __launch_bounds__(WAVE) void k_synth_frames(
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off,
  const float* __restrict__ f0, const int* __restrict__ nhar,
  const float* __restrict__ ampl, const float* __restrict__ phse, int maxnhar,
  float thop, float fs, int nwin, int L, const float* __restrict__ win,
  const float* __restrict__ cyc_shift, float* __restrict__ frames) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const float f = f0[g];
  if(!(f > 0)) return;
  int u, i; frame_owner(frm_utt, frm_off, g, & u, & i);
  float* out = frames + (size_t)g * nwin;
  synth_frame<NT>(g, i, f, nhar, ampl, phse, maxnhar, thop, fs, nwin, L, win, cyc_shift,
    (float2*)g_lds, lane, [&](int t, float v) { out[t] = v; });
}

// K3 + K4 fused (offline path): harmonic frames are overlap-added in LDS and never reach HBM.
// Same unit scheme as k_noise_filter_ola: a wavefront owns frames [i0, i1) of one utterance and
// the samples [lo(i0), lo(i1)), lo(i) = start of frame i's window (0 / length at the utterance
// ends); it walks the frames from `halo` before i0, adds each voiced frame into a ring of R >= nwin
// samples and writes a sample once the next frame starts beyond it: ascending frame order per
// sample, as layer0.c:135-140.  mode 0: out = x - sum (the analysis residual, layer0.c:500-501);
// mode 1: out = sum (y_sin) and, when mix != NULL, mix = sum + x with x = y_noise (the final mix).
template <int NT>
__global__ __launch_bounds__(WAVE) void k_synth_ola(
  const int4* __restrict__ units, int halo, int R,
  const int* __restrict__ frm_off, const int* __restrict__ nfrm,
  const int* __restrict__ out_off, const int* __restrict__ out_len,
  const float* __restrict__ f0, const int* __restrict__ nhar,
  const float* __restrict__ ampl, const float* __restrict__ phse, int maxnhar,
  float thop, float fs, int nwin, int L, const float* __restrict__ win, int lds_harmonics,
  const float* __restrict__ x, float* __restrict__ out, int mode, float* __restrict__ mix) {
  const int lane = threadIdx.x;
  float2* A = (float2*)g_lds;
  float* ring = (float*)(A + lds_harmonics + 4);     // sample s at ring[s & (R - 1)]
  for(int t = lane; t < R; t += WAVE) ring[t] = 0.0f;
  const int4 unit = units[xcd_frame(blockIdx.x, gridDim.x)];
  if(unit.w != 0) return;                            // padding unit (groups of four per utterance, k_synth_ola4)
  const int u = unit.x, i0 = unit.y, i1 = unit.z;
  const int nf = nfrm[u], fo = frm_off[u], len = out_len[u];
  const size_t oo = (size_t)out_off[u];
  const int own_lo = i0 == 0 ? 0 : min(max(lp::center(i0, thop, fs) - nwin / 2, 0), len);
  const int own_hi = i1 >= nf ? len : min(max(lp::center(i1, thop, fs) - nwin / 2, 0), len);
  const int j0 = max(0, i0 - halo);
  int flushed = lp::center(j0, thop, fs) - nwin / 2; // the ring holds samples [flushed, flushed + R)
  const float* xb = (x && len > 0) ? x + oo : nullptr;
  // samples [flushed, target) are complete: write the owned ones, clear their ring slots.
  // Four rows of 64 samples per round, the loads of a round issued together.
  // x (mode 0: the signal, mode 1: y_noise) of the NEXT flush is requested a frame ahead: xq[k] = x[pf_base + lane + 64 k]
  // (clamped), so that the flush of frame j + 1 does not wait a memory round trip for it
  float xq[4] = {0.0f, 0.0f, 0.0f, 0.0f}; int pf_base = INT_MIN;
  auto prefetch_x = [&](int from) {
    pf_base = from;
    if(! xb) return;
#pragma unroll
    for(int k = 0; k < 4; k ++) xq[k] = xb[min(max(from + lane + WAVE * k, 0), len - 1)];
  };
  auto advance = [&](int target) {
    for(; flushed < target; flushed = min(flushed + 4 * WAVE, target)) {
      float rv[4], xv[4]; bool own[4];
      const bool pre = flushed == pf_base;
#pragma unroll
      for(int k = 0; k < 4; k ++) {
        const int s = flushed + lane + WAVE * k;
        const bool ok = s < target;
        own[k] = ok && s >= own_lo && s < own_hi;
        rv[k] = ring[s & (R - 1)];
        xv[k] = pre ? xq[k] : (xb ? xb[own[k] ? s : 0] : 0.0f);
        if(ok) ring[s & (R - 1)] = 0.0f; else rv[k] = 0.0f;
      }
#pragma unroll
      for(int k = 0; k < 4; k ++) {
        const int s = flushed + lane + WAVE * k;
        if(! own[k]) continue;
        if(mode == 0) out[oo + s] = xv[k] - rv[k];           // residual
        else {
          out[oo + s] = rv[k];
          if(mix) mix[oo + s] = rv[k] + xv[k];               // y = y_sin + y_noise (layer0.c:657-659)
        }
      }
    }
  };
  for(int j = j0; j < i1; j ++) {
    const float f = f0[fo + j];
    if(!(f > 0)) continue;
    const int st = lp::center(j, thop, fs) - nwin / 2;
    advance(st);
    __syncthreads();
    synth_frame<NT>(fo + j, j, f, nhar, ampl, phse, maxnhar, thop, fs, nwin, L, win, nullptr, A, lane,
      [&](int t, float v) { ring[(st + t) & (R - 1)] += v; });
    __syncthreads();
  }
  advance(own_hi);
}

// =====================================================================
// k_synth_ola4 (round 4): the same fused harmonic frames + overlap-add for the common geometry (one column tile:
// nwin <= 496, at most 128 harmonics), four wavefronts per workgroup on four consecutive units of ONE utterance.
// What a frame's GEMM needs besides its amplitudes -- the row phasors V[row][h] = e^{j th_h rho_row} and the column
// phasors B[col][h] = e^{j th_h col} -- are functions of (F0, lane, k-step) only.  On configs 2 / 3 and on every flat
// stretch of an F0 track the frames a workgroup walks share ONE F0 (bit-identical), so wavefront 0 runs the phasor
// recurrences of synth_frame ONCE, writes (V, B) of every k-step to an LDS table (nks x 64 lanes x float4), and every
// frame of the group whose F0 has the table's bits reads a k-step's operands with one ds_read_b128 instead of
// rotating two phasors (8 of the 12 VALU instructions of a k-step) and re-seeding them from four float64 sincos per
// frame.  The table holds exactly the values the recurrence produces (same seeds, same rotations), and the products
// use the same source expressions (syn_pr / syn_npi), so a frame's samples do not depend on which path computed it
// (tests/test_gpu_synth_tables.py: bit-identical with llsm_gpu_synth_tables(0)); frames of another F0 and frames with
// more k-steps than the table take the recurrence path in place.
// Also hoisted out of the frame: the Hann window values and ring offsets of a lane's 8 output samples.
// =====================================================================
#ifndef SO4_WAVES
#define SO4_WAVES 4                                // wavefronts (units) per workgroup
#endif
#ifndef SO4_WPE
#define SO4_WPE 4                                  // wavefronts per SIMD the register budget is cut for
#endif
#define SYN_TAB_MAXKS 32                             // k-steps (x 4 harmonics) the phasor table can hold
struct So4Args {
  const int4* units; int halo, R;
  const int* frm_off; const int* nfrm; const int* out_off; const int* out_len;
  const float* f0; const int* nhar; const float* ampl; const float* phse; int maxnhar;
  float thop, fs; int nwin, L; const float* win; int lds_harmonics, nks_tab, use_tab;
  const float* x; float* out; int mode; float* mix;
};

__global__ __launch_bounds__(SO4_WAVES * WAVE, SO4_WPE) void k_synth_ola4(const So4Args P) {
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
  const int R = P.R, nwin = P.nwin, L = P.L, maxnhar = P.maxnhar;
  const float thop = P.thop, fs = P.fs;
  float4* tab = (float4*)g_lds;                       // [nks_tab][64]: (vr, vi, bx, by) of k-step ks for this lane
  float2* A = (float2*)(tab + (size_t)P.nks_tab * WAVE) + (size_t)wv * (P.lds_harmonics + 4);
  float* ring = (float*)((float2*)(tab + (size_t)P.nks_tab * WAVE) + (size_t)SO4_WAVES * (P.lds_harmonics + 4)) + (size_t)wv * R;
  for(int t = lane; t < R; t += WAVE) ring[t] = 0.0f;
  const int grp = xcd_frame(blockIdx.x, gridDim.x);
  const int4 unit0 = P.units[SO4_WAVES * grp];
  const int4 unit = P.units[SO4_WAVES * grp + wv];
  const int u = unit.x, i0 = unit.y, i1 = unit.z;
  const int nf = P.nfrm[u], fo = P.frm_off[u], len = P.out_len[u];
  const int row = lane & 15, q = lane >> 4;
  // ---- the group's table F0: the first voiced frame among the first 64 frames the group walks (every wavefront
  //      evaluates the same loads, so no exchange is needed); 0: no table for this group
  float f_tab = 0.0f;
  if(P.use_tab) {
    const int jg = max(0, unit0.y - P.halo);
    const int jj = jg + lane;
    const float fl = jj < nf ? P.f0[P.frm_off[unit0.x] + jj] : 0.0f;
    const unsigned long long vm = __ballot(fl > 0);
    if(vm) f_tab = __shfl(fl, __ffsll((long long)vm) - 1, WAVE);
  }
  if(f_tab > 0 && wv == 0) {
    // the recurrences of synth_frame (NT = 1, cb = 0), stored instead of consumed
    const double turn1 = (double)f_tab / (double)fs;
    const int rho = L * (row - 8) + L / 2;
    const double ta = turn1 * (double)rho, tb = turn1 * (double)row;
    float u4r, u4i, s4r, s4i;
    cs_turns(4.0 * ta, & u4r, & u4i);
    cs_turns(4.0 * tb, & s4r, & s4i);
    float vr = 1.0f, vi = 0.0f, bx = 1.0f, by = 0.0f;
    for(int ks = 0; ks < P.nks_tab; ks ++) {
      const int h = 4 * ks + q;
      if((ks & (SYN_RESEED - 1)) == 0) {
        cs_turns(ta * (double)(h + 1), & vr, & vi);
        cs_turns(tb * (double)(h + 1), & bx, & by);
      }
      tab[ks * WAVE + lane] = make_float4(vr, vi, bx, by);
      syn_rot(vr, vi, u4r, u4i);
      syn_rot(bx, by, s4r, s4i);
    }
  }
  __syncthreads();                                   // the only workgroup barrier: from here on the wavefronts run apart
  if(unit.w != 0) return;                            // padding unit of a group
  const size_t oo = (size_t)P.out_off[u];
  const int own_lo = i0 == 0 ? 0 : min(max(lp::center(i0, thop, fs) - nwin / 2, 0), len);
  const int own_hi = i1 >= nf ? len : min(max(lp::center(i1, thop, fs) - nwin / 2, 0), len);
  const int j0 = max(0, i0 - P.halo);
  int flushed = lp::center(j0, thop, fs) - nwin / 2; // the ring holds samples [flushed, flushed + R)
  const float* xb = (P.x && len > 0) ? P.x + oo : nullptr;
  float* out = P.out; float* mix = P.mix; const int mode = P.mode;
  // x (mode 0: the signal, mode 1: y_noise) of the NEXT flush is requested a frame ahead: xq[k] = x[pf_base + lane + 64 k]
  // (clamped), so that the flush of frame j + 1 does not wait a memory round trip for it
  float xq[4] = {0.0f, 0.0f, 0.0f, 0.0f}; int pf_base = INT_MIN;
  auto prefetch_x = [&](int from) {
    pf_base = from;
    if(! xb) return;
#pragma unroll
    for(int k = 0; k < 4; k ++) xq[k] = xb[min(max(from + lane + WAVE * k, 0), len - 1)];
  };
  auto advance = [&](int target) {
    for(; flushed < target; flushed = min(flushed + 4 * WAVE, target)) {
      float rv[4], xv[4]; bool own[4];
      const bool pre = flushed == pf_base;
#pragma unroll
      for(int k = 0; k < 4; k ++) {
        const int s = flushed + lane + WAVE * k;
        const bool ok = s < target;
        own[k] = ok && s >= own_lo && s < own_hi;
        rv[k] = ring[s & (R - 1)];
        xv[k] = pre ? xq[k] : (xb ? xb[own[k] ? s : 0] : 0.0f);
        if(ok) ring[s & (R - 1)] = 0.0f; else rv[k] = 0.0f;
      }
#pragma unroll
      for(int k = 0; k < 4; k ++) {
        const int s = flushed + lane + WAVE * k;
        if(! own[k]) continue;
        if(mode == 0) out[oo + s] = xv[k] - rv[k];           // residual
        else {
          out[oo + s] = rv[k];
          if(mix) mix[oo + s] = rv[k] + xv[k];               // y = y_sin + y_noise (layer0.c:657-659)
        }
      }
    }
  };
  // ---- per-lane output geometry (frame-invariant): D[row a = 4 q + r][col = lane & 15] is offset b = col from the
  //      centre of row a; window index tc +- b
  const int half = nwin / 2;
  int tP[4], tM[4]; float wP[4], wM[4];
#pragma unroll
  for(int r = 0; r < 4; r ++) {
    const int tc = L * (4 * q + r - 8) + L / 2 + half;
    const int tp = tc + row, tm = tc - row;
    const bool okp = row < L / 2 && tp >= 0 && tp < nwin;
    const bool okm = row >= 1 && row <= L / 2 && tm >= 0 && tm < nwin;
    tP[r] = okp ? tp : -1; tM[r] = okm ? tm : -1;
    wP[r] = okp ? P.win[tp] : 0.0f; wM[r] = okm ? P.win[tm] : 0.0f;
  }
  const unsigned f_tab_bits = __float_as_uint(f_tab);
  // The parameter row of frame j + 1 (F0, harmonic count, this lane's two amplitudes and phases) is requested while
  // frame j is computed: f0 -> nhar -> ampl / phse were three dependent memory round trips per frame.
  struct Row { float f; int K; float a0, a1, p0, p1; };
  auto load_row = [&](int j) {
    Row r; r.f = 0.0f; r.K = 0; r.a0 = r.a1 = r.p0 = r.p1 = 0.0f;
    if(j < i1) {
      const int g = fo + j;
      r.f = P.f0[g]; r.K = P.nhar[g];
      const float* ar = P.ampl + (size_t)g * maxnhar; const float* pr = P.phse + (size_t)g * maxnhar;
      if(lane < maxnhar) { r.a0 = ar[lane]; r.p0 = pr[lane]; }
      if(lane + WAVE < maxnhar) { r.a1 = ar[lane + WAVE]; r.p1 = pr[lane + WAVE]; }
    }
    return r;
  };
  Row cur = load_row(j0);
  for(int j = j0; j < i1; j ++) {
    const Row rw = cur;
    cur = load_row(j + 1);
    const float f = rw.f;
    if(!(f > 0)) continue;
    const int st = lp::center(j, thop, fs) - nwin / 2;
    advance(st);
    prefetch_x(st);                                  // (the next flush starts where this one ended)
    int K = rw.K; if(K > 2048) K = 2048; if(K > maxnhar) K = maxnhar; if(K < 0) K = 0;
    const int Kp = (K + 3) & ~3, nks = Kp / 4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the previous frame's reads of A are done: one wavefront, in order)
    __builtin_amdgcn_wave_barrier();
    {
      // syn_stage from the prefetched row: a_k e^{j (phi_k - corr (k + 1))}, zero beyond K
      const float corr = syn_corr(j, thop, fs, f);
#pragma unroll
      for(int m = 0; m < 2; m ++) {
        const int k = lane + WAVE * m;
        if(k < Kp) {
          float2 v = make_float2(0.0f, 0.0f);
          if(k < K) {
            const double phd = (double)(m ? rw.p1 : rw.p0) - (double)corr * (k + 1.0);
            float sn, cs; cs_turns(phd * 0.15915494309189533577, & cs, & sn);
            const float a = m ? rw.a1 : rw.a0;
            v = make_float2(a * cs, a * sn);
          }
          A[k] = v;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    f32x4 accE = (f32x4){0, 0, 0, 0}, accO = (f32x4){0, 0, 0, 0};
    if(__float_as_uint(f) == f_tab_bits && nks <= P.nks_tab) {
      const float4* tl = tab + lane;
      const float2* Aq = A + q;
      int ks = 0;
      for(; ks + 4 <= nks; ks += 4) {
        float2 a[4]; float4 t[4];
#pragma unroll
        for(int m = 0; m < 4; m ++) { a[m] = Aq[4 * (ks + m)]; t[m] = tl[(ks + m) * WAVE]; }
#pragma unroll
        for(int m = 0; m < 4; m ++) {
          accE = __builtin_amdgcn_mfma_f32_16x16x4f32(syn_pr(a[m], t[m].x, t[m].y), t[m].z, accE, 0, 0, 0);
          accO = __builtin_amdgcn_mfma_f32_16x16x4f32(syn_npi(a[m], t[m].x, t[m].y), t[m].w, accO, 0, 0, 0);
        }
      }
      for(; ks < nks; ks ++) {
        const float2 a = Aq[4 * ks]; const float4 t = tl[ks * WAVE];
        accE = __builtin_amdgcn_mfma_f32_16x16x4f32(syn_pr(a, t.x, t.y), t.z, accE, 0, 0, 0);
        accO = __builtin_amdgcn_mfma_f32_16x16x4f32(syn_npi(a, t.x, t.y), t.w, accO, 0, 0, 0);
      }
    } else {
      const double turn1 = (double)f / (double)fs;
      const int rho = L * (row - 8) + L / 2;
      const double ta = turn1 * (double)rho, tb = turn1 * (double)row;
      float u4r, u4i, s4r, s4i;
      cs_turns(4.0 * ta, & u4r, & u4i);
      cs_turns(4.0 * tb, & s4r, & s4i);
      float vr = 1.0f, vi = 0.0f, bx = 1.0f, by = 0.0f;
      for(int ks = 0; ks < nks; ks ++) {
        const int h = 4 * ks + q;
        if((ks & (SYN_RESEED - 1)) == 0) {
          cs_turns(ta * (double)(h + 1), & vr, & vi);
          cs_turns(tb * (double)(h + 1), & bx, & by);
        }
        const float2 a = A[h];
        accE = __builtin_amdgcn_mfma_f32_16x16x4f32(syn_pr(a, vr, vi), bx, accE, 0, 0, 0);
        accO = __builtin_amdgcn_mfma_f32_16x16x4f32(syn_npi(a, vr, vi), by, accO, 0, 0, 0);
        syn_rot(vr, vi, u4r, u4i);
        syn_rot(bx, by, s4r, s4i);
      }
    }
#pragma unroll
    for(int r = 0; r < 4; r ++) {
      const float e = accE[r], o = accO[r];
      if(tP[r] >= 0) ring[(st + tP[r]) & (R - 1)] += (e + o) * wP[r];
      if(tM[r] >= 0) ring[(st + tM[r]) & (R - 1)] += (e - o) * wM[r];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  advance(own_hi);
}

// ---------------------------------------------------------------- launchers
// Frames per unit = total frames / this divisor (about one unit per resident wavefront): the table kernel keeps 4
// wavefronts per SIMD resident (4096 on the chip), the one-wavefront kernel 8.  Measured on config 2 (tools/kbench.py,
// LLSM_GPU_SIN_UNIT): 25-frame units 0.845, 50-frame units 0.812, 40-frame units 1.22 ms per step for the table kernel.
static bool synth_ola_uses_groups(int nwin, int lds_harmonics) {
  const int T = ((nwin + 15) / 16 + 2 + 31) / 32;
  return T == 1 && lds_harmonics <= 4 * SYN_TAB_MAXKS;
}
int synth_ola_unit_div(int nwin, int lds_harmonics) { return synth_ola_uses_groups(nwin, lds_harmonics) ? 4096 : 8192; }
int synth_ola_group_units(void) { return SO4_WAVES; }   // units per workgroup of k_synth_ola4: the host pads every utterance to groups
int launch_synth_frames(LaunchCtx* P, const BatchDev& d, int nwin, const float* win,
  const float* cyc_shift, float* frames, int lds_harmonics) {
  if(d.nframes == 0) return 0;
  // row length L = 32 T - 2 samples (16 rows cover nwin): L/2 + 1 = 16 T offsets from the row
  // centre, in passes of NT <= 4 column tiles
  int T = ((nwin + 15) / 16 + 2 + 31) / 32;
  int NT = T;
  if(T > 4) { T = (T + 3) / 4 * 4; NT = 4; }
  const int L = 32 * T - 2;
  const size_t lds = (lds_harmonics + 4) * sizeof(float2);
#define SF_ARGS d.frm_utt, d.frm_off, d.f0, d.nhar, d.ampl, d.phse, d.maxnhar, d.thop, d.fs, nwin, L, win, \
    cyc_shift, frames
  switch(NT) {
    case 1: LAUNCH("k_synth_frames", (k_synth_frames<1>), dim3(d.nframes), dim3(WAVE), lds, SF_ARGS); break;
    case 2: LAUNCH("k_synth_frames", (k_synth_frames<2>), dim3(d.nframes), dim3(WAVE), lds, SF_ARGS); break;
    case 3: LAUNCH("k_synth_frames", (k_synth_frames<3>), dim3(d.nframes), dim3(WAVE), lds, SF_ARGS); break;
    default: LAUNCH("k_synth_frames", (k_synth_frames<4>), dim3(d.nframes), dim3(WAVE), lds, SF_ARGS); break;
  }
#undef SF_ARGS
  return 0;
}

// Fused harmonic frames + overlap-add over the units of a batch (see k_synth_ola).
int launch_synth_ola(LaunchCtx* P, const BatchDev& d, const int4* units, int nunits, int halo,
  int nwin, const float* win, int lds_harmonics, const int* out_off, const int* out_len,
  const float* x, float* out, int mode, float* mix) {
  if(nunits == 0) return 0;
  int T = ((nwin + 15) / 16 + 2 + 31) / 32;
  int NT = T;
  if(T > 4) { T = (T + 3) / 4 * 4; NT = 4; }
  const int L = 32 * T - 2;
  int R = 64; while(R < nwin) R <<= 1;
  if(synth_ola_uses_groups(nwin, lds_harmonics) && nunits % SO4_WAVES == 0) {
    So4Args a;
    a.units = units; a.halo = halo; a.R = R; a.frm_off = d.frm_off; a.nfrm = d.nfrm; a.out_off = out_off; a.out_len = out_len;
    a.f0 = d.f0; a.nhar = d.nhar; a.ampl = d.ampl; a.phse = d.phse; a.maxnhar = d.maxnhar; a.thop = d.thop; a.fs = d.fs;
    a.nwin = nwin; a.L = L; a.win = win; a.lds_harmonics = lds_harmonics; a.nks_tab = (lds_harmonics + 3) / 4;
    a.use_tab = d.synth_tables; a.x = x; a.out = out; a.mode = mode; a.mix = mix;
    const size_t lds4 = (size_t)a.nks_tab * WAVE * sizeof(float4)
      + SO4_WAVES * ((lds_harmonics + 4) * sizeof(float2) + R * sizeof(float));
    LAUNCH("k_synth_ola4", k_synth_ola4, dim3(nunits / SO4_WAVES), dim3(SO4_WAVES * WAVE), lds4, a);
    return 0;
  }
  const size_t lds = (lds_harmonics + 4) * sizeof(float2) + R * sizeof(float);
#define SO_ARGS units, halo, R, d.frm_off, d.nfrm, out_off, out_len, d.f0, d.nhar, d.ampl, d.phse, d.maxnhar, \
    d.thop, d.fs, nwin, L, win, lds_harmonics, x, out, mode, mix
  switch(NT) {
    case 1: LAUNCH("k_synth_ola", (k_synth_ola<1>), dim3(nunits), dim3(WAVE), lds, SO_ARGS); break;
    case 2: LAUNCH("k_synth_ola", (k_synth_ola<2>), dim3(nunits), dim3(WAVE), lds, SO_ARGS); break;
    case 3: LAUNCH("k_synth_ola", (k_synth_ola<3>), dim3(nunits), dim3(WAVE), lds, SO_ARGS); break;
    default: LAUNCH("k_synth_ola", (k_synth_ola<4>), dim3(nunits), dim3(WAVE), lds, SO_ARGS); break;
  }
#undef SO_ARGS
  return 0;
}
