// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the layer-0
// analysis / synthesis path of libllsm2_amd.
//
// Execution model used throughout: one 64-lane wavefront owns one frame or one pair of frames
// (blockDim = 64; pairs always belong to one utterance, BatchDev::pairs), frame data lives in
// registers (wave_fft.h, the MFMA kernels) or LDS, cross-lane sums use the wavefront shuffle
// butterfly.  Overlap-add never uses atomics and sums a sample's frames in ascending frame
// order -- the addition order of the reference's sequential loops: on chip by the kernel that
// produces the frames (k_synth_ola, k_noise_filter_ola), or as a gather over the frames covering
// a sample (envelope frames, fallback paths, llsmrt).
//
// Phase arithmetic convention (DESIGN.md "phase precision"): every angle of
// the form 2*pi*f*k*(t - c) is formed in float64 *turns*, reduced to
// [-0.5, 0.5] and only then handed to the float32 sincospi; the bulk
// multiply-accumulate work is float32.
//
// Each kernel cites the reference code it replaces (file:line in the
// reference tree).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <climits>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include "kernels.h"
#include "plan.h"

namespace lp = llsm_plan;

// The library is built with -ffp-contract=off so that the index plan (plan.h, included above:
// single correctly rounded operations shared with the host) can never be fused.  The DSP
// arithmetic below has no such constraint and is VALU-bound: let a*b + c become one FMA
// (a complex multiply is 4 instructions instead of 6, and is more accurate).
#pragma clang fp contract(fast)

#include "dev_common.h"
#include "synth_frame.h"                             // f32x4, cs_rot, xcd_frame, synth_frame (llsmrt hop kernels)
#include "launch.h"
// Timing experiments (tools/kbench.py --ablate IIR_FAKE_L2=1 / IIR_GEN_EXPERIMENT=1, RT2_TIMING) live OUTSIDE the product
// sources, in tools/kbench_experiments.h, and only a build that also passes -DLLSM_KBENCH_EXPERIMENTS (kbench does) can
// reach them: the product translation unit holds three empty hooks, and a stray -D of one of the switches is a
// compile error instead of a library that computes garbage (IIR_FAKE_L2 does, by design).
#if defined(LLSM_KBENCH_EXPERIMENTS)
#include "../../tools/kbench_experiments.h"
#else
#if defined(IIR_FAKE_L2) || defined(IIR_GEN_EXPERIMENT) || defined(RT2_TIMING) || defined(HT_TABLE_EXPERIMENT) || defined(KAL_BREAK)
#error "timing-experiment switches need -DLLSM_KBENCH_EXPERIMENTS (tools/kbench.py); the product library is never built with them"
#endif
#define RT2_T(i)
#define IIR_EXP_JOB(job, jobs, j)
#define IIR_EXP_GEN(fwd, square, src, gen_src, idx0, q) false
#define HT_EXP_TWIDDLE(NT, ks, tt, wr, wi) false
#endif

extern __shared__ __attribute__((aligned(16))) unsigned char g_lds[];

// Conventions of the un-vendored ciglet primitives that the reference's code cannot confirm (DESIGN.md
// section 6), switchable so that parity can be re-established the day a real ciglet build says otherwise
// (llsm_gpu_set_convention; the oracle has the same switches).  Defaults = the definitions of DESIGN.md.
__device__ DevConventions g_conv = {3, 0, 0, 0.13397922601295542f, 0};
int llsm_kernels_set_conventions(const DevConventions& c) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_conv), & c, sizeof(c)) == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------ helpers
// Sum each of NV per-lane values over the 64 lanes with the halving butterfly: at every step
// a lane keeps one half of its values and hands the other half to its partner, so the cost is
// ~2 NV exchanges instead of 6 NV.  On return lane l holds, in v[0], the total of value index
// wave_reduce_index<NV>(l); NV must be a power of two <= 64.
template <int NV>
DEV void wave_reduce_many(float (&v)[NV], int lane) {
  int n = NV, bit = WAVE / 2;
#pragma unroll
  for(int st = 0; st < 6; st ++) {
    if(n > 1) {
      const bool hi = (lane & bit) != 0;
#pragma unroll
      for(int i = 0; i < NV / 2; i ++) {
        if(i < n / 2) {
          // opaque copies: otherwise the select of two array elements becomes a select of the
          // INDEX, i.e. dynamic register indexing expanded into an NV-way compare chain
          float lo_v = v[i], hi_v = v[i + n / 2];
          asm volatile("" : "+v"(lo_v), "+v"(hi_v));
          const float send = hi ? lo_v : hi_v;
          const float keep = hi ? hi_v : lo_v;
          v[i] = keep + __shfl_xor(send, bit, WAVE);
        }
      }
      n >>= 1;
    } else {
      v[0] += __shfl_xor(v[0], bit, WAVE);
    }
    bit >>= 1;
  }
}
// index (0 .. NV-1) of the value lane `lane` ends up with: the kept half at step st adds n/2
template <int NV>
DEV int wave_reduce_index(int lane) {
  int idx = 0, n = NV, bit = WAVE / 2;
#pragma unroll
  for(int st = 0; st < 6; st ++) {
    if(n > 1) { if(lane & bit) idx += n / 2; n >>= 1; }
    bit >>= 1;
  }
  return idx;
}
// Guarded load without a branch: the address is clamped into the array and the value is
// discarded when the index is outside (a `cond ? p[i] : 0` compiles to one exec-mask branch per
// load, which serialises the 32-64 staging loads of the register-resident kernels).
// p must be valid for at least one element.
DEV float ld_guard(const float* __restrict__ p, int idx, int n, bool ok) {
  const bool in = ok && idx >= 0 && idx < n;
  const float v = p[in ? idx : 0];
  return in ? v : 0.0f;
}
// Range-checked loads in hardware: a raw buffer descriptor that covers elements [lo, hi) of p;
// every load outside it (including negative indices: the byte offset is unsigned) returns 0
// without touching memory.  One instruction per load, no compare / select -- for the windowed
// frame loads, where "inside the window" and "inside the signal" intersect to one range per frame.
typedef __amdgpu_buffer_rsrc_t buf_t;
DEV buf_t buf_range(const float* p, int lo, int hi) {
  const int n = hi > lo ? hi - lo : 0;
  return __builtin_amdgcn_make_buffer_rsrc((void*)(p + lo), 0, n * 4, 0x00020000);
}
// The byte offset is made opaque so that the compiler cannot split it into register + immediate
// offset: with a NEGATIVE register part (window clipped at the start of the signal) and a positive
// immediate the hardware range check does not always see the in-range sum (measured: wrong zeros in
// k_harm_env on the first frames of an utterance; tools/ubench/buf_wrap.hip covers only the
// small-negative case, which works).
DEV float ld_range(buf_t r, int idx_minus_lo) {
  int off = idx_minus_lo * 4;
  asm volatile("" : "+v"(off));
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}


DEV float ld_range_b(buf_t r, int byte_off) {       // the same with the byte offset formed by the caller
  asm volatile("" : "+v"(byte_off));
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

DEV float blackman_at(int t, int n) {           // symmetric, DESIGN.md "windows"
  if(n == 1) return 1.0f;
  float u = (float)t / (float)(n - 1);
  return 0.42f - 0.5f * cospif(2.0f * u) + 0.08f * cospif(4.0f * u);
}
DEV float hann_at(int t, int n) {
  if(n == 1) return 1.0f;
  float u = (float)t / (float)(n - 1);
  return 0.5f - 0.5f * cospif(2.0f * u);
}


// The spectral kernels transform TWO real frames per complex FFT.  Which two is fixed by a host-built table
// (frames i, i + 1 of the SAME utterance, i even; a trailing odd frame goes alone): an utterance's rows then do
// not depend on what else is in the batch or where it sits (the partner leaks into a frame at rounding level).
// pairs == NULL: consecutive global frames (llsmrt: streams).  g1 == nframes marks "no second frame".
DEV void pair_of(const int2* __restrict__ pairs, int p, int nframes, int& g0, int& g1) {
  if(pairs) { const int2 q = pairs[p]; g0 = q.x; g1 = q.y < 0 ? nframes : q.y; }
  else { g0 = 2 * p; g1 = 2 * p + 1; }
}

// =====================================================================
// K1  harmonic analysis of the speech signal (HOT LOOP A) on the f32 MFMA
// replaces llsm_harmonic_analysis / llsm_harmonic_czt, dsputils.c:145-228:
//   X_h = sum_t w[t] x[c - n/2 + t] e^{-j w0 h (t - n/2)},  h = 1..nhar
//   ampl = |X_h| * 2 / sum(w),  phse = arg X_h
// Every frame has its own w0, so this is not a shared-operand GEMM; it becomes
// one through a two-level factorisation of the time index, t = L*a + b with
// a in [0,16), b in [0,L):
//   X_h = sum_a e^{-j w0 h (L a - n/2)} * S[a][h],
//   S[a][h] = sum_b xw[L a + b] * e^{-j w0 h b}          (a 16 x L x 2nhar GEMM)
// The inner GEMM runs on v_mfma_f32_16x16x4_f32 (exact f32, bitwise an fmaf
// chain), split into the even and odd parts of every row about its centre (half
// the columns, see k_harm_speech below): A = windowed even / odd sample sums formed
// in registers straight from global memory (HarmRow), B = per-frame twiddles
// generated in registers by a 4-sample phasor step (this VALU work ADDS to the MFMA
// cycles on gfx950, it does not hide under them), 7 harmonic tiles x (cos, -sin) =
// 14 accumulators per pass of 112 harmonics.  The outer 16-term sum is VALU work
// plus two cross-lane adds.  Seeds and steps of every phasor come from
// float64-reduced phases.
// =====================================================================
#define HM_ROWS 16
#define HM_TILES 7

// NT harmonic tiles (16 harmonics each, cos and -sin accumulators) of one frame:
// inner GEMM over the L columns, then the outer 16-row sum; results in Pr/Pi[0..NT).

// Per-lane source of the A operands (row = lane & 15, k = (lane >> 4) + 4 ks): the even / odd
// parts of the windowed row about its centre are formed straight from global memory,
//   E[k] = xw[rho + k] + xw[rho - k],  O[k] = xw[rho + k] - xw[rho - k],
// with the Blackman window 0.34 - 0.5 c + 0.16 c^2 evaluated from c = cos(alpha +- beta_k):
// alpha (row centre) is fixed per lane, beta_k advances by a 4-sample rotation.
struct HarmRow {
  buf_t rng; int lo;             // readable samples of the utterance inside the window: [lo, hi), zero elsewhere
  int t0;                        // window index of the row centre: rho + n/2
  int org;                       // signal index of window sample 0
  int n, L;
  float ca, sa;                  // cos, sin(2 pi t0 / (n - 1))
  float stc, sts;                // cos, sin(2 pi 4 / (n - 1))
};
#define HM_CHUNK 4               // k-steps whose operands are loaded together

// C k-steps of the inner GEMM: operands of all C steps are loaded before the first use
template <int NT, int C>
DEV void harm_steps(const HarmRow& R, int ks0, int q, float& cb, float& sb, float& wsum,
  float (&wr)[NT], float (&wi)[NT], const float (&rc)[NT], const float (&rs)[NT],
  f32x4 (&are)[NT], f32x4 (&aim)[NT]) {
  const int L = R.L;
  float xp[C], xm[C];
#pragma unroll
  for(int j = 0; j < C; j ++) {
    const int k = q + ks0 + 4 * j;
    const int tp = R.t0 + k, tm = R.t0 - k;
    xp[j] = ld_range(R.rng, R.org + tp - R.lo);      // slots outside the row get window weight 0 below
    xm[j] = ld_range(R.rng, R.org + tm - R.lo);
  }
  float ev[C], ov[C];
#pragma unroll
  for(int j = 0; j < C; j ++) {
    const int k = q + ks0 + 4 * j;
    const int tp = R.t0 + k, tm = R.t0 - k;
    const float cc = R.ca * cb, ss = R.sa * sb;
    const float cp = cc - ss, cm = cc + ss;          // cos(alpha + beta), cos(alpha - beta)
    float wp = R.n > 1 ? fmaf(0.16f * cp, cp, fmaf(-0.5f, cp, 0.34f)) : 1.0f;
    float wm = R.n > 1 ? fmaf(0.16f * cm, cm, fmaf(-0.5f, cm, 0.34f)) : 1.0f;
    if(!(k < L / 2 && tp >= 0 && tp < R.n)) wp = 0.0f;
    if(!(k > 0 && k <= L / 2 && tm >= 0 && tm < R.n)) wm = 0.0f;
    wsum += wp + wm;
    const float vp = xp[j] * wp, vm = xm[j] * wm;
    ev[j] = vp + vm; ov[j] = vp - vm;
    cs_rot(cb, sb, R.stc, R.sts);
  }
#pragma unroll
  for(int j = 0; j < C; j ++) {
#pragma unroll
    for(int tt = 0; tt < NT; tt ++) {
      are[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[j], wr[tt], are[tt], 0, 0, 0);   // sum E cos
      aim[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ov[j], wi[tt], aim[tt], 0, 0, 0);   // -sum O sin
      const float nr = fmaf(wr[tt], rc[tt], wi[tt] * rs[tt]);
      const float ni = fmaf(wi[tt], rc[tt], -wr[tt] * rs[tt]);
      wr[tt] = nr; wi[tt] = ni;
    }
  }
}

template <int NT>
DEV float harm_block(const HarmRow& R, int KC, double turn1, int h0, int col, int q,
  float* Pr, float* Pi) {
  // Phasors of tile tt belong to harmonic hh = h0 + 16 tt + col + 1: e^{-j 2 pi turn1 hh m} for
  // m = q (B seed), 4 (B step), rho_a (outer seed), L (outer step).  Tile 0 comes from
  // float64-reduced phases; tile tt + 1 is tile tt rotated by the per-lane constant
  // e^{-j 2 pi 16 turn1 m} (<= 6 rotations: error ~ 4e-7, and 8 instead of 4 NT cs_turns).
  const double fk0 = turn1 * (double)(h0 + col + 1), fd = turn1 * 16.0;
  const int L = R.L;
  float wr[NT], wi[NT], rc[NT], rs[NT];
  f32x4 are[NT], aim[NT];
  {
    float c0, s0, c1, s1, d0c, d0s, d1c, d1s;
    cs_turns(fk0 * (double)q, & c0, & s0);
    cs_turns(fk0 * 4.0, & c1, & s1);
    cs_turns(fd * (double)q, & d0c, & d0s);
    cs_turns(fd * 4.0, & d1c, & d1s);
#pragma unroll
    for(int tt = 0; tt < NT; tt ++) {
      wr[tt] = c0; wi[tt] = -s0;                     // e^{-j 2 pi fk k}, k = q
      rc[tt] = c1; rs[tt] = s1;                      // 4-sample step
      are[tt] = (f32x4){0, 0, 0, 0}; aim[tt] = (f32x4){0, 0, 0, 0};
      cs_rot(c0, s0, d0c, d0s); cs_rot(c1, s1, d1c, d1s);
    }
  }
  float cb, sb;                                      // cos, sin(2 pi k / (n - 1)), k = q + 4 ks
  cs_turns((double)q / (double)(R.n > 1 ? R.n - 1 : 1), & cb, & sb);
  float wsum = 0.0f;
  int ks0 = 0;
  // the MFMA loops contain no branches (a guarded MFMA makes the compiler shuttle every
  // accumulator through VGPRs each step): full chunks, then single steps
  for(; ks0 + 4 * HM_CHUNK <= KC; ks0 += 4 * HM_CHUNK)
    harm_steps<NT, HM_CHUNK>(R, ks0, q, cb, sb, wsum, wr, wi, rc, rs, are, aim);
  for(; ks0 < KC; ks0 += 4)
    harm_steps<NT, 1>(R, ks0, q, cb, sb, wsum, wr, wi, rc, rs, are, aim);
  // outer sum over the 16 rows: lane holds rows a = 4q + r (r = 0..3) of column `col`
  float vc, vs, sc, ss, d2c, d2s, d3c, d3s;
  const int rho0 = L * (4 * q - HM_ROWS / 2) + L / 2;     // centre of row a = 4 q, relative to the window centre
  cs_turns(fk0 * (double)rho0, & vc, & vs);
  cs_turns(fk0 * (double)L, & sc, & ss);
  cs_turns(fd * (double)rho0, & d2c, & d2s);
  cs_turns(fd * (double)L, & d3c, & d3s);
#pragma unroll
  for(int tt = 0; tt < NT; tt ++) {
    float vr = vc, vi = -vs;                       // e^{-j 2 pi fk rho_a}
    float pr = 0, pi = 0;
#pragma unroll
    for(int r = 0; r < 4; r ++) {
      const float sr = are[tt][r], si = aim[tt][r];
      pr = fmaf(vr, sr, fmaf(-vi, si, pr));
      pi = fmaf(vr, si, fmaf(vi, sr, pi));
      const float nr = fmaf(vr, sc, vi * ss), ni = fmaf(vi, sc, -vr * ss);
      vr = nr; vi = ni;
    }
    pr += __shfl_xor(pr, 16, WAVE); pi += __shfl_xor(pi, 16, WAVE);
    pr += __shfl_xor(pr, 32, WAVE); pi += __shfl_xor(pi, 32, WAVE);
    Pr[tt] = pr; Pi[tt] = pi;
    cs_rot(vc, vs, d2c, d2s); cs_rot(sc, ss, d3c, d3s);
  }
  return wsum;
}

// ---------------------------------------------------------------------
// K1t  the same analysis for TILES of frames that share one F0 (the fixed-F0
// shared-operand GEMM).  Frames of one utterance with bit-identical F0 have the same window length n,
// the same harmonic count and the same twiddles, so 16 of them are the 16 ROWS of the MFMA and the
// K dimension is the even/odd-folded window itself:
//   X_h(frame r) = sum_k E_r[k] cos(th_h k) - j sum_k O_r[k] sin(th_h k),   k = 0 .. n/2,
//   E_r[k] = w[n/2 + k] x[c_r + k] + w[n/2 - k] x[c_r - k],  O_r[k] = w[n/2 + k] x[c_r + k] - w[n/2 - k] x[c_r - k]
// (w = 0 outside the window, w[n/2 - 0] counted once).  The B operands cos / -sin(th_h k) are ONE phasor
// recurrence per (harmonic tile, lane) shared by the 16 frames -- the per-frame kernel spends 2000 VALU
// instructions per frame on twiddles, float64 seeds and the outer 16-row sum, all of which exist only
// because it must treat every frame's F0 as its own --, the window comes from a table built once per tile
// in LDS, and there is no outer sum: the accumulators ARE the harmonics.  Phasors are re-seeded from
// float64-reduced phases every HT_SEG k-steps.
//
// Which frames go this way is decided per 16-ALIGNED BLOCK of an utterance's frames from that block's F0
// values alone (harm_tile_of), identically by this kernel and by k_harm_speech, which skips them: a
// frame's result therefore does not depend on what else is in the batch.
// ---------------------------------------------------------------------
#define HT_MINROWS 8                               // below this a tile costs more than its frames one by one
#define HT_KCAP_MAX (48 * 1024 / 8)                // window-table slots of the largest LDS provision (launch_harm_speech)
#ifndef HT_SEG
#define HT_SEG 46                                  // k-steps between exact phasor re-seeds
#endif
#ifndef HT_CHUNK
#define HT_CHUNK 2                                 // 4: 19 spilled registers at 4 wavefronts / SIMD and 2.5 % slower
#endif
#ifndef HT_SCHED
#define HT_SCHED 1
#endif
#ifndef HT_PREFETCH
#define HT_PREFETCH 1                              // -2.7 % (tools/kbench.py)
#endif

// The tile of one block: the first run of >= HT_MINROWS consecutive voiced frames with bit-identical F0
// whose folded window fits the LDS provision (kcap table slots).  Wave-uniform result; lanes 0..15 each
// look at one frame.  Returns false when the block has no tile.
DEV bool harm_tile_of(const float* __restrict__ f0, int g_first, int count, int lane, int kcap,
  float fs, float rel, int maxnhar, int* start, int* rows, float* f_tile) {
  const float f = lane < count ? f0[g_first + lane] : 0.0f;
  const float fprev = __shfl_up(f, 1, WAVE);
  const bool st = lane < count && (lane == 0 || __float_as_uint(f) != __float_as_uint(fprev));
  const unsigned S = (unsigned)__ballot(st) & 0xFFFFu;            // run starts (bit 0 is always set)
  const int l = lane & 15;
  const int s = 31 - __clz((int)(S & ((2u << l) - 1u)));           // start of this lane's run
  const unsigned above = (S >> (l + 1)) << (l + 1);
  const int e = above ? __ffs((int)above) - 1 : count;             // its end
  bool ok = lane < count && f > 0 && e - s >= HT_MINROWS;
  if(ok) {
    const int n = lp::hwin(f, fs, rel);
    ok = n >= 2 && n / 2 + 4 <= kcap && lp::nhar(f, fs, maxnhar) >= 1;
  }
  const unsigned long long E = __ballot(ok);
  if(E == 0) return false;
  const int lead = __ffsll((unsigned long long)E) - 1;
  // v_readlane with a scalar lane index: the results are wave-uniform for the compiler too (SGPRs), so that
  // everything derived from them -- the utterance, its buffer descriptor -- stays scalar
  *start = __builtin_amdgcn_readlane(s, lead); *rows = __builtin_amdgcn_readlane(e - s, lead);
  *f_tile = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(f), lead));
  return true;
}

// C k-steps of NT harmonic tiles: loads of all C steps first, then the folded operands, then the MFMAs
// operands of C k-steps (k = k0 + 4 j): the two signal samples and the two window values of lane (row, q)
template <int C>
struct HarmTileOps { float xp[C], xm[C], vp[C], vm[C]; };
template <int C>
DEV void harm_tile_load(HarmTileOps<C>& o, buf_t rng, int cidx, const float* __restrict__ wp, const float* __restrict__ wm,
  int k0, int kmax) {
#pragma unroll
  for(int j = 0; j < C; j ++) {
    const int k = min(k0 + 4 * j, kmax);             // a prefetch past the last step re-reads the last slot
    o.xp[j] = ld_range(rng, cidx + k);
    o.xm[j] = ld_range(rng, cidx - k);
    o.vp[j] = wp[k]; o.vm[j] = wm[k];
  }
}
// C k-steps of NT harmonic tiles: the folded operands, then the MFMAs
template <int NT, int C>
DEV void harm_tile_steps(const HarmTileOps<C>& o, int ks, float (&wr)[NT], float (&wi)[NT], const float (&rc)[NT],
  const float (&rs)[NT], f32x4 (&are)[NT], f32x4 (&aim)[NT]) {
  float ev[C], ov[C];
#pragma unroll
  for(int j = 0; j < C; j ++) {
    const float a = o.xp[j] * o.vp[j];
    ev[j] = fmaf(o.xm[j], o.vm[j], a);
    ov[j] = fmaf(-o.xm[j], o.vm[j], a);
  }
#pragma unroll
  for(int j = 0; j < C; j ++) {
#pragma unroll
    for(int tt = 0; tt < NT; tt ++) {
      are[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[j], wr[tt], are[tt], 0, 0, 0);   // sum E cos
      aim[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ov[j], wi[tt], aim[tt], 0, 0, 0);   // -sum O sin
      if(HT_EXP_TWIDDLE(NT, ks + j, tt, wr[tt], wi[tt])) continue;   // (timing experiment hook: false in the product)
      const float nr = fmaf(wr[tt], rc[tt], wi[tt] * rs[tt]);
      const float ni = fmaf(wi[tt], rc[tt], -wr[tt] * rs[tt]);
      wr[tt] = nr; wi[tt] = ni;
    }
#if HT_SCHED == 1
    // one run of MFMAs, then one run of VALU work per k-step: every MFMA <-> VALU alternation costs ~9 cycles of issue
    // on gfx950 (tools/ubench/mfma_valu: 14 MFMA + 28 VALU = 620 cycles interleaved, 545 grouped at 3 wavefronts / SIMD)
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 4 * NT, 0);
#endif
  }
}

// Harmonics h0 + 1 .. h0 + 16 NT of the tile's 16 rows.  The workgroup's four wavefronts split the folded window:
// wavefront w accumulates k-steps [w nq, (w + 1) nq) from its own float64-reduced seeds; the four partial GEMMs
// are then summed in the fixed order ((p0 + p1) + p2) + p3 through LDS as a reduce-scatter -- wavefront j ends up
// with a quarter of the (harmonic, frame) sums and does the |.| / arg work for exactly those.  One tile is thus
// a unit of four wavefronts, a quarter as long as the whole GEMM: the launch balances over the chip where
// one-wavefront tiles came in 3.25 rounds of resident wavefronts (measured: 0.79 -> see DESIGN.md).
template <int NT>
DEV void harm_tile_block(buf_t rng, int cidx, const float* __restrict__ wp, const float* __restrict__ wm,
  float* __restrict__ red, int nks, double turn1, int h0, int K, int wv, int col, int q, float scale,
  int g0, int rows, int maxnhar, float* __restrict__ ampl, float* __restrict__ phse) {
  const double fk0 = turn1 * (double)(h0 + col + 1), fd = turn1 * 16.0;
  float wr[NT], wi[NT], rc[NT], rs[NT];
  f32x4 are[NT], aim[NT];
  {
    float c1, s1, d1c, d1s;
    cs_turns(fk0 * 4.0, & c1, & s1);
    cs_turns(fd * 4.0, & d1c, & d1s);
#pragma unroll
    for(int tt = 0; tt < NT; tt ++) {
      rc[tt] = c1; rs[tt] = s1;                      // 4-sample step of tile tt
      are[tt] = (f32x4){0, 0, 0, 0}; aim[tt] = (f32x4){0, 0, 0, 0};
      cs_rot(c1, s1, d1c, d1s);
    }
  }
  const int nq = (nks + 3) / 4;
  const int kbeg = min(nks, wv * nq), kend = min(nks, kbeg + nq);
  for(int ks0 = kbeg; ks0 < kend; ks0 += HT_SEG) {
    const int ks1 = min(kend, ks0 + HT_SEG);
    {
      // exact seeds at k = q + 4 ks0: tile 0 from the float64-reduced phase, tile tt + 1 = tile tt rotated by the
      // per-lane constant e^{-j 2 pi 16 turn1 k} (as harm_block)
      const double kk = (double)(q + 4 * ks0);
      float c0, s0, d0c, d0s;
      cs_turns(fk0 * kk, & c0, & s0);
      cs_turns(fd * kk, & d0c, & d0s);
#pragma unroll
      for(int tt = 0; tt < NT; tt ++) {
        wr[tt] = c0; wi[tt] = -s0;
        cs_rot(c0, s0, d0c, d0s);
      }
    }
    int ks = ks0;
    const int kmax = q + 4 * (nks - 1);
#if HT_PREFETCH
    // the operands of chunk c + 1 are requested before the MFMAs of chunk c: their latency hides under the wavefront's
    // own matrix work instead of waiting for the other wavefronts of the SIMD to cover it
    HarmTileOps<HT_CHUNK> nxt;
    harm_tile_load<HT_CHUNK>(nxt, rng, cidx, wp, wm, q + 4 * ks, kmax);
    for(; ks + HT_CHUNK <= ks1; ks += HT_CHUNK) {
      const HarmTileOps<HT_CHUNK> cur = nxt;
      harm_tile_load<HT_CHUNK>(nxt, rng, cidx, wp, wm, q + 4 * (ks + HT_CHUNK), kmax);
      harm_tile_steps<NT, HT_CHUNK>(cur, ks, wr, wi, rc, rs, are, aim);
    }
#else
    for(; ks + HT_CHUNK <= ks1; ks += HT_CHUNK) {
      HarmTileOps<HT_CHUNK> cur;
      harm_tile_load<HT_CHUNK>(cur, rng, cidx, wp, wm, q + 4 * ks, kmax);
      harm_tile_steps<NT, HT_CHUNK>(cur, ks, wr, wi, rc, rs, are, aim);
    }
#endif
    for(; ks < ks1; ks ++) {
      HarmTileOps<1> cur;
      harm_tile_load<1>(cur, rng, cidx, wp, wm, q + 4 * ks, kmax);
      harm_tile_steps<NT, 1>(cur, ks, wr, wi, rc, rs, are, aim);
    }
  }
  // reduce-scatter: complex sum c = 4 tt + r (harmonic tile tt, accumulator row r), quarter j = sums [NT j, NT j + NT).
  // red: [4 wavefronts][2 NT values][64 lanes]
  const int lane = col + 16 * q;
#pragma unroll
  for(int j = 0; j < 4; j ++) {
    if(wv != j) {
#pragma unroll
      for(int i = 0; i < NT; i ++) {
        const int c = NT * j + i;
        red[(wv * 2 * NT + 2 * i) * WAVE + lane] = are[c >> 2][c & 3];
        red[(wv * 2 * NT + 2 * i + 1) * WAVE + lane] = aim[c >> 2][c & 3];
      }
    }
    __syncthreads();
    if(wv == j) {
#pragma unroll
      for(int i = 0; i < NT; i ++) {
        const int c = NT * j + i;
        float pr = 0.0f, pi = 0.0f;
#pragma unroll
        for(int w = 0; w < 4; w ++) {                // fixed order p0 + p1 + p2 + p3, own part from registers
          const float vr = w == j ? are[c >> 2][c & 3] : red[(w * 2 * NT + 2 * i) * WAVE + lane];
          const float vi = w == j ? aim[c >> 2][c & 3] : red[(w * 2 * NT + 2 * i + 1) * WAVE + lane];
          pr = w == 0 ? vr : pr + vr; pi = w == 0 ? vi : pi + vi;
        }
        // D[row = 4 q + r][col]: harmonic h0 + 16 tt + col + 1 of frame g0 + row
        const int h = h0 + 16 * (c >> 2) + col + 1, row = 4 * q + (c & 3);
        if(row < rows && h <= K) {
          const size_t o = (size_t)(g0 + row) * maxnhar + (h - 1);
          ampl[o] = sqrtf(pr * pr + pi * pi) * scale;
          phse[o] = atan2f(pi, pr);
        }
      }
    }
    __syncthreads();
  }
}

#ifndef HT_WPE
#define HT_WPE 4                                   // an EVEN number of wavefronts per SIMD: 3 measure 7 % slower than 2 or 4 (mfma_valu)
#endif
#define HT_NT (4 * WAVE)
// One workgroup of four wavefronts per 16-frame block of one utterance (hblocks[p]).
__global__ __launch_bounds__(HT_NT, HT_WPE) void k_harm_speech_tile(
  const int2* __restrict__ hblocks, int kcap,
  const float* __restrict__ x, const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off,
  const float* __restrict__ f0, float thop, float fs, float rel_winsize, int maxnhar,
  int* __restrict__ nhar_out, float* __restrict__ ampl, float* __restrict__ phse) {
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int2 blk = hblocks[xcd_frame(blockIdx.x, gridDim.x)];      // (first global frame, frames) of the block
  int s, rows; float f;
  if(! harm_tile_of(f0, blk.x, blk.y, lane, kcap, fs, rel_winsize, maxnhar, & s, & rows, & f)) return;
  const int g0 = blk.x + s;
  const int u = frm_utt[g0], i0 = g0 - frm_off[u];
  const int n = lp::hwin(f, fs, rel_winsize), half = n / 2;
  const int K = lp::nhar(f, fs, maxnhar);
  const int nks = (half + 4) / 4;                    // k = 0 .. half in steps of 4 slots
  float* wp = (float*)g_lds;                         // w[half + k], 0 beyond the window
  float* wm = wp + kcap;                             // w[half - k], k >= 1
  float* red = wm + kcap;                            // [4][2 HM_TILES][64] partial sums + 4 window sums
  float* wsums = red + 4 * 2 * HM_TILES * WAVE;
  // Blackman window 0.34 - 0.5 c + 0.16 c^2, c = cos(2 pi t / (n - 1)) at t = half +- k by angle addition
  float wsum = 0.0f;
  {
    const double inv = 1.0 / (double)(n > 1 ? n - 1 : 1);
    float ca, sa; cs_turns((double)half * inv, & ca, & sa);
    for(int k = tid; k < 4 * nks; k += HT_NT) {
      float cb, sb; cs_turns((double)k * inv, & cb, & sb);
      const float cc = ca * cb, ss = sa * sb;
      const float cp = cc - ss, cm = cc + ss;
      float a = fmaf(0.16f * cp, cp, fmaf(-0.5f, cp, 0.34f));
      float b = fmaf(0.16f * cm, cm, fmaf(-0.5f, cm, 0.34f));
      if(!(half + k < n)) a = 0.0f;
      if(!(k >= 1 && k <= half)) b = 0.0f;
      wp[k] = a; wm[k] = b;
      wsum += a + b;
    }
  }
  wsum = wave_sum(wsum);
  if(lane == 0) wsums[wv] = wsum;
  __syncthreads();
  const float scale = 2.0f / (((wsums[0] + wsums[1]) + wsums[2]) + wsums[3]);
  const int col = lane & 15, q = lane >> 4;
  const int cidx = lp::center(i0 + (col < rows ? col : 0), thop, fs);   // idle rows repeat row 0 (results discarded)
  const buf_t rng = buf_range(x + x_off[u], 0, nx[u]);   // zero outside the utterance
  const double turn1 = (double)f / (double)fs;
  for(int h0 = 0; h0 < K; h0 += 16 * HM_TILES) {
    const int ntile = min(HM_TILES, (K - h0 + 15) / 16);
#define HT_CALL(NT) harm_tile_block<NT>(rng, cidx, wp, wm, red, nks, turn1, h0, K, wv, col, q, scale, g0, rows, maxnhar, ampl, phse)
    switch(ntile) {
      case 7: HT_CALL(7); break;
      case 6: HT_CALL(6); break;
      case 5: HT_CALL(5); break;
      case 4: HT_CALL(4); break;
      case 3: HT_CALL(3); break;
      case 2: HT_CALL(2); break;
      default: HT_CALL(1); break;
    }
#undef HT_CALL
  }
  for(int r = wv; r < rows; r += 4)
    for(int k = K + lane; k < maxnhar; k += WAVE) {
      ampl[(size_t)(g0 + r) * maxnhar + k] = 0; phse[(size_t)(g0 + r) * maxnhar + k] = 0;
    }
  if(tid < rows) nhar_out[g0 + tid] = K;
}

#define HS_WPE 3                                   // <= 168 VGPRs, no spills: 3 wavefronts / SIMD hide the operand loads
// one frame by one wavefront
DEV void harm_frame(int g, int u, int i, int lane,
  const float* __restrict__ x, const int* __restrict__ x_off, const int* __restrict__ nx,
  const float* __restrict__ f0, float thop, float fs, float rel_winsize, int maxnhar,
  int* __restrict__ nhar_out, float* __restrict__ ampl, float* __restrict__ phse) {
  const float f = f0[g];
  float* arow = ampl + (size_t)g * maxnhar;
  float* prow = phse + (size_t)g * maxnhar;
  if(!(f > 0)) {
    if(lane == 0) nhar_out[g] = 0;
    for(int k = lane; k < maxnhar; k += WAVE) { arow[k] = 0; prow[k] = 0; }
    return;
  }
  const int n = lp::hwin(f, fs, rel_winsize);
  const int c = lp::center(i, thop, fs);
  const int K = lp::nhar(f, fs, maxnhar);
  // Row a covers the L samples tau in [rho_a - L/2, rho_a + L/2), rho_a = L (a - 8) + L/2, tau = t - n/2.
  // About its centre the row splits into an even part E[k] = xw[rho + k] + xw[rho - k] and an odd
  // part O[k] = xw[rho + k] - xw[rho - k], k = 0 .. L/2 (E[0] = xw[rho]; k = L/2 holds only
  // tau = rho - L/2), and
  //   S[a][h] = sum_b xw[rho_a + b] e^{-j th_h b} = sum_k E[k] cos(th_h k) - j sum_k O[k] sin(th_h k):
  // half the columns, i.e. half the MFMAs and half the twiddle generation of the plain product.
  // Lane (row = lane & 15, q = lane >> 4) is exactly the MFMA A-operand owner of (row, k = q + 4 ks),
  // so E and O are formed in registers from global memory: no LDS, no barrier.
  const int L = ((n + HM_ROWS - 1) / HM_ROWS + 7) & ~7;   // samples per row, multiple of 8
  const int KC = (L / 2 + 4) & ~3;                        // columns k = 0 .. L/2, padded to a multiple of 4
  const int half = n / 2;
  const int col = lane & 15, q = lane >> 4;
  const double turn1 = (double)f / (double)fs;      // cycles per sample of the fundamental
  HarmRow R;
  R.n = n; R.L = L;
  R.t0 = L * (col - HM_ROWS / 2) + L / 2 + half;
  R.org = c - half;
  R.lo = max(R.org, 0);
  R.rng = buf_range(x + x_off[u], R.lo, min(R.org + n, nx[u]));
  {
    const double inv = 1.0 / (double)(n > 1 ? n - 1 : 1);
    cs_turns((double)R.t0 * inv, & R.ca, & R.sa);
    cs_turns(4.0 * inv, & R.stc, & R.sts);
  }
  float wsum = 0.0f;
  for(int h0 = 0; h0 < K; h0 += 16 * HM_TILES) {
    const int ntile = min(HM_TILES, (K - h0 + 15) / 16);
    float Pr[HM_TILES + 1], Pi[HM_TILES + 1];
#pragma unroll
    for(int tt = 0; tt <= HM_TILES; tt ++) { Pr[tt] = 0; Pi[tt] = 0; }
    // the MFMA loop is instantiated per tile count so that it contains no branches
    // (a guarded MFMA makes the compiler shuttle every accumulator through VGPRs each step)
    switch(ntile) {
      case 7: wsum = harm_block<7>(R, KC, turn1, h0, col, q, Pr, Pi); break;
      case 6: wsum = harm_block<6>(R, KC, turn1, h0, col, q, Pr, Pi); break;
      case 5: wsum = harm_block<5>(R, KC, turn1, h0, col, q, Pr, Pi); break;
      case 4: wsum = harm_block<4>(R, KC, turn1, h0, col, q, Pr, Pi); break;
      case 3: wsum = harm_block<3>(R, KC, turn1, h0, col, q, Pr, Pi); break;
      case 2: wsum = harm_block<2>(R, KC, turn1, h0, col, q, Pr, Pi); break;
      default: wsum = harm_block<1>(R, KC, turn1, h0, col, q, Pr, Pi); break;
    }
    const float scale = 2.0f / wave_sum(wsum);       // 2 / sum of the window (every lane adds its slots)
    // every lane group now holds all tiles; group q finishes tiles q and q + 4
#pragma unroll
    for(int jj = 0; jj < 2; jj ++) {
      const float pr = q == 0 ? Pr[4 * jj] : q == 1 ? Pr[4 * jj + 1] : q == 2 ? Pr[4 * jj + 2] : Pr[4 * jj + 3];
      const float pi = q == 0 ? Pi[4 * jj] : q == 1 ? Pi[4 * jj + 1] : q == 2 ? Pi[4 * jj + 2] : Pi[4 * jj + 3];
      const int tt = 4 * jj + q;
      const int h = h0 + 16 * tt + col + 1;
      if(tt < ntile && h <= K) {
        arow[h - 1] = sqrtf(pr * pr + pi * pi) * scale;
        prow[h - 1] = atan2f(pi, pr);
      }
    }
  }
  for(int k = K + lane; k < maxnhar; k += WAVE) { arow[k] = 0; prow[k] = 0; }
  if(lane == 0) nhar_out[g] = K;
}

// every frame its own wavefront (no tile kernel in front: llsm_gpu_shared_f0_tiles(0))
__global__ __launch_bounds__(WAVE, HS_WPE) void k_harm_speech(
  const float* __restrict__ x, const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off,
  const float* __restrict__ f0, float thop, float fs, float rel_winsize, int maxnhar,
  int* __restrict__ nhar_out, float* __restrict__ ampl, float* __restrict__ phse) {
  const int g = xcd_frame(blockIdx.x, gridDim.x), lane = threadIdx.x;
  int u, i; frame_owner(frm_utt, frm_off, g, & u, & i);
  harm_frame(g, u, i, lane, x, x_off, nx, f0, thop, fs, rel_winsize, maxnhar, nhar_out, ampl, phse);
}
// The frames k_harm_speech_tile leaves: one workgroup of four wavefronts per 16-frame block, wavefront w takes the
// block's frames outside its tile in turn (w, w + 4, ...).  A block whose frames all sit in the tile costs one
// workgroup that exits at once -- a sixteenth of the launches a grid over frames would spend on it.
__global__ __launch_bounds__(4 * WAVE, HS_WPE) void k_harm_speech_rest(
  const int2* __restrict__ hblocks, int kcap,
  const float* __restrict__ x, const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off,
  const float* __restrict__ f0, float thop, float fs, float rel_winsize, int maxnhar,
  int* __restrict__ nhar_out, float* __restrict__ ampl, float* __restrict__ phse) {
  const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int2 blk = hblocks[xcd_frame(blockIdx.x, gridDim.x)];
  int s = 0, rows = 0; float ft;
  if(! harm_tile_of(f0, blk.x, blk.y, lane, kcap, fs, rel_winsize, maxnhar, & s, & rows, & ft)) { s = 0; rows = 0; }
  const int u = frm_utt[blk.x], i0 = blk.x - frm_off[u];
  for(int m = wv; m < blk.y - rows; m += 4) {
    const int li = m < s ? m : m + rows;             // m-th frame of the block outside [s, s + rows)
    harm_frame(blk.x + li, u, i0 + li, lane, x, x_off, nx, f0, thop, fs, rel_winsize, maxnhar, nhar_out, ampl, phse);
  }
}

// =====================================================================
// K2  harmonic analysis of the squared sub-band signals + short-time mean
// replaces the per-channel body of llsm_analyze_noise_envelope,
// layer0.c:433-458 (llsm_harmonic_analysis with maxnhar_e harmonics,
// llsm_compute_dc dsputils.c:117-124).  One wavefront per frame; here the
// 64 lanes split the WINDOW (only <= 8 harmonics are wanted) and the
// per-harmonic sums are reduced with the shuffle butterfly.
// =====================================================================
#ifndef HE_DC2
#define HE_DC2 1                                    // the short-time mean adds whole pairs (see the loop)
#endif
#ifndef HE_MID2
#define HE_MID2 1                                   // the centre sample of an odd window is zeroed at its mirror load, not selected per channel
#endif
#if HE_DC2 && ! HE_MID2
#error "HE_DC2 needs HE_MID2 (the pair sum must hold the centre sample once)"
#endif
#ifndef HE_CHEB
#define HE_CHEB 0                                   // 1: harmonic phasors by the three-term recurrence c_k = 2 c_1 c_(k-1) - c_(k-2): 0.01 ms faster and, fed
#endif                                              // with CARRIED phasors, 7e-4 rad on an envelope phase (seed 904505; rotation: 3e-5) against a bound of 1e-3: off
#ifndef HE_CARRY
#define HE_CARRY 1                                  // 1: phasors seeded once per frame and rotated from trip to trip (0: re-seeded every 4 pairs)
#endif
#ifndef HE_WPE
#define HE_WPE 5                                   // <= 102 VGPRs: the <4, 4> form holds 99, no spills (0.577 -> 0.549 ms against 4; 3: 0.584; a budget of 85 spills 71 registers: 2.9x slower)
#endif
template <int NCH, int ME>
__global__ __launch_bounds__(WAVE, (NCH * ME <= 16 ? HE_WPE : 1)) void k_harm_env(
  const float* __restrict__ ce, size_t ce_stride,       // channel c at ce + c*ce_stride
  const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off,
  const float* __restrict__ f0, float thop, float fs, float rel_winsize,
  int nch, int me, int* __restrict__ nhar_e_out, float* __restrict__ edc,
  float* __restrict__ eamp, float* __restrict__ ephs) {
  const int g = xcd_frame(blockIdx.x, gridDim.x), lane = threadIdx.x;
  int u, i; frame_owner(frm_utt, frm_off, g, & u, & i);
  const float f = f0[g];
  const int c0 = lp::center(i, thop, fs);
  const int nxu = nx[u];
  const size_t xo = (size_t)x_off[u];
  // ---- short-time mean (every frame): llsm_compute_dc, window of ndc samples about the centre.
  // For voiced frames that window lies inside the harmonic-analysis window (2 vs rel_winsize
  // periods), so its sums ride along the main loop below; otherwise they get their own pass.
  const int ndc = lp::dcwin(f > 0 ? f : 0.0f, thop, fs);
  const int bdc = c0 - ndc / 2;
  const int n_h = f > 0 ? lp::hwin(f, fs, rel_winsize) : 0;
  const bool dc_inside = f > 0 && bdc >= c0 - n_h / 2 && bdc + ndc <= c0 - n_h / 2 + n_h;
  float dacc[NCH];
#pragma unroll
  for(int c = 0; c < NCH; c ++) dacc[c] = 0;
  if(! dc_inside) {
#pragma unroll 4
    for(int j = lane; j < ndc; j += WAVE) {
      int idx = bdc + j;
      if(idx >= 0 && idx < nxu) {
#pragma unroll
        for(int c = 0; c < NCH; c ++)
          if(c < nch) dacc[c] += ce[(size_t)c * ce_stride + xo + idx];
      }
    }
#pragma unroll
    for(int c = 0; c < NCH; c ++) {
      float s = wave_sum(dacc[c]);
      if(lane == 0 && c < nch) edc[(size_t)g * nch + c] = s / (float)ndc;
    }
  }
  float* arow = eamp + (size_t)g * nch * me;
  float* prow = ephs + (size_t)g * nch * me;
  if(!(f > 0)) {
    if(lane == 0) nhar_e_out[g] = 0;
    for(int k = lane; k < nch * me; k += WAVE) { arow[k] = 0; prow[k] = 0; }
    return;
  }
  const int n = lp::hwin(f, fs, rel_winsize);
  const int K = lp::nhar(f, fs, me);
  const int half = n / 2, base = c0 - half;
  const double turn1 = (double)f / (double)fs;
  float are[NCH][ME], aim[NCH][ME];
#pragma unroll
  for(int c = 0; c < NCH; c ++)
#pragma unroll
    for(int k = 0; k < ME; k ++) { are[c][k] = 0; aim[c][k] = 0; }
  float wsum = 0;
  // The Blackman window is symmetric about (n - 1) / 2, so samples t- = p and t+ = n - 1 - p
  // share one window value and, with the phase referred to that centre (tau' = t - (n-1)/2, put
  // back onto n/2 by one rotation per harmonic at the end), one phasor:
  //   w x+ e^{-j th tau'} + w x- e^{+j th tau'} = E cos(th tau') - j O sin(th tau'),
  //   E = w (x+ + x-), O = w (x+ - x-):  half the multiply-adds and phasor work of the plain sum.
  // Four pairs per lane and trip (p = p0 + 64 q): their 2 x NCH x 4 loads are issued together;
  // window phase and phasor are seeded per trip from float64-reduced phases and rotated by 64.
  const double inv_n1 = 1.0 / (double)(n > 1 ? n - 1 : 1);
  const int npair = (n + 1) / 2;
  float wstc, wsts, zstc, zsts;
  cs_turns((double)WAVE * inv_n1, & wstc, & wsts);
  cs_turns(turn1 * (double)WAVE, & zstc, & zsts);
  // per-channel readable range: window [base, base + n) inside the signal [0, nxu), zero elsewhere
  const int rlo = max(base, 0), rhi = min(base + n, nxu);
  buf_t rng[NCH];
#pragma unroll
  for(int c = 0; c < NCH; c ++)
    rng[c] = buf_range(ce + (size_t)min(c, nch - 1) * ce_stride + xo, rlo, c < nch ? rhi : rlo);
  // short-time mean riding along (dc_inside): sample base + pp lies in [bdc, bdc + ndc) iff pp >= dcA, its mirror image
  // base + n - 1 - pp iff pp >= dcB (the other two bounds hold for every pair: the mean's window is centred inside the
  // analysis window) -- a pair with both inside adds the sum the even part needs anyway, the <= 1 pair index between the
  // two thresholds takes a rare branch.  The centre sample of an odd window (pp = (n - 1) / 2, its own mirror image) is
  // loaded once: the mirror load gets an offset outside the range (0), and its odd part meets sin(0) = 0.
  const int dcA = bdc - base, dcB = base + n - bdc - ndc;
  const int dcBoth = dc_inside ? max(dcA, dcB) : INT_MAX, dcOne = dc_inside ? min(dcA, dcB) : INT_MAX;
#if HE_CARRY
  // window phase and phasor of pair p0 = lane from float64-reduced phases; every later pair of the lane (64 further each) by
  // rotation -- at most ceil(npair / 64) - 1 steps (a dozen at 4 periods of 100 Hz), 6e-8 each
  float wc, wsn, z1c, z1s;
  cs_turns((double)lane * inv_n1, & wc, & wsn);
  cs_turns(turn1 * 0.5 * (double)(n - 1 - 2 * lane), & z1c, & z1s);
#endif
  for(int p0 = lane; p0 < npair; p0 += WAVE * 4) {
    float vm[4][NCH], vp[4][NCH];
#pragma unroll
    for(int q = 0; q < 4; q ++) {
      const int pp = p0 + q * WAVE, im_ = base + pp, ip_ = base + n - 1 - pp;   // pp >= npair: not used below
#if HE_MID2
      const int op_ = 2 * pp == n - 1 ? -1 : ip_ - rlo;
#else
      const int op_ = ip_ - rlo;
#endif
#pragma unroll
      for(int c = 0; c < NCH; c ++) {
        vm[q][c] = ld_range(rng[c], im_ - rlo);
        vp[q][c] = ld_range(rng[c], op_);
      }
    }
#if ! HE_CARRY
    float wc, wsn, z1c, z1s;
    cs_turns((double)p0 * inv_n1, & wc, & wsn);
    cs_turns(turn1 * 0.5 * (double)(n - 1 - 2 * p0), & z1c, & z1s);   // th tau' of the pair, turns
#endif
#pragma unroll
    for(int q = 0; q < 4; q ++) {
      const int pp = p0 + q * WAVE;
      if(pp < npair) {
        const bool mid = 2 * pp == n - 1;            // odd n: the centre sample pairs with itself
        float ev[NCH], on[NCH];
#if HE_DC2
#pragma unroll
        for(int c = 0; c < NCH; c ++) ev[c] = vp[q][c] + vm[q][c];
        {
          const float fb = pp >= dcBoth ? 1.0f : 0.0f;
#pragma unroll
          for(int c = 0; c < NCH; c ++) dacc[c] = fmaf(ev[c], fb, dacc[c]);
          if(pp >= dcOne && pp < dcBoth) {
#pragma unroll
            for(int c = 0; c < NCH; c ++) dacc[c] += pp >= dcA ? vm[q][c] : (HE_MID2 || ! mid ? vp[q][c] : 0.0f);
          }
        }
#else
        if(dc_inside) {                              // short-time mean rides along (see above)
          const int im_ = base + pp, ip_ = base + n - 1 - pp;
          const bool dm = im_ >= bdc && im_ < bdc + ndc, dp = ! mid && ip_ >= bdc && ip_ < bdc + ndc;
#pragma unroll
          for(int c = 0; c < NCH; c ++) dacc[c] += (dm ? vm[q][c] : 0.0f) + (dp ? vp[q][c] : 0.0f);
        }
#endif
        // 0.42 - 0.5 cos a + 0.08 cos 2a with cos 2a = 2 cos^2 a - 1
        const float w = n > 1 ? fmaf(wc, fmaf(wc, 0.16f, -0.5f), 0.34f) : 1.0f;
        wsum += mid ? w : 2.0f * w;
#pragma unroll
        for(int c = 0; c < NCH; c ++) {
#if HE_MID2
          ev[c] = (HE_DC2 ? ev[c] : vp[q][c] + vm[q][c]) * w;
          on[c] = (vm[q][c] - vp[q][c]) * w;         // -O (centre sample: meets sin 0 = 0 below)
#else
          ev[c] = mid ? vm[q][c] * w : (HE_DC2 ? ev[c] : vp[q][c] + vm[q][c]) * w;
          on[c] = mid ? 0.0f : (vm[q][c] - vp[q][c]) * w;      // -O
#endif
        }
#if HE_CHEB
        // cos, sin(k th tau'), k = 1 .. ME, by the three-term recurrence c_k = 2 c_1 c_(k-1) - c_(k-2) (likewise s_k): one
        // fused multiply-add per value instead of a complex rotation; ME <= 8 steps from exact seeds
        const float tc = z1c + z1c;
        float zr = z1c, zi = z1s, zrp = 1.0f, zip = 0.0f;   // (c_k, s_k) and (c_(k-1), s_(k-1)); k = 0: (1, 0)
#pragma unroll
        for(int k = 0; k < ME; k ++) {
#pragma unroll
          for(int c = 0; c < NCH; c ++) {
            are[c][k] = fmaf(ev[c], zr, are[c][k]);
            aim[c][k] = fmaf(on[c], zi, aim[c][k]);
          }
          const float nr = fmaf(tc, zr, -zrp), ni = fmaf(tc, zi, -zip);
          zrp = zr; zip = zi; zr = nr; zi = ni;
        }
#else
        float zr = z1c, zi = z1s;                    // cos, sin(k th tau')
#pragma unroll
        for(int k = 0; k < ME; k ++) {
#pragma unroll
          for(int c = 0; c < NCH; c ++) {
            are[c][k] = fmaf(ev[c], zr, are[c][k]);
            aim[c][k] = fmaf(on[c], zi, aim[c][k]);
          }
          const float nr = zr * z1c - zi * z1s, ni = zr * z1s + zi * z1c;
          zr = nr; zi = ni;
        }
#endif
      }
      float t1 = wc * wstc - wsn * wsts, t2 = wc * wsts + wsn * wstc; wc = t1; wsn = t2;
      t1 = z1c * zstc + z1s * zsts; t2 = z1s * zstc - z1c * zsts; z1c = t1; z1s = t2;   // tau' -= 64
    }
  }
  // X_k = X'_k e^{-j th_k ((n-1)/2 - n/2)}: half a sample for even n, nothing for odd n
  const double back = -0.5 * turn1 * (double)(n - 1 - 2 * half);
  wsum = wave_sum(wsum);
  const float scale = 2.0f / wsum;
  if(dc_inside) {
#pragma unroll
    for(int c = 0; c < NCH; c ++) {
      float s = wave_sum(dacc[c]);
      if(lane == 0 && c < nch) edc[(size_t)g * nch + c] = s / (float)ndc;
    }
  }
  if constexpr (2 * NCH * ME <= WAVE) {
    // all NCH x ME complex sums at once; lane l ends up owning one (c, k, re|im) and one lane
    // per pair does ONE sqrt / atan2 instead of lane 0 doing NCH x ME of them
    float red[2 * NCH * ME];
#pragma unroll
    for(int c = 0; c < NCH; c ++)
#pragma unroll
      for(int k = 0; k < ME; k ++) { red[2 * (c * ME + k)] = are[c][k]; red[2 * (c * ME + k) + 1] = aim[c][k]; }
    wave_reduce_many<2 * NCH * ME>(red, lane);
    const int idx = wave_reduce_index<2 * NCH * ME>(lane);
    // the lowest bit of the value index follows lane bit LOWBIT: the other part of the complex
    // sum is one cross-lane read away
    constexpr int LOWBIT = WAVE / (2 * NCH * ME);
    const float mine = red[0];
    const float other = __shfl_xor(mine, LOWBIT, WAVE);
    const bool is_im = (idx & 1) != 0;
    const int pair = idx >> 1, c = pair / ME, k = pair % ME;
    float re = is_im ? other : mine, im = is_im ? mine : other;
    { float cr, sr; cs_turns(back * (double)(k + 1), & cr, & sr);
      const float t1 = re * cr - im * sr, t2 = re * sr + im * cr; re = t1; im = t2; }
    const bool writer = ! is_im && (lane & (LOWBIT - 1)) == 0;       // one lane per pair
    if(writer && c < nch && k < me) {
      const bool live = k < K;
      arow[c * me + k] = live ? sqrtf(re * re + im * im) * scale : 0.0f;
      prow[c * me + k] = live ? atan2f(im, re) : 0.0f;
    }
  } else {
#pragma unroll
    for(int c = 0; c < NCH; c ++)
#pragma unroll
      for(int k = 0; k < ME; k ++) {
        float re = wave_sum(are[c][k]), im = wave_sum(aim[c][k]);
        { float cr, sr; cs_turns(back * (double)(k + 1), & cr, & sr);
          const float t1 = re * cr - im * sr, t2 = re * sr + im * cr; re = t1; im = t2; }
        if(lane == 0 && c < nch && k < me) {
          bool live = k < K;
          arow[c * me + k] = live ? sqrtf(re * re + im * im) * scale : 0.0f;
          prow[c * me + k] = live ? atan2f(im, re) : 0.0f;
        }
      }
  }
  if(lane == 0) nhar_e_out[g] = K;
}


// =====================================================================
// K5  zero-phase Chebyshev band filters, wave-parallel block IIR in float64
// replaces chebyfilt / llsm_subband_energy (dsputils.c:51-70, 230-235) and the
// band-limiting of llsm_generate_bandlimited_noise (dsputils.c:389-390).
// filtfilt contract: odd extension by pad = min(15, n-1) samples,
// steady-state initial conditions scaled by the first sample of each pass,
// forward then backward transposed-direct-form-II passes.
//
// One wavefront per signal.  The (extended) signal is cut into tiles of
// 64 lanes x IIR_SEG samples; lane m runs the recursion over its own
// contiguous segment from a ZERO state (registers only), the 64 segment end
// states are combined by a Kogge-Stone scan over lanes with the precomputed
// powers of the state-transition matrix (A^SEG)^(2^d), and each lane adds the
// zero-input response of its true initial state through the table
// H[i] = e0^T A^i.  Sequential depth per pass: n/64 + 6 instead of n.
// All recursion arithmetic is float64 (the recursion is the precision-critical
// part of the envelope analysis; it is nowhere near the fp64 roof).
// =====================================================================
DEV double shfl_up_d(double v, int d) { return __shfl_up(v, d, WAVE); }

// rows of IIR_SEG + 4 floats: 16-byte aligned rows, and 16 lanes x ds_read_b128 at a stride of
// 28 (20) dwords cover the 64 banks exactly once
#define IIR_LDS_STRIDE (IIR_SEG % 8 == 4 ? IIR_SEG : IIR_SEG + 4)   // = 4 mod 8
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // float4 at any 4-byte boundary
typedef float f4a __attribute__((ext_vector_type(4)));
// global address space views (pointers read from a job table are generic: flat_load / flat_store, which
// also count against the LDS counter; these compile to global_load / global_store)
typedef const __attribute__((address_space(1))) float* gcfp;
typedef __attribute__((address_space(1))) float* gfp;
typedef const __attribute__((address_space(1))) f4u* gcf4p;
typedef __attribute__((address_space(1))) f4u* gf4p;
#define IIR_TILE (WAVE * IIR_SEG)

// LDS: the transposition buffer only (the block tables are scalar operands, see iir_pass).
struct IirLds {
  float seg[WAVE * IIR_LDS_STRIDE];                 // loads in (float32), results out (float32)
};

// extended-signal accessors of the two passes (the forward-pass output tmp is float32: it is
// an intermediate of the zero-phase pair and costs 4 instead of 8 bytes of HBM traffic per
// sample and pass; the recursion itself stays float64)
DEV float fwd_at(const float* __restrict__ x, int n, int pad, int ne, int t) {
  if(t >= ne) return 0.0f;
  if(t < pad) return 2.0f * x[0] - x[pad - t];
  if(t >= pad + n) return 2.0f * x[n - 1] - x[n - 2 - (t - pad - n)];
  return x[t - pad];
}
DEV float bwd_at(const float* __restrict__ tmp, int ne, int r) { return r < ne ? tmp[ne - 1 - r] : 0.0f; }

// One pass over `ne` samples.  FWD: reads the odd-extended input, writes tmp[t].
// !FWD: reads tmp reversed, writes the central n samples of the (re-reversed) result to dst.
// Global loads and stores are coalesced 16-byte accesses (lane l moves the 4 samples
// base + 4 (l + 64 r) ..) and are transposed through LDS with 16-byte LDS accesses so that every
// lane owns IIR_SEG consecutive samples; the loads of tile k+1 are issued before the recursion of
// tile k runs (software prefetch).  The first and last tile of a pass (odd extension, ragged end)
// take a scalar guarded path.
// One tile of one section, in registers: v[] (this lane's IIR_SEG consecutive samples, float64) is replaced by the
// section's output; c[] is the state carried from tile to tile.  The block tables are wave-uniform: they are read with
// SCALAR loads straight into SGPR operands of the float64 FMAs.  (As LDS broadcasts they were 104 of the 116
// ds_read_b128 of a tile: 1 KB of LDS return bandwidth each, which made the LDS pipe, not the VALU, the limit of this
// kernel.)  The pointer is made opaque at every use so that the loads stay where they are used instead of being
// hoisted out of the tile loop into hundreds of spilled SGPRs.
typedef const __attribute__((address_space(4))) FiltSectionD* SecK;   // constant address space: s_load
#define IIR_TAB_M(d) ({ asm volatile("" : "+s"(sp)); sp -> M[d]; })
#define IIR_TAB_H(i) ({ asm volatile("" : "+s"(sp)); sp -> H[i]; })
DEV void iir_tile(SecK sp, double (&v)[IIR_SEG], double (&c)[4], int lane) {
  const double b0 = sp -> b[0], b1 = sp -> b[1], b2 = sp -> b[2], b3 = sp -> b[3], b4 = sp -> b[4];
  const double a1 = sp -> a[1], a2 = sp -> a[2], a3 = sp -> a[3], a4 = sp -> a[4];
  // ---- response of this lane's segment from a zero state -- lane 0 from the carried state, so that the carry
  // needs no separate absorption step -- in transposed direct form II: 9 float64 operations per sample, the
  // end state (what the scan propagates) falls out of the recursion, and the dependent chain is two FMAs per
  // sample (y -> t0 -> y'), well inside the issue time of the nine.
  // The scalar loads of a table are issued one stage AHEAD of their use (M[0] before the recursion, M[d + 1]
  // before the arithmetic of stage d, the first rows of H before the last stage): a scalar load issued where it is
  // used stalls the wavefront for the scalar-cache latency thirteen times per tile.
  double mt[16];
  { const auto Mp = IIR_TAB_M(0);
#pragma unroll
    for(int k = 0; k < 16; k ++) mt[k] = Mp[k]; }
  double z0 = lane == 0 ? c[0] : 0.0, z1 = lane == 0 ? c[1] : 0.0, z2 = lane == 0 ? c[2] : 0.0, z3 = lane == 0 ? c[3] : 0.0;
#pragma unroll
  for(int i = 0; i < IIR_SEG; i ++) {
    const double xi = v[i];
    const double yi = fma(b0, xi, z0);
    z0 = fma(-a1, yi, fma(b1, xi, z1));
    z1 = fma(-a2, yi, fma(b2, xi, z2));
    z2 = fma(-a3, yi, fma(b3, xi, z3));
    z3 = fma(-a4, yi, b4 * xi);
    v[i] = yi;
  }
  // Kogge-Stone scan of end states: E[m] += (A^SEG)^(2^d) E[m - 2^d]   (branch-free: lanes below 2^d add zero)
  double ht[16];                                     // H rows 0 .. 3, fetched during the last stage
#pragma unroll
  for(int d = 0; d < 6; d ++) {
    const int off = 1 << d;
    double u0 = shfl_up_d(z0, off), u1 = shfl_up_d(z1, off);
    double u2 = shfl_up_d(z2, off), u3 = shfl_up_d(z3, off);
    const bool on = lane >= off;
    u0 = on ? u0 : 0.0; u1 = on ? u1 : 0.0; u2 = on ? u2 : 0.0; u3 = on ? u3 : 0.0;
    double m[16];
#pragma unroll
    for(int k = 0; k < 16; k ++) m[k] = mt[k];
    if(d < 5) {
      const auto Mp = IIR_TAB_M(d < 5 ? d + 1 : 5);
#pragma unroll
      for(int k = 0; k < 16; k ++) mt[k] = Mp[k];
    } else {
#pragma unroll
      for(int r = 0; r < 4; r ++) {
        const auto hp = IIR_TAB_H(r);
#pragma unroll
        for(int k = 0; k < 4; k ++) ht[4 * r + k] = hp[k];
      }
    }
    z0 = fma(m[0], u0, fma(m[1], u1, fma(m[2], u2, fma(m[3], u3, z0))));
    z1 = fma(m[4], u0, fma(m[5], u1, fma(m[6], u2, fma(m[7], u3, z1))));
    z2 = fma(m[8], u0, fma(m[9], u1, fma(m[10], u2, fma(m[11], u3, z2))));
    z3 = fma(m[12], u0, fma(m[13], u1, fma(m[14], u2, fma(m[15], u3, z3))));
  }
  // true initial state of this lane's segment = end state of the previous lane
  double s0 = shfl_up_d(z0, 1), s1 = shfl_up_d(z1, 1), s2 = shfl_up_d(z2, 1), s3 = shfl_up_d(z3, 1);
  if(lane == 0) { s0 = 0.0; s1 = 0.0; s2 = 0.0; s3 = 0.0; }   // lane 0 ran from the true state already
  c[0] = __shfl(z0, WAVE - 1, WAVE); c[1] = __shfl(z1, WAVE - 1, WAVE);
  c[2] = __shfl(z2, WAVE - 1, WAVE); c[3] = __shfl(z3, WAVE - 1, WAVE);
  // ---- zero-input correction
#pragma unroll
  for(int i = 0; i < IIR_SEG; i += 4) {
    double h[16];
#pragma unroll
    for(int k = 0; k < 16; k ++) h[k] = ht[k];
    if(i + 4 < IIR_SEG) {                            // rows i + 4 .. i + 7 for the next round
#pragma unroll
      for(int r = 0; r < 4; r ++) {
        const auto hp = IIR_TAB_H(i + 4 + r);
#pragma unroll
        for(int k = 0; k < 4; k ++) ht[4 * r + k] = hp[k];
      }
    }
#pragma unroll
    for(int k = 0; k < 4; k ++)
      v[i + k] = fma(h[4 * k], s0, fma(h[4 * k + 1], s1, fma(h[4 * k + 2], s2, fma(h[4 * k + 3], s3, v[i + k]))));
  }
}

// One pass over `ne` samples through NSEC sections (1: one direction of a filtfilt; 2: the same direction of the two
// sections of a band-pass, back to back in registers -- see k_filtfilt).  FWD: reads the odd-extended input, writes tmp[t].
// !FWD: reads tmp reversed, writes the samples [wlo, whi) of the central n samples of the (re-reversed) result to dst.
// Global loads and stores are coalesced 16-byte accesses (lane l moves the 4 samples
// base + 4 (l + 64 r) ..) and are transposed through LDS with 16-byte LDS accesses so that every
// lane owns IIR_SEG consecutive samples; the loads of tile k+1 are issued before the recursion of
// tile k runs (software prefetch).  The first and last tile of a pass (odd extension, ragged end)
// take a scalar guarded path.
template <bool FWD, int NSEC>
DEV void iir_pass(const FiltSectionD* __restrict__ secA, const FiltSectionD* __restrict__ secB, IirLds* L,
  const float* __restrict__ src, int ne, int n, int pad, float* __restrict__ tmp, float* __restrict__ dst,
  bool square, int wlo, int whi, int lane, const float* gen_src = nullptr) {
  SecK spA = (SecK)(unsigned long long)secA, spB = (SecK)(unsigned long long)secB;
  const double init = (double)(FWD ? fwd_at(src, n, pad, ne, 0) : bwd_at(tmp, ne, 0));
  double cA[4], cB[4];                               // carried states: steady state of a constant input `init`
#pragma unroll
  for(int k = 0; k < 4; k ++) cA[k] = secA -> zi[k] * init;
  if(NSEC == 2) {
    // the second section starts from the steady state of the FIRST sample the first one puts out (lfilter_zi x y[0])
    const double y0 = fma(secA -> b[0], init, cA[0]);
#pragma unroll
    for(int k = 0; k < 4; k ++) cB[k] = secB -> zi[k] * y0;
  }
  float nxt[IIR_SEG];
  bool have_nxt = false;                             // nxt holds the tile about to be processed
  __syncthreads();
  for(int base = 0; base < ne; base += IIR_TILE) {
    // ---- tile -> LDS, transposed.  Interior tiles were prefetched into registers; the
    // first / last tiles (odd extension, ragged end) go through the generic accessor.
    if(have_nxt) {
#pragma unroll
      for(int r = 0; r < IIR_SEG / 4; r ++) {          // chunk c = 4 samples of one lane row
        const int c = r * WAVE + lane;
        *(f4a*)& L -> seg[(c / (IIR_SEG / 4)) * IIR_LDS_STRIDE + 4 * (c % (IIR_SEG / 4))] =
          f4a{nxt[4 * r], nxt[4 * r + 1], nxt[4 * r + 2], nxt[4 * r + 3]};
      }
    } else {
      // first / last tile: all loads of the tile in one branch-free batch through a range-checked
      // descriptor (zero outside the signal), then the <= 15 odd-extension samples are patched
      const buf_t rng = FWD ? buf_range(src, 0, n) : buf_range(tmp, 0, ne);
#pragma unroll
      for(int r = 0; r < IIR_SEG; r ++) {
        const int t = base + r * WAVE + lane;
        nxt[r] = ld_range(rng, FWD ? t - pad : ne - 1 - t);
      }
#pragma unroll
      for(int r = 0; r < IIR_SEG; r ++) {
        const int e = r * WAVE + lane, t = base + e;
        float v = nxt[r];
        if(FWD && (t < pad || (t >= pad + n && t < ne))) v = fwd_at(src, n, pad, ne, t);
        L -> seg[(e / IIR_SEG) * IIR_LDS_STRIDE + (e % IIR_SEG)] = v;
      }
    }
    __syncthreads();
    // ---- prefetch the next tile if it is interior: plain coalesced loads that fly while
    // the recursion below runs
    {
      const int nb = base + IIR_TILE;
      have_nxt = FWD ? (nb >= pad && nb + IIR_TILE <= pad + n) : (nb + IIR_TILE <= ne);
      if(have_nxt) {
        // lane l, round r: the 4 samples t = nb + 4 (64 r + l) .. + 3 (reversed in memory when !FWD)
        const gcfp p = (gcfp)(unsigned long long)(FWD ? src + (nb - pad) + 4 * lane : tmp + (ne - 1 - nb - 3) - 4 * lane);
#pragma unroll
        for(int r = 0; r < IIR_SEG / 4; r ++) {
          { f4u qg;                                    // (empty hook in the product build: see the top of this file)
            if(IIR_EXP_GEN(FWD, square, src, gen_src, nb - pad + 4 * lane + 4 * WAVE * r, qg)) {
              nxt[4 * r] = qg.x; nxt[4 * r + 1] = qg.y; nxt[4 * r + 2] = qg.z; nxt[4 * r + 3] = qg.w;
              continue;
            } }
          const f4u q = *(gcf4p)(FWD ? p + 4 * WAVE * r : p - 4 * WAVE * r);
          nxt[4 * r] = FWD ? q.x : q.w; nxt[4 * r + 1] = FWD ? q.y : q.z;
          nxt[4 * r + 2] = FWD ? q.z : q.y; nxt[4 * r + 3] = FWD ? q.w : q.x;
        }
      }
    }
    double v[IIR_SEG];
#pragma unroll
    for(int i = 0; i < IIR_SEG; i += 4) {
      const f4a q = *(const f4a*)& L -> seg[lane * IIR_LDS_STRIDE + i];
      v[i] = (double)q.x; v[i + 1] = (double)q.y; v[i + 2] = (double)q.z; v[i + 3] = (double)q.w;
    }
    iir_tile(spA, v, cA, lane);
    if(NSEC == 2) iir_tile(spB, v, cB, lane);          // the first section's output never leaves float64 registers
    // ---- result back into LDS (own row: no hazard with other lanes)
#pragma unroll
    for(int i = 0; i < IIR_SEG; i += 4)
      *(f4a*)& L -> seg[lane * IIR_LDS_STRIDE + i] = f4a{(float)v[i], (float)v[i + 1], (float)v[i + 2], (float)v[i + 3]};
    __syncthreads();
    // ---- coalesced store
    // tiles that lie wholly inside the stored range: 16-byte stores, no guards
    if(FWD ? (base + IIR_TILE <= ne) : (base >= pad + (n - whi) && base + IIR_TILE <= pad + n - wlo)) {
      const gfp q = (gfp)(unsigned long long)(FWD ? tmp + base + 4 * lane : dst + (ne - 1 - pad - base - 3) - 4 * lane);
#pragma unroll
      for(int r = 0; r < IIR_SEG / 4; r ++) {
        const int c = r * WAVE + lane;
        f4a y = *(const f4a*)& L -> seg[(c / (IIR_SEG / 4)) * IIR_LDS_STRIDE + 4 * (c % (IIR_SEG / 4))];
        if(! FWD && square) y = y * y;
        if(FWD) *(gf4p)(q + 4 * WAVE * r) = f4u{y.x, y.y, y.z, y.w};
        else *(gf4p)(q - 4 * WAVE * r) = f4u{y.w, y.z, y.y, y.x};
      }
      __syncthreads();
      continue;
    }
#pragma unroll
    for(int r = 0; r < IIR_SEG; r ++) {
      const int e = r * WAVE + lane;
      const float y = L -> seg[(e / IIR_SEG) * IIR_LDS_STRIDE + (e % IIR_SEG)];
      const int t = base + e;
      if(FWD) { if(t < ne) tmp[t] = y; }
      else {
        // reversed index t  <->  extended index ne - 1 - t  <->  dst[.. - pad]
        const int te = ne - 1 - t;
        if(te >= pad + wlo && te < pad + whi) dst[te - pad] = square ? y * y : y;
      }
    }
    __syncthreads();
  }
}

#ifndef IIR_WPE
#define IIR_WPE 3                                  // wavefronts per SIMD the register budget is cut for
#endif
__global__ __launch_bounds__(WAVE, IIR_WPE) void k_filtfilt(const FiltJob* __restrict__ jobs, int njobs,
  const FiltSectionD* __restrict__ sections) {
  const int j = blockIdx.x, lane = threadIdx.x;
  if(j >= njobs) return;
  FiltJob job = jobs[j];
  if(job.n <= 1) return;
  IIR_EXP_JOB(job, jobs, j);                          // (empty hook in the product build: see the top of this file)
  IirLds* L = (IirLds*)g_lds;
  const int n = job.n, pad = min(job.pad > 0 ? job.pad : 15, n - 1), ne = n + 2 * pad;
  const int wlo = job.whi > job.wlo ? job.wlo : 0, whi = job.whi > job.wlo ? job.whi : n;   // (0, 0): the whole signal
  if(job.fused && job.sec1 >= 0) {
    // Band-pass with both sections per pass: F_hp F_lp, then B_hp B_lp, instead of chebyfilt's F_hp B_hp F_lp B_lp
    // (dsputils.c:54-59).  The four operators commute, so away from the ends the result is the same to rounding, with
    // half the passes over the signal (4 plane transfers instead of 8) and no float32 intermediate between the
    // sections.  Near the ends the two orders differ by the second filtfilt's own padding and initial conditions -- a
    // transient that dies with the slowest pole; the host gives this job only the interior [wlo, whi) to write and
    // covers the ends with two short jobs in the reference's order (engine.cpp build_jobs).
    const FiltSectionD *sa = sections + job.sec0, *sb = sections + job.sec1;
    iir_pass<true, 2>(sa, sb, L, job.src, ne, n, pad, job.tmp, job.dst, false, 0, n, lane, job.square ? nullptr : job.src);
    iir_pass<false, 2>(sa, sb, L, job.src, ne, n, pad, job.tmp, job.dst, job.square != 0, wlo, whi, lane);
    return;
  }
  const int nsec = job.sec1 < 0 ? 1 : 2;
  for(int si = 0; si < nsec; si ++) {                // chebyfilt: high-pass then low-pass (dsputils.c:54-59)
    const FiltSectionD* sec = sections + (si == 0 ? job.sec0 : job.sec1);
    const float* src = si == 0 ? job.src : job.mid;
    float* dst = (si == nsec - 1) ? job.dst : job.mid;
    const bool last = si == nsec - 1;
    const bool square = last && job.square != 0;
    iir_pass<true, 1>(sec, sec, L, src, ne, n, pad, job.tmp, dst, square, 0, n, lane, job.square ? nullptr : job.src);
    iir_pass<false, 1>(sec, sec, L, src, ne, n, pad, job.tmp, dst, square, last ? wlo : 0, last ? whi : n, lane);
  }
}

// =====================================================================
// Wavefront FFT in LDS, IN PLACE (one N-point float2 buffer), mixed radix-4 /
// radix-2.  fft_dif: natural order in, BIT-REVERSED order out, forward,
// unnormalised.  ifft_dit: bit-reversed in, natural out, inverse, unscaled
// (callers fold the 1/N in).  A forward/inverse pair therefore needs no
// reordering pass, and consumers of a spectrum index it through brevN().
// Twiddles come from an LDS table tw[m] = e^{-2 pi i m / NT}, m < NT/2; an
// M-point transform uses it with stride NT/M.  (fft = unnormalised e^{-j},
// ifft = 1/N: the contract the reference needs from ciglet, SURVEY Appendix A.)
//
// All FFT kernels transform TWO real frames per complex FFT (z = a + j b):
// the spectra are separated with A[k] = (Z[k] + conj Z[M-k]) / 2,
// B[k] = (Z[k] - conj Z[M-k]) / 2j, and Hermitian spectra are recombined as
// Ya + j Yb so that one inverse FFT returns both real frames.
// =====================================================================

// spectra of the two real frames packed in the bit-reversed spectrum Z (length M): k in [0, M/2]
DEV void unpack_pair(const float2* Z, int M, int logM, int k, float2* A, float2* B) {
  const float2 zk = Z[brevN(k, logM)], zn = Z[brevN((M - k) & (M - 1), logM)];
  *A = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
  *B = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
}

// =====================================================================
// K6  log-power spectral envelope per frame (feeds the Kalman process
// variance) -- replaces layer0.c:325-345: llsm_compute_spectrogram
// (dsputils.c:96-115, Hann window of 3 periods, nfft_spgm) + spec2env +
// "*2" + bin decimation to the PSD grid.  Per PAIR of frames, all in one LDS
// buffer: forward FFT of N (both real frames), log spectra written back in
// place, inverse FFT of N (both real cepstra), liftering folded onto the
// decimated output grid (only every fold-th envelope bin is wanted,
// layer0.c:341), forward FFT of N/fold.  Persistent: each wavefront walks
// frame pairs.  LDS: N float2 + N/2 float2 twiddles.
// =====================================================================
__global__ __launch_bounds__(WAVE) void k_spgm_env(
  const float* __restrict__ x, const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off,
  const float* __restrict__ f0, int nframes, float thop, float fs, int nwin_psd,
  int N, int logN, int nfft_psd, float norm_base,
  const float2* __restrict__ tw_glob, int tw_nmax, float* __restrict__ env_out,
  const int2* __restrict__ pairs, int npair) {
  const int lane = threadIdx.x;
  float2* X = (float2*)g_lds;
  float2* tw = X + N;
  load_twiddles(tw, tw_glob, N, tw_nmax, lane);
  const int nspec = nfft_psd / 2 + 1;
  // fold the third transform when the output grid is an integer decimation of the bins
  const int fold = (N >= nfft_psd) ? N / nfft_psd : 1;
  const int M3 = N / fold;
  int logM3 = 0; while((1 << logM3) < M3) logM3 ++;
  const float invN = 1.0f / (float)N;
  const int wgx = xcd_frame(blockIdx.x, gridDim.x);  // chunk index: neighbouring chunks share an XCD
  const int per = (npair + gridDim.x - 1) / gridDim.x;
  for(int p = wgx * per; p < min(npair, (wgx + 1) * per); p ++) {
    int gg[2]; pair_of(pairs, p, nframes, gg[0], gg[1]);
    float f0n[2], normalizer[2];
    // stage both frames: zero-phase placement (frame centre at index 0), time-aliased if ws > N
    const float* xsp[2]; int nxu2[2], cc[2], wsz[2];
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      const int g = gg[e];
      f0n[e] = 200.0f / fs; normalizer[e] = 0; xsp[e] = x; nxu2[e] = 0; cc[e] = 0; wsz[e] = 0;
      if(g >= nframes) continue;
      int u, i; frame_owner(frm_utt, frm_off, g, & u, & i);
      const float f = f0[g];
      wsz[e] = lp::spgmwin(f > 0 ? f : 0.0f, fs, nwin_psd);
      cc[e] = lp::center(i, thop, fs);
      xsp[e] = x + x_off[u]; nxu2[e] = nx[u];
      f0n[e] = (f > 0 ? f : 200.0f) / fs;
      normalizer[e] = norm_base / (float)wsz[e];
    }
    for(int p0 = lane; p0 < N; p0 += WAVE * 8) {     // 16 independent loads in flight per lane
      float va[8], vb[8];
#pragma unroll
      for(int q8 = 0; q8 < 8; q8 ++) {
        const int pos = p0 + q8 * WAVE;
        const int ja = (pos + wsz[0] / 2) & (N - 1), jb = (pos + wsz[1] / 2) & (N - 1);
        const int ia = cc[0] - wsz[0] / 2 + ja, ib = cc[1] - wsz[1] / 2 + jb;
        va[q8] = (pos < N && ja < wsz[0] && ia >= 0 && ia < nxu2[0]) ? xsp[0][ia] : 0.0f;
        vb[q8] = (pos < N && jb < wsz[1] && ib >= 0 && ib < nxu2[1]) ? xsp[1][ib] : 0.0f;
      }
#pragma unroll
      for(int q8 = 0; q8 < 8; q8 ++) {
        const int pos = p0 + q8 * WAVE;
        if(pos < N) {
          const int ja = (pos + wsz[0] / 2) & (N - 1), jb = (pos + wsz[1] / 2) & (N - 1);
          float a = ja < wsz[0] ? va[q8] * hann_at(ja, wsz[0]) : 0.0f;
          float b = jb < wsz[1] ? vb[q8] * hann_at(jb, wsz[1]) : 0.0f;
          // window longer than the FFT: add the time-aliased remainder (rare: F0 < 3 fs / N)
          for(int j = ja + N; j < wsz[0]; j += N) {
            const int idx = cc[0] - wsz[0] / 2 + j;
            if(idx >= 0 && idx < nxu2[0]) a += xsp[0][idx] * hann_at(j, wsz[0]);
          }
          for(int j = jb + N; j < wsz[1]; j += N) {
            const int idx = cc[1] - wsz[1] / 2 + j;
            if(idx >= 0 && idx < nxu2[1]) b += xsp[1][idx] * hann_at(j, wsz[1]);
          }
          X[pos] = make_float2(a, b);
        }
      }
    }
    __syncthreads();
    fft_dif(X, tw, 1, N, logN, lane);
    // log magnitude of both frames, written back over the (bit-reversed) bin pair k, N-k
    for(int k = lane; k <= N / 2; k += WAVE) {
      float2 A, B; unpack_pair(X, N, logN, k, & A, & B);
      const float La = __logf(__builtin_amdgcn_sqrtf(A.x * A.x + A.y * A.y) * normalizer[0] + 1e-10f);
      const float Lb = __logf(__builtin_amdgcn_sqrtf(B.x * B.x + B.y * B.y) * normalizer[1] + 1e-10f);
      X[brevN(k, logN)] = make_float2(La, Lb);
      X[brevN((N - k) & (N - 1), logN)] = make_float2(La, Lb);
    }
    __syncthreads();
    ifft_dit(X, tw, 1, N, logN, lane);              // both real cepstra (x N)
    // lifter with sinc(q f0) (both frames), folded to M3 points; thread q owns bins q + m M3.
    // sin(pi f0n q') for q' = lane + 64 i by phasor rotation (seeded from reduced phases).
    {
      float rca, rsa, rcb, rsb;                      // rotation by 64 quefrency bins
      cs_turns(0.5 * (double)f0n[0] * (double)WAVE, & rca, & rsa);
      cs_turns(0.5 * (double)f0n[1] * (double)WAVE, & rcb, & rsb);
      for(int jf = 0; jf < fold; jf ++) {            // bins m = q + jf M3 fold onto q
        // quefrency of bin m: m (m <= N/2) or N - m; sin(pi f0n qq) by rotation over q:
        // qq = off + sgn q with (off, sgn) = (jf M3, +1) or (N - jf M3, -1)
        const bool up = jf * M3 + (M3 - 1) <= N / 2;
        const bool mixed = ! up && jf * M3 <= N / 2;  // the fold straddles N/2: no recurrence
        const int off = up ? jf * M3 : N - jf * M3;
        const float sgn = up ? 1.0f : -1.0f;
        float ca, sa, cb2, sb2;
        cs_turns(0.5 * (double)f0n[0] * (double)(off + (up ? lane : -lane)), & ca, & sa);
        cs_turns(0.5 * (double)f0n[1] * (double)(off + (up ? lane : -lane)), & cb2, & sb2);
        for(int q = lane; q < M3; q += WAVE) {
          const int m = q + jf * M3;
          const int qq = m <= N / 2 ? m : N - m;
          float la = invN, lb = invN;
          if(qq > 0) {
            const float sna = mixed ? sinpif((float)qq * f0n[0]) : sa;
            const float snb = mixed ? sinpif((float)qq * f0n[1]) : sb2;
            la = invN * sna * __builtin_amdgcn_rcpf(3.14159265358979f * (float)qq * f0n[0]);
            lb = invN * snb * __builtin_amdgcn_rcpf(3.14159265358979f * (float)qq * f0n[1]);
          }
          const float2 cv = X[m];
          const float2 acc = jf == 0 ? make_float2(0.0f, 0.0f) : X[q];
          X[q] = make_float2(fmaf(cv.x, la, acc.x), fmaf(cv.y, lb, acc.y));
          // advance by +-64 bins: e^{j(a +- d)}
          float t1 = ca * rca - sgn * sa * rsa, t2 = sgn * ca * rsa + sa * rca; ca = t1; sa = t2;
          t1 = cb2 * rcb - sgn * sb2 * rsb; t2 = sgn * cb2 * rsb + sb2 * rcb; cb2 = t1; sb2 = t2;
        }
      }
    }
    __syncthreads();
    fft_dif(X, tw, N / M3, M3, logM3, lane);
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      if(gg[e] >= nframes) continue;
      for(int j = lane; j < nspec; j += WAVE) {
        // bin idx = j * N / nfft_psd of the N-point envelope (layer0.c:341) = bin idx/fold of E
        const int idx = ((int)((long long)j * N / nfft_psd) / fold) & (M3 - 1);
        const float2 v = X[brevN(idx, logM3)];
        env_out[(size_t)gg[e] * nspec + j] = (e == 0 ? v.x : v.y) * 2.0f;
      }
    }
    __syncthreads();
  }
}

#include "wave_fft.h"

// wave_fft diagnostic: one wavefront per transform, data straight from / to global memory
template <int LOGN>
__global__ __launch_bounds__(WAVE, 2) void k_wf_selftest(const float2* __restrict__ in,
  float2* __restrict__ out, int inverse) {
  constexpr int N = 1 << LOGN, P = N / WAVE;
  const int lane = threadIdx.x;
  float2* lds = (float2*)g_lds;
  WfTw<LOGN> tw; wf_init(tw, lane);
  float xr[P], xi[P];
  const float2* src = in + (size_t)blockIdx.x * N;
#pragma unroll
  for(int m = 0; m < P; m ++) { const float2 v = src[lane + WAVE * m]; xr[m] = v.x; xi[m] = v.y; }
  if(inverse) wave_fft<LOGN>(xi, xr, tw, lds, lane);
  else wave_fft<LOGN>(xr, xi, tw, lds, lane);
  float2* dst = out + (size_t)blockIdx.x * N;
#pragma unroll
  for(int m = 0; m < P; m ++) dst[lane + WAVE * m] = make_float2(xr[m], xi[m]);
}

// K6 on the register-resident wavefront FFT (N = 2^LOGN = nfft_spgm, fold = 2^LOGF =
// N / nfft_psd): the frame pair stays in registers from the global load of the samples to
// the global store of the envelope; LDS only carries the exchanges inside the transforms.
// Element lane + 64 m of every length-N (or M3) sequence is register m of lane `lane`.
#ifndef SPGM_EDGE_F64
#define SPGM_EDGE_F64 1                             // 0: bins 0 and N/2 of the spectrogram always as the float32 transform returns them (rounds 1 - 5)
#endif
#define SPGM_SEED_LDS (2 * WAVE * sizeof(float4))  // the seed cache of k_spgm_env_wf: two float4 per lane behind the exchange buffer
#define SPGM_EDGE_THRESH 12.7f                      // a DC / Nyquist bin this many nepers (110 dB) under the bin of the frame's F0: recompute exactly
// The exact DC and Nyquist sums of one Hann-windowed frame (window of ws samples centred on sample c of xs[0, nxe)), by one
// wavefront: float64 window, products and sums.  Only the FIX instantiation of k_spgm_env_wf contains it: inside the
// ordinary kernel -- inlined behind a rare branch, or as a real call -- its register needs made the compiler spill a
// hundred values of the common path (profiles/r06_b_*).
DEV void spgm_exact_edges(const float* __restrict__ xs, int nxe, int c, int ws, int lane, double* dc, double* ny) {
  const int half = ws / 2;
  const double inv = 1.0 / (double)(ws - 1);
  double sd = 0.0, sn = 0.0;
  for(int j = lane; j < ws; j += WAVE) {             // window sample j sits at transform position (j - half) mod N
    const int idx = c - half + j;
    if(idx < 0 || idx >= nxe) continue;
    const double t = (0.5 - 0.5 * cospi(2.0 * (double)j * inv)) * (double)xs[idx];
    sd += t; sn += ((j - half) & 1) ? -t : t;
  }
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) { sd += __shfl_xor(sd, o, WAVE); sn += __shfl_xor(sn, o, WAVE); }
  *dc = sd; *ny = sn;
}
// FIX = false: every frame pair; pairs with a DC / Nyquist bin under the threshold are appended to fix_list (pair index,
// mask of the frames concerned).  FIX = true (second launch, usually a handful of pairs): the listed pairs once more,
// bit for bit the same arithmetic, with the exact bins put in place of the transform's.
template <int LOGN, int LOGF, bool FIX>
__global__ __launch_bounds__(WAVE, 2) void k_spgm_env_wf(
  const float* __restrict__ x, const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off,
  const float* __restrict__ f0, int nframes, float thop, float fs, int nwin_psd,
  float norm_base, float* __restrict__ env_out, const int2* __restrict__ pairs, int npair,
  int2* __restrict__ fix_list, int* __restrict__ fix_count) {
  constexpr int N = 1 << LOGN, P = N / WAVE, LOGM = LOGN - LOGF, M3 = 1 << LOGM, P3 = M3 / WAVE;
  const int lane = threadIdx.x;
  float2* lds = (float2*)g_lds;
  WfTw<LOGN> twN; wf_init(twN, lane);
  WfTw<LOGM> twM; wf_init(twM, lane);
  constexpr int nspec = M3 / 2 + 1;
  const float invN = 1.0f / (float)N;
  // Seed cache (round 6).  The phasor seeds of a frame -- Hann window (three float64-reduced sine / cosine pairs and a
  // float64 division) and lifter (three more) -- depend on (F0, window length, lane) only, and on configs 2 / 3 and over any
  // flat stretch of an F0 track the frames a wavefront walks repeat them.  The seeds of frame a of the last pair that
  // missed are kept: the eight per-lane values in 2 KB of LDS behind the exchange buffer (all that 8 wavefronts per CU
  // leave of the 160 KB), the five wave-uniform ones in scalar registers.  A frame whose (F0 bits, window length) match
  // reads them back (two ds_read_b128) instead of evaluating 6 cs_turns + 1 division: the values are the ones it would
  // have computed -- same expressions, same inputs -- so the result does not depend on the hit.
  float4* seedc = (float4*)(lds + (wf_lds_elems<LOGN>() > wf_lds_elems<LOGM>() ? wf_lds_elems<LOGN>() : wf_lds_elems<LOGM>()));
  unsigned key_f = 0u; int key_ws = -1;              // (no frame has F0 bits 0: unvoiced frames carry 200 Hz / fs)
  int k_stc = 0, k_sts = 0, k_rc = 0, k_rs = 0, k_sc = 0;   // float bits, wave-uniform
  auto sfl = [](float v) { return __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)); };
  auto ufl = [](int v) { return __builtin_bit_cast(float, v); };
  const int wgx = xcd_frame(blockIdx.x, gridDim.x);
  const int nwork = FIX ? min(*fix_count, npair) : npair;
  const int per = (nwork + gridDim.x - 1) / gridDim.x;
  for(int pw = wgx * per; pw < min(nwork, (wgx + 1) * per); pw ++) {
    const int p = FIX ? fix_list[pw].x : pw;
    const int fixmask = FIX ? fix_list[pw].y : 0;
    int gg[2]; pair_of(pairs, p, nframes, gg[0], gg[1]);
#ifdef SPGM_SINGLE_EXPERIMENT                         // (experiment: every frame transformed beside an EMPTY partner, twice the work)
    const int g_both[2] = {gg[0], gg[1]};
    for(int rep = 0; rep < 2; rep ++) {
    gg[0] = g_both[rep]; gg[1] = nframes;
    if(gg[0] >= nframes) continue;
#endif
    float f0n[2], normalizer[2];
    const float* xsp[2]; int nxu[2], cc[2], wsz[2];
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      const int g = gg[e];
      f0n[e] = 200.0f / fs; normalizer[e] = 0; xsp[e] = x; nxu[e] = 0; cc[e] = 0; wsz[e] = 0;
      if(g >= nframes) continue;
      int u, i; frame_owner(frm_utt, frm_off, g, & u, & i);
      const float f = f0[g];
      wsz[e] = lp::spgmwin(f > 0 ? f : 0.0f, fs, nwin_psd);
      cc[e] = lp::center(i, thop, fs);
      xsp[e] = x + x_off[u]; nxu[e] = nx[u];
      f0n[e] = (f > 0 ? f : 200.0f) / fs;
      normalizer[e] = norm_base / (float)wsz[e];
    }
    // The DC and the Nyquist bin of a real frame are REAL sums (sum v, sum (-1)^t v) that change sign from frame to frame:
    // now and then a frame catches one 100 dB and more below its harmonics (seed 123208, frame 43: -147 dB), where the
    // float32 transform -- and the float32 window recurrence before it -- return their own rounding: the envelope there
    // came out 1.3 nepers off, the Kalman process variance with it, the smoothed PSD of the next frames by 1.95 dB
    // (tools/psd_bisect.py --product; profiles/r06_a_psd_bisect_123208.txt).  A pair with such a bin (detected on the
    // transform's output against the bin of the frame's F0; about one frame in a thousand) is listed and done again by the FIX launch
    // with the two bins of that frame formed exactly -- float64 Hann window, products and sums.  The float64 oracle
    // itself moves by +-30 % there under a one-ulp change of the input; this puts the product inside that band.
    // Both frames' samples are requested FIRST, branch-free (a window longer than the transform gets an empty range here and
    // its time-aliased sum below): round 5 loaded frame a, windowed it, then loaded frame b -- two memory round trips per
    // pair in a kernel with two wavefronts per SIMD; now the second frame's, and the seed look-up, ride under the first's.
    float xr[P], xi[P];
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      const int ws = wsz[e], half = ws / 2, c = cc[e];
      // window samples j in [0, ws) sit at signal samples c - half + j, clipped to [0, nxe)
      const int lo = max(c - half, 0), hi = ws <= N ? min(c - half + ws, nxu[e]) : lo;
      const buf_t rng = buf_range(xsp[e], lo, hi);
#pragma unroll
      for(int m = 0; m < P; m ++) {
        const int sp = lane + WAVE * m - (m >= P / 2 ? N : 0);
        const float ld = ld_range(rng, c + sp - lo);
        if(e == 0) xr[m] = ld; else xi[m] = ld;
      }
    }
    // seeds of frame a: from the cache, refilled when its (F0, window) differ from the entry's
    {
      const unsigned fb = __builtin_amdgcn_readfirstlane(__float_as_uint(f0n[0]));
      if(fb != key_f || wsz[0] != key_ws) {
        const int ws = wsz[0], half = ws / 2;
        const double inv = 1.0 / (double)(ws > 1 ? ws - 1 : 1);
        float stc, sts, c1, s1, c2, s2;
        cs_turns((double)WAVE * inv, & stc, & sts);
        cs_turns((double)(lane + half) * inv, & c1, & s1);           // m = 0
        cs_turns((double)(lane + half - N / 2) * inv, & c2, & s2);   // m = P / 2
        float rc, rs, ca, sa, cb, sb;
        cs_turns(0.5 * (double)f0n[0] * (double)WAVE, & rc, & rs);
        cs_turns(0.5 * (double)f0n[0] * (double)lane, & ca, & sa);             // qq = lane + 64 m
        cs_turns(0.5 * (double)f0n[0] * (double)(N / 2 - lane), & cb, & sb);   // qq = N - lane - 64 m
        seedc[lane] = make_float4(c1, s1, c2, s2);
        seedc[WAVE + lane] = make_float4(ca, sa, cb, sb);
        k_stc = sfl(stc); k_sts = sfl(sts); k_rc = sfl(rc); k_rs = sfl(rs); k_sc = sfl(invN / (3.14159265358979f * f0n[0]));
        key_f = fb; key_ws = ws;
      }
    }
    const bool seed_hit[2] = {true, __builtin_amdgcn_readfirstlane(__float_as_uint(f0n[1])) == key_f && wsz[1] == key_ws};
    float edge_log[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};   // FIX: exact log magnitudes of (bin 0, bin N/2) of a listed frame
    int edge_mask = 0;
    // zero-phase placement: position pos holds window sample j = sp + ws/2 with sp = pos
    // (first half) or pos - N (second half).  Hann window 0.5 - 0.5 cos(2 pi j / (ws - 1))
    // by phasor rotation over m (64 samples), one float64-reduced seed per half.
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      const int ws = wsz[e], half = ws / 2, c = cc[e], nxe = nxu[e];
      const float* xs = xsp[e];
      float (& v)[P] = e == 0 ? xr : xi;               // (loaded above)
      if(ws <= N) {
        float stc, sts, c1, s1, c2, s2;
        if(seed_hit[e]) {
          const float4 sd = seedc[lane];
          c1 = sd.x; s1 = sd.y; c2 = sd.z; s2 = sd.w; stc = ufl(k_stc); sts = ufl(k_sts);
        } else {
          const double inv = 1.0 / (double)(ws > 1 ? ws - 1 : 1);
          cs_turns((double)WAVE * inv, & stc, & sts);
          cs_turns((double)(lane + half) * inv, & c1, & s1);           // m = 0
          cs_turns((double)(lane + half - N / 2) * inv, & c2, & s2);   // m = P / 2
        }
#pragma unroll
        for(int m = 0; m < P; m ++) {
          const int j = lane + WAVE * m - (m >= P / 2 ? N : 0) + half;
          float& wc = m < P / 2 ? c1 : c2;
          float& wsn = m < P / 2 ? s1 : s2;
          const float w = ws > 1 ? 0.5f - 0.5f * wc : 1.0f;
          v[m] *= w;                                 // already 0 outside the window (ld_range)
          const float t1 = wc * stc - wsn * sts, t2 = wc * sts + wsn * stc; wc = t1; wsn = t2;
        }
      } else {
        // window longer than the transform (F0 < 3 fs / N): time-aliased sum, staged through LDS
        float* stage = (float*)lds;
        for(int pos = lane; pos < N; pos += WAVE) {
          float acc = 0.0f;
          for(int j = (pos + half) & (N - 1); j < ws; j += N) {
            const int idx = c - half + j;
            if(idx >= 0 && idx < nxe) acc += xs[idx] * hann_at(j, ws);
          }
          stage[pos] = acc;
        }
        __syncthreads();
#pragma unroll
        for(int m = 0; m < P; m ++) v[m] = stage[lane + WAVE * m];
        __syncthreads();
      }
#if SPGM_EDGE_F64
      if(FIX && ((fixmask >> e) & 1)) {
        double sd, sn;
        spgm_exact_edges(xs, nxe, c, ws, lane, & sd, & sn);
        edge_log[e][0] = __logf((float)fabs(sd) * normalizer[e] + 1e-10f);
        edge_log[e][1] = __logf((float)fabs(sn) * normalizer[e] + 1e-10f);
      }
#endif
    }
    wave_fft<LOGN>(xr, xi, twN, lds, lane);
    {                                                // log magnitude spectra of both frames
      constexpr int H = P / 2;
      float mr[H + 1], mi[H + 1];
      wave_mirror_lo<P>(xr, mr, lane);
      wave_mirror_lo<P>(xi, mi, lane);
#pragma unroll
      for(int m = 0; m <= H; m ++) {                 // bins k <= N/2 (m = H: lane 0 only matters)
        const float ar = 0.5f * (xr[m] + mr[m]), ai = 0.5f * (xi[m] - mi[m]);
        const float br = 0.5f * (xi[m] + mi[m]), bi = -0.5f * (xr[m] - mr[m]);
#ifdef SPGM_PRECISE_LOG                               // (experiment: correctly rounded sqrt / log instead of the hardware approximations)
        xr[m] = logf(sqrtf(ar * ar + ai * ai) * normalizer[0] + 1e-10f);
        xi[m] = logf(sqrtf(br * br + bi * bi) * normalizer[1] + 1e-10f);
#else
        xr[m] = __logf(__builtin_amdgcn_sqrtf(ar * ar + ai * ai) * normalizer[0] + 1e-10f);
        xi[m] = __logf(__builtin_amdgcn_sqrtf(br * br + bi * bi) * normalizer[1] + 1e-10f);
#endif
      }
#if SPGM_EDGE_F64
      if constexpr (! FIX) {
        // log magnitudes: bins 0 .. 63 sit in register 0 of the 64 lanes, bins 0 and N/2 in registers 0 and H of lane 0.
        // The yardstick is the bin of the frame's F0 (its fundamental: a strong bin of a voiced frame; 200 Hz for an
        // unvoiced one, as spec2env assumes): three cross-lane reads into scalar registers, nothing kept in vector registers
        int mask = 0;
#pragma unroll
        for(int e = 0; e < 2; e ++) {
          const int k0 = min(WAVE - 1, max(1, (int)(f0n[e] * (float)N + 0.5f)));
          const float top = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e == 0 ? xr[0] : xi[0]), k0));
          const float d0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, e == 0 ? xr[0] : xi[0])));
          const float dn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, e == 0 ? xr[H] : xi[H])));
          if(gg[e] < nframes && wsz[e] <= N && wsz[e] > 1 && fminf(d0, dn) < top - SPGM_EDGE_THRESH) mask |= 1 << e;
        }
        edge_mask = mask;                              // (wave-uniform; appended to the list at the end of the pair, where registers are free)
      } else if(lane == 0) {
        if(fixmask & 1) { xr[0] = edge_log[0][0]; xr[H] = edge_log[0][1]; }
        if(fixmask & 2) { xi[0] = edge_log[1][0]; xi[H] = edge_log[1][1]; }
      }
#endif
      wave_reflect<P>(xr, xr, lane);                 // log spectra are even: L[N - k] = L[k]
      wave_reflect<P>(xi, xi, lane);
    }
    wave_fft<LOGN>(xi, xr, twN, lds, lane);          // inverse: both real cepstra (x N)
    // lifter sinc(qq f0) / N, qq = min(q, N - q), folded onto M3 points (q mod M3 is in-lane);
    // sin(pi f0n qq) by phasor rotation over m (64 quefrency bins), seeded from reduced phases
    float er[P3], ei[P3];
    {
      float rc[2], rs[2], ca[2], sa[2], cb[2], sb[2], sc[2];
#pragma unroll
      for(int e = 0; e < 2; e ++) {
        if(seed_hit[e]) {
          const float4 sd = seedc[WAVE + lane];
          ca[e] = sd.x; sa[e] = sd.y; cb[e] = sd.z; sb[e] = sd.w; rc[e] = ufl(k_rc); rs[e] = ufl(k_rs); sc[e] = ufl(k_sc);
        } else {
          cs_turns(0.5 * (double)f0n[e] * (double)WAVE, & rc[e], & rs[e]);
          cs_turns(0.5 * (double)f0n[e] * (double)lane, & ca[e], & sa[e]);             // qq = lane + 64 m
          cs_turns(0.5 * (double)f0n[e] * (double)(N / 2 - lane), & cb[e], & sb[e]);   // qq = N - lane - 64 m
          sc[e] = invN / (3.14159265358979f * f0n[e]);
        }
      }
#pragma unroll
      for(int m = 0; m < P; m ++) {
        const int q = lane + WAVE * m;
        const int qq = m < P / 2 ? q : N - q;
        const float rq = __builtin_amdgcn_rcpf((float)qq);
        float la, lb;
        if(m < P / 2) {
          la = sc[0] * sa[0] * rq; lb = sc[1] * sa[1] * rq;
#pragma unroll
          for(int e = 0; e < 2; e ++) {              // advance by +64 bins
            const float t1 = ca[e] * rc[e] - sa[e] * rs[e], t2 = ca[e] * rs[e] + sa[e] * rc[e];
            ca[e] = t1; sa[e] = t2;
          }
        } else {
          la = sc[0] * sb[0] * rq; lb = sc[1] * sb[1] * rq;
#pragma unroll
          for(int e = 0; e < 2; e ++) {              // advance by -64 bins
            const float t1 = cb[e] * rc[e] + sb[e] * rs[e], t2 = sb[e] * rc[e] - cb[e] * rs[e];
            cb[e] = t1; sb[e] = t2;
          }
        }
        if(qq == 0) { la = invN; lb = invN; }
        const int mm = m & (P3 - 1);
        if(m < P3) { er[mm] = xr[m] * la; ei[mm] = xi[m] * lb; }
        else { er[mm] = fmaf(xr[m], la, er[mm]); ei[mm] = fmaf(xi[m], lb, ei[mm]); }
      }
    }
    wave_fft<LOGM>(er, ei, twM, lds, lane);
    // envelope bin j = lane + 64 mm of frame a in er, of frame b in ei; nspec = M3 / 2 + 1 bins
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      if(gg[e] >= nframes) continue;
      float* row = env_out + (size_t)gg[e] * nspec;
#pragma unroll
      for(int mm = 0; mm <= P3 / 2; mm ++) {
        const int j = lane + WAVE * mm;
        if(j < nspec) row[j] = (e == 0 ? er[mm] : ei[mm]) * 2.0f;
      }
    }
    if(! FIX && edge_mask && fix_list && lane == 0)
      fix_list[atomicAdd(fix_count, 1)] = make_int2(p, edge_mask);   // (at most one entry per pair: never beyond npair)
#ifdef SPGM_SINGLE_EXPERIMENT
    }
#endif
  }
}

// =====================================================================
// K7  residual PSD per frame (HOT LOOP C) -- replaces layer0.c:354-360 with
// llsm_estimate_psd / llsm_fft_to_psd (dsputils.c:237-265): Blackman window
// of nwin samples, FFT of nfft, |X|^2 / sum(w^2), log(max(1e-10, .)).
// Two frames per complex FFT.
// =====================================================================
__global__ __launch_bounds__(WAVE) void k_psd_frames(
  const float* __restrict__ xres, const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off, int nframes,
  float thop, float fs, int nwin, const float* __restrict__ win, float inv_wpow,
  int N, int logN, const float2* __restrict__ tw_glob, int tw_nmax,
  float* __restrict__ psd_log, const int2* __restrict__ pairs, int npair, float2* __restrict__ gscr) {
  // gscr != NULL (N > 8192): X of this workgroup in global scratch (N float2), twiddles from the global table
  const int lane = threadIdx.x;
  float2* X = gscr ? gscr + (size_t)blockIdx.x * N : (float2*)g_lds;
  float2* twl = (float2*)g_lds + N;
  const float2* tw = gscr ? tw_glob : twl;
  const int tws = gscr ? tw_nmax / N : 1;
  if(! gscr) load_twiddles(twl, tw_glob, N, tw_nmax, lane);
  const int nspec = N / 2 + 1;
  const int wgx = xcd_frame(blockIdx.x, gridDim.x);
  const int per = (npair + gridDim.x - 1) / gridDim.x;
  for(int p = wgx * per; p < min(npair, (wgx + 1) * per); p ++) {
    const float* xs[2]; int nxu[2], base[2];
    int gg[2]; pair_of(pairs, p, nframes, gg[0], gg[1]);
    const bool two = gg[1] < nframes;               // a lone frame transforms beside its own copy
    if(! two) gg[1] = gg[0];
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      const int g = gg[e];
      int u, i; frame_owner(frm_utt, frm_off, g, & u, & i);
      xs[e] = xres + x_off[u]; nxu[e] = nx[u];
      base[e] = lp::center(i, thop, fs) - nwin / 2;
    }
    for(int t0 = lane; t0 < N; t0 += WAVE * 8) {
      float va[8], vb[8], wv[8];
#pragma unroll
      for(int q8 = 0; q8 < 8; q8 ++) {
        const int t = t0 + q8 * WAVE;
        const int ia = base[0] + t, ib = base[1] + t;
        const bool in = t < nwin;
        wv[q8] = in ? win[t] : 0.0f;
        va[q8] = (in && ia >= 0 && ia < nxu[0]) ? xs[0][ia] : 0.0f;
        vb[q8] = (in && ib >= 0 && ib < nxu[1]) ? xs[1][ib] : 0.0f;
      }
#pragma unroll
      for(int q8 = 0; q8 < 8; q8 ++) {
        const int t = t0 + q8 * WAVE;
        if(t < N) X[t] = make_float2(va[q8] * wv[q8], vb[q8] * wv[q8]);
      }
    }
    __syncthreads();
    fft_dif(X, tw, tws, N, logN, lane);
    for(int k = lane; k < nspec; k += WAVE) {
      float2 A, B; unpack_pair(X, N, logN, k, & A, & B);
      psd_log[(size_t)gg[0] * nspec + k] = logf(fmaxf(1e-10f, (A.x * A.x + A.y * A.y) * inv_wpow));
      if(two)
        psd_log[(size_t)gg[1] * nspec + k] = logf(fmaxf(1e-10f, (B.x * B.x + B.y * B.y) * inv_wpow));
    }
    __syncthreads();
  }
}

// K7 on the register-resident wavefront FFT (N = 2^LOGN)
template <int LOGN>
__global__ __launch_bounds__(WAVE, 2) void k_psd_frames_wf(
  const float* __restrict__ xres, const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off, int nframes,
  float thop, float fs, int nwin, const float* __restrict__ win, float inv_wpow,
  float* __restrict__ psd_log, const int2* __restrict__ pairs, int npair) {
  constexpr int N = 1 << LOGN, P = N / WAVE, H = P / 2, nspec = N / 2 + 1;
  const int lane = threadIdx.x;
  float2* lds = (float2*)g_lds;
  WfTw<LOGN> tw; wf_init(tw, lane);
  float wv[P];                                       // the window is the same for every frame pair
#pragma unroll
  for(int m = 0; m < P; m ++) wv[m] = ld_guard(win, lane + WAVE * m, nwin, true);
  const int wgx = xcd_frame(blockIdx.x, gridDim.x);
  const int per = (npair + gridDim.x - 1) / gridDim.x;
  for(int p = wgx * per; p < min(npair, (wgx + 1) * per); p ++) {
    const float* xs[2]; int nxu[2], base[2];
    int gg[2]; pair_of(pairs, p, nframes, gg[0], gg[1]);
    const bool two = gg[1] < nframes;               // a lone frame transforms beside its own copy
    if(! two) gg[1] = gg[0];
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      const int g = gg[e];
      int u, i; frame_owner(frm_utt, frm_off, g, & u, & i);
      xs[e] = xres + x_off[u]; nxu[e] = nx[u];
      base[e] = lp::center(i, thop, fs) - nwin / 2;
    }
    float xr[P], xi[P];
    {
      const int lo0 = max(base[0], 0), lo1 = max(base[1], 0);
      const buf_t r0 = buf_range(xs[0], lo0, min(base[0] + nwin, nxu[0]));
      const buf_t r1 = buf_range(xs[1], lo1, min(base[1] + nwin, nxu[1]));
#pragma unroll
      for(int m = 0; m < P; m ++) {
        const int t = lane + WAVE * m;
        xr[m] = ld_range(r0, base[0] + t - lo0);
        xi[m] = ld_range(r1, base[1] + t - lo1);
      }
    }
#pragma unroll
    for(int m = 0; m < P; m ++) { xr[m] *= wv[m]; xi[m] *= wv[m]; }
    wave_fft<LOGN>(xr, xi, tw, lds, lane);
    float mr[H + 1], mi[H + 1];
    wave_mirror_lo<P>(xr, mr, lane);
    wave_mirror_lo<P>(xi, mi, lane);
    float* rowa = psd_log + (size_t)gg[0] * nspec;
    float* rowb = psd_log + (size_t)gg[1] * nspec;
#pragma unroll
    for(int m = 0; m <= H; m ++) {
      const int k = lane + WAVE * m;
      if(m < H || lane == 0) {
        const float ar = 0.5f * (xr[m] + mr[m]), ai = 0.5f * (xi[m] - mi[m]);
        const float br = 0.5f * (xi[m] + mi[m]), bi = -0.5f * (xr[m] - mr[m]);
        rowa[k] = logf(fmaxf(1e-10f, (ar * ar + ai * ai) * inv_wpow));
        if(two) rowb[k] = logf(fmaxf(1e-10f, (br * br + bi * bi) * inv_wpow));
      }
    }
  }
}

// =====================================================================
// K1b  peak-picking harmonic analysis (LLSM_AOPTION_HMPP) -- replaces
// llsm_harmonic_analysis's HMPP branch (dsputils.c:196-213):
// llsm_compute_spectrogram with a Blackman window of the frame's own length
// and ONE fft size per llsm_harmonic_analysis call (= per utterance and
// signal, llsm_get_fftsize dsputils.c:318-326), log magnitude (+1e-8), then
// llsm_harmonic_peakpicking (dsputils.c:126-143): arg-max within +-0.3 f0 of
// each harmonic, parabolic refinement (qifft), exp; phase linearly
// interpolated between the two bins around the refined peak, not unwrapped.
// One wavefront per frame; lanes = FFT points, then lanes = harmonics.
// nsig signals are analysed per launch (speech: 1; sub-band energies: nch).
// =====================================================================
__global__ __launch_bounds__(256) void k_utt_fftsize(
  const float* __restrict__ f0, const int* __restrict__ frm_off, const int* __restrict__ nfrm,
  float fs, float rel_winsize, int nmax, int* __restrict__ nfft_u) {
  const int u = blockIdx.x;
  __shared__ float red[256];
  float m = 1000.0f;                                 // dsputils.c:319
  for(int i = threadIdx.x; i < nfrm[u]; i += 256) {
    float f = f0[frm_off[u] + i];
    if(f > 0 && f < m) m = f;
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for(int o = 128; o > 0; o >>= 1) {
    if((int)threadIdx.x < o) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if(threadIdx.x == 0) {
    const int w = lp::hwin(red[0], fs, rel_winsize);
    int n = 1; while(n < w) n <<= 1;                 // pow(2, ceil(log2(max_winsize)))
    nfft_u[u] = n > nmax ? nmax : n;
  }
}

// one frame g on NT threads; bufA: N float2, lm: 2 (N / 2 + 1) floats (LDS, or global scratch for N beyond the LDS);
// tw / tws: twiddle table and its stride
template <int NT>
DEV void harm_pp_frame(int g, int lane, float2* bufA, const float2* tw, int tws, float* lm, int N,
  const float* __restrict__ sig, size_t sig_stride, int nsig,
  const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off, float f, float thop, float fs,
  float rel_winsize, int maxnhar, float norm_base,
  int* __restrict__ nhar_out, float* __restrict__ ampl, float* __restrict__ phse) {
  int u, i; frame_owner(frm_utt, frm_off, g, & u, & i);
  int logN = 0; while((1 << logN) < N) logN ++;
  float* ph = lm + (N / 2 + 1);
  const int ws = lp::hwin(f, fs, rel_winsize);
  const int c = lp::center(i, thop, fs);
  const int K = lp::nhar(f, fs, maxnhar);
  const int half = ws / 2;
  const int nxu = nx[u];
  const float normalizer = norm_base / (float)ws;
  for(int sidx = 0; sidx < nsig; sidx ++) {
    const float* xs = sig + (size_t)sidx * sig_stride + x_off[u];
    for(int pos = lane; pos < N; pos += NT) {
      float acc = 0;
      for(int j = (pos + half) % N; j < ws; j += N) {
        int idx = c - half + j;
        if(idx >= 0 && idx < nxu) acc += xs[idx] * blackman_at(j, ws);
      }
      bufA[pos] = make_float2(acc, 0.0f);
    }
    __syncthreads();
    fft_dif<NT>(bufA, tw, tws, N, logN, lane);
    for(int k = lane; k <= N / 2; k += NT) {
      const float2 v = bufA[brevN(k, logN)];
      lm[k] = logf(sqrtf(v.x * v.x + v.y * v.y) * normalizer + 1e-8f);
      ph[k] = atan2f(v.y, v.x);
    }
    __syncthreads();
    float* arow = ampl + ((size_t)g * nsig + sidx) * maxnhar;
    float* prow = phse + ((size_t)g * nsig + sidx) * maxnhar;
    for(int h = lane + 1; h <= maxnhar; h += NT) {
      float a = 0, p = 0;
      if(h <= K) {
        int l = lp::iround((double)lp::fmul(lp::fdiv(lp::fmul(f, (float)h - 0.3f), fs), (float)N));
        int r = lp::iround((double)lp::fmul(lp::fdiv(lp::fmul(f, (float)h + 0.3f), fs), (float)N));
        l = max(1, l); r = min(N / 2 - 1, r);
        int pk = l;
        for(int j = l; j <= r; j ++) if(lm[j] > lm[pk]) pk = j;
        const float ya = lm[pk - 1], yb = lm[pk], yc = lm[pk + 1];
        const float a1 = (ya + yc) * 0.5f - yb, a2 = yc - yb - a1;
        float xq = a1 == 0 ? 0.0f : -a2 / a1 * 0.5f;
        if(xq < -1.0f || xq > 1.0f) xq = 0.0f;
        const float pf = (float)pk + xq;
        a = expf(a1 * xq * xq + a2 * xq + yb);
        const int kb = (int)pf;
        const float fr = pf - (float)kb;               // fmod(peak_freq, 1.0)
        p = ph[kb] + (ph[kb + 1] - ph[kb]) * fr;
      }
      arow[h - 1] = a; prow[h - 1] = p;
    }
    __syncthreads();
  }
  if(lane == 0) nhar_out[g] = K;
}

__global__ __launch_bounds__(WAVE) void k_harm_pp(
  const float* __restrict__ sig, size_t sig_stride, int nsig,
  const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off,
  const float* __restrict__ f0, const int* __restrict__ nfft_u, float thop, float fs,
  float rel_winsize, int maxnhar, float norm_base, const float2* __restrict__ tw_glob, int tw_nmax,
  int lds_n, int* __restrict__ nhar_out, float* __restrict__ ampl, float* __restrict__ phse) {
  const int g = xcd_frame(blockIdx.x, gridDim.x), lane = threadIdx.x;
  const float f = f0[g];
  const int N = nfft_u[frm_utt[g]];
  if(!(f > 0) || N > lds_n) {                       // (N > lds_n: k_harm_pp_big fills these rows afterwards)
    if(lane == 0) nhar_out[g] = 0;
    for(int k = lane; k < nsig * maxnhar; k += WAVE) {
      ampl[(size_t)g * nsig * maxnhar + k] = 0; phse[(size_t)g * nsig * maxnhar + k] = 0;
    }
    return;
  }
  float2* bufA = (float2*)g_lds;
  float2* tw = bufA + lds_n;
  float* lm = (float*)(tw + lds_n / 2);               // log magnitude, then phase
  load_twiddles(tw, tw_glob, N, tw_nmax, lane);
  harm_pp_frame<WAVE>(g, lane, bufA, tw, 1, lm, N, sig, sig_stride, nsig, x_off, nx, frm_utt, frm_off, f, thop, fs,
    rel_winsize, maxnhar, norm_base, nhar_out, ampl, phse);
}

// The frames whose transform does not fit the LDS (N > lds_n: F0 below 21.6 Hz at 44.1 kHz): persistent workgroups of
// 256 threads, buffers in global scratch (N float2 + 2 (N / 2 + 1) floats per workgroup: L2-resident), twiddles from
// the big global table.  dsputils.c:196-213, 318-326 take any power of two; so does this.
#define HPP_BIG_NT 256
__global__ __launch_bounds__(HPP_BIG_NT) void k_harm_pp_big(
  const float* __restrict__ sig, size_t sig_stride, int nsig,
  const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off, int nframes,
  const float* __restrict__ f0, const int* __restrict__ nfft_u, float thop, float fs,
  float rel_winsize, int maxnhar, float norm_base, const float2* __restrict__ tw_big, int tw_big_nmax,
  int lds_n, int nmax, float2* __restrict__ gscr, int* __restrict__ nhar_out, float* __restrict__ ampl, float* __restrict__ phse) {
  const int lane = threadIdx.x;
  float2* bufA = gscr + (size_t)blockIdx.x * (size_t)(nmax + nmax / 2 + 2);
  float* lm = (float*)(bufA + nmax);
  for(int g = blockIdx.x; g < nframes; g += gridDim.x) {
    const float f = f0[g];
    const int N = nfft_u[frm_utt[g]];
    if(!(f > 0) || N <= lds_n || N > nmax) continue;  // (uniform per workgroup)
    harm_pp_frame<HPP_BIG_NT>(g, lane, bufA, tw_big, tw_big_nmax / N, lm, N, sig, sig_stride, nsig, x_off, nx, frm_utt, frm_off,
      f, thop, fs, rel_winsize, maxnhar, norm_base, nhar_out, ampl, phse);
    __syncthreads();
  }
}

// =====================================================================
// K8 + K9  Kalman filter + RTS smoother along time and resampling to the PSD grid
// replaces layer0.c:361-385 (process variance from the 3-frame moving variance of the
// envelope, R = pi^2/6, smoothed + Euler gamma, residual) and layer0.c:388-408 (interp1 of the
// smoothed log-PSD and of the residual onto linspace(0, fnyq, npsd), to dB).
// The smoother is independent per FFT bin and an output point j only ever reads the two bins
// floor(pos_j), floor(pos_j) + 1, so a lane owns ONE OUTPUT POINT and runs just those two
// chains (2 x 256 chains over the 513 bins at the defaults: nearly every bin, each once); the full-resolution smoothed planes are never
// written.  The kernel is HBM-bound, so the forward filter keeps only a CHECKPOINT of its
// state every 8 frames (ck: [chunk][4 npsd], ~F/8 rows in all) and the backward (RTS) pass
// recomputes the 8 filtered states of a chunk from the checkpoint before smoothing them.
// Loads of a chunk are independent of the recursion and are issued together.
// =====================================================================
// The two chains of an output point (bins k0 and k1) run as the two halves of float2 values:
// explicit vector arithmetic gives v_pk_* instructions (the library is built without SLP
// vectorisation, which pays everywhere except here), component-wise the same IEEE operations.
#ifdef KAL_F64                                        // (experiment: the recursions in float64; rows and checkpoints stay float32)
typedef double kal1;
typedef double kal2 __attribute__((ext_vector_type(2)));
#else
typedef float kal1;
typedef float kal2 __attribute__((ext_vector_type(2)));
#endif
#ifndef KAL_SPLIT
#define KAL_SPLIT 0                                  // 1: one BIN chain per lane (lanes 2 j and 2 j + 1 carry the two bins point j interpolates between);
#endif                                               // 0: both in one lane as packed pairs (rounds 1 - 5)
// The split form was an experiment: the launch has n_utt x npsd points (132 k at 1 024 utterances = two wavefronts per SIMD of
// 200 x 2 dependent steps each) and keeps the VALU a fifth busy; one bin per lane doubles the wavefronts (four per SIMD,
// half the registers each) at the same instruction count per wavefront.  Measured 0.54 ms against 0.46: a switch only.
#if KAL_SPLIT
typedef kal1 kalv;
#define KALV(x) ((kal1)(x))
#else
typedef kal2 kalv;
#define KALV(x) ((kal2){(kal1)(x), (kal1)(x)})
#endif
struct KalState { kalv xk, p, Q; };
// bins are neighbours (k1 = k0 + 1, or k1 = k0 at the last point): one 8-byte load at p[k1 - 1]
struct __attribute__((packed, aligned(4))) KalPair { float a, b; };
#ifndef KAL_ABL
#define KAL_ABL 0                                    // timing ablations (tools/kbench.py): 1 forward pass only, 2 no exp / log at the output, 4 no loads, 8 no stores
#endif
DEV kalv kal_ld(const float* __restrict__ p, size_t at, bool same) {
#if KAL_ABL & 4
  return KALV((float)(at & 1023) * 1e-3f);
#endif
#if KAL_SPLIT
  (void)same;
  return (kal1)p[at];                                // (`at` already points at this lane's bin)
#else
  // (a, b) as loaded: at the last point (k1 == k0) the chain in x runs on bin k1 - 1 and is not looked at -- the outputs take
  // y there.  Round 5 selected (b, b) HERE: the select sat behind an s_waitcnt right after each load (four loads in
  // flight, every "prefetched" row waited for at once).
  (void)same;
  const KalPair v = *(const KalPair*)(p + at);
  return (kal2){(kal1)v.a, (kal1)v.b};
#endif
}
#ifndef KAL_FAST_OUT
#define KAL_FAST_OUT 1                               // the output point's exp / log10 on the hardware's exp2 / log2
#endif
#ifndef KAL_RCP
#define KAL_RCP 1                                    // gains as numerator x reciprocal (hardware 1-ulp reciprocal + one Newton step) instead of an IEEE division
#endif
// num / den for den > 0.  An IEEE float32 division is 11 dependent instructions (~60 cycles, tools/ubench/dep_latency.hip)
// on the one chain that bounds this kernel (the covariance recursion: every step waits for the previous gain); the
// reciprocal form is 5 (~40).  Both are good to an ulp; the filter contracts such errors.
DEV kalv kal_ratio(kalv num, kalv den) {
#if KAL_RCP && ! defined(KAL_F64)
#if KAL_SPLIT
  kalv r = __builtin_amdgcn_rcpf(den);
#else
  kalv r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
#endif
  const kalv er = (kal1)1.0f - den * r;
  r = r + r * er;
  return num * r;
#else
  return num / den;
#endif
}
// process variance = the 3-frame moving variance of the envelope, m2 / 3 - m1^2 / 9 (layer0.c:366-375), evaluated as the
// mean squared deviation from the 3-frame mean: in float32 the reference's difference of two numbers of the size of the
// squared log level (~ 200) rounds at the size of a small variance itself; this form does not cancel (smooth stretches
// come 2 - 5 x closer to the float64 oracle, profiles/r04_r_psd_tails.txt; the rare 0.1 dB tails have another origin)
DEV kalv kal_q(kalv e_prev, kalv e_cur, kalv e_next) {
  const kalv mean = (e_prev + e_cur + e_next) * (kal1)(1.0f / 3.0f);
  const kalv da = e_prev - mean, db = e_cur - mean, dc = e_next - mean;
  return __builtin_elementwise_max(KALV(1e-8f), (da * da + db * db + dc * dc) * (kal1)(1.0f / 3.0f));
}
// one filter step with the process variance Q of its frame; first: frame 0 of the utterance
DEV void kal_upd(KalState& s, bool first, kalv Q, kalv z) {
  const kal1 R = (kal1)1.6449340668482264;            // LOGCHI2VAR = pi^2/6
  s.Q = Q;
  if(first) {
    s.xk = z; s.p = KALV(R);                     // the first observation is the state ...
    if(g_conv.kalman_init == 1) {                     // ... or also the first update: prior (z0, R0), then the filter step
      const kalv pp = s.p + s.Q;
      s.p = ((kal1)1.0f - kal_ratio(pp, pp + R)) * pp;
    }
  }
  else {
    const kalv pp = s.p + s.Q;
#ifdef KAL_BREAK                                      // (a deliberately WRONG build: the parity contract must fail it -- tests/gpu_common.py)
    const kalv kg = (kal1)1.5f * kal_ratio(pp, pp + R);
#else
    const kalv kg = kal_ratio(pp, pp + R);
#endif
    s.xk = s.xk + kg * (z - s.xk);
    s.p = ((kal1)1.0f - kg) * pp;
  }
}
DEV void kal_step(KalState& s, int i, kalv e_prev, kalv e_cur, kalv e_next, kalv z) {
  kal_upd(s, i == 0, kal_q(e_prev, e_cur, e_next), z);
}

#ifndef KAL_WPE
#define KAL_WPE (KAL_SPLIT ? 4 : 2)                               // wavefronts per SIMD the register budget is cut for: the flat numbering leaves two wavefronts per SIMD in all at 1 024 utterances (3: 168 registers, spills with per-lane frame counts)
#endif
__global__ __launch_bounds__(128, KAL_WPE) void k_kalman(
  const float* __restrict__ env, const float* __restrict__ psd_log, float* __restrict__ ck,
  const int* __restrict__ frm_off, const int* __restrict__ nfrm, int n_utt, int nspec, int npsd, float fs,
  float* __restrict__ psd, float* __restrict__ psdres, int* __restrict__ has_psdres) {
  // One thread per (utterance, output point), numbered flat: with a grid of (points / 128, utterances) the 129 points of
  // the default grid made a second workgroup per utterance with ONE live lane -- half of the launch's wavefronts, each
  // as long as a full one.  A wavefront may straddle two utterances: frame counts and offsets are per lane.
#if KAL_SPLIT
  const int flat2 = blockIdx.x * 128 + threadIdx.x;  // lanes 2 j and 2 j + 1: the bins k1 - 1 and k1 of point j
  const int flat = flat2 >> 1, cb = flat2 & 1;
#else
  const int flat = blockIdx.x * 128 + threadIdx.x;
  const int cb = 0;
#endif
  if(flat >= n_utt * npsd) return;
  const int u = flat / npsd, j = flat - u * npsd;
  const int n = nfrm[u];
  if(n <= 0) return;
  // the two bins this output point interpolates between (layer0.c:388-396)
  const float fnyq = fs / 2.0f;
  const float xq = npsd > 1 ? fnyq * (float)j / (float)(npsd - 1) : 0.0f;
  const float pos = xq / fnyq * (float)(nspec - 1 + g_conv.interp1u_excl);   // interp1u: inclusive / exclusive right end
  int k0 = (int)floorf(pos);
  float r = 0.0f;
  int k1;
  if(k0 >= nspec - 1) { k0 = nspec - 1; k1 = k0; }
  else { if(k0 < 0) k0 = 0; k1 = k0 + 1; r = pos - (float)k0; }
  const size_t ns = (size_t)nspec, fo = (size_t)frm_off[u];
  const bool same = k1 == k0;
  const size_t op = fo * ns + (size_t)max(k1 - 1, 0) + (size_t)cb;   // pair base: bins (k1 - 1, k1) (split: this lane's bin of the two)
  constexpr int KC = KAL_CHUNK;                        // frames per chunk (checkpoint spacing, rows per request)
  // checkpoints: chunk c of utterance u at row (frm_off[u] / KC + u + c) of 4 npsd floats
  // (xa, pa, xb, pb per output point); rows of different utterances cannot overlap because
  // floor((fo + n) / 8) - floor(fo / 8) + 1 >= ceil(n / 8)
  const size_t cstride = (size_t)4 * npsd;
  float* ckp = ck + (fo / KC + (size_t)u) * cstride + (size_t)4 * j + (size_t)2 * cb;
  KalState S = {KALV(0), KALV(0), KALV(0)};
  {
    kalv e_prev = kal_ld(env, op, same), e_cur = e_prev;           // clamped at i = -1
    // The rows of chunk i0 + 8 are requested before chunk i0 is computed (double buffer in registers): with many
    // utterances in flight other wavefronts cover the load latency anyway, with ONE utterance per call (the drop-in
    // llsm_analyze) the chain load -> 8 steps -> load was 60 % of this kernel's time.
    // Rows are requested TWO chunks ahead, into THREE register buffers that take turns (the chunk loop is unrolled three
    // times): nothing is copied.  Rounds 2 - 5 kept "current" and "next" arrays and copied next -> current at the end of every
    // turn -- a register move that has to wait for the load it moves, so the rows requested at the top of a turn were
    // waited for at its bottom.  MEASURED (round 6, profiles/r06_*kalman*): a single utterance (llsm_analyze, 1 154
    // frames) 0.53 -> 0.49 ms; the batch of 1 024 did not move (0.45 ms) -- nor did it with the loads' select removed,
    // the per-frame branches gone (55 -> 27 dependent instructions per frame), reciprocal gains, or twice the wavefronts
    // (one bin per lane, KAL_SPLIT: slower).  Its timing ablations say 0.20 ms arithmetic + 0.17 loads + 0.09 stores.
    // What bounds it is HBM: the planes are 513 bins wide (420 MB each), read once forward and 1.1 times backward, plus
    // 420 MB of outputs = 2.6 GB per launch at 1 024 utterances = 5.6 TB/s in 0.46 ms (LAB.md round 6, item 5).
    struct Rows { kalv e[KC], z[KC]; };
    auto load_rows = [&](Rows& r, int c0) {               // env at c0 + 1 .. c0 + KC, log PSD at c0 .. c0 + KC - 1 (clamped)
#pragma unroll
      for(int q = 0; q < KC; q ++) {
        const size_t in = (size_t)min(n - 1, c0 + q + 1) * ns, ic = (size_t)min(n - 1, c0 + q) * ns;
        r.e[q] = kal_ld(env, op + in, same);
        r.z[q] = kal_ld(psd_log, op + ic, same);
      }
    };
    // one chunk: rows of chunk i0 + 16 requested into `ahead`, chunk i0 computed from `cur`
    auto chunk = [&](int i0, const Rows& cur, Rows& ahead) {
      if(i0 + 2 * KC < n) load_rows(ahead, i0 + 2 * KC);
      // A chunk that every live lane runs in full has no per-step exec branch: its eight process variances are formed up
      // front (independent of the state) and the compiler interleaves them, and the state's linear part, with the one
      // chain that cannot be shortened -- covariance -> gain -> covariance.  Round 5's form tested i < n at every step:
      // eight basic blocks per chunk, each one dependent run of ~55 instructions.
      if(__builtin_amdgcn_ballot_w64(i0 + KC > n) == 0) {
        kalv Qs[KC];
        Qs[0] = kal_q(e_prev, e_cur, cur.e[0]); Qs[1] = kal_q(e_cur, cur.e[0], cur.e[1]);
#pragma unroll
        for(int q = 2; q < KC; q ++) Qs[q] = kal_q(cur.e[q - 2], cur.e[q - 1], cur.e[q]);
        if(i0 == 0) kal_upd(S, true, Qs[0], cur.z[0]); else kal_upd(S, false, Qs[0], cur.z[0]);
#pragma unroll
        for(int q = 1; q < KC; q ++) kal_upd(S, false, Qs[q], cur.z[q]);
        e_prev = cur.e[KC - 2]; e_cur = cur.e[KC - 1];
      } else {
#pragma unroll
        for(int q = 0; q < KC; q ++) {
          const int i = i0 + q;
          if(i < n) {
            kal_step(S, i, e_prev, e_cur, cur.e[q], cur.z[q]);
            e_prev = e_cur; e_cur = cur.e[q];
          }
        }
      }
      float* c = ckp + (size_t)(i0 / KC) * cstride;  // state after frame min(i0 + KC - 1, n - 1)
#if KAL_SPLIT
      *(float2*)c = make_float2((float)S.xk, (float)S.p);
#else
      *(float4*)c = make_float4((float)S.xk.x, (float)S.p.x, (float)S.xk.y, (float)S.p.y);
#endif
    };
    Rows r0, r1, r2;
    load_rows(r0, 0); load_rows(r1, KC);
    for(int i0 = 0; i0 < n; i0 += 3 * KC) {
      chunk(i0, r0, r2);
      if(i0 + KC < n) chunk(i0 + KC, r1, r0);
      if(i0 + 2 * KC < n) chunk(i0 + 2 * KC, r2, r1);
    }
  }
#if KAL_ABL & 1
  if((float)(S.xk + S.p)[0 * KAL_SPLIT] == 1.2345f) psd[flat] = 0.0f;
  return;
#endif
  kalv sm = S.xk;                                    // smoothed values at i = n - 1
  kalv qn = KALV(0);                                 // Q of the first frame of the later chunk
  const int i_last = ((n - 1) / KC) * KC;
  struct RowsB { kalv e[KC + 2], z[KC]; float4 cpt; };   // env at i0 - 1 .. i0 + KC (clamped), log PSD at i0 .. i0 + KC - 1, checkpoint before chunk i0
  auto fetch = [&](int i0, RowsB& rb) {
#pragma unroll
    for(int q = 0; q < KC + 2; q ++) {
      const size_t ic = (size_t)min(n - 1, max(0, i0 - 1 + q)) * ns;
      rb.e[q] = kal_ld(env, op + ic, same);
    }
#pragma unroll
    for(int q = 0; q < KC; q ++) {
      const size_t ic = (size_t)min(n - 1, i0 + q) * ns;
      rb.z[q] = kal_ld(psd_log, op + ic, same);
    }
    rb.cpt = make_float4(0, 0, 0, 0);
#if KAL_SPLIT
    if(i0 > 0) { const float2 c2 = *(const float2*)(ckp + (size_t)(i0 / KC - 1) * cstride); rb.cpt = make_float4(c2.x, c2.y, 0.0f, 0.0f); }
#else
    if(i0 > 0) rb.cpt = *(const float4*)(ckp + (size_t)(i0 / KC - 1) * cstride);
#endif
  };
  // smoothed log-PSD (+ EULERGAMMA bias removal) and residual at the two bins, interpolated
  auto put = [&](int i, kalv smv, kalv zv) {
    const kalv m = smv + (kal1)0.57721566f, rs = zv - smv;
#if KAL_SPLIT
    // the even lane of a pair writes the point: its own bin (k1 - 1) and the odd lane's (k1), one cross-lane read each
    const kal1 my = __shfl_xor(m, 1, WAVE), ry = __shfl_xor(rs, 1, WAVE);
    if(cb) return;
    const float a = same ? (float)my : (float)(m + (my - m) * (kal1)r);
    const float b = same ? (float)ry : (float)(rs + (ry - rs) * (kal1)r);
#else
    const float a = same ? (float)m.y : (float)(m.x + (m.y - m.x) * (kal1)r);
    const float b = same ? (float)rs.y : (float)(rs.x + (rs.y - rs.x) * (kal1)r);
#endif
    const size_t g = (fo + (size_t)i) * npsd + j;
#if KAL_ABL & 8
    if(a == 1.2345f) psd[g] = b;
#elif KAL_ABL & 2
    psdres[g] = b; psd[g] = a;
#elif KAL_FAST_OUT
    // hardware exp2 / log2 (1 ulp) and a multiplication instead of the library's correctly rounded exp / log10 and an IEEE
    // division: 82 -> 30 instructions per output point in a kernel whose wavefronts run alone on their SIMDs; the level
    // moves by < 1e-5 dB (|a| log2(e) rounds at 2e-6 relative)
    psdres[g] = b * (10.0f / 2.3025851f);
    psd[g] = (10.0f * 0.30102999566f) * __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a * 1.44269504089f) * (44100.0f / fs) + 1e-12f);
#else
    psdres[g] = b / 2.3025851f * 10.0f;
    psd[g] = 10.0f * log10f(expf(a) * 44100.0f / fs + 1e-12f);
#endif
    if(j == 0) has_psdres[fo + i] = 1;
  };
  // one chunk of the backward pass: rows of chunk i0 - 16 requested into `ahead` (three buffers in turn, as in the forward
  // pass), chunk i0 recomputed from its checkpoint and smoothed from `cur`
  auto chunk_b = [&](int i0, const RowsB& cur, RowsB& ahead) {
    if(i0 >= 2 * KC) fetch(i0 - 2 * KC, ahead);
#if KAL_SPLIT
    if(i0 > 0) { S.xk = (kal1)cur.cpt.x; S.p = (kal1)cur.cpt.y; }
#else
    if(i0 > 0) { S.xk = (kalv){cur.cpt.x, cur.cpt.z}; S.p = (kalv){cur.cpt.y, cur.cpt.w}; }
#endif
    kalv xf[KC], pf[KC], qf[KC];
    if(__builtin_amdgcn_ballot_w64(i0 + KC > n) == 0) {
      // a full chunk on every live lane: the filter steps again (variances up front, as above), then the smoother's gains
      // for all eight frames at once -- they depend on the filter's covariances only --, which leaves a chain of two
      // instructions per frame for the smoothed value itself, and the eight outputs side by side.  The chunk that ends
      // the utterance starts from sm = xk of its last frame: that frame's "step" xf + cg (sm - xf) returns sm unchanged
      // (cg = pf / (pf + 0) = 1 times an exact 0), the condition i < n - 1 of the general form below
      kalv Qs[KC];
#pragma unroll
      for(int q = 0; q < KC; q ++) Qs[q] = kal_q(cur.e[q], cur.e[q + 1], cur.e[q + 2]);
      if(i0 == 0) kal_upd(S, true, Qs[0], cur.z[0]); else kal_upd(S, false, Qs[0], cur.z[0]);
      xf[0] = S.xk; pf[0] = S.p; qf[0] = S.Q;
#pragma unroll
      for(int q = 1; q < KC; q ++) { kal_upd(S, false, Qs[q], cur.z[q]); xf[q] = S.xk; pf[q] = S.p; qf[q] = S.Q; }
      kalv cg[KC], smv[KC];
#pragma unroll
      for(int q = 0; q < KC; q ++) cg[q] = kal_ratio(pf[q], pf[q] + (q == KC - 1 ? qn : qf[q == KC - 1 ? KC - 1 : q + 1]));
#pragma unroll
      for(int q = KC - 1; q >= 0; q --) { sm = xf[q] + cg[q] * (sm - xf[q]); smv[q] = sm; }
#pragma unroll
      for(int q = KC - 1; q >= 0; q --) put(i0 + q, smv[q], cur.z[q]);
    } else {
#pragma unroll
      for(int q = 0; q < KC; q ++) {
        const int i = i0 + q;
        if(i < n) kal_step(S, i, cur.e[q], cur.e[q + 1], cur.e[q + 2], cur.z[q]);
        xf[q] = S.xk; pf[q] = S.p; qf[q] = S.Q;
      }
#pragma unroll
      for(int q = KC - 1; q >= 0; q --) {
        const int i = i0 + q;
        if(i < n) {
          if(i < n - 1) {
            const kalv nq = q == KC - 1 ? qn : qf[q == KC - 1 ? KC - 1 : q + 1];
            const kalv cg = kal_ratio(pf[q], pf[q] + nq);
            sm = xf[q] + cg * (sm - xf[q]);
          }
          put(i, sm, cur.z[q]);
        }
      }
    }
    qn = qf[0];
  };
  RowsB b0, b1, b2;
  fetch(i_last, b0);
  if(i_last >= KC) fetch(i_last - KC, b1);
  for(int i0 = i_last; i0 >= 0; i0 -= 3 * KC) {
    chunk_b(i0, b0, b2);
    if(i0 >= KC) chunk_b(i0 - KC, b1, b0);
    if(i0 >= 2 * KC) chunk_b(i0 - 2 * KC, b2, b1);
  }
}

// =====================================================================
// S1  Gaussian white-noise templates -- replaces llsm_generate_white_noise
// (dsputils.c:353-361) with the counter generator of plan.h.  Template of
// (utterance u, channel c) has n_ext(u) = min(20000, ny_u) + 128 samples; the
// reference's own wrap-around of the extension (dsputils.c:358-359) is kept.
// =====================================================================
#ifndef WHITE_FAST
#define WHITE_FAST 1
#endif
__global__ __launch_bounds__(256) void k_white(
  float* __restrict__ white, int ntemplate_ext, const int* __restrict__ out_len,
  int nch, unsigned long long seed) {
  const int u = blockIdx.z, c = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = min(20000, out_len[u]) + 128;
  if(i >= n) return;
  const int nt = min(20000, n);
  const int src = i < nt ? i : (i - nt) % nt;
  float u1, u2;
  lp::rng_uniforms((seed + (unsigned long long)u) * 16ULL + (unsigned long long)c,
    (unsigned long long)src, & u1, & u2);
#if WHITE_FAST
  // Box-Muller on the hardware's log2 / sqrt / cos-of-turns (1 ulp; |error| of a sample ~ 1e-6 of sigma): the correctly
  // rounded libm forms were 2/3 of this kernel's ~60 instructions per sample (it is VALU-bound, not bound by its 0.33 GB)
  white[((size_t)u * nch + c) * ntemplate_ext + i] =
    __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);   // -2 ln u1 = -2 ln 2 log2 u1
#else
  white[((size_t)u * nch + c) * ntemplate_ext + i] = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
#endif
}

// =====================================================================
// S2  noise-envelope frames -- replaces the per-frame body of
// llsm_synthesize_noise_envelope, layer0.c:296-304: small harmonic frame of
// the channel's envelope model + edc, floored at 1e-8, times Hann(nwin_env).
// One wavefront per frame, all channels.  Row (g, c) of envf[F][nch][nwin].
// =====================================================================
// `lane` of `nthr` threads work on frame g (k_env_frames: one wavefront; k_rt_front: three of a workgroup's four)
template <int NCH, int ME>
DEV void env_frame_body(int g, int lane, int nthr,
  const float* __restrict__ f0, const int* __restrict__ nhar_e,
  const float* __restrict__ eamp, const float* __restrict__ ephs,
  const float* __restrict__ edc, int nch, int me, float fs, int nwin,
  const float* __restrict__ win, float* __restrict__ envf, float* out_rows = nullptr,
  const float* par = nullptr, float f_par = 0, int nhe_par = 0) {
  // out_rows != NULL: the nch rows of this frame go there (LDS of k_rt_hop2) instead of into envf
  // par != NULL: the frame's parameters come from there -- [nch] edc, then [nch][me] (a cos phi, a sin phi) -- with f_par,
  // nhe_par instead of f0[g], nhar_e[g] (k_rt_hop2 forms them once per stream, not once per thread)
  const float f = par ? f_par : f0[g];
  const int K = f > 0 ? min(par ? nhe_par : nhar_e[g], me) : 0;
  const double turn1 = (double)f / (double)fs;
  const int half = nwin / 2;
  // a_k e^{j phi_k} of every channel in registers; the phasor powers e^{j k w0 (t - n/2)} are
  // shared by the channels, seeded per 4 samples from float64-reduced phases and rotated by 64
  float ar[NCH][ME], ai[NCH][ME], off[NCH];
#pragma unroll
  for(int c = 0; c < NCH; c ++) {
    off[c] = c < nch ? (par ? par[c] : edc[(size_t)g * nch + c]) : 0.0f;
#pragma unroll
    for(int k = 0; k < ME; k ++) {
      ar[c][k] = 0; ai[c][k] = 0;
      if(c < nch && k < K) {
        if(par) { ar[c][k] = par[nch + 2 * (c * me + k)]; ai[c][k] = par[nch + 2 * (c * me + k) + 1]; }
        else {
          const float a = eamp[((size_t)g * nch + c) * me + k], ph = ephs[((size_t)g * nch + c) * me + k];
          float sn, co; sincosf(ph, & sn, & co);
          ar[c][k] = a * co; ai[c][k] = a * sn;
        }
      }
    }
  }
  float stc, sts; cs_turns(turn1 * (double)nthr, & stc, & sts);
  float* out0 = out_rows ? out_rows : envf + (size_t)g * nch * nwin;
  for(int t0 = lane; t0 < nwin; t0 += nthr * 4) {
    float z1r, z1i; cs_turns(turn1 * (double)(t0 - half), & z1r, & z1i);
#pragma unroll
    for(int q = 0; q < 4; q ++) {
      const int t = t0 + q * nthr;
      if(t < nwin) {
        float y[NCH];
#pragma unroll
        for(int c = 0; c < NCH; c ++) y[c] = 0.0f;
        float zr = z1r, zi = z1i;
#pragma unroll
        for(int k = 0; k < ME; k ++) {               // ar = ai = 0 beyond K
#pragma unroll
          for(int c = 0; c < NCH; c ++) y[c] += ar[c][k] * zr - ai[c][k] * zi;
          const float nr = zr * z1r - zi * z1i, ni = zr * z1i + zi * z1r;
          zr = nr; zi = ni;
        }
        const float w = win[t];
#pragma unroll
        for(int c = 0; c < NCH; c ++)
          if(c < nch) out0[(size_t)c * nwin + t] = fmaxf(y[c] + off[c], 1e-8f) * w;
      }
      const float t1 = z1r * stc - z1i * sts, t2 = z1r * sts + z1i * stc; z1r = t1; z1i = t2;
    }
  }
}
template <int NCH, int ME>
__global__ __launch_bounds__(WAVE) void k_env_frames(
  const float* __restrict__ f0, const int* __restrict__ nhar_e,
  const float* __restrict__ eamp, const float* __restrict__ ephs,
  const float* __restrict__ edc, int nch, int me, float fs, int nwin,
  const float* __restrict__ win, float* __restrict__ envf) {
  env_frame_body<NCH, ME>(blockIdx.x, threadIdx.x, WAVE, f0, nhar_e, eamp, ephs, edc, nch, me, fs, nwin, win, envf);
}
// S3 (offline path): complex envelope amplitudes a_k e^{j phi_k} of every (frame, channel,
// harmonic), zero beyond nhar_e / for unvoiced frames, so that k_excite_env needs no
// per-sample sincos.  One thread per element.
__global__ __launch_bounds__(256) void k_env_params(
  const float* __restrict__ f0, const int* __restrict__ nhar_e,
  const float* __restrict__ eamp, const float* __restrict__ ephs, int nframes, int nch, int me,
  float2* __restrict__ cplx) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if(tid >= (size_t)nframes * nch * me) return;
  const int g = (int)(tid / ((size_t)nch * me)), k = (int)(tid % me);
  const int K = f0[g] > 0 ? min(nhar_e[g], me) : 0;
  float2 v = make_float2(0.0f, 0.0f);
  if(k < K) {
    float sn, co; sincosf(ephs[tid], & sn, & co);
    const float a = eamp[tid];
    v = make_float2(a * co, a * sn);
  }
  cplx[tid] = v;
}

// S3 + S3b fused (offline path) -- replaces layer0.c:296-310 (envelope frames, Hann window,
// overlap-add) and layer0.c:535-555 (template tiling, sqrt-envelope modulation, channel sum):
// the envelope frames are never materialised.  Thread per output sample; the (frame, offset)
// pairs whose overlap-add position round((i-1) thop fs + j) equals this sample come from a
// per-batch table built on the host with plan.h (they depend on thop, fs and the window only,
// not on the utterance), at most EXC_HITS of them in ascending (i, j) order -- the
// accumulation order of the reference's frame loop.
#define EXC_HITS 3
#ifndef EXC_FAST
#define EXC_FAST 1                                  // hardware sin / cos / sqrt in k_excite_env (0: float64-reduced phases, libm sqrt: rounds 1 - 5)
#endif
#define EXC_SLOTS 8                                 // envelope frames staged per block of 256 samples
template <int NCH, int ME>
__global__ __launch_bounds__(256) void k_excite_env(
  const float* __restrict__ colored, int ntemplate_ext, const int2* __restrict__ hits,
  const float2* __restrict__ cplx, const float* __restrict__ edc, const float* __restrict__ f0,
  int nwin_env, const float* __restrict__ win, int nch, int me, int nch_active,
  const int* __restrict__ frm_off, const int* __restrict__ nfrm,
  const int* __restrict__ out_off, const int* __restrict__ out_len, float thop, float fs,
  float* __restrict__ yexc) {
  // parameters of the frames this block's samples can touch: env_ola(i, j) = round((i-1) hop + j),
  // j < 2 hop, puts sample p under frames floor(p / hop) and floor(p / hop) + 1 (+- rounding)
  __shared__ float2 s_cp[EXC_SLOTS][NCH * ME];
  __shared__ float s_off[EXC_SLOTS][NCH];
  __shared__ float s_turn[EXC_SLOTS];               // f0 / fs, <= 0 when unvoiced
  const int u = blockIdx.y;
  const int b0 = blockIdx.x * 256, idx = b0 + threadIdx.x;
  const int ny = out_len[u];
  if(b0 >= ny) return;
  const int nf = nfrm[u], fo = frm_off[u];
  const float hop = lp::fmul(thop, fs);
  const int imin = max(0, (int)((float)b0 / hop) - 1);
  for(int t = threadIdx.x; t < EXC_SLOTS * NCH * ME; t += 256) {
    const int sl = t / (NCH * ME), r = t % (NCH * ME), c = r / ME, k = r % ME;
    const int i = imin + sl;
    float2 v = make_float2(0.0f, 0.0f);
    if(i < nf && c < nch && k < me) v = cplx[((size_t)(fo + i) * nch + c) * me + k];
    s_cp[sl][r] = v;
  }
  if(threadIdx.x < EXC_SLOTS * NCH) {
    const int sl = threadIdx.x / NCH, c = threadIdx.x % NCH, i = imin + sl;
    s_off[sl][c] = (i < nf && c < nch) ? edc[(size_t)(fo + i) * nch + c] : 0.0f;
  }
  if(threadIdx.x < EXC_SLOTS) {
    const int i = imin + threadIdx.x;
    s_turn[threadIdx.x] = i < nf ? f0[fo + i] / fs : 0.0f;
  }
  __syncthreads();
  if(idx >= ny) return;
  const int ntemplate = min(20000, ny);
  int b; float r;
  const int a = lp::stretch_index(idx, ntemplate, ny, 128, & b, & r);
  const int half = nwin_env / 2;
  float e[NCH];
#pragma unroll
  for(int c = 0; c < NCH; c ++) e[c] = 0.0f;
#pragma unroll
  for(int hh = 0; hh < EXC_HITS; hh ++) {
    const int2 hit = hits[(size_t)idx * EXC_HITS + hh];
    if(hit.x < 0 || hit.x >= nf) continue;
    const int sl = hit.x - imin, j = hit.y;
    const float w = win[j];
    if(sl >= 0 && sl < EXC_SLOTS) {
      const float tn = s_turn[sl];
      float z1r = 1.0f, z1i = 0.0f;
#if EXC_FAST
      // |j - half| <= nwin / 2 and tn <= 1/2: the float32 product is good to 6e-8 of its (<= ~100) turns and the hardware
      // sine / cosine take turns directly (|error| ~ 1e-6 of an envelope value whose square root modulates NOISE; the
      // float64 phase reduction + polynomial of cs_turns was a tenth of this kernel's instructions)
      if(tn > 0) { const float ph = tn * (float)(j - half); z1r = __builtin_amdgcn_cosf(ph); z1i = __builtin_amdgcn_sinf(ph); }
#else
      if(tn > 0) cs_turns((double)tn * (double)(j - half), & z1r, & z1i);
#endif
      float zr[ME], zi[ME];                          // e^{j k th}, k = 1 .. ME
      zr[0] = z1r; zi[0] = z1i;
#pragma unroll
      for(int k = 1; k < ME; k ++) {
        zr[k] = zr[k - 1] * z1r - zi[k - 1] * z1i; zi[k] = zr[k - 1] * z1i + zi[k - 1] * z1r;
      }
#pragma unroll
      for(int c = 0; c < NCH; c ++) {
        float y = 0.0f;
#pragma unroll
        for(int k = 0; k < ME; k ++) {               // amplitudes are zero beyond nhar_e
          const float2 av = s_cp[sl][c * ME + k];
          y += av.x * zr[k] - av.y * zi[k];
        }
        e[c] += fmaxf(y + s_off[sl][c], 1e-8f) * w;
      }
    } else {
      // hop shorter than 256 / (EXC_SLOTS - 3) samples: frame not staged, read it from HBM
      const int g = fo + hit.x;
      const float f = f0[g];
      float z1r = 1.0f, z1i = 0.0f;
      if(f > 0) cs_turns((double)(f / fs) * (double)(j - half), & z1r, & z1i);
      for(int c = 0; c < nch; c ++) {
        float y = 0.0f, zr = z1r, zi = z1i;
        for(int k = 0; k < me; k ++) {
          const float2 av = cplx[((size_t)g * nch + c) * me + k];
          y += av.x * zr - av.y * zi;
          const float nr = zr * z1r - zi * z1i, ni = zr * z1i + zi * z1r;
          zr = nr; zi = ni;
        }
        const float val = fmaxf(y + edc[(size_t)g * nch + c], 1e-8f) * w;
#pragma unroll
        for(int cc = 0; cc < NCH; cc ++) if(cc == c) e[cc] += val;
      }
    }
  }
  const float xf = b >= 0 ? __frsqrt_rn(2.0f * r * (r - 1.0f) + 1.0f) : 1.0f;
  float acc = 0;
#pragma unroll
  for(int c = 0; c < NCH; c ++) {
    if(c < nch_active) {
      const float* tpl = colored + ((size_t)u * nch + c) * ntemplate_ext;
      float v = tpl[a];
      if(b >= 0) {
        v *= 1.0f - r;
        v += tpl[b] * r;
        v *= xf;
      }
#if EXC_FAST
      v *= __builtin_amdgcn_sqrtf(e[c]);               // e >= 1e-8 w > 0 or exactly 0: the hardware root (1 ulp) has no special case to miss
#else
      v *= sqrtf(e[c]);
#endif
      acc += v;
    }
  }
  yexc[(size_t)out_off[u] + idx] = acc;
}


// S3 + S3b, second form (round 5): the same sums, arranged by TEMPLATE position.  k_excite_env above walks the output
// samples, so every template sample is fetched once per tile of the stretched noise (2.23 tiles at 1 s: 925 MB of
// fetches for 330 MB of templates, profiles/r04_zz_traffic.json) and every sample pays its own index division, float64
// phase reductions and 16 complex amplitudes from LDS per frame it lies under.  Here a thread owns FOUR consecutive
// residues rho .. rho + 3 of the tiling period T = ntemplate - 128 (T and the 128-sample cross-fade are multiples of
// four) and walks the tiles p = rho + t T: the templates are loaded once (16-byte loads) and stay in registers; the four
// samples lie under the same frames, so the frame's amplitudes are read from LDS once per four samples, the phasor of
// the first sample is seeded from a float64-reduced phase and the next three are one rotation by e^{j 2 pi f0 / fs}
// each; the output goes out as one 16-byte store.  stretch_index's closed form becomes the tile loop itself.
// Same (frame, offset) table and the same accumulation order per sample as above; the results differ from it by the
// float32 rounding of the three rotations (~1e-7 of an envelope value).
#ifndef EXC4_WPE
#define EXC4_WPE 4                                  // wavefronts per SIMD the register budget is cut for
#endif
#define EXC4_SLOTS 16                               // envelope frames staged per tile of 1024 samples (hop >= 79 samples; shorter hops: HBM path)
template <int NCH, int ME>
__global__ __launch_bounds__(256, EXC4_WPE) void k_excite_env4(
  const float* __restrict__ colored, int ntemplate_ext, const int2* __restrict__ hits,
  const float2* __restrict__ cplx, const float* __restrict__ edc, const float* __restrict__ f0,
  int nwin_env, const float* __restrict__ win, int nch, int me, int nch_active,
  const int* __restrict__ frm_off, const int* __restrict__ nfrm,
  const int* __restrict__ out_off, const int* __restrict__ out_len, float thop, float fs,
  float* __restrict__ yexc) {
  __shared__ __attribute__((aligned(16))) float2 s_cp[EXC4_SLOTS][NCH * ME];
  __shared__ float s_off[EXC4_SLOTS][NCH];
  __shared__ float s_turn[EXC4_SLOTS];              // f0 / fs, <= 0 when unvoiced
  __shared__ float2 s_w[EXC4_SLOTS];                // e^{j 2 pi f0 / fs}
  const int u = blockIdx.y;
  const int ny = out_len[u];
  const int nt = min(20000, ny);
  const bool tiled = ny > nt;
  const int T = nt - 128;                           // tiling period (plan.h stretch_index)
  const int R = tiled ? T : ny;                     // residues
  const int r0 = blockIdx.x * 1024;
  if(r0 >= R) return;
  const int rho = r0 + 4 * threadIdx.x;
  const int nf = nfrm[u], fo = frm_off[u];
  const float hop = lp::fmul(thop, fs);
  const int half = nwin_env / 2;
  const int n_ext = nt + 128;                       // samples of a template row (k_white)
  // templates of the four residues, every active channel (and, inside the cross-fade, of T + rho ..)
  float tv[NCH][4];
#pragma unroll
  for(int c = 0; c < NCH; c ++) {
#pragma unroll
    for(int q = 0; q < 4; q ++) tv[c][q] = 0.0f;
    if(c < nch_active) {
      const float* tpl = colored + ((size_t)u * nch + c) * ntemplate_ext;
      if(rho + 3 < n_ext) { const f4u v = *(const f4u*)(tpl + rho); tv[c][0] = v.x; tv[c][1] = v.y; tv[c][2] = v.z; tv[c][3] = v.w; }
      else {
#pragma unroll
        for(int q = 0; q < 4; q ++) if(rho + q < n_ext) tv[c][q] = tpl[rho + q];
      }
    }
  }
  const int ntile = tiled ? (ny - r0 + T - 1) / T : 1;   // tiles t with r0 + t T < ny (the same for the whole block)
  for(int t = 0; t < ntile; t ++) {
    const int b0 = r0 + t * T;                      // first output sample of the block in this tile
    const int imin = max(0, (int)((float)b0 / hop) - 1);
    __syncthreads();                                // (the previous tile's readers are done)
    for(int k2 = threadIdx.x; k2 < EXC4_SLOTS * NCH * ME; k2 += 256) {
      const int sl = k2 / (NCH * ME), r = k2 % (NCH * ME), c = r / ME, k = r % ME;
      const int i = imin + sl;
      float2 v = make_float2(0.0f, 0.0f);
      if(i < nf && c < nch && k < me) v = cplx[((size_t)(fo + i) * nch + c) * me + k];
      s_cp[sl][r] = v;
    }
    if(threadIdx.x < EXC4_SLOTS * NCH) {
      const int sl = threadIdx.x / NCH, c = threadIdx.x % NCH, i = imin + sl;
      s_off[sl][c] = (i < nf && c < nch) ? edc[(size_t)(fo + i) * nch + c] : 0.0f;
    }
    if(threadIdx.x < EXC4_SLOTS) {
      const int i = imin + threadIdx.x;
      const float tn = i < nf ? f0[fo + i] / fs : 0.0f;
      s_turn[threadIdx.x] = tn;
      float wr = 1.0f, wi = 0.0f;
      if(tn > 0) cs_turns((double)tn, & wr, & wi);
      s_w[threadIdx.x] = make_float2(wr, wi);
    }
    __syncthreads();
    const int p0 = rho + t * T;
    if(rho >= R || p0 >= ny) continue;
    float e[4][NCH];
#pragma unroll
    for(int q = 0; q < 4; q ++)
#pragma unroll
      for(int c = 0; c < NCH; c ++) e[q][c] = 0.0f;
#pragma unroll 1
    for(int hh = 0; hh < EXC_HITS; hh ++) {                // (not unrolled: one set of amplitude registers, 154 -> VGPRs of one pass)
      int fi = -2, prevj = 0;
      float tn = 0, wr = 1.0f, wi = 0.0f, z1r = 1.0f, z1i = 0.0f;
#pragma unroll
      for(int q = 0; q < 4; q ++) {
        if(rho + q >= R || p0 + q >= ny) continue;
        const int2 hit = hits[(size_t)(p0 + q) * EXC_HITS + hh];
        if(hit.x < 0 || hit.x >= nf) continue;
        const int sl = hit.x - imin, j = hit.y;
        const float w = win[j];
        if(sl >= 0 && sl < EXC4_SLOTS) {
          const bool seed = hit.x != fi;
          if(seed) { fi = hit.x; tn = s_turn[sl]; const float2 wv = s_w[sl]; wr = wv.x; wi = wv.y; }
          if(seed || j != prevj + 1) {
            z1r = 1.0f; z1i = 0.0f;
            if(tn > 0) cs_turns((double)tn * (double)(j - half), & z1r, & z1i);
          } else { const float t1 = z1r * wr - z1i * wi, t2 = z1r * wi + z1i * wr; z1r = t1; z1i = t2; }
          prevj = j;
          float zr[ME], zi[ME];                      // e^{j k th}, k = 1 .. ME
          zr[0] = z1r; zi[0] = z1i;
          // explicit fused multiply-adds (the file is compiled with contraction off: a complex multiply-add written
          // with * and + is four instructions per term, and the 16 terms of a frame are most of this kernel)
#pragma unroll
          for(int k = 1; k < ME; k ++) {
            zr[k] = fmaf(zr[k - 1], z1r, -(zi[k - 1] * z1i)); zi[k] = fmaf(zr[k - 1], z1i, zi[k - 1] * z1r);
          }
#pragma unroll
          for(int c = 0; c < NCH; c ++) {
            float y = s_off[sl][c];
#pragma unroll
            for(int k = 0; k < ME; k += 2) {         // amplitudes are zero beyond nhar_e; two per 16-byte LDS broadcast
              const float4 av = *(const float4*)& s_cp[sl][c * ME + k];
              y = fmaf(av.x, zr[k], y); y = fmaf(-av.y, zi[k], y);
              y = fmaf(av.z, zr[k + 1], y); y = fmaf(-av.w, zi[k + 1], y);
            }
            e[q][c] = fmaf(fmaxf(y, 1e-8f), w, e[q][c]);
          }
        } else {
          // hop shorter than 1024 / (EXC4_SLOTS - 3) samples: frame not staged, read it from HBM
          const int g = fo + hit.x;
          const float f = f0[g];
          float y1r = 1.0f, y1i = 0.0f;
          if(f > 0) cs_turns((double)(f / fs) * (double)(j - half), & y1r, & y1i);
          for(int c = 0; c < nch; c ++) {
            float y = 0.0f, zr = y1r, zi = y1i;
            for(int k = 0; k < me; k ++) {
              const float2 av = cplx[((size_t)g * nch + c) * me + k];
              y += av.x * zr - av.y * zi;
              const float nr = zr * y1r - zi * y1i, ni = zr * y1i + zi * y1r;
              zr = nr; zi = ni;
            }
            const float val = fmaxf(y + edc[(size_t)g * nch + c], 1e-8f) * w;
#pragma unroll
            for(int cc = 0; cc < NCH; cc ++) if(cc == c) e[q][cc] += val;
          }
        }
      }
    }
    // template sample of output p0 + q (plan.h stretch_index with p = rho + q + t T): the residue's own sample, or -- in
    // the first 128 residues of a later tile -- the cross-fade of the template's tail into its head
    const bool fade_zone = t > 0 && rho < 128;
    const bool applied = fade_zone && (t == 1 ? true : ny >= nt + (t - 1) * T);
    float acc[4];
#pragma unroll
    for(int q = 0; q < 4; q ++) {
      const float r = (float)(rho + q) / 128.0f;
      const float xf = applied ? __frsqrt_rn(2.0f * r * (r - 1.0f) + 1.0f) : 1.0f;
      float a = 0;
#pragma unroll
      for(int c = 0; c < NCH; c ++) {
        if(c < nch_active) {
          // (the template's tail T + rho .. is only ever read here: 128 residues of the later tiles)
          float v = fade_zone ? colored[((size_t)u * nch + c) * ntemplate_ext + T + rho + q] : tv[c][q];
          if(applied) {
            v *= 1.0f - r;
            v += tv[c][q] * r;
            v *= xf;
          }
          v *= sqrtf(e[q][c]);
          a += v;
        }
      }
      acc[q] = a;
    }
    float* dst = yexc + (size_t)out_off[u] + p0;
    if(rho + 3 < R && p0 + 3 < ny) *(f4u*)dst = f4u{acc[0], acc[1], acc[2], acc[3]};
    else {
#pragma unroll
      for(int q = 0; q < 4; q ++) if(rho + q < R && p0 + q < ny) dst[q] = acc[q];
    }
  }
}


// =====================================================================
// S4  per-frame spectral noise shaping (HOT LOOP E) -- replaces the frame body
// of llsm_filter_noise, layer0.c:581-619: Hann-windowed excitation frame,
// zero-padded centred FFT, PSD, 7-tap smoothing, target PSD (psd + PSDRES -
// LOG2IN(0.375)) interpolated to the FFT grid, gain, Hermitian completion,
// inverse FFT, 16-sample fades.  Output row g of nframes[F][N]; live[g] = 0
// for frames under the -100 dB floor (layer0.c:584-585).
// Two frames per complex FFT in both directions (see fft_dif / ifft_dit).
// rt != 0: llsmrt.c:441-477 variant -- the frame comes from a per-stream
// excitation buffer (yexc[g][nwin]) instead of the utterance signal.
// =====================================================================
DEV float target_db(const float* __restrict__ prow, const float* __restrict__ rrow, bool hasres,
  int npsd, float fq, float fnyq_conf) {
  // interp1 of (psd [+ PSDRES - LOG2IN(LOGRESBIAS)]) on linspace(0, fnyq_conf, npsd)
  const float pos = fq / fnyq_conf * (float)(npsd - 1);
  int q = (int)floorf(pos);
  if(q >= npsd - 1) return prow[npsd - 1] + (hasres ? rrow[npsd - 1] - 1.6286014f : 0.0f);
  if(q < 0) q = 0;
  const float rr = pos - (float)q;
  const float t0 = prow[q] + (hasres ? rrow[q] - 1.6286014f : 0.0f);
  const float t1 = prow[q + 1] + (hasres ? rrow[q + 1] - 1.6286014f : 0.0f);
  return t0 + (t1 - t0) * rr;
}

// max over the NT threads of a workgroup (one wavefront: shuffles; more: through LDS scratch `red`, NT / 64 floats)
template <int NT>
DEV float block_max(float v, float* red, int tid) {
  v = wave_max(v);
  if(NT == WAVE) return v;
  __syncthreads();
  if((tid & (WAVE - 1)) == 0) red[tid >> 6] = v;
  __syncthreads();
  float m = red[0];
#pragma unroll
  for(int w = 1; w < NT / WAVE; w ++) m = fmaxf(m, red[w]);
  return m;
}

// One frame pair (gg[0], gg[1] >= nframes: absent) by NT threads; X / tw / P: LDS (N, N/2, N/2 + 1 float2), red: NT / 64 floats.
// Every thread of the workgroup must call it (barriers inside).
// lds_frames != NULL (rt): frame e of the pair is lds_frames + e * nwin instead of yexc + g * nwin.
// nframes_out == NULL: the filtered frames stay in X (x: frame 0, y: frame 1; unscaled, unfaded -- the caller applies
// rt_noise_sample) and nothing is written to `live`.  Returns bit e set when frame e was filtered.
template <int NT>
DEV int noise_filter_pair(const int (&gg)[2], int tid, float2* X, const float2* tw, float2* P, float* red, float2* Tdb,
  const float* __restrict__ yexc, const int* __restrict__ out_off, const int* __restrict__ out_len,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off, int nframes,
  const float* __restrict__ psd, const float* __restrict__ psdres,
  const int* __restrict__ has_psdres, int npsd, float fnyq_conf,
  float thop, float fs, int nwin, const float* __restrict__ win, float inv_wsqr,
  int N, int logN, float* __restrict__ nframes_out, int* __restrict__ live, int rt, const float* lds_frames = nullptr,
  const float* pk_ready = nullptr, int tws = 1) {
  // tws: stride of the twiddle table (1: the LDS copy of N / 2 entries; tw_nmax / N: the global table, big transforms)
  // pk_ready != NULL (k_rt_hop2): Tdb is filled already and the largest level of frame e is the maximum of
  // pk_ready[4 e .. 4 e + 3] -- the level rows are not read again
  const int lane = tid;
  const int nspec = N / 2 + 1;
  const int nfade = 16;
  const float fn_syn = fs / 2.0f;
  const float invN = 1.0f / (float)N;
  bool alive[2]; const float* xs[2]; int nxu[2], base[2];
#pragma unroll
  for(int e = 0; e < 2; e ++) {
    const int g = gg[e];
    alive[e] = false; xs[e] = yexc; nxu[e] = 0; base[e] = 0;
    float pk = -3.0e38f;
    if(pk_ready) pk = fmaxf(fmaxf(pk_ready[4 * e], pk_ready[4 * e + 1]), fmaxf(pk_ready[4 * e + 2], pk_ready[4 * e + 3]));
    else if(g < nframes) {
      const float* prow = psd + (size_t)g * npsd;
      const float* rrow = psdres + (size_t)g * npsd;
      const bool hr = has_psdres[g] != 0;
      for(int j = lane; j < npsd; j += NT) {
        const float t = prow[j];
        pk = fmaxf(pk, t);
        // Tdb (LDS, npsd float2): the target level psd [+ PSDRES - LOG2IN(LOGRESBIAS)] of both frames, read once
        if(Tdb) { const float v = t + (hr ? rrow[j] - 1.6286014f : 0.0f); if(e == 0) Tdb[j].x = v; else Tdb[j].y = v; }
      }
    } else if(Tdb && e == 1)
      for(int j = lane; j < npsd; j += NT) Tdb[j].y = Tdb[j].x;
    if(! pk_ready) pk = block_max<NT>(pk, red, tid);   // (uniform control flow: every thread gets here)
    if(g >= nframes) continue;
    alive[e] = !(pk < -100.0f);
    if(lane == 0 && nframes_out) live[g] = alive[e] ? 1 : 0;
    if(rt) { xs[e] = lds_frames ? lds_frames + (size_t)e * nwin : yexc + (size_t)g * nwin; nxu[e] = nwin; base[e] = 0; }
    else {
      int u, i; frame_owner(frm_utt, frm_off, g, & u, & i);
      xs[e] = yexc + out_off[u]; nxu[e] = out_len[u];
      base[e] = lp::center(i, thop, fs) - nwin / 2;
    }
  }
  if(! alive[0] && ! alive[1]) return 0;
  const int shift = N / 2 - nwin / 2;              // x_re[j - nwin/2 + nfft/2]
  for(int t0 = lane; t0 < N; t0 += NT * 8) {
    float va[8], vb[8], wv[8];
#pragma unroll
    for(int q8 = 0; q8 < 8; q8 ++) {
      const int j = t0 + q8 * NT - shift;
      const bool in = j >= 0 && j < nwin;
      const int ia = base[0] + j, ib = base[1] + j;
      wv[q8] = in ? win[j] : 0.0f;
      va[q8] = (in && alive[0] && ia >= 0 && ia < nxu[0]) ? xs[0][ia] : 0.0f;
      vb[q8] = (in && alive[1] && ib >= 0 && ib < nxu[1]) ? xs[1][ib] : 0.0f;
    }
#pragma unroll
    for(int q8 = 0; q8 < 8; q8 ++) {
      const int t = t0 + q8 * NT;
      if(t < N) X[t] = make_float2(va[q8] * wv[q8], vb[q8] * wv[q8]);
    }
  }
  __syncthreads();
  RT2_T(9);
  fft_dif<NT>(X, tw, tws, N, logN, lane);
  RT2_T(10);
  for(int k = lane; k < nspec; k += NT) {
    float2 A, B; unpack_pair(X, N, logN, k, & A, & B);
    P[k] = make_float2((A.x * A.x + A.y * A.y) * inv_wsqr, (B.x * B.x + B.y * B.y) * inv_wsqr);
  }
  __syncthreads();
  RT2_T(11);
  const float* prow0 = psd + (size_t)gg[0] * npsd;
  const float* rrow0 = psdres + (size_t)gg[0] * npsd;
  const bool hr0 = has_psdres[gg[0]] != 0;
  const int g1 = alive[1] ? gg[1] : gg[0];
  const float* prow1 = psd + (size_t)g1 * npsd;
  const float* rrow1 = psdres + (size_t)g1 * npsd;
  const bool hr1 = has_psdres[g1] != 0;
  // filtered spectra, recombined as Ya + j Yb, written back over the bin pair (k, N-k): a thread reads and writes
  // only its own pair of bins
  for(int k0 = 0; k0 < nspec - 1; k0 += NT) {
    const int k = k0 + lane;
    const bool on = k < nspec - 1;
    float2 A = make_float2(0, 0), B = make_float2(0, 0);
    if(on) {
      const int mh = g_conv.mavg_half;
      const int lo = max(0, k - mh), hi = min(nspec - 1, k + mh);
      float ea = 0, eb = 0;
      for(int q = lo; q <= hi; q ++) { const float2 pv = P[q]; ea += pv.x; eb += pv.y; }
      const float inv = 1.0f / (float)(hi - lo + 1);
      ea *= inv; eb *= inv;
      const float fq = (float)k * fn_syn / (float)(nspec - 1);
      float ta, tb;
      if(Tdb) {                                      // interp1 on linspace(0, fnyq_conf, npsd), as target_db
        const float pos = fq / fnyq_conf * (float)(npsd - 1);
        int q = (int)floorf(pos);
        if(q >= npsd - 1) { const float2 t = Tdb[npsd - 1]; ta = t.x; tb = t.y; }
        else {
          if(q < 0) q = 0;
          const float rr = pos - (float)q;
          const float2 t0 = Tdb[q], t1 = Tdb[q + 1];
          ta = t0.x + (t1.x - t0.x) * rr; tb = t0.y + (t1.y - t0.y) * rr;
        }
      } else {
        ta = target_db(prow0, rrow0, hr0, npsd, fq, fnyq_conf); tb = target_db(prow1, rrow1, hr1, npsd, fq, fnyq_conf);
      }
      const float Ha = expf(ta * (2.3025851f / 20.0f)) / sqrtf(ea * 44100.0f / fs + 1e-8f);
      const float Hb = expf(tb * (2.3025851f / 20.0f)) / sqrtf(eb * 44100.0f / fs + 1e-8f);
      unpack_pair(X, N, logN, k, & A, & B);
      A.x *= Ha; A.y *= Ha; B.x *= Hb; B.y *= Hb;
      if(k == 0) { A.y = 0; B.y = 0; }              // real signals: DC bin is real
    }
    if(on) {
      // Ya[k] + j Yb[k]  and  conj(Ya[k]) + j conj(Yb[k]) at the mirror bin
      X[brevN(k, logN)] = make_float2(A.x - B.y, A.y + B.x);
      if(k > 0) X[brevN(N - k, logN)] = make_float2(A.x + B.y, -A.y + B.x);
      if(k == nspec - 2)                             // x[nspec-1] = x[nspec-2] (layer0.c:611-612);
        X[brevN(nspec - 1, logN)] = make_float2(A.x, B.x);   // only its real part reaches the output
    }
  }
  __syncthreads();
  RT2_T(12);
  ifft_dit<NT>(X, tw, tws, N, logN, lane);
  const int mask = (alive[0] ? 1 : 0) | (alive[1] ? 2 : 0);
  if(! nframes_out) return mask;                     // (ifft_dit ends with a barrier)
#pragma unroll
  for(int e = 0; e < 2; e ++) {
    if(! alive[e]) continue;
    float* out = nframes_out + (size_t)gg[e] * N;
    for(int t = lane; t < N; t += NT) {
      float v = (e == 0 ? X[t].x : X[t].y) * invN;
      if(t < nfade) v *= (float)t / (float)nfade;
      if(t >= N - nfade) v *= 1.0f - (float)(N - 1 - t) / (float)nfade;
      out[t] = v;
    }
  }
  __syncthreads();
  return mask;
}
// sample t of a filtered frame as noise_filter_pair stores it: 1 / N and the 16-sample fades at both ends
DEV float rt_noise_sample(float x, int t, int N) {
  const int nfade = 16;
  float v = x * (1.0f / (float)N);
  if(t < nfade) v *= (float)t / (float)nfade;
  if(t >= N - nfade) v *= 1.0f - (float)(N - 1 - t) / (float)nfade;
  return v;
}

__global__ __launch_bounds__(WAVE) void k_noise_filter(
  const float* __restrict__ yexc, const int* __restrict__ out_off, const int* __restrict__ out_len,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off, int nframes,
  const float* __restrict__ psd, const float* __restrict__ psdres,
  const int* __restrict__ has_psdres, int npsd, float fnyq_conf,
  float thop, float fs, int nwin, const float* __restrict__ win, float inv_wsqr,
  int N, int logN, const float2* __restrict__ tw_glob, int tw_nmax,
  float* __restrict__ nframes_out, int* __restrict__ live, int rt,
  const int2* __restrict__ pairs, int npair, float2* __restrict__ gscr) {
  // gscr != NULL (transforms beyond the LDS: N > 8192): X and P of this workgroup live in global scratch
  // (N + N / 2 + 1 float2 per workgroup, L2-resident) and the twiddles are read from the global table
  const int lane = threadIdx.x;
  float2* X = gscr ? gscr + (size_t)blockIdx.x * (N + N / 2 + 1) : (float2*)g_lds;
  float2* twl = (float2*)g_lds + N;
  float2* P = gscr ? X + N : twl + N / 2;             // nspec (PSD of frame a, frame b)
  float* red = gscr ? (float*)g_lds : (float*)(P + N / 2 + 1);
  const float2* tw = gscr ? tw_glob : twl;
  const int tws = gscr ? tw_nmax / N : 1;
  if(! gscr) load_twiddles(twl, tw_glob, N, tw_nmax, lane);
  const int wgx = xcd_frame(blockIdx.x, gridDim.x);
  const int per = (npair + gridDim.x - 1) / gridDim.x;
  for(int p = wgx * per; p < min(npair, (wgx + 1) * per); p ++) {
    int gg[2];
    pair_of(pairs, p, nframes, gg[0], gg[1]);
    noise_filter_pair<WAVE>(gg, lane, X, tw, P, red, nullptr, yexc, out_off, out_len, frm_utt, frm_off, nframes, psd, psdres, has_psdres,
      npsd, fnyq_conf, thop, fs, nwin, win, inv_wsqr, N, logN, nframes_out, live, rt, nullptr, nullptr, tws);
  }
}

// S4 on the register-resident wavefront FFT (N = 2^LOGN): same arithmetic as k_noise_filter;
// the frame pair stays in registers, LDS carries the transform exchanges and the padded
// power spectrum the 7-bin smoother reads (aliased with the exchange buffer).
#define NF_WPE 2                                   // 256 VGPRs: the 1024-point transform needs them
template <int LOGN>
__global__ __launch_bounds__(WAVE, (LOGN >= 11 ? 1 : NF_WPE)) void k_noise_filter_wf(
  const float* __restrict__ yexc, const int* __restrict__ out_off, const int* __restrict__ out_len,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off, int nframes,
  const float* __restrict__ psd, const float* __restrict__ psdres,
  const int* __restrict__ has_psdres, int npsd, float fnyq_conf,
  float thop, float fs, int nwin, const float* __restrict__ win, float inv_wsqr,
  float* __restrict__ nframes_out, int* __restrict__ live, int rt,
  const int2* __restrict__ pairs, int npair) {
  constexpr int N = 1 << LOGN, P = N / WAVE, H = P / 2, nspec = N / 2 + 1;
  const int lane = threadIdx.x;
  float2* lds = (float2*)g_lds;
  float2* Pw = lds;                                  // nspec + 6 entries, bin k at Pw[k + 3]
  float2* Tdb = lds + wf_lds_elems<LOGN>();          // target level (dB) of frames a, b on the PSD grid
  WfTw<LOGN> tw; wf_init(tw, lane);
  const int nfade = 16;
  const float fn_syn = fs / 2.0f;
  const float invN = 1.0f / (float)N;
  // the analysis window is the same for every frame pair this wavefront walks: load it once
  const int shift = N / 2 - nwin / 2;                // x_re[j - nwin/2 + nfft/2]
  float* Wl = (float*)(Tdb + npsd);                  // ... and kept in LDS (16 registers fewer: no spills)
#pragma unroll
  for(int m = 0; m < P; m ++) Wl[lane + WAVE * m] = ld_guard(win, lane + WAVE * m - shift, nwin, true);
  const int wgx = xcd_frame(blockIdx.x, gridDim.x);
  const int per = (npair + gridDim.x - 1) / gridDim.x;
  const int p_end = min(npair, (wgx + 1) * per);
  // Per-pair metadata (wave-uniform: scalar loads).  The metadata of pair p + 1 is fetched while
  // pair p computes, so that a pair starts with ONE round of independent vector loads (signal
  // samples and PSD rows) instead of a psd -> liveness -> owner -> offsets -> samples chain.
  struct Meta { size_t off[2]; int nxu[2], base[2], hr[2], g[2]; bool valid[2]; };
  auto pair_meta = [&](int p) {
    Meta M;
    M.g[0] = M.g[1] = nframes;
    if(p < p_end) pair_of(pairs, p, nframes, M.g[0], M.g[1]);
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      const int g = M.g[e];
      M.valid[e] = p < p_end && g < nframes;
      const int gc = M.valid[e] ? g : 0;
      M.hr[e] = has_psdres[gc];
      if(rt) { M.off[e] = (size_t)gc * nwin; M.nxu[e] = nwin; M.base[e] = 0; }
      else {
        int u, i; frame_owner(frm_utt, frm_off, gc, & u, & i);
        M.off[e] = out_off[u]; M.nxu[e] = out_len[u];
        M.base[e] = lp::center(i, thop, fs) - nwin / 2;
      }
      if(! M.valid[e]) M.nxu[e] = 0;
    }
    return M;
  };
  Meta nxt = pair_meta(wgx * per);
  for(int p = wgx * per; p < p_end; p ++) {
    const Meta cur = nxt;
    const int gg[2] = {cur.g[0], cur.g[1]};
    float xr[P], xi[P];
#pragma unroll
    for(int m = 0; m < P; m ++) {
      const int j = lane + WAVE * m - shift;
      const bool in = j >= 0 && j < nwin;
      xr[m] = ld_guard(yexc + cur.off[0], cur.base[0] + j, cur.nxu[0], in);
      xi[m] = ld_guard(yexc + cur.off[1], cur.base[1] + j, cur.nxu[1], in);
    }
    // psd [+ PSDRES - LOG2IN(LOGRESBIAS)] of both frames -> LDS; the peak of psd decides liveness
    float pk0 = -3.0e38f, pk1 = -3.0e38f;
    {
      const size_t r0 = (size_t)gg[0] * npsd, r1 = (size_t)(cur.valid[1] ? gg[1] : gg[0]) * npsd;
      for(int j = lane; j < npsd; j += WAVE) {
        float t0 = psd[r0 + j], t1 = psd[r1 + j];
        pk0 = fmaxf(pk0, t0); pk1 = fmaxf(pk1, t1);
        if(cur.hr[0]) t0 += psdres[r0 + j] - 1.6286014f;
        if(cur.hr[1]) t1 += psdres[r1 + j] - 1.6286014f;
        Tdb[j] = make_float2(t0, t1);
      }
    }
    nxt = pair_meta(p + 1);
    pk0 = wave_max(pk0); pk1 = wave_max(pk1);
    const bool alive[2] = {!(pk0 < -100.0f), cur.valid[1] && !(pk1 < -100.0f)};
    if(lane == 0) { live[gg[0]] = alive[0] ? 1 : 0; if(cur.valid[1]) live[gg[1]] = alive[1] ? 1 : 0; }
    if(! alive[0] && ! alive[1]) { __syncthreads(); continue; }
#pragma unroll
    for(int m = 0; m < P; m ++) {
      const float w = Wl[lane + WAVE * m];
      xr[m] *= alive[0] ? w : 0.0f; xi[m] *= alive[1] ? w : 0.0f;
    }
    wave_fft<LOGN>(xr, xi, tw, lds, lane);
    float mr[H + 1], mi[H + 1];
    wave_mirror_lo<P>(xr, mr, lane);
    wave_mirror_lo<P>(xi, mi, lane);
    // spectra A, B of the two frames for the bins k <= N/2, in place: A -> (xr, xi), B -> (mr, mi);
    // their powers go to LDS, zero padded by 3 on both sides, for the 7-bin smoother
    if(lane < 3) { Pw[lane] = make_float2(0.0f, 0.0f); Pw[nspec + 3 + lane] = make_float2(0.0f, 0.0f); }
#pragma unroll
    for(int m = 0; m <= H; m ++) {
      const float ar = 0.5f * (xr[m] + mr[m]), ai = 0.5f * (xi[m] - mi[m]);
      const float br = 0.5f * (xi[m] + mi[m]), bi = -0.5f * (xr[m] - mr[m]);
      xr[m] = ar; xi[m] = ai; mr[m] = br; mi[m] = bi;
      if(m < H || lane == 0)
        Pw[3 + lane + WAVE * m] = make_float2((ar * ar + ai * ai) * inv_wsqr, (br * br + bi * bi) * inv_wsqr);
    }
    __syncthreads();
    // interp1 of the target on linspace(0, fnyq_conf, npsd) at fq = k fn_syn / (nspec - 1)
    const float cpos = fn_syn / ((float)(nspec - 1) * fnyq_conf) * (float)(npsd - 1);
    const int mavg_h = g_conv.mavg_half;                      // half width of the periodogram smoother (moving_avg)
    const float esc = 44100.0f / fs;
    // bins k < N/2: gain = target / smoothed source; Z[k] = Ya + j Yb stays here, the
    // conjugate-symmetric Z[N - k] is parked in (mr, mi) for the lane that owns that bin
    float nyq_r = 0.0f, nyq_i = 0.0f;
    int lv = lane;                                   // opaque per pair: the per-bin grid positions and
    asm volatile("" : "+v"(lv));                     // weights are recomputed, not hoisted (and spilled)
#pragma unroll
    for(int m = 0; m < H; m ++) {
      const int k = lv + WAVE * m;
      float ea = 0, eb = 0;
#pragma unroll
      for(int q = 0; q < 7; q ++) { const float2 pv = Pw[k + q]; const bool in = abs(q - 3) <= mavg_h; ea += in ? pv.x : 0.0f; eb += in ? pv.y : 0.0f; }
      const int lo = max(0, k - mavg_h), hi = min(nspec - 1, k + mavg_h);
      const float inv = 1.0f / (float)(hi - lo + 1);
      ea *= inv; eb *= inv;
      const float pos = (float)k * cpos;
      int q = (int)pos;                              // pos >= 0
      float ta, tb;
      if(q >= npsd - 1) { const float2 t = Tdb[npsd - 1]; ta = t.x; tb = t.y; }
      else {
        const float rr = pos - (float)q;
        const float2 t0 = Tdb[q], t1 = Tdb[q + 1];
        ta = t0.x + (t1.x - t0.x) * rr; tb = t0.y + (t1.y - t0.y) * rr;
      }
      // 10^(t/20) / sqrt(e 44100/fs + 1e-8): hardware exp2 / rsq (1 ulp), the argument scaling
      // costs |t| 7e-9 relative -- inside the stated 1e-4 synthesis tolerance by three orders
      const float ha = __expf(ta * (2.3025851f / 20.0f)) * __frsqrt_rn(fmaf(ea, esc, 1e-8f));
      const float hb = __expf(tb * (2.3025851f / 20.0f)) * __frsqrt_rn(fmaf(eb, esc, 1e-8f));
      float ar = xr[m] * ha, ai = xi[m] * ha, br = mr[m] * hb, bi = mi[m] * hb;
      if(m == 0 && lane == 0) { ai = 0.0f; bi = 0.0f; }           // real signals: DC bin is real
      if(m == H - 1) { nyq_r = __shfl(ar, WAVE - 1, WAVE); nyq_i = __shfl(br, WAVE - 1, WAVE); }
      xr[m] = ar - bi; xi[m] = ai + br;
      mr[m] = ar + bi; mi[m] = br - ai;
    }
    __syncthreads();
    // x[nspec-1] = x[nspec-2] (layer0.c:611-612); only its real part reaches the output
    if(lane == 0) { xr[H] = nyq_r; xi[H] = nyq_i; }
    wave_reflect<P>(mr, xr, lane);
    wave_reflect<P>(mi, xi, lane);
    wave_fft<LOGN>(xi, xr, tw, lds, lane);           // inverse (x N): frame a in xr, frame b in xi
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      if(! alive[e]) continue;
      float* out = nframes_out + (size_t)gg[e] * N;
#pragma unroll
      for(int m = 0; m < P; m ++) {
        const int t = lane + WAVE * m;
        float v = (e == 0 ? xr[m] : xi[m]) * invN;
        if(m == 0 && t < nfade) v *= (float)t / (float)nfade;
        if(m == P - 1 && t >= N - nfade) v *= 1.0f - (float)(N - 1 - t) / (float)nfade;
        out[t] = v;
      }
    }
  }
}

// S4 + the noise half of S5 fused (offline path, N <= 2048; one wavefront per SIMD at 2048): the shaped frames never reach HBM.
// A wavefront owns one UNIT = frames [i0, i1) of one utterance (i0 even) and the output samples
// [lo(i0), lo(i1)), lo(i) = start of frame i's N-sample output window (0 / ny at the utterance
// ends).  It walks the frame pairs from `halo` frames before i0 (every earlier frame that still
// reaches into its samples), overlap-adds each shaped frame into an N-sample ring in LDS and
// writes a sample out as soon as the next frame starts beyond it.  Every sample is therefore
// summed by ONE wavefront in ascending frame order -- the order of the reference's sequential
// loop (layer0.c:620-624) -- and written exactly once; the halo frames are computed twice
// (halo / unit length of extra work) instead of exchanging partial sums between wavefronts.
// Frame pairs are (i, i + 1) with i even WITHIN the utterance, so the result of an utterance does
// not depend on where it sits in the batch.
// Round 6: three wavefronts per SIMD for N <= 1024 (round 5: SQ_WAIT_ANY = 49 % of the wave-cycles at two, VALU busy 63 %).
// What that needed: <= 168 VGPRs and <= 13 KB of LDS per wavefront, so (a) the analysis window lives in registers (it is
// read at lane + 64 m only), (b) the target rows ride in registers through the forward transform and land INSIDE the
// exchange buffer behind the power spectrum (ALIAS; rows of up to NF_TQ * 64 points, else the round-5 placement),
// (c) the periodogram smoother has a 7-tap form (mavg_half = 3, the default convention) without the general loop's
// selects, whose sums keep that loop's order (bit-identical), the 1 / count factor a constant away from the spectrum's ends.
#ifndef NF_ABL
#define NF_ABL 0                                     // timing ablations (tools/kbench.py): 1 no sample loads, 2 no forward / 8 no inverse transform, 4 no gain loop, 16 no overlap-add
#endif
#ifndef NF_OLA_WPE
#define NF_OLA_WPE 2                                 // 3: 168 registers -> ~90 spilled, 1.49 ms against 0.85 (profiles/r06_a_*)
#endif
#define NF_TQ 4                                      // most target-row values per lane held in registers (ALIAS form): npsd <= 256; TQ = 2 up to 128 points
template <int LOGN, int MH>                          // MH = 3: the 7-tap smoother (default convention); 0: general (mavg_h <= 3)
DEV void nf_gain_loop(float (&xr)[(1 << LOGN) / WAVE], float (&xi)[(1 << LOGN) / WAVE],
  float (&mr)[(1 << LOGN) / WAVE / 2 + 1], float (&mi)[(1 << LOGN) / WAVE / 2 + 1],
  const float2* Pw, const float2* Tdb, int npsd, float cpos, float esc, int mavg_h, int lane, float& nyq_r, float& nyq_i) {
  constexpr int N = 1 << LOGN, P = N / WAVE, H = P / 2, nspec = N / 2 + 1;
  int lv = lane;                                     // opaque per pair: the per-bin grid positions and
  asm volatile("" : "+v"(lv));                       // weights are recomputed, not hoisted (and spilled)
#pragma unroll
  for(int m = 0; m < H; m ++) {
    const int k = lv + WAVE * m;
    float ea = 0, eb = 0;
    if constexpr (MH == 3) {
      // bins k - 3 .. k + 3 (zero beyond the ends), summed in ascending order like the general loop; the
      // count is 7 except at the three lowest and the two highest bins of the half spectrum
#pragma unroll
      for(int q = 0; q < 7; q ++) { const float2 pv = lds_rd64(Pw + k + q); ea += pv.x; eb += pv.y; }
      float inv = 1.0f / 7.0f;
      if(m == 0 || m == H - 1) { const int lo = max(0, k - 3), hi = min(nspec - 1, k + 3); inv = 1.0f / (float)(hi - lo + 1); }
      ea *= inv; eb *= inv;
    } else {
#pragma unroll
      for(int q = 0; q < 7; q ++) { const float2 pv = lds_rd64(Pw + k + q); const bool in = abs(q - 3) <= mavg_h; ea += in ? pv.x : 0.0f; eb += in ? pv.y : 0.0f; }
      const int lo = max(0, k - mavg_h), hi = min(nspec - 1, k + mavg_h);
      const float inv = 1.0f / (float)(hi - lo + 1);
      ea *= inv; eb *= inv;
    }
    const float pos = (float)k * cpos;
    int q = (int)pos;                                // pos >= 0
    float ta, tb;
    // (a branch-free form of this -- clamped points and a select, no exec branch per register -- let the compiler fit the
    //  kernel into 167 registers and made it 20 % SLOWER, 0.59 -> 0.71 ms: round 6, visit v30)
    if(q >= npsd - 1) { const float2 t = Tdb[npsd - 1]; ta = t.x; tb = t.y; }
    else {
      const float rr = pos - (float)q;
      const float2 t0 = Tdb[q], t1 = Tdb[q + 1];
      ta = t0.x + (t1.x - t0.x) * rr; tb = t0.y + (t1.y - t0.y) * rr;
    }
    // 10^(t/20) / sqrt(e 44100/fs + 1e-8): hardware exp2 / rsq (1 ulp), the argument scaling
    // costs |t| 7e-9 relative -- inside the stated 1e-4 synthesis tolerance by three orders
    const float ha = __expf(ta * (2.3025851f / 20.0f)) * __frsqrt_rn(fmaf(ea, esc, 1e-8f));
    const float hb = __expf(tb * (2.3025851f / 20.0f)) * __frsqrt_rn(fmaf(eb, esc, 1e-8f));
    float ar = xr[m] * ha, ai = xi[m] * ha, br = mr[m] * hb, bi = mi[m] * hb;
    if(m == 0 && lane == 0) { ai = 0.0f; bi = 0.0f; }           // real signals: DC bin is real
    if(m == H - 1) { nyq_r = __shfl(ar, WAVE - 1, WAVE); nyq_i = __shfl(br, WAVE - 1, WAVE); }
    // Z[k] = Ya + j Yb stays here, the conjugate-symmetric Z[N - k] is parked in (mr, mi) for the lane that owns that bin
    xr[m] = ar - bi; xi[m] = ai + br;
    mr[m] = ar + bi; mi[m] = br - ai;
  }
}

template <int LOGN, bool ALIAS, int TQ>
__global__ __launch_bounds__(WAVE, (LOGN >= 11 ? 1 : (ALIAS ? NF_OLA_WPE : NF_WPE))) void k_noise_filter_ola(
  const int4* __restrict__ units, int nunits, int halo,
  const float* __restrict__ yexc, const int* __restrict__ out_off, const int* __restrict__ out_len,
  const int* __restrict__ frm_off, const int* __restrict__ nfrm,
  const float* __restrict__ psd, const float* __restrict__ psdres,
  const int* __restrict__ has_psdres, int npsd, float fnyq_conf,
  float thop, float fs, int nwin, const float* __restrict__ win, int wsym, float inv_wsqr,
  float* __restrict__ ynoise) {
  constexpr int N = 1 << LOGN, P = N / WAVE, H = P / 2, nspec = N / 2 + 1;
  const int lane = threadIdx.x;
  float2* lds = (float2*)g_lds;
  float2* Pw = lds;                                  // nspec + 6 entries, bin k at Pw[k + 3]
  // target level (dB) of frames a, b on the PSD grid: behind the power spectrum inside the exchange buffer (ALIAS: free
  // between the two transforms, which is when it is read) or behind the exchange buffer
  float2* Tdb = ALIAS ? lds + (nspec + 6) : lds + wf_lds_elems<LOGN>();
  float* ring = ALIAS ? (float*)(lds + wf_lds_elems<LOGN>()) : (float*)(Tdb + npsd);   // overlap-add accumulator, sample s at ring[s & (N - 1)]
  WfTw<LOGN> tw; wf_init(tw, lane);
  const int nfade = 16;
  const float fn_syn = fs / 2.0f;
  const float invN = 1.0f / (float)N;
  const int shift = N / 2 - nwin / 2;                // x_re[j - nwin/2 + nfft/2]
  // analysis window, sample lane + 64 m of the padded frame.  Two wavefronts per SIMD: in registers.  ALIAS (three): its
  // first half in LDS behind the ring -- win[j] == win[wsym - j] exactly (engine.cpp make_hann mirrors the table) --, 1 KB
  // instead of 16 registers
  float wv[ALIAS ? 1 : P];
  float* Wh = ring + N;
  const int whalf = wsym / 2;
  if constexpr (ALIAS) { for(int j = lane; j <= whalf; j += WAVE) Wh[j] = j < nwin ? win[j] : 0.0f; }
#pragma unroll
  for(int m = 0; m < P; m ++) {
    if constexpr (! ALIAS) wv[m] = ld_guard(win, lane + WAVE * m - shift, nwin, true);
    ring[lane + WAVE * m] = 0.0f;
  }
  const int4 unit = units[xcd_frame(blockIdx.x, gridDim.x)];
  const int u = unit.x, i0 = unit.y, i1 = unit.z;
  const int nf = nfrm[u], fo = frm_off[u], ny = out_len[u];
  const size_t yo = (size_t)out_off[u];
  const float* xs = yexc + yo;
  float* yn = ynoise + yo;
  // frame i adds its sample t at output sample center(i) - N/2 + t
  const int own_lo = i0 == 0 ? 0 : min(max(lp::center(i0, thop, fs) - N / 2, 0), ny);
  const int own_hi = i1 >= nf ? ny : min(max(lp::center(i1, thop, fs) - N / 2, 0), ny);
  const int j0 = max(0, i0 - halo) & ~1;
  int flushed = lp::center(j0, thop, fs) - N / 2;    // the ring holds samples [flushed, flushed + N)
  // samples [flushed, target) are complete: write the owned ones, clear their ring slots
  auto advance = [&](int target) {
    for(int s = flushed + lane; s < target; s += WAVE) {
      const float v = ring[s & (N - 1)];
      ring[s & (N - 1)] = 0.0f;
      if(s >= own_lo && s < own_hi) yn[s] = v;
    }
    flushed = max(flushed, target);
  };
  const float cpos = fn_syn / ((float)(nspec - 1) * fnyq_conf) * (float)(npsd - 1);
  const int mavg_h = g_conv.mavg_half;               // half width of the periodogram smoother (moving_avg)
  const float esc = 44100.0f / fs;
  // Every global load of a pair -- its 2 P samples per lane and (ALIAS) its target rows -- is ISSUED one pair ahead, right after
  // the previous pair's inverse transform and before its overlap-add, and CONSUMED at the top of the pair's own turn: the
  // registers that hold them are live only across the overlap-add (the transforms' peak register need is untouched) and no
  // turn waits for HBM.  Round 5's loop loaded at the top of the turn and read the target rows behind two branches per
  // value (q < npsd, then has_psdres): four dependent HBM latencies per pair, which the timing ablations of round 6
  // (tools/kbench.py NF_ABL: all arithmetic removed, 0.43 of 0.78 ms left) showed to be half of the kernel.
  float sxr[P], sxi[P];                              // staged samples of the pair about to be processed
  float sp0[TQ], sp1[TQ], sr0[TQ], sr1[TQ];          // staged psd / PSDRES row values (ALIAS), point lane + 64 i
  int hr_stg[2] = {0, 0};                            // has_psdres of the staged pair
  auto stage = [&](int jn, const int (&hrn)[2]) {     // (called for jn >= i1 too, with empty ranges: a conditional call would keep the
    const bool v0 = jn < i1, v1 = jn + 1 < i1;        //  old values alive through the whole turn)
    const int c0 = lp::center(jn, thop, fs), c1 = lp::center(jn + 1, thop, fs);
    const int b0 = c0 - nwin / 2, b1 = c1 - nwin / 2;               // window sample w at signal sample b + w
    const int lo0 = max(b0, 0), lo1 = max(b1, 0);
    const buf_t r0 = buf_range(xs, lo0, v0 ? min(b0 + nwin, ny) : lo0);
    const buf_t r1 = buf_range(xs, lo1, v1 ? min(b1 + nwin, ny) : lo1);
    // byte offsets as (per-pair base) + 256 m: the bases are opaque, so the 16 per-register offsets lane + 64 m - shift are
    // not kept as loop invariants (16 VGPRs that the transforms need)
    int o0 = (lane - shift + b0 - lo0) * 4, o1 = (lane - shift + b1 - lo1) * 4;
    asm volatile("" : "+v"(o0), "+v"(o1));
#pragma unroll
    for(int m = 0; m < P; m ++) {
#if NF_ABL & 1
      sxr[m] = (float)(o0 + m + jn) * 1e-3f; sxi[m] = (float)(o1 + m - jn) * 1e-3f; (void)r0; (void)r1;
#else
      sxr[m] = ld_range_b(r0, o0 + 4 * WAVE * m);
      sxi[m] = ld_range_b(r1, o1 + 4 * WAVE * m);
#endif
    }
    if constexpr (ALIAS) {
      // rows through range-checked descriptors: points beyond npsd, and the PSDRES row of a frame without one, read as 0
      const size_t g0 = (size_t)(fo + jn) * npsd, g1 = (size_t)(fo + (v1 ? jn + 1 : jn)) * npsd;
      const int np = v0 ? npsd : 0;
      const buf_t p0 = buf_range(psd + g0, 0, np), p1 = buf_range(psd + g1, 0, np);
      const buf_t q0 = buf_range(psdres + g0, 0, hrn[0] ? np : 0), q1 = buf_range(psdres + g1, 0, hrn[1] ? np : 0);
#pragma unroll
      for(int i = 0; i < TQ; i ++) {
        const int q = lane + WAVE * i;
        sp0[i] = ld_range(p0, q); sp1[i] = ld_range(p1, q);
        sr0[i] = ld_range(q0, q); sr1[i] = ld_range(q1, q);
      }
    }
    hr_stg[0] = hrn[0]; hr_stg[1] = hrn[1];
  };
  int hr_nxt[2];
#pragma unroll
  for(int e = 0; e < 2; e ++) hr_nxt[e] = has_psdres[fo + min(j0 + e, nf - 1)];
  stage(j0, hr_nxt);
  for(int j = j0; j < i1; j += 2) {
    const bool valid1 = j + 1 < i1;
    const int gg[2] = {fo + j, fo + (valid1 ? j + 1 : j)};
    const int cen[2] = {lp::center(j, thop, fs), lp::center(j + 1, thop, fs)};
    const int hr[2] = {hr_stg[0], hr_stg[1]};
    float xr[P], xi[P];
#pragma unroll
    for(int m = 0; m < P; m ++) { xr[m] = sxr[m]; xi[m] = sxi[m]; }
    // psd [+ PSDRES - LOG2IN(LOGRESBIAS)] of both frames (-> LDS now, or after the forward transform); the peak of psd decides liveness
    float pk0 = -3.0e38f, pk1 = -3.0e38f;
    float tq0[TQ], tq1[TQ];
    {
      if constexpr (ALIAS) {
#pragma unroll
        for(int i = 0; i < TQ; i ++) {
          const int q = lane + WAVE * i;
          float t0 = -3.0e38f, t1 = -3.0e38f;
          if(q < npsd) {
            t0 = sp0[i]; t1 = sp1[i];
            pk0 = fmaxf(pk0, t0); pk1 = fmaxf(pk1, t1);
            if(hr[0]) t0 += sr0[i] - 1.6286014f;
            if(hr[1]) t1 += sr1[i] - 1.6286014f;
          }
          tq0[i] = t0; tq1[i] = t1;
        }
      } else {
        const size_t r0 = (size_t)gg[0] * npsd, r1 = (size_t)gg[1] * npsd;
        for(int q = lane; q < npsd; q += WAVE) {
          float t0 = psd[r0 + q], t1 = psd[r1 + q];
          pk0 = fmaxf(pk0, t0); pk1 = fmaxf(pk1, t1);
          if(hr[0]) t0 += psdres[r0 + q] - 1.6286014f;
          if(hr[1]) t1 += psdres[r1 + q] - 1.6286014f;
          Tdb[q] = make_float2(t0, t1);
        }
      }
    }
#pragma unroll
    for(int e = 0; e < 2; e ++) hr_nxt[e] = has_psdres[fo + min(j + 2 + e, nf - 1)];
    pk0 = wave_max(pk0); pk1 = wave_max(pk1);
    const bool alive[2] = {!(pk0 < -100.0f), valid1 && !(pk1 < -100.0f)};
    if(! alive[0] && ! alive[1]) { stage(j + 2, hr_nxt); continue; }
    {
      int js = lane - shift;                         // opaque: the table positions are recomputed per pair, not kept
      asm volatile("" : "+v"(js));
#pragma unroll
      for(int m = 0; m < P; m ++) {
        float w;
        if constexpr (ALIAS) { const int j = js + WAVE * m; w = Wh[max(0, min(j, wsym - j))]; }   // (samples beyond the window are 0 already)
        else w = wv[m];
        xr[m] *= alive[0] ? w : 0.0f; xi[m] *= alive[1] ? w : 0.0f;
      }
    }
#if !(NF_ABL & 2)
    wave_fft<LOGN>(xr, xi, tw, lds, lane);
#endif
    float mr[H + 1], mi[H + 1];
    wave_mirror_lo<P>(xr, mr, lane);
    wave_mirror_lo<P>(xi, mi, lane);
    if(lane < 3) { Pw[lane] = make_float2(0.0f, 0.0f); Pw[nspec + 3 + lane] = make_float2(0.0f, 0.0f); }
    if constexpr (ALIAS) {
#pragma unroll
      for(int i = 0; i < TQ; i ++) if(lane + WAVE * i < npsd) Tdb[lane + WAVE * i] = make_float2(tq0[i], tq1[i]);
    }
#pragma unroll
    for(int m = 0; m <= H; m ++) {
      const float ar = 0.5f * (xr[m] + mr[m]), ai = 0.5f * (xi[m] - mi[m]);
      const float br = 0.5f * (xi[m] + mi[m]), bi = -0.5f * (xr[m] - mr[m]);
      xr[m] = ar; xi[m] = ai; mr[m] = br; mi[m] = bi;
      if(m < H || lane == 0)
        Pw[3 + lane + WAVE * m] = make_float2((ar * ar + ai * ai) * inv_wsqr, (br * br + bi * bi) * inv_wsqr);
    }
    __syncthreads();
    float nyq_r = 0.0f, nyq_i = 0.0f;
#if !(NF_ABL & 4)
    if(mavg_h == 3) nf_gain_loop<LOGN, 3>(xr, xi, mr, mi, Pw, Tdb, npsd, cpos, esc, mavg_h, lane, nyq_r, nyq_i);
    else nf_gain_loop<LOGN, 0>(xr, xi, mr, mi, Pw, Tdb, npsd, cpos, esc, mavg_h, lane, nyq_r, nyq_i);
#endif
    __syncthreads();
    if(lane == 0) { xr[H] = nyq_r; xi[H] = nyq_i; }
    wave_reflect<P>(mr, xr, lane);
    wave_reflect<P>(mi, xi, lane);
#if !(NF_ABL & 8)
    wave_fft<LOGN>(xi, xr, tw, lds, lane);           // inverse (x N): frame a in xr, frame b in xi
#endif
    __builtin_amdgcn_sched_barrier(0);               // (the loads must not be scheduled up into the transform: its registers are all taken)
    stage(j + 2, hr_nxt);                            // the next pair's loads fly while this pair is overlap-added
    __builtin_amdgcn_sched_barrier(0);
    // overlap-add, frame a then frame b (ascending order per sample)
    int lo_ = lane;
    asm volatile("" : "+v"(lo_));
#pragma unroll
    for(int e = 0; e < 2; e ++) {
      if(! alive[e]) continue;
      const int st = cen[e] - N / 2;
#if NF_ABL & 16
      { float sacc = 0; for(int m = 0; m < P; m ++) sacc += (e == 0 ? xr[m] : xi[m]); if(sacc == 1.2345f) yn[lane] = sacc; continue; }
#endif
      advance(st);
      // read all slots, then write all: the P slots are distinct, which the compiler cannot see
      // through the wrap-around (a slot-by-slot += would wait for LDS P times; ds_add_f32 is slower)
      float acc[P];
#pragma unroll
      for(int m = 0; m < P; m ++) acc[m] = ring[(st + lo_ + WAVE * m) & (N - 1)];
#pragma unroll
      for(int m = 0; m < P; m ++) {
        const int t = lo_ + WAVE * m;
        float v = (e == 0 ? xr[m] : xi[m]) * invN;
        if(m == 0 && t < nfade) v *= (float)t / (float)nfade;
        if(m == P - 1 && t >= N - nfade) v *= 1.0f - (float)(N - 1 - t) / (float)nfade;
        acc[m] += v;
      }
#pragma unroll
      for(int m = 0; m < P; m ++) ring[(st + lo_ + WAVE * m) & (N - 1)] = acc[m];
    }
  }
  advance(own_hi);
}

// =====================================================================
// S5 (unfused path: noise-filter transforms above 2048 points)  overlap-add gather of the shaped
// noise frames + final mix -- replaces layer0.c:620-624 and 657-659: y_noise = OLA,
// y = y_sin + y_noise (y_sin is already in place, written by k_synth_ola).  Thread per sample.
// =====================================================================
__global__ __launch_bounds__(256) void k_ola_noise_mix(
  const float* __restrict__ nframes_in, const int* __restrict__ live, int N,
  const int* __restrict__ frm_off, const int* __restrict__ nfrm,
  const int* __restrict__ out_off, const int* __restrict__ out_len,
  float thop, float fs, const float* __restrict__ ysin,
  float* __restrict__ ynoise, float* __restrict__ y) {
  const int u = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if(idx >= out_len[u]) return;
  const int nf = nfrm[u], fo = frm_off[u];
  const float hop = lp::fmul(thop, fs);
  const size_t o = (size_t)out_off[u] + idx;
  const int ilo = max(0, (int)((float)(idx - N / 2) / hop) - 1);
  const int ihi = min(nf - 1, (int)((float)(idx + N / 2) / hop) + 1);
  float acc = 0;
  // candidates in groups of 8, branch-free within a group: all loads of the gather are in
  // flight together (latency-bound kernel); ascending frame order is kept
  for(int i0 = ilo; i0 <= ihi; i0 += 8) {
    int lv[8]; float gv[8];
#pragma unroll
    for(int q = 0; q < 8; q ++) lv[q] = live[fo + min(i0 + q, nf - 1)];
#pragma unroll
    for(int q = 0; q < 8; q ++) {
      const int i = i0 + q;
      const int j = idx - lp::center(i, thop, fs) + N / 2;
      const bool ok = i <= ihi && j >= 0 && j < N;
      gv[q] = nframes_in[ok ? (size_t)(fo + i) * N + j : 0];
      if(!(ok && lv[q])) gv[q] = 0.0f;
    }
#pragma unroll
    for(int q = 0; q < 8; q ++) acc += gv[q];
  }
  ynoise[o] = acc;
  y[o] = ysin[o] + acc;
}

// =====================================================================
// A0  F0 refinement -- replaces llsm_refine_f0 (dsputils.c:72-94).  ciglet's
// ifdetector is opaque; OUR estimator (DESIGN.md "F0 refinement"): for
// harmonic j = 1..3 the phase advance between two Hann-windowed single-bin
// DFTs one sample apart, window span nh = round(4/fres) samples; estimates
// within 10 % of the input F0 are averaged.  One wavefront per frame.
// =====================================================================
__global__ __launch_bounds__(WAVE) void k_refine_f0(
  const float* __restrict__ x, const int* __restrict__ x_off, const int* __restrict__ nx,
  const int* __restrict__ frm_utt, const int* __restrict__ frm_off,
  float thop, float fs, float* __restrict__ f0) {
  const int g = xcd_frame(blockIdx.x, gridDim.x), lane = threadIdx.x;
  const float f = f0[g];
  if(!(f > 0)) return;
  int u, i; frame_owner(frm_utt, frm_off, g, & u, & i);
  const float* xs = x + x_off[u];
  const int nxu = nx[u];
  const int c = lp::center(i, thop, fs);
  const double fres = (double)f / (double)fs;
  const int nh = (int)round(4.0 / fres);
  const int base = c - nh / 2;
  float favg = 0; int nf = 0;
  for(int j = 1; j <= 3; j ++) {
    const double fc = fres * (double)j;
    float c0r = 0, c0i = 0, c1r = 0, c1i = 0;
    for(int t = lane; t < nh - 1; t += WAVE) {
      float w = hann_at(t, nh - 1);
      float co, si; cs_turns(fc * (double)t, & co, & si);
      int i0 = base + t, i1 = base + t + 1;
      float a = (i0 >= 0 && i0 < nxu) ? xs[i0] * w : 0.0f;
      float b = (i1 >= 0 && i1 < nxu) ? xs[i1] * w : 0.0f;
      c0r = fmaf(a, co, c0r); c0i = fmaf(-a, si, c0i);
      c1r = fmaf(b, co, c1r); c1i = fmaf(-b, si, c1i);
    }
    c0r = wave_sum(c0r); c0i = wave_sum(c0i); c1r = wave_sum(c1r); c1i = wave_sum(c1i);
    float pr = c1r * c0r + c1i * c0i, pi = c1i * c0r - c1r * c0i;
    float fj = atan2f(pi, pr) / 6.28318530717958647692f / (float)j;
    if(fabsf(fj - f / fs) < f * 0.1f / fs) { favg += fj; nf ++; }
  }
  if(lane == 0 && nf > 0) f0[g] = favg / (float)nf * fs;
}

// =====================================================================
// llsmrt kernels (harmonic-model path of llsmrt.c).  A "group" is S streams
// advancing in lock step; ring buffers (buffer.h:32-138) live in HBM as
// [S][cap] arrays and are addressed with the reference's negative-lag rule
// data[(curr + lag + cap) % cap]; `curr` values are tracked on the host.
// =====================================================================
DEV int ring_at(int curr, int lag, int cap) { return ((curr + lag) % cap + cap) % cap; }

// R0  circular noise templates -- llsm_make_exc_template (llsmrt.c:93-107):
// tile the band-limited template to ntemplate samples (dsputils.c:363-383,
// closed form) and make it circular with a 32-sample cross-fade (llsmrt.c:80-91).
__global__ __launch_bounds__(256) void k_rt_template(
  const float* __restrict__ colored, int ntemplate_ext, int nch, int nch_active,
  int ntemplate, float* __restrict__ tpl) {
  const int s = blockIdx.z, c = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if(j >= ntemplate) return;
  float* out = tpl + ((size_t)s * nch + c) * ntemplate;
  if(c >= nch_active) { out[j] = 0; return; }
  const float* src = colored + ((size_t)s * nch + c) * ntemplate_ext;
  const int nx = min(20000, ntemplate);
  auto at = [&](int p) {
    int b; float r;
    const int a = lp::stretch_index(p, nx, ntemplate, 128, & b, & r);
    float v = src[a];
    if(b >= 0) { v *= 1.0f - r; v += src[b] * r; v /= sqrtf(2.0f * r * (r - 1.0f) + 1.0f); }
    return v;
  };
  float y = at(j);
  if(j < 32) {
    const float r = (float)j / 32.0f;
    y *= 1.0f - r;
    y += at(ntemplate - 32 + j) * r;
    y /= sqrtf(2.0f * r * (r - 1.0f) + 1.0f);
  }
  out[j] = y;
}

// R1  llsm_update_cycle's appendblank (llsmrt.c:124-127) + the ring adds of
// feed_modcomps / feed_sinusoids (llsmrt.c:266, 287-288).  One block per stream.
__device__ __forceinline__ void rt_rings_body(
  float* mod, float* sinr, float* noiser, int cap, int nch,
  int mod_curr, int sin_curr, int noise_curr, int nhop, int nwin,
  const float* envf, const float* frames_sin, const float* f0, const int* has_nm, const int* nhar,
  int s = blockIdx.x, int tid = threadIdx.x, bool on = true) {
  // (s, tid, on: k_rt_hop runs two streams per workgroup, 256 threads each; a half without a stream only keeps the barriers)
  for(int i = tid; on && i < nhop; i += 256) {
    for(int c = 0; c < nch; c ++) mod[((size_t)s * nch + c) * cap + ring_at(mod_curr, -nhop + i, cap)] = 0;
    sinr[(size_t)s * cap + ring_at(sin_curr, -nhop + i, cap)] = 0;
    noiser[(size_t)s * cap + ring_at(noise_curr, -nhop + i, cap)] = 0;
  }
  __syncthreads();
  if(! on) return;
  if(has_nm[s])
    for(int c = 0; c < nch; c ++)
      for(int t = tid; t < nwin; t += 256)
        mod[((size_t)s * nch + c) * cap + ring_at(mod_curr, -nwin + t, cap)] += envf[((size_t)s * nch + c) * nwin + t];
  if(f0[s] > 0 && nhar[s] >= 0)
    for(int t = tid; t < nwin; t += 256)
      sinr[(size_t)s * cap + ring_at(sin_curr, -nwin + t, cap)] += frames_sin[(size_t)s * nwin + t];
}
__global__ __launch_bounds__(256) void k_rt_rings(
  float* __restrict__ mod, float* __restrict__ sinr, float* __restrict__ noiser, int cap, int nch,
  int mod_curr, int sin_curr, int noise_curr, int nhop, int nwin,
  const float* __restrict__ envf, const float* __restrict__ frames_sin,
  const float* __restrict__ f0, const int* __restrict__ has_nm, const int* __restrict__ nhar) {
  rt_rings_body(mod, sinr, noiser, cap, nch, mod_curr, sin_curr, noise_curr, nhop, nwin, envf, frames_sin, f0, has_nm, nhar);
}

// R2  llsm_run_excitation_buffers (llsmrt.c:134-147) + the gather of the
// 2*nhop-sample frame the filter works on (llsmrt.c:447-448).
// exc_curr is the ring position AFTER the append of nx samples.
__device__ __forceinline__ void rt_excite_body(
  const float* mod, const float* tpl, float* excr,
  int cap, int nch, int ntemplate, int mod_curr, int exc_curr, int exc_cycle, int curr_nhop,
  int nx, int nwin_frame, float* exc_frame, int s = blockIdx.x, int tid = threadIdx.x, bool on = true) {
  for(int i = tid; on && i < nx; i += 256) {
    float acc = 0;
    for(int c = 0; c < nch; c ++) {
      const float m = mod[((size_t)s * nch + c) * cap + ring_at(mod_curr, -curr_nhop - nx + i, cap)];
      acc += sqrtf(m) * tpl[((size_t)s * nch + c) * ntemplate + (exc_cycle + i) % ntemplate];
    }
    excr[(size_t)s * cap + ring_at(exc_curr, -nx + i, cap)] = acc;
  }
  __syncthreads();
  if(exc_frame && on)
    for(int j = tid; j < nwin_frame; j += 256)
      exc_frame[(size_t)s * nwin_frame + j] = excr[(size_t)s * cap + ring_at(exc_curr, -nwin_frame + j, cap)];
}
__global__ __launch_bounds__(256) void k_rt_excite(
  const float* __restrict__ mod, const float* __restrict__ tpl, float* __restrict__ excr,
  int cap, int nch, int ntemplate, int mod_curr, int exc_curr, int exc_cycle, int curr_nhop,
  int nx, int nwin_frame, float* __restrict__ exc_frame) {
  rt_excite_body(mod, tpl, excr, cap, nch, ntemplate, mod_curr, exc_curr, exc_cycle, curr_nhop, nx, nwin_frame, exc_frame);
}
// R1 + R2 of one hop in one launch: both are one block per stream, and the excitation reads only its own stream's
// envelope ring, which the same block has just advanced (one launch less in the dependent chain of a feed).
__global__ __launch_bounds__(256) void k_rt_rings_excite(
  float* mod, float* sinr, float* noiser, int cap, int nch, int mod_curr, int sin_curr, int noise_curr, int nhop, int nwin,
  const float* envf, const float* frames_sin, const float* f0, const int* has_nm, const int* nhar,
  const float* tpl, float* excr, int ntemplate, int exc_curr, int exc_cycle, float* exc_frame) {
  rt_rings_body(mod, sinr, noiser, cap, nch, mod_curr, sin_curr, noise_curr, nhop, nwin, envf, frames_sin, f0, has_nm, nhar);
  __threadfence_block();
  __syncthreads();
  rt_excite_body(mod, tpl, excr, cap, nch, ntemplate, mod_curr, exc_curr, exc_cycle, nhop, nhop, nwin, exc_frame);
}

// R3  noise-ring add (llsmrt.c:475) + llsm_rtsynth_buffer_feed_mix reads
// (llsmrt.c:483-486): out[s][0][i] = sinusoid ring, out[s][1][i] = noise ring.
__global__ __launch_bounds__(256) void k_rt_mix(
  float* __restrict__ noiser, const float* __restrict__ sinr, int cap, int noise_curr, int sin_curr,
  int sin_pos, int nfft, const float* __restrict__ nframes_in, const int* __restrict__ live,
  int next_nhop, int out_stride, float* __restrict__ out) {
  const int s = blockIdx.x, tid = threadIdx.x;
  if(live[s])
    for(int t = tid; t < nfft; t += 256)
      noiser[(size_t)s * cap + ring_at(noise_curr, -nfft + t, cap)] += nframes_in[(size_t)s * nfft + t];
  __syncthreads();
  for(int i = tid; i < next_nhop; i += 256) {
    out[((size_t)s * 2 + 0) * out_stride + i] = sinr[(size_t)s * cap + ring_at(sin_curr, sin_pos + i, cap)];
    out[((size_t)s * 2 + 1) * out_stride + i] = noiser[(size_t)s * cap + ring_at(noise_curr, -nfft + i, cap)];
  }
}

// A stream's rows of the hop from the pinned host block (RtRows) into its device rows, by the 256 threads `tid` of the
// workgroup that owns stream s; the caller puts a workgroup barrier behind it.
DEV void rt_stage_rows(const RtRows& host, int s, int tid, const float* f0, const int* nhar_e, const float* eamp,
  const float* ephs, const float* edc, int nch, int me, const int* nhar, const float* ampl, const float* phse, int maxnhar,
  const float* cyc_shift, const int* has_nm, int npsd, float* psd_dev, const float* f0_sin, RtPbpOp* ops_dev) {
  // The hop's rows are still in the pinned host block: ONE round trip over the link for the whole workgroup -- the
  // counts beside every row (the noise level row of k_rt_back included), the harmonic rows speculatively at 256 slots
  // -- and a second one only for a frame with more harmonics than that, instead of one trip per dependent load
  // further down.  The device rows are this workgroup's own; nobody else reads them in this launch.
  // (every load is issued before the first store: the compiler cannot tell the rows apart and would not move a load
  // above a store it follows)
  const float f = host.f0[s], cy = host.cyc[s];
  const int K = host.nhar[s], nhe = host.nhar_e[s], nm = host.has_nm[s];
  float a0 = 0, p0 = 0, e0 = 0, ea0 = 0, ep0 = 0, lv[4] = {0, 0, 0, 0}, fsin = 0; int opw = 0;
  if(host.f0sin) { fsin = host.f0sin[s]; if(tid < 8) opw = ((const int*)(host.ops + s))[tid]; }
  if(tid < maxnhar) { a0 = host.ampl[(size_t)s * maxnhar + tid]; p0 = host.phse[(size_t)s * maxnhar + tid]; }
  if(tid < nch) e0 = host.edc[(size_t)s * nch + tid];
  if(tid < nch * me) { ea0 = host.eamp[(size_t)s * nch * me + tid]; ep0 = host.ephs[(size_t)s * nch * me + tid]; }
#pragma unroll
  for(int j = 0; j < 4; j ++) if(tid + 256 * j < npsd) lv[j] = host.psd[(size_t)s * npsd + tid + 256 * j];
  float* ampl_w = (float*)ampl + (size_t)s * maxnhar; float* phse_w = (float*)phse + (size_t)s * maxnhar;
  const int Kc = K < 0 ? 0 : (K > maxnhar ? maxnhar : K);
  if(tid < Kc) { ampl_w[tid] = a0; phse_w[tid] = p0; }
  if(tid < nch) ((float*)edc)[(size_t)s * nch + tid] = e0;
  if(tid < nch * me) { ((float*)eamp)[(size_t)s * nch * me + tid] = ea0; ((float*)ephs)[(size_t)s * nch * me + tid] = ep0; }
#pragma unroll
  for(int j = 0; j < 4; j ++) if(tid + 256 * j < npsd) psd_dev[(size_t)s * npsd + tid + 256 * j] = lv[j];
  if(tid == 0) {
    ((float*)f0)[s] = f; ((float*)cyc_shift)[s] = cy;
    ((int*)nhar)[s] = K; ((int*)nhar_e)[s] = nhe; ((int*)has_nm)[s] = nm;
    if(host.f0sin) ((float*)f0_sin)[s] = fsin;
  }
  if(host.f0sin && tid < 8) ((int*)(ops_dev + s))[tid] = opw;
  for(int k = tid + 256; k < Kc; k += 256) {           // (longer rows than the first trip covers: rare)
    ampl_w[k] = host.ampl[(size_t)s * maxnhar + k]; phse_w[k] = host.phse[(size_t)s * maxnhar + k];
  }
  for(int k = tid + 1024; k < npsd; k += 256) psd_dev[(size_t)s * npsd + k] = host.psd[(size_t)s * npsd + k];
}

// R-front  one hop of one stream up to the excitation frame in ONE launch: envelope frames (three wavefronts) beside
// the harmonic frame (the fourth, on the MFMA), then the ring adds and the excitation step (rt_rings_body /
// rt_excite_body).  Replaces the k_env_frames -> k_synth_frames -> k_rt_rings_excite chain of a feed: the first two are
// independent and used to be two dependent launches of one wavefront per stream each.
template <int NCH, int ME, int NTS>
__global__ __launch_bounds__(256) void k_rt_front(
  const float* __restrict__ f0, const int* __restrict__ nhar_e, const float* __restrict__ eamp,
  const float* __restrict__ ephs, const float* __restrict__ edc, int nch, int me, float fs, int nwin,
  const float* __restrict__ win, float* __restrict__ envf,
  const float* __restrict__ f0_sin, const int* __restrict__ nhar, const float* __restrict__ ampl,
  const float* __restrict__ phse, int maxnhar, float thop, int L, const float* __restrict__ cyc_shift,
  float* __restrict__ frames_sin,
  float* mod, float* sinr, float* noiser, int cap, int mod_curr, int sin_curr, int noise_curr, int nhop,
  const int* __restrict__ has_nm, const float* __restrict__ tpl, float* excr, int ntemplate, int exc_curr,
  int exc_cycle, float* exc_frame, RtRows host, int npsd, float* psd_dev, RtPbpArgs pbp) {
  const int s = blockIdx.x, tid = threadIdx.x;
  if(host.f0) {
    rt_stage_rows(host, s, tid, f0, nhar_e, eamp, ephs, edc, nch, me, nhar, ampl, phse, maxnhar, cyc_shift, has_nm, npsd, psd_dev,
      f0_sin, pbp.ops);
    __threadfence_block();
    __syncthreads();
  }
  if(tid < WAVE) {
    const float f = f0_sin[s];
    if(f > 0) {
      float* out = frames_sin + (size_t)s * nwin;
      auto sink = [&](int t, float v) { out[t] = v; };
      synth_frame<NTS, decltype(sink), true>(s, 0, f, nhar, ampl, phse, maxnhar, thop, fs, nwin, L, win, cyc_shift,
        (float2*)g_lds, tid, sink);
    }
  } else
    env_frame_body<NCH, ME>(s, tid - WAVE, 256 - WAVE, f0, nhar_e, eamp, ephs, edc, nch, me, fs, nwin, win, envf);
  __threadfence_block();
  __syncthreads();
  rt_rings_body(mod, sinr, noiser, cap, nch, mod_curr, sin_curr, noise_curr, nhop, nwin, envf, frames_sin, f0_sin, has_nm, nhar);
  __threadfence_block();
  __syncthreads();
  rt_excite_body(mod, tpl, excr, cap, nch, ntemplate, mod_curr, exc_curr, exc_cycle, nhop, nhop, nwin, exc_frame);
  if(pbp.ops) {                                      // (k_rt_pbp's launch folded in: it touches the sinusoid ring and the dual buffer only)
    __threadfence_block();
    __syncthreads();
    rt_pbp_body(pbp.ops, pbp.frwd, pbp.bkwd, cap, pbp.dual_curr, sinr, sin_curr, nhop, win, pbp.pulse_out, pbp.pulse_stride, s, tid, true);
  }
}

// R-back  the rest of the hop for a PAIR of streams (2 p, 2 p + 1: they share one complex transform) in one launch of
// 256 threads: the noise filter on the LDS transform (four wavefronts instead of the one of k_noise_filter_wf: the hop
// waits for this chain), then noise-ring add and the hop's output samples of both streams (k_rt_mix).
__global__ __launch_bounds__(256) void k_rt_back(
  const float* __restrict__ exc_frame, int S, const float* __restrict__ psd, const float* __restrict__ psdres,
  const int* __restrict__ has_psdres, int npsd, float fnyq_conf, float thop, float fs, int nwin,
  const float* __restrict__ win, float inv_wsqr, int N, int logN, const float2* __restrict__ tw_glob, int tw_nmax,
  float* __restrict__ nframes, int* __restrict__ live,
  float* __restrict__ noiser, const float* __restrict__ sinr, int cap, int noise_curr, int sin_curr, int sin_pos,
  int next_nhop, int out_stride, float* __restrict__ out) {
  const int tid = threadIdx.x;
  float2* X = (float2*)g_lds;
  float2* tw = X + N;
  float2* P = tw + N / 2;
  float* red = (float*)(P + N / 2 + 1);
  float2* Tdb = (float2*)(red + 16);
  load_twiddles<256>(tw, tw_glob, N, tw_nmax, tid);
  const int gg[2] = {2 * (int)blockIdx.x, 2 * (int)blockIdx.x + 1};
  noise_filter_pair<256>(gg, tid, X, tw, P, red, Tdb, exc_frame, nullptr, nullptr, nullptr, nullptr, S, psd, psdres, has_psdres,
    npsd, fnyq_conf, thop, fs, nwin, win, inv_wsqr, N, logN, nframes, live, 1);
  __threadfence_block();
  __syncthreads();
#pragma unroll
  for(int e = 0; e < 2; e ++) {
    const int s = gg[e];
    if(s >= S) continue;
    if(live[s])
      for(int t = tid; t < N; t += 256)
        noiser[(size_t)s * cap + ring_at(noise_curr, -N + t, cap)] += nframes[(size_t)s * N + t];
  }
  __threadfence_block();
  __syncthreads();
#pragma unroll
  for(int e = 0; e < 2; e ++) {
    const int s = gg[e];
    if(s >= S) continue;
    for(int i = tid; i < next_nhop; i += 256) {
      out[((size_t)s * 2 + 0) * out_stride + i] = sinr[(size_t)s * cap + ring_at(sin_curr, sin_pos + i, cap)];
      out[((size_t)s * 2 + 1) * out_stride + i] = noiser[(size_t)s * cap + ring_at(noise_curr, -N + i, cap)];
    }
  }
}

// R-hop  k_rt_front and k_rt_back as ONE launch: a workgroup of 512 threads owns a PAIR of streams (2 p, 2 p + 1).
// Each half of it (256 threads) takes one stream through k_rt_front's steps; the whole of it then runs the noise filter
// of the pair (they share one complex transform) and writes both streams' samples.  Same device functions, same order of
// arithmetic per stream: the samples equal the two-launch hop's bit for bit.  (Pulse-by-pulse buffers add their pulses
// between the two halves of a hop and keep the two launches.)
template <int NCH, int ME, int NTS>
__global__ __launch_bounds__(512) void k_rt_hop(
  const float* __restrict__ f0, const int* __restrict__ nhar_e, const float* __restrict__ eamp,
  const float* __restrict__ ephs, const float* __restrict__ edc, int nch, int me, float fs, int nwin,
  const float* __restrict__ win, float* __restrict__ envf,
  const float* __restrict__ f0_sin, const int* __restrict__ nhar, const float* __restrict__ ampl,
  const float* __restrict__ phse, int maxnhar, float thop, int L, const float* __restrict__ cyc_shift,
  float* __restrict__ frames_sin,
  float* mod, float* sinr, float* noiser, int cap, int mod_curr, int sin_curr, int noise_curr, int nhop,
  const int* __restrict__ has_nm, const float* __restrict__ tpl, float* excr, int ntemplate, int exc_curr,
  int exc_cycle, float* exc_frame, RtRows host, int npsd, float* psd_dev, int lds_half,
  int S, const float* psdres, const int* has_psdres, float fnyq_conf, float inv_wsqr, int N, int logN,
  const float2* __restrict__ tw_glob, int tw_nmax, float* __restrict__ nframes, int* __restrict__ live,
  int sin_pos, int next_nhop, int out_stride, float* __restrict__ out, RtPbpArgs pbp) {
  const int half = threadIdx.x >> 8, tid = threadIdx.x & 255;
  const int s = 2 * (int)blockIdx.x + half;
  const bool on = s < S;
  if(host.f0) {
    if(on) rt_stage_rows(host, s, tid, f0, nhar_e, eamp, ephs, edc, nch, me, nhar, ampl, phse, maxnhar, cyc_shift, has_nm, npsd, psd_dev,
      f0_sin, pbp.ops);
    __threadfence_block();
    __syncthreads();
  }
  if(on) {
    if(tid < WAVE) {
      const float f = f0_sin[s];
      if(f > 0) {
        float* o = frames_sin + (size_t)s * nwin;
        auto sink = [&](int t, float v) { o[t] = v; };
        synth_frame<NTS, decltype(sink), true>(s, 0, f, nhar, ampl, phse, maxnhar, thop, fs, nwin, L, win, cyc_shift,
          (float2*)g_lds + (size_t)half * lds_half, tid, sink);
      }
    } else
      env_frame_body<NCH, ME>(s, tid - WAVE, 256 - WAVE, f0, nhar_e, eamp, ephs, edc, nch, me, fs, nwin, win, envf);
  }
  __threadfence_block();
  __syncthreads();
  rt_rings_body(mod, sinr, noiser, cap, nch, mod_curr, sin_curr, noise_curr, nhop, nwin, envf, frames_sin, f0_sin, has_nm, nhar, s, tid, on);
  __threadfence_block();
  __syncthreads();
  rt_excite_body(mod, tpl, excr, cap, nch, ntemplate, mod_curr, exc_curr, exc_cycle, nhop, nhop, nwin, exc_frame, s, tid, on);
  __threadfence_block();
  __syncthreads();
  if(pbp.ops) {
    rt_pbp_body(pbp.ops, pbp.frwd, pbp.bkwd, cap, pbp.dual_curr, sinr, sin_curr, nhop, win, pbp.pulse_out, pbp.pulse_stride, s, tid, on);
    __threadfence_block();
    __syncthreads();
  }
  // ---- k_rt_back's steps on all 512 threads
  const int t5 = threadIdx.x;
  float2* X = (float2*)g_lds;
  float2* tw = X + N;
  float2* P = tw + N / 2;
  float* red = (float*)(P + N / 2 + 1);
  float2* Tdb = (float2*)(red + 16);
  load_twiddles<512>(tw, tw_glob, N, tw_nmax, t5);
  const int gg[2] = {2 * (int)blockIdx.x, 2 * (int)blockIdx.x + 1};
  noise_filter_pair<512>(gg, t5, X, tw, P, red, Tdb, exc_frame, nullptr, nullptr, nullptr, nullptr, S, psd_dev, psdres, has_psdres,
    npsd, fnyq_conf, thop, fs, nwin, win, inv_wsqr, N, logN, nframes, live, 1);
  __threadfence_block();
  __syncthreads();
  if(on) {
    if(live[s])
      for(int t = tid; t < N; t += 256)
        noiser[(size_t)s * cap + ring_at(noise_curr, -N + t, cap)] += nframes[(size_t)s * N + t];
  }
  __threadfence_block();
  __syncthreads();
  if(on)
    for(int i = tid; i < next_nhop; i += 256) {
      out[((size_t)s * 2 + 0) * out_stride + i] = sinr[(size_t)s * cap + ring_at(sin_curr, sin_pos + i, cap)];
      out[((size_t)s * 2 + 1) * out_stride + i] = noiser[(size_t)s * cap + ring_at(noise_curr, -N + i, cap)];
    }
}

// R-hop2  k_rt_hop with the hop's temporaries on chip.  k_rt_hop moves every intermediate of a hop through global
// memory between barriers (envelope frames -> ring adds -> excitation -> gathered frame -> filtered frame -> ring add ->
// output: about ten dependent round trips of 1 - 2 us around ten microseconds of arithmetic).  Here
//   - everything the hop will read that does not depend on the host's rows -- the ring cells it updates, the noise
//     template cells of the excitation step, the older half of the excitation frame, the sinusoid samples it hands out,
//     the transform's twiddles -- is requested FIRST, so those round trips fly beside the trip over the link;
//   - the envelope frames, the harmonic frame, the excitation frame and the filtered frames live in LDS;
//   - a ring cell is read once (prefetched) and written once: "zero the new cells, then add" and "add, then read back for
//     the output" become one pass each.
// Per cell the arithmetic and its order are those of rt_rings_body / rt_excite_body / k_rt_back: the samples equal
// k_rt_hop's bit for bit.  Limits (else k_rt_hop): a window of at most 1024 samples, a transform of at most 2048 points.
#define RT2_JW 4                                   // window slots per thread (nwin <= 1024)
#define RT2_JN 4                                   // transform slots per thread and stream (N <= 1024; 2048: two rounds)
struct RtRingBase { int mod0, sin0, exc0, noi0, out0, tpl0; };
// index of slot base + t of a ring of `cap` cells (0 <= base < cap, 0 <= t < cap): no division per cell
DEV int ring_step(int base, int t, int cap) { const int i = base + t; return i >= cap ? i - cap : i; }

template <int NCH, int ME, int NTS>
__global__ __launch_bounds__(512) void k_rt_hop2(
  const float* __restrict__ f0, const int* __restrict__ nhar_e, const float* __restrict__ eamp,
  const float* __restrict__ ephs, const float* __restrict__ edc, int nch, int me, float fs, int nwin,
  const float* __restrict__ win,
  const float* __restrict__ f0_sin, const int* __restrict__ nhar, const float* __restrict__ ampl,
  const float* __restrict__ phse, int maxnhar, float thop, int L, const float* __restrict__ cyc_shift,
  float* mod, float* sinr, float* noiser, int cap, int mod_curr, int sin_curr, int noise_curr, int nhop,
  const int* __restrict__ has_nm, const float* __restrict__ tpl, float* excr, int ntemplate, int exc_curr,
  int exc_cycle, RtRows host, int npsd, float* psd_dev, int lds_half,
  int S, const float* psdres, const int* has_psdres, float fnyq_conf, float inv_wsqr, int N, int logN,
  const float2* __restrict__ tw_glob, int tw_nmax,
  int sin_pos, int next_nhop, int out_stride, float* __restrict__ out, RtPbpArgs pbp, int early_out, RtRingBase rb) {
  const int half = threadIdx.x >> 8, tid = threadIdx.x & 255, t5 = threadIdx.x;
  const int s = 2 * (int)blockIdx.x + half;
  const bool on = s < S;
  RT2_T(0);
  // ---- LDS: [X (N) | the two harmonic-amplitude scratches (2 lds_half)] tw P red Tdb | envelope, harmonic, excitation
  // frames of the two streams | their envelope parameters
  const int nx0 = N > 2 * lds_half ? N : 2 * lds_half;
  float2* X = (float2*)g_lds;
  float2* tw = X + nx0;
  float2* P = tw + N / 2;
  float* red = (float*)(P + N / 2 + 1);
  float2* Tdb = (float2*)(red + 16);
  float* fl = (float*)(Tdb + npsd);
  float* envl = fl + (size_t)half * nch * nwin;                        // this stream's [nch][nwin]
  float* sinl = fl + (size_t)2 * nch * nwin + (size_t)half * nwin;
  float* excl0 = fl + (size_t)2 * nch * nwin + (size_t)2 * nwin;       // [2][nwin]: the pair's excitation frames
  float* excl = excl0 + (size_t)half * nwin;
  float* par = excl0 + (size_t)2 * nwin + (size_t)half * (nch + 2 * nch * me);   // [nch] edc, [nch][me] (a cos, a sin)
  float* winl = excl0 + (size_t)2 * nwin + (size_t)2 * (nch + 2 * nch * me);      // the window, once
  float* pkp = winl + nwin;                                                       // [2][4] largest levels per wavefront
  float2* A = X + (size_t)half * lds_half;
  // ring positions of the window's / the transform's first cell (formed by the launcher: six pairs of integer divisions
  // per thread otherwise): the cells of a hop are consecutive from there
  const int mod0 = rb.mod0, sin0 = rb.sin0, exc0 = rb.exc0, noi0 = rb.noi0, out0 = rb.out0, tpl0 = rb.tpl0;
  float* mod_s = mod + (size_t)s * nch * cap; float* sin_s = sinr + (size_t)s * cap;
  float* exc_s = excr + (size_t)s * cap; float* noi_s = noiser + (size_t)s * cap;
  // ---- requests that do not wait for the host's rows
  float m_old[2][NCH], tp[2][NCH], s_old[2] = {0, 0}, ex_old[2] = {0, 0}, n_old[2 * RT2_JN], s_out[2] = {0, 0};
#pragma unroll
  for(int j = 0; j < 2; j ++)
#pragma unroll
    for(int c = 0; c < NCH; c ++) { m_old[j][c] = 0; tp[j][c] = 0; }
#pragma unroll
  for(int j = 0; j < 2 * RT2_JN; j ++) n_old[j] = 0;
  if(on) {
#pragma unroll
    for(int j = 0; j < 2; j ++) {
      const int t = tid + 256 * j;                   // the older half of the window (t < nhop) and hop sample i = t
      if(t < nhop) {
        const int it = tpl0 + t >= ntemplate ? tpl0 + t - ntemplate : tpl0 + t;
#pragma unroll
        for(int c = 0; c < NCH; c ++)
          if(c < nch) {
            m_old[j][c] = mod_s[(size_t)c * cap + ring_step(mod0, t, cap)];
            tp[j][c] = tpl[((size_t)s * nch + c) * ntemplate + it];
          }
        s_old[j] = sin_s[ring_step(sin0, t, cap)];
        ex_old[j] = exc_s[ring_step(exc0, t, cap)];
      }
      if(early_out && t < next_nhop) s_out[j] = sin_s[ring_step(out0, t, cap)];
    }
#pragma unroll
    for(int j = 0; j < 2 * RT2_JN; j ++) {
      const int t = tid + 256 * j;                   // cells older than this hop's (those become zero: appendblank)
      if(t < N - nhop) n_old[j] = noi_s[ring_step(noi0, t, cap)];
    }
  }
  float2 tw_v[2]; float win_v[2];                    // twiddles (N / 2 <= 1024) and window (nwin <= 1024), stored below
  {
    const int stride = tw_nmax / N;
#pragma unroll
    for(int j = 0; j < 2; j ++) {
      const int k = t5 + 512 * j;
      tw_v[j] = k < N / 2 ? tw_glob[k * stride] : make_float2(0.0f, 0.0f);
      win_v[j] = k < nwin ? win[k] : 0.0f;
    }
  }
  RT2_T(1);
  // ---- the stream's rows, from the pinned block (one trip over the link for the whole workgroup, see rt_stage_rows) or
  // from the device rows a copy has filled: counts, harmonic amplitudes A_h straight into the LDS scratch of synth_frame,
  // envelope parameters a e^{j phi} formed ONCE (env_frame_body alone: sixteen sincosf per thread), noise levels
  float f_true = 0, f_syn = 0; int Kraw = -1, K = 0, nhe = 0, nmv = 0;
  const bool dir = host.f0 != nullptr;
  {
    // noise levels of the pair into Tdb (x: first stream, y: second; an absent second stream mirrors the first, as
    // noise_filter_pair does) and their per-wavefront maxima; has_psdres is all zero in llsmrt (PSDRES is folded into the
    // rows on the host, llsmrt.c:513-520), so the level is the row itself
    const int sp = on ? s : s - 1;
    const float* r_psd = dir ? host.psd : psd_dev;
    float lv[4], pk = -3.0e38f;
#pragma unroll
    for(int j = 0; j < 4; j ++) lv[j] = tid + 256 * j < npsd ? r_psd[(size_t)sp * npsd + tid + 256 * j] : 0.0f;
#pragma unroll
    for(int j = 0; j < 4; j ++) {
      const int k = tid + 256 * j;
      if(k < npsd) {
        pk = fmaxf(pk, lv[j]);
        if(half == 0) Tdb[k].x = lv[j]; else Tdb[k].y = lv[j];
        if(dir && on) psd_dev[(size_t)s * npsd + k] = lv[j];       // (kept current for the two-launch form)
      }
    }
    pk = wave_max(pk);
    if((tid & (WAVE - 1)) == 0) pkp[4 * half + (tid >> 6)] = pk;
#pragma unroll
    for(int j = 0; j < 2; j ++) {
      const int k = t5 + 512 * j;
      if(k < N / 2) tw[k] = tw_v[j];
      if(k < nwin) winl[k] = win_v[j];
    }
  }
  if(on) {
    const float* r_ampl = dir ? host.ampl : ampl; const float* r_phse = dir ? host.phse : phse;
    f_true = (dir ? host.f0 : f0)[s];
    const float cy = (dir ? host.cyc : cyc_shift)[s];
    Kraw = (dir ? host.nhar : nhar)[s]; nhe = (dir ? host.nhar_e : nhar_e)[s]; nmv = (dir ? host.has_nm : has_nm)[s];
    f_syn = dir ? (host.f0sin ? host.f0sin[s] : f_true) : f0_sin[s];
    float a0 = 0, p0 = 0, e0 = 0, ea0 = 0, ep0 = 0; int opw = 0;
    if(tid < maxnhar) { a0 = r_ampl[(size_t)s * maxnhar + tid]; p0 = r_phse[(size_t)s * maxnhar + tid]; }
    if(tid < nch) e0 = (dir ? host.edc : edc)[(size_t)s * nch + tid];
    if(tid < nch * me) { ea0 = (dir ? host.eamp : eamp)[(size_t)s * nch * me + tid]; ep0 = (dir ? host.ephs : ephs)[(size_t)s * nch * me + tid]; }
    if(dir && host.f0sin && tid < 8) opw = ((const int*)(host.ops + s))[tid];
    K = Kraw; if(K > 2048) K = 2048; if(K > maxnhar) K = maxnhar; if(K < 0) K = 0;
    const int Kp = (K + 3) & ~3;
    if(tid < Kp) A[tid] = tid < K ? synth_amplitude(a0, p0, tid, cy, f_syn) : make_float2(0.0f, 0.0f);
    for(int k = tid + 256; k < Kp; k += 256)           // (more harmonics than the first trip covers: rare)
      A[k] = k < K ? synth_amplitude(r_ampl[(size_t)s * maxnhar + k], r_phse[(size_t)s * maxnhar + k], k, cy, f_syn) : make_float2(0.0f, 0.0f);
    if(tid < nch) par[tid] = e0;
    if(tid < nch * me) { float sn, co; sincosf(ep0, & sn, & co); par[nch + 2 * tid] = ea0 * co; par[nch + 2 * tid + 1] = ea0 * sn; }
    if(dir && host.f0sin && tid < 8) ((int*)(pbp.ops + s))[tid] = opw;   // (rt_pbp_body reads the device row)
  }
  __threadfence_block();
  __syncthreads();
  RT2_T(2);
  if(on && early_out) {
#pragma unroll
    for(int j = 0; j < 2; j ++) { const int t = tid + 256 * j; if(t < next_nhop) out[((size_t)s * 2 + 0) * out_stride + t] = s_out[j]; }
  }
  // ---- envelope frames (three wavefronts) beside the harmonic frame (the fourth), into LDS
  const bool voiced = on && f_syn > 0 && Kraw >= 0, nm = on && nmv != 0;
  if(on) {
    if(tid < WAVE) {
      if(f_syn > 0) {
        auto sink = [&](int t, float v) { sinl[t] = v; };
        synth_frame<NTS, decltype(sink), true>(s, 0, f_syn, nhar, ampl, phse, maxnhar, thop, fs, nwin, L, winl, cyc_shift, A, tid, sink, K);
      }
    } else
      env_frame_body<NCH, ME>(s, tid - WAVE, 256 - WAVE, f0, nhar_e, eamp, ephs, edc, nch, me, fs, nwin, winl, nullptr, envl, par, f_true, nhe);
  }
  RT2_T(3);
  __syncthreads();
  RT2_T(4);
  // ---- ring cells: the older half of the window gains this frame, the newer half (blank) becomes it; excitation of the hop
  if(on) {
#pragma unroll
    for(int j = 0; j < RT2_JW; j ++) {
      const int t = tid + 256 * j;
      if(t < nwin) {
        const bool old_half = t < nhop;              // (j < 2 there: nhop <= 512)
        float mnew[NCH];
#pragma unroll
        for(int c = 0; c < NCH; c ++) {
          mnew[c] = 0;
          if(c < nch) {
            float v = old_half ? m_old[j < 2 ? j : 0][c] : 0.0f;
            if(nm) v += envl[(size_t)c * nwin + t];
            mnew[c] = v;
            if(nm || ! old_half) mod_s[(size_t)c * cap + ring_step(mod0, t, cap)] = v;
          }
        }
        {
          float v = old_half ? s_old[j < 2 ? j : 0] : 0.0f;
          if(voiced) v += sinl[t];
          if(voiced || ! old_half) sin_s[ring_step(sin0, t, cap)] = v;
        }
        if(old_half) {                               // rt_excite_body, hop sample i = t; and the frame the filter works on
          float acc = 0;
#pragma unroll
          for(int c = 0; c < NCH; c ++) if(c < nch) acc += sqrtf(mnew[c]) * tp[j < 2 ? j : 0][c];
          exc_s[ring_step(exc0, nhop + t, cap)] = acc;
          excl[nhop + t] = acc;
          excl[t] = ex_old[j < 2 ? j : 0];
        }
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  RT2_T(5);
  if(pbp.ops) {
    rt_pbp_body(pbp.ops, pbp.frwd, pbp.bkwd, cap, pbp.dual_curr, sinr, sin_curr, nhop, win, pbp.pulse_out, pbp.pulse_stride, s, tid, on);
    __threadfence_block();
    __syncthreads();
  }
  RT2_T(6);
  // ---- the pair's noise filter on all 512 threads; the filtered frames stay in X
  const int gg[2] = {2 * (int)blockIdx.x, 2 * (int)blockIdx.x + 1};
  const int alive = noise_filter_pair<512>(gg, t5, X, tw, P, red, Tdb, nullptr, nullptr, nullptr, nullptr, nullptr, S, psd_dev, psdres, has_psdres,
    npsd, fnyq_conf, thop, fs, nwin, winl, inv_wsqr, N, logN, nullptr, nullptr, 1, excl0, pkp);
  RT2_T(7);
  // ---- noise ring: blank cells take the frame, older ones gain it; the hop's samples
  if(on) {
    const bool live = (alive >> half) & 1;
#pragma unroll
    for(int j = 0; j < 2 * RT2_JN; j ++) {
      const int t = tid + 256 * j;
      if(t < N) {
        const bool blank = t >= N - nhop;
        float v = blank ? 0.0f : n_old[j];
        if(live) v += rt_noise_sample(half == 0 ? X[t].x : X[t].y, t, N);
        if(live || blank) noi_s[ring_step(noi0, t, cap)] = v;
        if(t < next_nhop) out[((size_t)s * 2 + 1) * out_stride + t] = v;
      }
    }
    if(! early_out)
      for(int i = tid; i < next_nhop; i += 256) out[((size_t)s * 2 + 0) * out_stride + i] = sin_s[ring_step(out0, i, cap)];
  }
  RT2_T(8);
}

// ---------------------------------------------------------------- launchers
int launch_refine_f0(LaunchCtx* P, const BatchDev& d) {
  if(d.nframes == 0) return 0;
  LAUNCH("k_refine_f0", k_refine_f0, dim3(d.nframes), dim3(WAVE), 0,
    d.x, d.x_off, d.nx, d.frm_utt, d.frm_off, d.thop, d.fs, d.f0);
  return 0;
}

// min_f0: lowest voiced F0 the batch can hold (sizes the window table of the tile kernel; 0: unknown, largest table)
int launch_harm_speech(LaunchCtx* P, const BatchDev& d, float min_f0) {
  if(d.nframes == 0) return 0;
  // The window table of the tile kernel is provisioned from the lowest F0 the batch can hold, capped at 48 KB.  Every
  // frame of the batch that fits the CAP fits this provision too, so which frames form a tile does not depend on the
  // batch an utterance sits in.  min_f0 <= 0: unknown (F0 written through its device pointer) -> the full cap.
  int kcap = 0;
  if(d.hblocks && d.nhblocks > 0) {
    kcap = min_f0 > 0 ? (lp::hwin(min_f0, d.fs, d.rel_winsize) / 2 + 8) & ~3 : HT_KCAP_MAX;
    if(kcap > HT_KCAP_MAX) kcap = HT_KCAP_MAX;       // lower F0 than this provision: those frames stay with the per-frame kernel
  }
  if(kcap > 0) {
    LAUNCH("k_harm_speech_tile", k_harm_speech_tile, dim3(d.nhblocks), dim3(HT_NT), (size_t)kcap * 8 + (4 * 2 * HM_TILES * WAVE + 4) * sizeof(float),
      d.hblocks, kcap, d.x, d.x_off, d.nx, d.frm_utt, d.frm_off, d.f0, d.thop, d.fs, d.rel_winsize, d.maxnhar,
      d.nhar, d.ampl, d.phse);
    LAUNCH("k_harm_speech_rest", k_harm_speech_rest, dim3(d.nhblocks), dim3(4 * WAVE), 0,
      d.hblocks, kcap, d.x, d.x_off, d.nx, d.frm_utt, d.frm_off, d.f0, d.thop, d.fs, d.rel_winsize, d.maxnhar,
      d.nhar, d.ampl, d.phse);
    return 0;
  }
  LAUNCH("k_harm_speech", k_harm_speech, dim3(d.nframes), dim3(WAVE), 0,
    d.x, d.x_off, d.nx, d.frm_utt, d.frm_off, d.f0, d.thop, d.fs, d.rel_winsize, d.maxnhar,
    d.nhar, d.ampl, d.phse);
  return 0;
}

int launch_harm_env(LaunchCtx* P, const BatchDev& d, const float* ce, size_t ce_stride) {
  if(d.nframes == 0) return 0;
#define HE_ARGS ce, ce_stride, d.x_off, d.nx, d.frm_utt, d.frm_off, d.f0, d.thop, d.fs, \
    d.rel_winsize, d.nchannel, d.maxnhar_e, d.nhar_e, d.edc, d.eenv_ampl, d.eenv_phse
  if(d.nchannel <= 4 && d.maxnhar_e <= 4)
    LAUNCH("k_harm_env", (k_harm_env<4, 4>), dim3(d.nframes), dim3(WAVE), 0, HE_ARGS);
  else if(d.nchannel <= 4 && d.maxnhar_e <= 8)
    LAUNCH("k_harm_env", (k_harm_env<4, 8>), dim3(d.nframes), dim3(WAVE), 0, HE_ARGS);
  else if(d.nchannel <= 8 && d.maxnhar_e <= 8)
    LAUNCH("k_harm_env", (k_harm_env<8, 8>), dim3(d.nframes), dim3(WAVE), 0, HE_ARGS);
  else return -1001;
#undef HE_ARGS
  return 0;
}


int launch_filtfilt(LaunchCtx* P, const FiltJob* jobs, int njobs, const FiltSectionD* sections) {
  if(njobs == 0) return 0;
  LAUNCH("k_filtfilt", k_filtfilt, dim3(njobs), dim3(WAVE),
    sizeof(IirLds), jobs, njobs, sections);
  return 0;
}

static int fft_grid(int np) { return np < 2048 ? np : 2048; }
// Workgroups of a persistent one-wavefront kernel that walks `np` work items: as many as the device keeps resident for
// THIS instantiation (its register and LDS use decide: 2 wavefronts per SIMD for the 2048-point transforms, 4 for the
// 1024-point ones -- a fixed 2048 left half the slots of the latter empty: k_psd_frames_wf 0.248 -> 0.194 ms), at least 2048.
template <class K>
static int persistent_grid(K kernel, size_t lds, int np) {
  static std::mutex mx; static std::map<std::pair<const void*, size_t>, int> cache;
  int resident = 0;
  {
    std::lock_guard<std::mutex> lock(mx);
    auto it = cache.find({(const void*)kernel, lds});
    if(it != cache.end()) resident = it -> second;
    else {
      int per_cu = 0, dev = 0, cus = 0;
      if(hipOccupancyMaxActiveBlocksPerMultiprocessor(& per_cu, kernel, WAVE, lds) != hipSuccess) { per_cu = 0; (void)hipGetLastError(); }
      if(hipGetDevice(& dev) != hipSuccess || hipDeviceGetAttribute(& cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { cus = 0; (void)hipGetLastError(); }
      resident = per_cu * cus;
      cache[{(const void*)kernel, lds}] = resident;
    }
  }
  const int g = resident > 2048 ? resident : 2048;
  return np < g ? np : g;
}
static int npairs_of(const BatchDev& d) { return d.pairs ? d.npairs : (d.nframes + 1) / 2; }

int launch_spgm_env(LaunchCtx* P, const BatchDev& d, int nwin_psd, int N, int logN,
  int nfft_psd, float norm_base, const float2* tw, int tw_nmax, float* env_out, int2* fix_list, int* fix_count, int which) {
  // which: 1 the transforms of every pair, 2 the listed pairs again with exact edge bins (its own launch, so that the
  // caller can put it beside the next kernel on a second stream), 3 both in a row
  if(d.nframes == 0) return 0;
  // register-resident transform when N / nfft_psd is a fold of 1, 2 or 4 and N <= 2048
  // (4096 points = 128 data VGPRs per lane spill at 2 waves / SIMD: the LDS kernel serves those)
  int logF = -1;
  if(nfft_psd <= N && N % nfft_psd == 0) { logF = 0; while((nfft_psd << logF) < N) logF ++; }
#define WF_CASE(LN, LF) \
  if(logN == LN && logF == LF) { \
    constexpr int e1 = wf_lds_elems<LN>(), e2 = wf_lds_elems<LN - LF>(); \
    if(which & 1) \
    LAUNCH("k_spgm_env_wf", (k_spgm_env_wf<LN, LF, false>), dim3(persistent_grid(k_spgm_env_wf<LN, LF, false>, sizeof(float2) * (e1 > e2 ? e1 : e2) + SPGM_SEED_LDS, npairs_of(d))), dim3(WAVE), \
      sizeof(float2) * (e1 > e2 ? e1 : e2) + SPGM_SEED_LDS, d.x, d.x_off, d.nx, d.frm_utt, d.frm_off, d.f0, \
      d.nframes, d.thop, d.fs, nwin_psd, norm_base, env_out, d.pairs, npairs_of(d), fix_list, fix_count); \
    if((which & 2) && fix_list && SPGM_EDGE_F64) /* the listed pairs again, exact edge bins (a few dozen wavefronts find work, if any) */ \
      LAUNCH("k_spgm_env_fix", (k_spgm_env_wf<LN, LF, true>), dim3(npairs_of(d) < 256 ? npairs_of(d) : 256), dim3(WAVE), \
        sizeof(float2) * (e1 > e2 ? e1 : e2) + SPGM_SEED_LDS, d.x, d.x_off, d.nx, d.frm_utt, d.frm_off, d.f0, \
        d.nframes, d.thop, d.fs, nwin_psd, norm_base, env_out, d.pairs, npairs_of(d), fix_list, fix_count); \
    return 0; \
  }
  WF_CASE(9, 0) WF_CASE(9, 1)
  WF_CASE(10, 0) WF_CASE(10, 1) WF_CASE(10, 2)
  WF_CASE(11, 0) WF_CASE(11, 1) WF_CASE(11, 2)
#undef WF_CASE
  if(!(which & 1)) return 0;
  size_t lds = (size_t)(N + N / 2) * sizeof(float2);
  LAUNCH("k_spgm_env", k_spgm_env, dim3(fft_grid(npairs_of(d))), dim3(WAVE), lds,
    d.x, d.x_off, d.nx, d.frm_utt, d.frm_off, d.f0, d.nframes, d.thop, d.fs, nwin_psd,
    N, logN, nfft_psd, norm_base, tw, tw_nmax, env_out, d.pairs, npairs_of(d));
  return 0;
}

int launch_wf_selftest(LaunchCtx* P, int logN, const float2* in, float2* out, int count, int inverse) {
#define WF_CASE(LN) \
  if(logN == LN) { \
    LAUNCH("k_wf_selftest", (k_wf_selftest<LN>), dim3(count), dim3(WAVE), \
      sizeof(float2) * wf_lds_elems<LN>(), in, out, inverse); \
    return 0; \
  }
  WF_CASE(8) WF_CASE(9) WF_CASE(10) WF_CASE(11) WF_CASE(12)
#undef WF_CASE
  return -1;
}

int launch_psd_frames(LaunchCtx* P, const BatchDev& d, const float* xres, int nwin,
  const float* win, float inv_wpow, int N, int logN, const float2* tw, int tw_nmax,
  float* psd_log) {
  if(d.nframes == 0) return 0;
#define WF_CASE(LN) \
  if(logN == LN) { \
    LAUNCH("k_psd_frames_wf", (k_psd_frames_wf<LN>), dim3(persistent_grid(k_psd_frames_wf<LN>, sizeof(float2) * wf_lds_elems<LN>(), npairs_of(d))), dim3(WAVE), \
      sizeof(float2) * wf_lds_elems<LN>(), xres, d.x_off, d.nx, d.frm_utt, d.frm_off, d.nframes, \
      d.thop, d.fs, nwin, win, inv_wpow, psd_log, d.pairs, npairs_of(d)); \
    return 0; \
  }
  WF_CASE(8) WF_CASE(9) WF_CASE(10) WF_CASE(11)
#undef WF_CASE
  if(N > LLSM_LDS_FFT_MAX) {                          // beyond the LDS: global scratch + the big twiddle table (engine.cpp)
    const int grid = std::min(fft_grid(npairs_of(d)), llsm_big_fft_grid((size_t)N));
    if(! P -> tw_big || N > P -> tw_big_nmax || P -> big_scratch_elems < (size_t)grid * N) return -1;
    LAUNCH("k_psd_frames", k_psd_frames, dim3(grid), dim3(WAVE), 64,
      xres, d.x_off, d.nx, d.frm_utt, d.frm_off, d.nframes, d.thop, d.fs, nwin, win, inv_wpow,
      N, logN, P -> tw_big, P -> tw_big_nmax, psd_log, d.pairs, npairs_of(d), P -> big_scratch);
    return 0;
  }
  size_t lds = (size_t)(N + N / 2) * sizeof(float2);
  if(lds > 64 * 1024 &&
     hipFuncSetAttribute((const void*)k_psd_frames, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  LAUNCH("k_psd_frames", k_psd_frames, dim3(fft_grid(npairs_of(d))), dim3(WAVE), lds,
    xres, d.x_off, d.nx, d.frm_utt, d.frm_off, d.nframes, d.thop, d.fs, nwin, win, inv_wpow,
    N, logN, tw, tw_nmax, psd_log, d.pairs, npairs_of(d), (float2*)nullptr);
  return 0;
}

int launch_kalman(LaunchCtx* P, const BatchDev& d, const float* env, const float* psd_log,
  float* ck, int nspec) {
  if(d.nframes == 0) return 0;
  LAUNCH("k_kalman", k_kalman, dim3((unsigned)(((size_t)d.npsd * d.n_utt * (KAL_SPLIT ? 2 : 1) + 127) / 128)), dim3(128), 0,
    env, psd_log, ck, d.frm_off, d.nfrm, d.n_utt, nspec, d.npsd, d.fs, d.psd, d.psdres, d.has_psdres);
  return 0;
}


int launch_white(LaunchCtx* P, const BatchDev& d, float* white, int ntemplate_ext,
  const int* out_len, unsigned long long seed) {
  if(d.n_utt == 0) return 0;
  LAUNCH("k_white", k_white, dim3((ntemplate_ext + 255) / 256, d.nchannel, d.n_utt), dim3(256), 0,
    white, ntemplate_ext, out_len, d.nchannel, seed);
  return 0;
}

int launch_env_frames(LaunchCtx* P, const BatchDev& d, float fs_syn, int nwin,
  const float* win, float* envf) {
  if(d.nframes == 0) return 0;
#define EF_ARGS d.f0, d.nhar_e, d.eenv_ampl, d.eenv_phse, d.edc, d.nchannel, d.maxnhar_e, fs_syn, nwin, win, envf
  if(d.nchannel <= 4 && d.maxnhar_e <= 4)
    LAUNCH("k_env_frames", (k_env_frames<4, 4>), dim3(d.nframes), dim3(WAVE), 0, EF_ARGS);
  else if(d.nchannel <= 4)
    LAUNCH("k_env_frames", (k_env_frames<4, 8>), dim3(d.nframes), dim3(WAVE), 0, EF_ARGS);
  else
    LAUNCH("k_env_frames", (k_env_frames<8, 8>), dim3(d.nframes), dim3(WAVE), 0, EF_ARGS);
#undef EF_ARGS
  return 0;
}

// The analysed rows of every frame gathered into ONE record per frame (packed.h): what the object path ships to the host so
// that an utterance's frames arrive as one contiguous block that the reference's frame objects are laid over in place
// (model.cpp llsm_frames_packed_finish) -- instead of eleven row arrays that the host re-scatters frame by frame.
// One wavefront per frame; 19 MB per 32 one-second utterances, ~10 us.
__global__ __launch_bounds__(WAVE) void k_pack_frames(int nframes, LlsmPackedLayout L,
  const float* __restrict__ f0, const int* __restrict__ nhar, const float* __restrict__ ampl, const float* __restrict__ phse,
  const float* __restrict__ psd, const float* __restrict__ psdres, const int* __restrict__ has_psdres,
  const float* __restrict__ edc, const int* __restrict__ nhar_e, const float* __restrict__ eamp, const float* __restrict__ ephs,
  float* __restrict__ out, float* const* __restrict__ dst_tab, const int* __restrict__ frm_utt, const int* __restrict__ frm_off) {
  const int g = blockIdx.x, lane = threadIdx.x;
  if(g >= nframes) return;
  // dst_tab != NULL: the record goes straight to its place in the utterance's page-locked host block (dst_tab[u], itself
  // a page-locked table: posted writes over the link, no copy engine -- a device-to-host copy costs ~0.1 - 0.3 ms whatever
  // its size on this stack, and 32 of them per block were 10 ms, profiles/r05_h); else into the device buffer `out`
  float* r;
  if(dst_tab) { const int u = frm_utt[g]; r = dst_tab[u] + (size_t)(g - frm_off[u]) * L.words; }
  else r = out + (size_t)g * L.words;
  // every piece of a record starts on a 16-byte boundary and is padded to a multiple of four words: 16-byte stores (the
  // link carries 64-byte-and-larger writes far better than 4-byte ones); source rows are read with 4-byte-aligned 16-byte loads
  auto piece = [&](int o_words, const float* src, int n) {       // n floats of `src` into words [o_words, o_words + up4(n)); the padding as zeros
    for(int k = 4 * lane; k < n; k += 4 * WAVE) {
      f4u v;
      if(k + 3 < n) v = *(const f4u*)(src + k);
      else { v.x = src[k]; v.y = k + 1 < n ? src[k + 1] : 0.0f; v.z = k + 2 < n ? src[k + 2] : 0.0f; v.w = 0.0f; }
      *(f4a*)(r + o_words + k) = f4a{v.x, v.y, v.z, v.w};
    }
  };
  if(lane == 0) {
    f4a h; h.x = f0[g]; h.y = __int_as_float(nhar[g]); h.z = __int_as_float(nhar_e[g]); h.w = __int_as_float(has_psdres[g]);
    *(f4a*)r = h;
    f4a q; q.x = 0.0f; q.y = 0.0f; q.z = 0.0f; q.w = __int_as_float(L.npsd);
    *(f4a*)(r + L.o_reshdr) = q;
  }
  piece(L.o_ampl, ampl + (size_t)g * L.maxnhar, L.maxnhar);
  piece(L.o_phse, phse + (size_t)g * L.maxnhar, L.maxnhar);
  piece(L.o_psd, psd + (size_t)g * L.npsd, L.npsd);
  piece(L.o_psdres, psdres + (size_t)g * L.npsd, L.npsd);
  piece(L.o_edc, edc + (size_t)g * L.nch, L.nch);
  piece(L.o_eamp, eamp + (size_t)g * L.nch * L.me, L.nch * L.me);
  piece(L.o_ephs, ephs + (size_t)g * L.nch * L.me, L.nch * L.me);
}
// ... and back: packed records (uploaded per utterance from the chunks' slabs) scattered into the rows the synthesis reads
__global__ __launch_bounds__(WAVE) void k_unpack_frames(int nframes, LlsmPackedLayout L, const float* __restrict__ in,
  const float* const* __restrict__ src_tab, const int* __restrict__ frm_utt, const int* __restrict__ frm_off,
  float* __restrict__ f0, int* __restrict__ nhar, float* __restrict__ ampl, float* __restrict__ phse,
  float* __restrict__ psd, float* __restrict__ psdres, int* __restrict__ has_psdres,
  float* __restrict__ edc, int* __restrict__ nhar_e, float* __restrict__ eamp, float* __restrict__ ephs) {
  const int g = blockIdx.x, lane = threadIdx.x;
  if(g >= nframes) return;
  const float* r;                                     // src_tab != NULL: read from the utterance's page-locked host block
  if(src_tab) { const int u = frm_utt[g]; r = src_tab[u] + (size_t)(g - frm_off[u]) * L.words; }
  else r = in + (size_t)g * L.words;
  const f4a h = *(const f4a*)r;
  const int nh = __float_as_int(h.y), ne = __float_as_int(h.z), hr = __float_as_int(h.w);
  if(lane == 0) { f0[g] = h.x; nhar[g] = nh; nhar_e[g] = ne; has_psdres[g] = hr; }
  // 16-byte loads of the record (its pieces are 16-byte aligned and padded); values beyond a frame's own counts are written
  // as zeros: what llsm_chunk_to_flat leaves there.  keep(k): element k of the piece is live
  auto piece = [&](int o_words, float* dst, int n, auto keep) {
    for(int k = 4 * lane; k < n; k += 4 * WAVE) {
      const f4a v = *(const f4a*)(r + o_words + k);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for(int q = 0; q < 4; q ++) if(k + q < n) dst[k + q] = keep(k + q) ? e[q] : 0.0f;
    }
  };
  piece(L.o_ampl, ampl + (size_t)g * L.maxnhar, L.maxnhar, [&](int k) { return k < nh; });
  piece(L.o_phse, phse + (size_t)g * L.maxnhar, L.maxnhar, [&](int k) { return k < nh; });
  piece(L.o_psd, psd + (size_t)g * L.npsd, L.npsd, [&](int) { return true; });
  piece(L.o_psdres, psdres + (size_t)g * L.npsd, L.npsd, [&](int) { return hr != 0; });
  piece(L.o_edc, edc + (size_t)g * L.nch, L.nch, [&](int) { return true; });
  piece(L.o_eamp, eamp + (size_t)g * L.nch * L.me, L.nch * L.me, [&](int k) { return (k % L.me) < ne; });
  piece(L.o_ephs, ephs + (size_t)g * L.nch * L.me, L.nch * L.me, [&](int k) { return (k % L.me) < ne; });
}
int launch_unpack_frames(LaunchCtx* P, const BatchDev& d, const LlsmPackedLayout& L, const float* in, const float* const* src_tab) {
  if(d.nframes == 0) return 0;
  LAUNCH("k_unpack_frames", k_unpack_frames, dim3((unsigned)d.nframes), dim3(WAVE), 0, d.nframes, L, in, src_tab, d.frm_utt, d.frm_off, d.f0, d.nhar, d.ampl, d.phse,
    d.psd, d.psdres, d.has_psdres, d.edc, d.nhar_e, d.eenv_ampl, d.eenv_phse);
  return 0;
}
int launch_pack_frames(LaunchCtx* P, const BatchDev& d, const LlsmPackedLayout& L, float* out, float* const* dst_tab) {
  if(d.nframes == 0) return 0;
  LAUNCH("k_pack_frames", k_pack_frames, dim3((unsigned)d.nframes), dim3(WAVE), 0, d.nframes, L, d.f0, d.nhar, d.ampl, d.phse,
    d.psd, d.psdres, d.has_psdres, d.edc, d.nhar_e, d.eenv_ampl, d.eenv_phse, out, dst_tab, d.frm_utt, d.frm_off);
  return 0;
}
// the three waveforms of utterance u straight into its page-locked output arrays: tab[3 u + k][p] = array_k[y_off[u] + p]
__global__ __launch_bounds__(256) void k_scatter_outputs(const float* __restrict__ y, const float* __restrict__ ysin,
  const float* __restrict__ ynoise, const int* __restrict__ y_off, const int* __restrict__ ny, float* const* __restrict__ tab) {
  const int u = blockIdx.y;
  const int n = ny[u];
  const size_t o = (size_t)y_off[u];
  const float* src[3] = {y + o, ysin + o, ynoise + o};
  // 16-byte stores into the (64-byte aligned) page-locked arrays, 4-byte-aligned 16-byte loads of the device rows
#pragma unroll
  for(int k = 0; k < 3; k ++) {
    float* d = tab[3 * u + k];
    if(! d) continue;
    for(int p = 4 * (blockIdx.x * 256 + threadIdx.x); p < n; p += 4 * gridDim.x * 256) {
      if(p + 3 < n) { const f4u v = *(const f4u*)(src[k] + p); *(f4a*)(d + p) = f4a{v.x, v.y, v.z, v.w}; }
      else for(int q = p; q < n; q ++) d[q] = src[k][q];
    }
  }
}
int launch_scatter_outputs(LaunchCtx* P, int n_utt, int max_ny, const float* y, const float* ysin, const float* ynoise,
  const int* y_off, const int* ny, float* const* tab) {
  if(n_utt == 0 || max_ny == 0) return 0;
  LAUNCH("k_scatter_outputs", k_scatter_outputs, dim3((unsigned)std::min((max_ny + 255) / 256, 64), (unsigned)n_utt), dim3(256), 0,
    y, ysin, ynoise, y_off, ny, tab);
  return 0;
}

int launch_env_params(LaunchCtx* P, const BatchDev& d, float2* cplx) {
  const size_t total = (size_t)d.nframes * d.nchannel * d.maxnhar_e;
  if(total == 0) return 0;
  LAUNCH("k_env_params", k_env_params, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
    d.f0, d.nhar_e, d.eenv_ampl, d.eenv_phse, d.nframes, d.nchannel, d.maxnhar_e, cplx);
  return 0;
}
// Envelope overlap-add plan (layer0.c:307): hits[p][0 .. 3) = the (frame, window sample) pairs with env_ola(i, j) == p,
// ascending in the frame index -- the order of the reference's sequential loops.  Thread per output sample: the frames
// whose window can reach p are floor((p - nwin) / hop) .. floor(p / hop) + 2, and within a frame env_ola is strictly
// increasing in j, so at most one j matches; it is looked for around p - floor(base_i) with the SAME float32
// expression the host and the oracle use (plan.h env_ola).  over != 0 afterwards: more than three frames on a sample.
__global__ __launch_bounds__(256) void k_env_plan(int max_ny, int max_nfrm, int nwin_env, float thop, float fs,
  int2* __restrict__ hits, int* __restrict__ over) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if(p >= max_ny) return;
  const float hop = lp::fmul(thop, fs);
  int ilo = (int)floorf((float)(p - nwin_env) / hop) - 1, ihi = (int)floorf((float)p / hop) + 3;
  if(ilo < 0) ilo = 0;
  if(ihi > max_nfrm - 1) ihi = max_nfrm - 1;
  int cnt = 0;
  int2 h[LLSM_EXC_HITS];
#pragma unroll
  for(int k = 0; k < LLSM_EXC_HITS; k ++) h[k] = make_int2(-1, -1);
  for(int i = ilo; i <= ihi; i ++) {
    const float base = lp::fmul(lp::fmul((float)(i - 1), thop), fs);
    const int jc = p - (int)floorf(base);
    for(int j = max(jc - 2, 0); j <= min(jc + 2, nwin_env - 1); j ++)
      if(lp::env_ola(i, j, thop, fs) == p) {
        if(cnt < LLSM_EXC_HITS) {
#pragma unroll
          for(int k = 0; k < LLSM_EXC_HITS; k ++) if(k == cnt) h[k] = make_int2(i, j);
        }
        cnt ++;
        break;
      }
  }
  if(cnt > LLSM_EXC_HITS) atomicOr(over, 1);
#pragma unroll
  for(int k = 0; k < LLSM_EXC_HITS; k ++) hits[(size_t)p * LLSM_EXC_HITS + k] = h[k];
}
int launch_env_plan(LaunchCtx* P, int max_ny, int max_nfrm, int nwin_env, float thop, float fs, int2* hits, int* over) {
  if(max_ny <= 0) return 0;
  LAUNCH("k_env_plan", k_env_plan, dim3((max_ny + 255) / 256), dim3(256), 0, max_ny, max_nfrm, nwin_env, thop, fs, hits, over);
  return 0;
}

int launch_excite_env(LaunchCtx* P, const BatchDev& d, const float* colored, int ntemplate_ext,
  const int2* hits, const float2* cplx, int nwin_env, const float* win, int nch_active,
  const int* out_off, const int* out_len, int max_len, float fs_syn, float* yexc) {
  if(d.n_utt == 0 || max_len == 0) return 0;
#define EX_ARGS colored, ntemplate_ext, hits, cplx, d.edc, d.f0, nwin_env, win, d.nchannel, d.maxnhar_e, \
    nch_active, d.frm_off, d.nfrm, out_off, out_len, d.thop, fs_syn, yexc
  // $LLSM_GPU_EXCITE4=1: by template position, four samples per thread (k_excite_env4).  Measured and NOT the default:
  // 0.625 ms against 0.587 - 0.596 ms for the per-sample kernel on the bench batch, with and without explicit FMAs
  // (profiles/r05_b, r05_c kbench lines; LAB.md round 5) -- half the fetches and a third fewer instructions, but 4
  // instead of 8 wavefronts per SIMD and three dependent table / LDS rounds per tile.  Kept for its test and as a base.
  const char* e4 = std::getenv("LLSM_GPU_EXCITE4");   // (read per launch: tests switch it)
  const bool by_template = e4 && e4[0] == '1';
  if(by_template && d.nchannel <= 4 && d.maxnhar_e <= 8) {
    const dim3 grid4((std::min(max_len, 20000) + 1023) / 1024, d.n_utt);
    if(d.maxnhar_e <= 4) LAUNCH("k_excite_env4", (k_excite_env4<4, 4>), grid4, dim3(256), 0, EX_ARGS);
    else LAUNCH("k_excite_env4", (k_excite_env4<4, 8>), grid4, dim3(256), 0, EX_ARGS);
    return 0;
  }
  const dim3 grid((max_len + 255) / 256, d.n_utt);
  if(d.nchannel <= 4 && d.maxnhar_e <= 4) LAUNCH("k_excite_env", (k_excite_env<4, 4>), grid, dim3(256), 0, EX_ARGS);
  else if(d.nchannel <= 4) LAUNCH("k_excite_env", (k_excite_env<4, 8>), grid, dim3(256), 0, EX_ARGS);
  else LAUNCH("k_excite_env", (k_excite_env<8, 8>), grid, dim3(256), 0, EX_ARGS);
#undef EX_ARGS
  return 0;
}

int launch_noise_filter(LaunchCtx* P, const BatchDev& d, const float* yexc,
  const int* out_off, const int* out_len, float fnyq_conf, float fs_syn, int nwin,
  const float* win, float inv_wsqr, int N, int logN, const float2* tw, int tw_nmax,
  float* nframes_out, int* live, int rt) {
  if(d.nframes == 0) return 0;
#define WF_CASE(LN) \
  if(logN == LN) { \
    LAUNCH("k_noise_filter_wf", (k_noise_filter_wf<LN>), dim3(fft_grid(rt ? (d.nframes + 1) / 2 : npairs_of(d)) * (NF_WPE / 2)), dim3(WAVE), \
      sizeof(float2) * (wf_lds_elems<LN>() + d.npsd) + (sizeof(float) << LN), yexc, out_off, out_len, d.frm_utt, d.frm_off, d.nframes, \
      d.psd, d.psdres, d.has_psdres, d.npsd, fnyq_conf, d.thop, fs_syn, nwin, win, inv_wsqr, \
      nframes_out, live, rt, rt ? nullptr : d.pairs, rt ? (d.nframes + 1) / 2 : npairs_of(d)); \
    return 0; \
  }
  WF_CASE(8) WF_CASE(9) WF_CASE(10) WF_CASE(11)      // 4096 and up: the LDS kernel (register budget)
#undef WF_CASE
  const int np = rt ? (d.nframes + 1) / 2 : npairs_of(d);
  if(N > LLSM_LDS_FFT_MAX) {                          // beyond the LDS: global scratch + the big twiddle table (engine.cpp)
    const int grid = std::min(fft_grid(np), llsm_big_fft_grid((size_t)N + N / 2 + 1));
    if(! P -> tw_big || N > P -> tw_big_nmax || P -> big_scratch_elems < (size_t)grid * (N + N / 2 + 1)) return -1;
    LAUNCH("k_noise_filter", k_noise_filter, dim3(grid), dim3(WAVE), 64,
      yexc, out_off, out_len, d.frm_utt, d.frm_off, d.nframes, d.psd, d.psdres, d.has_psdres,
      d.npsd, fnyq_conf, d.thop, fs_syn, nwin, win, inv_wsqr, N, logN, P -> tw_big, P -> tw_big_nmax,
      nframes_out, live, rt, rt ? nullptr : d.pairs, np, P -> big_scratch);
    return 0;
  }
  size_t lds = (size_t)(N + N / 2 + N / 2 + 1) * sizeof(float2) + 16 * sizeof(float);
  lds = (lds + 15) / 16 * 16;
  if(lds > 64 * 1024 &&
     hipFuncSetAttribute((const void*)k_noise_filter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  LAUNCH("k_noise_filter", k_noise_filter, dim3(fft_grid(np)), dim3(WAVE), lds,
    yexc, out_off, out_len, d.frm_utt, d.frm_off, d.nframes, d.psd, d.psdres, d.has_psdres,
    d.npsd, fnyq_conf, d.thop, fs_syn, nwin, win, inv_wsqr, N, logN, tw, tw_nmax,
    nframes_out, live, rt, rt ? nullptr : d.pairs, np, (float2*)nullptr);
  return 0;
}

// Fused noise filter + overlap-add; returns -2 when the transform size has no fused kernel
// (callers then use launch_noise_filter + the gathering mix).
int launch_noise_filter_ola(LaunchCtx* P, const BatchDev& d, const int4* units, int nunits, int halo,
  const float* yexc, const int* out_off, const int* out_len, float fnyq_conf, float fs_syn, int nwin,
  const float* win, int wsym, float inv_wsqr, int logN, float* ynoise) {
  if(nunits == 0) return 0;
#define NFO_ARGS units, nunits, halo, yexc, out_off, out_len, d.frm_off, d.nfrm, d.psd, d.psdres, d.has_psdres, d.npsd, fnyq_conf, \
      d.thop, fs_syn, nwin, win, wsym, inv_wsqr, ynoise
#define WF_CASE(LN) \
  if(logN == LN) { \
    /* target rows inside the exchange buffer when they fit there and in NF_TQ registers per lane */ \
    if(wsym > 0 && d.npsd <= NF_TQ * WAVE && (1 << (LN - 1)) + 7 + d.npsd <= wf_lds_elems<LN>()) { \
      if(d.npsd <= 2 * WAVE) { \
        LAUNCH("k_noise_filter_ola", (k_noise_filter_ola<LN, true, 2>), dim3(nunits), dim3(WAVE), \
          sizeof(float2) * wf_lds_elems<LN>() + (sizeof(float) << LN) + sizeof(float) * (wsym / 2 + 1), NFO_ARGS); \
      } else { \
        LAUNCH("k_noise_filter_ola", (k_noise_filter_ola<LN, true, NF_TQ>), dim3(nunits), dim3(WAVE), \
          sizeof(float2) * wf_lds_elems<LN>() + (sizeof(float) << LN) + sizeof(float) * (wsym / 2 + 1), NFO_ARGS); \
      } \
    } else { \
      LAUNCH("k_noise_filter_ola", (k_noise_filter_ola<LN, false, 1>), dim3(nunits), dim3(WAVE), \
        sizeof(float2) * (wf_lds_elems<LN>() + d.npsd) + (sizeof(float) << LN), NFO_ARGS); \
    } \
    return 0; \
  }
  WF_CASE(8) WF_CASE(9) WF_CASE(10) WF_CASE(11)
#undef WF_CASE
#undef NFO_ARGS
  return -2;
}

int launch_ola_noise_mix(LaunchCtx* P, const BatchDev& d, const float* nframes_in,
  const int* live, int N, const int* out_off, const int* out_len,
  int max_len, float fs_syn, const float* ysin, float* ynoise, float* y) {
  if(d.n_utt == 0 || max_len == 0) return 0;
  LAUNCH("k_ola_noise_mix", k_ola_noise_mix, dim3((max_len + 255) / 256, d.n_utt), dim3(256), 0,
    nframes_in, live, N, d.frm_off, d.nfrm, out_off, out_len, d.thop, fs_syn, ysin, ynoise, y);
  return 0;
}

int launch_rt_template(LaunchCtx* P, const float* colored, int ntemplate_ext, int nch, int nch_active,
  int ntemplate, int S, float* tpl) {
  LAUNCH("k_rt_template", k_rt_template, dim3((ntemplate + 255) / 256, nch, S), dim3(256), 0,
    colored, ntemplate_ext, nch, nch_active, ntemplate, tpl);
  return 0;
}
int launch_rt_rings(LaunchCtx* P, int S, float* mod, float* sinr, float* noiser, int cap, int nch,
  int mod_curr, int sin_curr, int noise_curr, int nhop, int nwin, const float* envf,
  const float* frames_sin, const float* f0, const int* has_nm, const int* nhar) {
  LAUNCH("k_rt_rings", k_rt_rings, dim3(S), dim3(256), 0, mod, sinr, noiser, cap, nch, mod_curr,
    sin_curr, noise_curr, nhop, nwin, envf, frames_sin, f0, has_nm, nhar);
  return 0;
}
int launch_rt_excite(LaunchCtx* P, int S, const float* mod, const float* tpl, float* excr, int cap,
  int nch, int ntemplate, int mod_curr, int exc_curr, int exc_cycle, int curr_nhop, int nx,
  int nwin_frame, float* exc_frame) {
  LAUNCH("k_rt_excite", k_rt_excite, dim3(S), dim3(256), 0, mod, tpl, excr, cap, nch, ntemplate,
    mod_curr, exc_curr, exc_cycle, curr_nhop, nx, nwin_frame, exc_frame);
  return 0;
}
int launch_rt_rings_excite(LaunchCtx* P, int S, float* mod, float* sinr, float* noiser, int cap, int nch,
  int mod_curr, int sin_curr, int noise_curr, int nhop, int nwin, const float* envf, const float* frames_sin,
  const float* f0, const int* has_nm, const int* nhar, const float* tpl, float* excr, int ntemplate, int exc_curr,
  int exc_cycle, float* exc_frame) {
  LAUNCH("k_rt_rings_excite", k_rt_rings_excite, dim3(S), dim3(256), 0, mod, sinr, noiser, cap, nch, mod_curr, sin_curr,
    noise_curr, nhop, nwin, envf, frames_sin, f0, has_nm, nhar, tpl, excr, ntemplate, exc_curr, exc_cycle, exc_frame);
  return 0;
}
int launch_rt_mix(LaunchCtx* P, int S, float* noiser, const float* sinr, int cap, int noise_curr,
  int sin_curr, int sin_pos, int nfft, const float* nframes_in, const int* live, int next_nhop,
  int out_stride, float* out) {
  LAUNCH("k_rt_mix", k_rt_mix, dim3(S), dim3(256), 0, noiser, sinr, cap, noise_curr, sin_curr,
    sin_pos, nfft, nframes_in, live, next_nhop, out_stride, out);
  return 0;
}

// llsmrt, one hop in two launches (k_rt_front, k_rt_back).  d: the per-stream rows as a batch of S one-frame "utterances".
int launch_rt_front(LaunchCtx* P, const BatchDev& d, int nwin, const float* win, const float* f0_sin, const float* cyc_shift,
  float* envf, float* frames_sin, int lds_harmonics, float* mod, float* sinr, float* noiser, int cap, int mod_curr,
  int sin_curr, int noise_curr, int nhop, const int* has_nm, const float* tpl, float* excr, int ntemplate, int exc_curr,
  int exc_cycle, float* exc_frame, const RtRows* host, const RtPbpArgs* pbp) {
  const int S = d.nframes;
  if(S == 0) return 0;
  RtRows hr; std::memset(& hr, 0, sizeof(hr));
  if(host) hr = *host;
  RtPbpArgs pa; std::memset(& pa, 0, sizeof(pa));
  if(pbp) pa = *pbp;
  int T = ((nwin + 15) / 16 + 2 + 31) / 32;
  int NT = T;
  if(T > 4) { T = (T + 3) / 4 * 4; NT = 4; }
  const int L = 32 * T - 2;
  const size_t lds = (lds_harmonics + 4) * sizeof(float2);
#define RF_ARGS d.f0, d.nhar_e, d.eenv_ampl, d.eenv_phse, d.edc, d.nchannel, d.maxnhar_e, d.fs, nwin, win, envf, \
    f0_sin, d.nhar, d.ampl, d.phse, d.maxnhar, d.thop, L, cyc_shift, frames_sin, mod, sinr, noiser, cap, mod_curr, sin_curr, \
    noise_curr, nhop, has_nm, tpl, excr, ntemplate, exc_curr, exc_cycle, exc_frame, hr, d.npsd, d.psd, pa
#define RF_CASE(NCH, ME) \
  switch(NT) { \
    case 1: LAUNCH("k_rt_front", (k_rt_front<NCH, ME, 1>), dim3(S), dim3(256), lds, RF_ARGS); break; \
    case 2: LAUNCH("k_rt_front", (k_rt_front<NCH, ME, 2>), dim3(S), dim3(256), lds, RF_ARGS); break; \
    case 3: LAUNCH("k_rt_front", (k_rt_front<NCH, ME, 3>), dim3(S), dim3(256), lds, RF_ARGS); break; \
    default: LAUNCH("k_rt_front", (k_rt_front<NCH, ME, 4>), dim3(S), dim3(256), lds, RF_ARGS); break; \
  }
  if(d.nchannel <= 4 && d.maxnhar_e <= 4) { RF_CASE(4, 4) }
  else if(d.nchannel <= 4) { RF_CASE(4, 8) }
  else { RF_CASE(8, 8) }
#undef RF_CASE
#undef RF_ARGS
  return 0;
}
int launch_rt_back(LaunchCtx* P, const BatchDev& d, const float* exc_frame, float fnyq_conf, float fs_syn, int nwin,
  const float* win, float inv_wsqr, int N, int logN, const float2* tw, int tw_nmax, float* nframes, int* live,
  float* noiser, const float* sinr, int cap, int noise_curr, int sin_curr, int sin_pos, int next_nhop, int out_stride,
  float* out) {
  const int S = d.nframes;
  if(S == 0) return 0;
  size_t lds = (size_t)(N + N / 2 + N / 2 + 1 + d.npsd) * sizeof(float2) + 16 * sizeof(float);
  lds = (lds + 15) / 16 * 16;
  LAUNCH("k_rt_back", k_rt_back, dim3((S + 1) / 2), dim3(256), lds, exc_frame, S, d.psd, d.psdres, d.has_psdres, d.npsd,
    fnyq_conf, d.thop, fs_syn, nwin, win, inv_wsqr, N, logN, tw, tw_nmax, nframes, live, noiser, sinr, cap, noise_curr,
    sin_curr, sin_pos, next_nhop, out_stride, out);
  return 0;
}

// llsmrt, one hop in ONE launch (harmonic-model buffers): the arguments of launch_rt_front and launch_rt_back
int launch_rt_hop(LaunchCtx* P, const BatchDev& d, int nwin, const float* win, const float* f0_sin, const float* cyc_shift,
  float* envf, float* frames_sin, int lds_harmonics, float* mod, float* sinr, float* noiser, int cap, int mod_curr,
  int sin_curr, int noise_curr, int nhop, const int* has_nm, const float* tpl, float* excr, int ntemplate, int exc_curr,
  int exc_cycle, float* exc_frame, const RtRows* host, float fnyq_conf, float inv_wsqr, int N, int logN, const float2* tw,
  int tw_nmax, float* nframes, int* live, int sin_pos, int next_nhop, int out_stride, float* out, const RtPbpArgs* pbp,
  bool on_chip) {
  const int S = d.nframes;
  if(S == 0) return 0;
  RtRows hr; std::memset(& hr, 0, sizeof(hr));
  if(host) hr = *host;
  RtPbpArgs pa; std::memset(& pa, 0, sizeof(pa));
  if(pbp) pa = *pbp;
  int T = ((nwin + 15) / 16 + 2 + 31) / 32;
  int NT = T;
  if(T > 4) { T = (T + 3) / 4 * 4; NT = 4; }
  const int L = 32 * T - 2;
  const int lds_half = lds_harmonics + 4;
  size_t lds_back = (size_t)(N + N / 2 + N / 2 + 1 + d.npsd) * sizeof(float2) + 16 * sizeof(float);
  size_t lds = std::max((size_t)2 * lds_half * sizeof(float2), lds_back);
  lds = (lds + 15) / 16 * 16;
  if(lds > 64 * 1024) return -1002;
  // the on-chip form (k_rt_hop2) where its per-thread slots and its LDS fit
  const bool hop2_ok = on_chip;
  {
    const int nx0 = std::max(N, 2 * lds_half);
    const int me_rt = d.maxnhar_e > 0 ? d.maxnhar_e : 1;
    size_t lds2 = (size_t)(nx0 + N / 2 + N / 2 + 1 + d.npsd) * sizeof(float2) + 16 * sizeof(float) +
      ((size_t)2 * d.nchannel * nwin + 5 * (size_t)nwin + 2 * (size_t)(d.nchannel + 2 * d.nchannel * me_rt) + 8) * sizeof(float);
    lds2 = (lds2 + 15) / 16 * 16;
    const int nhop_now = nwin / 2;
    if(hop2_ok && nwin == 2 * nhop_now && nwin <= 256 * RT2_JW && nhop_now <= 512 && N <= 256 * 2 * RT2_JN && N > nhop_now &&
       next_nhop <= 512 && next_nhop <= N && N < cap && nwin < cap && nhop_now < ntemplate && d.nchannel * me_rt <= 256 &&
       d.npsd <= 1024 &&
       lds2 <= 64 * 1024) {
      // the sinusoid samples a hop hands out lie behind the window it adds to (sin_pos is negative enough) unless the hop
      // length jumped; pulse-by-pulse buffers add to the ring at positions of their own: those read them at the end
      const int early_out = (sin_pos + next_nhop <= -nwin && ! pa.ops) ? 1 : 0;
      auto ring_host = [cap](int curr, int lag) { return ((curr + lag) % cap + cap) % cap; };
      RtRingBase rb;
      rb.mod0 = ring_host(mod_curr, -nwin); rb.sin0 = ring_host(sin_curr, -nwin); rb.exc0 = ring_host(exc_curr, -nwin);
      rb.noi0 = ring_host(noise_curr, -N); rb.out0 = ring_host(sin_curr, sin_pos); rb.tpl0 = exc_cycle % ntemplate;
#define RH2_ARGS d.f0, d.nhar_e, d.eenv_ampl, d.eenv_phse, d.edc, d.nchannel, d.maxnhar_e, d.fs, nwin, win, \
    f0_sin, d.nhar, d.ampl, d.phse, d.maxnhar, d.thop, L, cyc_shift, mod, sinr, noiser, cap, mod_curr, sin_curr, \
    noise_curr, nhop, has_nm, tpl, excr, ntemplate, exc_curr, exc_cycle, hr, d.npsd, d.psd, lds_half, \
    S, d.psdres, d.has_psdres, fnyq_conf, inv_wsqr, N, logN, tw, tw_nmax, sin_pos, next_nhop, out_stride, out, pa, early_out, rb
#define RH2_CASE(NCH, ME) \
  switch(NT) { \
    case 1: LAUNCH("k_rt_hop2", (k_rt_hop2<NCH, ME, 1>), dim3((S + 1) / 2), dim3(512), lds2, RH2_ARGS); break; \
    case 2: LAUNCH("k_rt_hop2", (k_rt_hop2<NCH, ME, 2>), dim3((S + 1) / 2), dim3(512), lds2, RH2_ARGS); break; \
    case 3: LAUNCH("k_rt_hop2", (k_rt_hop2<NCH, ME, 3>), dim3((S + 1) / 2), dim3(512), lds2, RH2_ARGS); break; \
    default: LAUNCH("k_rt_hop2", (k_rt_hop2<NCH, ME, 4>), dim3((S + 1) / 2), dim3(512), lds2, RH2_ARGS); break; \
  }
      if(d.nchannel <= 4 && d.maxnhar_e <= 4) { RH2_CASE(4, 4) }
      else if(d.nchannel <= 4) { RH2_CASE(4, 8) }
      else { RH2_CASE(8, 8) }
#undef RH2_CASE
#undef RH2_ARGS
      return 0;
    }
  }
#define RH_ARGS d.f0, d.nhar_e, d.eenv_ampl, d.eenv_phse, d.edc, d.nchannel, d.maxnhar_e, d.fs, nwin, win, envf, \
    f0_sin, d.nhar, d.ampl, d.phse, d.maxnhar, d.thop, L, cyc_shift, frames_sin, mod, sinr, noiser, cap, mod_curr, sin_curr, \
    noise_curr, nhop, has_nm, tpl, excr, ntemplate, exc_curr, exc_cycle, exc_frame, hr, d.npsd, d.psd, lds_half, \
    S, d.psdres, d.has_psdres, fnyq_conf, inv_wsqr, N, logN, tw, tw_nmax, nframes, live, sin_pos, next_nhop, out_stride, out, pa
#define RH_CASE(NCH, ME) \
  switch(NT) { \
    case 1: LAUNCH("k_rt_hop", (k_rt_hop<NCH, ME, 1>), dim3((S + 1) / 2), dim3(512), lds, RH_ARGS); break; \
    case 2: LAUNCH("k_rt_hop", (k_rt_hop<NCH, ME, 2>), dim3((S + 1) / 2), dim3(512), lds, RH_ARGS); break; \
    case 3: LAUNCH("k_rt_hop", (k_rt_hop<NCH, ME, 3>), dim3((S + 1) / 2), dim3(512), lds, RH_ARGS); break; \
    default: LAUNCH("k_rt_hop", (k_rt_hop<NCH, ME, 4>), dim3((S + 1) / 2), dim3(512), lds, RH_ARGS); break; \
  }
  if(d.nchannel <= 4 && d.maxnhar_e <= 4) { RH_CASE(4, 4) }
  else if(d.nchannel <= 4) { RH_CASE(4, 8) }
  else { RH_CASE(8, 8) }
#undef RH_CASE
#undef RH_ARGS
  return 0;
}

int launch_utt_fftsize(LaunchCtx* P, const BatchDev& d, int nmax, int* nfft_u) {
  if(d.n_utt == 0) return 0;
  LAUNCH("k_utt_fftsize", k_utt_fftsize, dim3(d.n_utt), dim3(256), 0, d.f0, d.frm_off, d.nfrm, d.fs,
    d.rel_winsize, nmax, nfft_u);
  return 0;
}
int launch_harm_pp(LaunchCtx* P, const BatchDev& d, const float* sig, size_t sig_stride, int nsig,
  const int* nfft_u, int maxnhar, float norm_base, const float2* tw, int tw_nmax, int lds_n,
  int* nhar_out, float* ampl, float* phse) {
  if(d.nframes == 0) return 0;
  size_t lds = (size_t)(lds_n + lds_n / 2 + lds_n / 2 + 2) * sizeof(float2);
  if(lds > 160 * 1024) return -1;
  if(lds > 64 * 1024 &&                                // 8192 points: 128 KB of the CU's 160 KB
     hipFuncSetAttribute((const void*)k_harm_pp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  LAUNCH("k_harm_pp", k_harm_pp, dim3(d.nframes), dim3(WAVE), lds, sig, sig_stride, nsig, d.x_off, d.nx,
    d.frm_utt, d.frm_off, d.f0, nfft_u, d.thop, d.fs, d.rel_winsize, maxnhar, norm_base, tw, tw_nmax,
    lds_n, nhar_out, ampl, phse);
  return 0;
}
// the frames of launch_harm_pp whose transform is larger than lds_n points (nmax: the largest size nfft_u can hold)
int launch_harm_pp_big(LaunchCtx* P, const BatchDev& d, const float* sig, size_t sig_stride, int nsig,
  const int* nfft_u, int maxnhar, float norm_base, int lds_n, int nmax, int* nhar_out, float* ampl, float* phse) {
  if(d.nframes == 0 || nmax <= lds_n) return 0;
  const int grid = std::min(d.nframes, llsm_big_fft_grid((size_t)nmax + nmax / 2 + 2));
  if(! P -> tw_big || nmax > P -> tw_big_nmax || P -> big_scratch_elems < (size_t)grid * (nmax + nmax / 2 + 2)) return -1;
  LAUNCH("k_harm_pp_big", k_harm_pp_big, dim3(grid), dim3(HPP_BIG_NT), 0, sig, sig_stride, nsig, d.x_off, d.nx,
    d.frm_utt, d.frm_off, d.nframes, d.f0, nfft_u, d.thop, d.fs, d.rel_winsize, maxnhar, norm_base,
    P -> tw_big, P -> tw_big_nmax, lds_n, nmax, P -> big_scratch, nhar_out, ampl, phse);
  return 0;
}
