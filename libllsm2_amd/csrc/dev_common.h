// dev_common.h -- device helpers shared by kernels.hip and l1_kernels.hip: wavefront reductions,
// exact-phase sincos, and the in-place LDS wavefront FFT (see the comment block above fft_dif in kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"                              // RtPbpOp

#define WAVE 64
#define DEV __device__ __forceinline__

DEV float wave_sum(float v) {
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

DEV float wave_max(float v) {
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
  return v;
}

// (cos, sin)(2*pi*turns), turns in float64.  The phase is reduced to [-1/2, 1/2] turns in
// float64, then to a quarter turn r in [-1/2, 1/2] (units of pi/2) in float32, where
// sin(pi r / 2) and cos(pi r / 2) are evaluated by their Taylor polynomials (truncation
// < 2e-9 on that range, i.e. below float32 rounding) and rotated back by quadrant.
DEV void cs_turns(double turns, float* c, float* s) {
  const float y = (float)((turns - rint(turns)) * 4.0);      // quarter turns, |y| <= 2
  const float k = rintf(y);
  const float r = y - k, r2 = r * r;
  float sn = fmaf(r2, 1.6044118478735982e-4f, -4.681754135318688e-3f);
  sn = fmaf(r2, sn, 7.969262624616704e-2f);
  sn = fmaf(r2, sn, -6.459640975062462e-1f);
  sn = fmaf(r2, sn, 1.5707963267948966f) * r;
  float cs = fmaf(r2, -2.5202042373060605e-5f, 9.1926027483942658e-4f);
  cs = fmaf(r2, cs, -2.0863480763352960e-2f);
  cs = fmaf(r2, cs, 2.5366950790104800e-1f);
  cs = fmaf(r2, cs, -1.2337005501361697f);
  cs = fmaf(r2, cs, 1.0f);
  const int q = (int)k & 3;                                  // rotate by q quarter turns
  const float c1 = (q & 1) ? -sn : cs, s1 = (q & 1) ? cs : sn;
  *c = (q & 2) ? -c1 : c1;
  *s = (q & 2) ? -s1 : s1;
}

// cs_turns with the quarter-turn reduction done in float64 as well: RELATIVE accuracy at every zero of the sine and
// of the cosine, not only at turns = 0.  (cs_turns converts to float32 before it subtracts the quarter-turn index, so
// sin(2 pi (1/2 + e)) comes back with an absolute error of ~1e-7 -- fine for a phasor, several per cent of the value
// when e ~ 1e-6 and the value is the numerator of a 0 / 0 ratio: the Dirichlet kernels of the layer-1 envelope.)
DEV void cs_turns_rel(double turns, float* c, float* s) {
  const double y = (turns - rint(turns)) * 4.0;
  const double kd = rint(y);
  const float r = (float)(y - kd), r2 = r * r;
  float sn = fmaf(r2, 1.6044118478735982e-4f, -4.681754135318688e-3f);
  sn = fmaf(r2, sn, 7.969262624616704e-2f);
  sn = fmaf(r2, sn, -6.459640975062462e-1f);
  sn = fmaf(r2, sn, 1.5707963267948966f) * r;
  float cs = fmaf(r2, -2.5202042373060605e-5f, 9.1926027483942658e-4f);
  cs = fmaf(r2, cs, -2.0863480763352960e-2f);
  cs = fmaf(r2, cs, 2.5366950790104800e-1f);
  cs = fmaf(r2, cs, -1.2337005501361697f);
  cs = fmaf(r2, cs, 1.0f);
  const int q = (int)kd & 3;
  const float c1 = (q & 1) ? -sn : cs, s1 = (q & 1) ? cs : sn;
  *c = (q & 2) ? -c1 : c1;
  *s = (q & 2) ? -s1 : s1;
}

DEV float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
DEV float2 caddf(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
DEV float2 csubf(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
DEV int brevN(int k, int logN) { return (int)(__brev((unsigned)k) >> (32 - logN)); }

// PAD (round 4): element i of the transform lives at X[i + (i >> 5)] -- one float2 of padding per 32 elements (= one sweep
// over the 64 banks).  The in-place stages access X at strides of 4^s elements; unpadded, the stages with a stride below 32
// elements put 4 - 8 lanes of a half-wavefront on one bank pair (measured: 63 % of the LDS cycles of k_pbp_pulse and
// k_l1_frame were bank conflicts, profiles/r04_a_l1_pmc_sq_set2.txt).  Callers that index X themselves use fft_px(i) as
// well and provide N + N / 32 elements.  Same arithmetic, same results; only the storage layout differs.
template <bool PAD> DEV int fft_px(int i) { return PAD ? i + (i >> 5) : i; }

template <int NT = WAVE, bool PAD = false>
DEV void fft_dif(float2* X, const float2* tw, int tw_stride, int M, int logM, int lane) {
  int span = M;
  if(logM & 1) {                                    // leading radix-2 stage, half = M/2
    const int h = M >> 1;
    for(int j = lane; j < h; j += NT) {
      const float2 a = X[fft_px<PAD>(j)], b = X[fft_px<PAD>(j + h)];
      X[fft_px<PAD>(j)] = caddf(a, b);
      X[fft_px<PAD>(j + h)] = cmulf(csubf(a, b), tw[j * tw_stride]);
    }
    __syncthreads();
    span = h;
  }
  const int q4 = M >> 2;
  for(; span >= 4; span >>= 2) {
    const int Q = span >> 2;
    const int twm = tw_stride * (M / span);         // e^{-2 pi i k / span} = tw[k * twm]
    for(int j = lane; j < q4; j += NT) {
      const int k = j & (Q - 1);
      const int b0 = ((j - k) << 2) + k;
      const int i0 = fft_px<PAD>(b0), i1 = fft_px<PAD>(b0 + Q), i2 = fft_px<PAD>(b0 + 2 * Q), i3 = fft_px<PAD>(b0 + 3 * Q);
      const float2 a0 = X[i0], a1 = X[i1], a2 = X[i2], a3 = X[i3];
      const float2 w1 = tw[k * twm], w2 = tw[2 * k * twm];
      const float2 w3 = cmulf(w1, w2);
      const float2 t0 = caddf(a0, a2), t1 = csubf(a0, a2), t2 = caddf(a1, a3);
      const float2 d = csubf(a1, a3);
      const float2 t3 = make_float2(d.y, -d.x);     // * (-j)
      X[i0] = caddf(t0, t2);
      X[i1] = cmulf(csubf(t0, t2), w2);
      X[i2] = cmulf(caddf(t1, t3), w1);
      X[i3] = cmulf(csubf(t1, t3), w3);
    }
    __syncthreads();
  }
}

template <int NT = WAVE, bool PAD = false>
DEV void ifft_dit(float2* X, const float2* tw, int tw_stride, int M, int logM, int lane) {
  const int q4 = M >> 2;
  int Q = 1;
  for(int st = 0; st < (logM >> 1); st ++, Q <<= 2) {
    const int twm = tw_stride * (M / (4 * Q));
    for(int j = lane; j < q4; j += NT) {
      const int k = j & (Q - 1);
      const int b0 = ((j - k) << 2) + k;
      const int i0 = fft_px<PAD>(b0), i1 = fft_px<PAD>(b0 + Q), i2 = fft_px<PAD>(b0 + 2 * Q), i3 = fft_px<PAD>(b0 + 3 * Q);
      const float2 x0 = X[i0], x1 = X[i1], x2 = X[i2], x3 = X[i3];
      float2 w1 = tw[k * twm], w2 = tw[2 * k * twm];
      w1.y = -w1.y; w2.y = -w2.y;                   // conjugate twiddles
      const float2 w3 = cmulf(w1, w2);
      const float2 p1 = cmulf(x1, w2), p2 = cmulf(x2, w1), p3 = cmulf(x3, w3);
      const float2 u0 = caddf(x0, p1), u1 = csubf(x0, p1), sm = caddf(p2, p3);
      const float2 d = csubf(p2, p3);
      const float2 dj = make_float2(-d.y, d.x);     // * (+j)
      X[i0] = caddf(u0, sm);
      X[i1] = caddf(u1, dj);
      X[i2] = csubf(u0, sm);
      X[i3] = csubf(u1, dj);
    }
    __syncthreads();
  }
  if(logM & 1) {                                    // trailing radix-2 stage, half = M/2
    const int h = M >> 1;
    for(int j = lane; j < h; j += NT) {
      float2 w = tw[j * tw_stride]; w.y = -w.y;
      const float2 a = X[fft_px<PAD>(j)], b = cmulf(X[fft_px<PAD>(j + h)], w);
      X[fft_px<PAD>(j)] = caddf(a, b);
      X[fft_px<PAD>(j + h)] = csubf(a, b);
    }
    __syncthreads();
  }
}

template <int NT = WAVE>
DEV void load_twiddles(float2* tw, const float2* __restrict__ tw_glob, int N, int tw_nmax, int lane) {
  const int stride = tw_nmax / N;                   // table holds e^{-2 pi i k / tw_nmax}
  for(int k = lane; k < N / 2; k += NT) tw[k] = tw_glob[k * stride];
}

// llsmrt pulse-by-pulse bookkeeping of one hop for stream s, by 256 threads `tid` (llsmrt.c:118-128, 380-419):
// dual-buffer forward, the new pulse group added, the windowed read into the sinusoid ring and the trapezoid catch-up at
// termination.  All streams of a group share the ring cursors (lock-step hops).  Every thread of the workgroup calls it
// (the barriers are unconditional; on = false: a half of k_rt_hop's workgroup without a stream).
DEV void rt_pbp_body(const RtPbpOp* __restrict__ ops, float* __restrict__ frwd, float* __restrict__ bkwd, int cap, int dual_curr,
  float* __restrict__ sinr, int sin_curr, int nhop, const float* __restrict__ win,
  const float* __restrict__ pulse_out, int pulse_stride, int s, int tid, bool on) {
  RtPbpOp op; op.add_off = op.add_size = op.rd_off = op.rd_on = op.term_off = op.term_size = op.pad0 = op.pad1 = 0;
  if(on) op = ops[s];
  float* fw = frwd + (size_t)s * cap; float* bk = bkwd + (size_t)s * cap; float* sr = sinr + (size_t)s * cap;
  for(int i = tid; on && i < nhop; i += 256) {                 // llsm_dualbuffer_forward, buffer.h:183-189
    const int idx = (dual_curr + i) % cap;
    bk[idx] = fw[idx]; fw[idx] = 0.0f;
  }
  const int curr = (dual_curr + nhop) % cap;
  __syncthreads();
  auto at = [&](int off) { return ((curr + off) % cap + cap) % cap; };
  if(op.add_size > 0) {                                        // llsm_dualbuffer_addchunk, buffer.h:193-204
    int before = op.add_off > 0 ? 0 : -op.add_off; if(before > op.add_size) before = op.add_size;
    const float* src = pulse_out + (size_t)s * pulse_stride;
    for(int i = tid; i < op.add_size; i += 256) {
      if(i < before) bk[at(op.add_off + i)] += src[i]; else fw[at(op.add_off + i)] += src[i];
    }
  }
  __syncthreads();
  auto rd = [&](int off, int size, int i) {                    // llsm_dualbuffer_readchunk, buffer.h:168-179
    int before = off > 0 ? 0 : -off; if(before > size) before = size;
    return i < before ? bk[at(off + i)] : fw[at(off + i)];
  };
  if(op.rd_on) {
    for(int j = tid; j < 2 * nhop; j += 256) {
      const int pos = ((sin_curr + op.rd_off + j) % cap + cap) % cap;
      sr[pos] += rd(op.rd_off, 2 * nhop, j) * win[j];
    }
  }
  __syncthreads();
  if(op.term_size > 0) {
    for(int j = tid; j < op.term_size; j += 256) {
      float v = rd(op.term_off, op.term_size, j);
      if(j < nhop) v *= win[j];
      if(j >= op.term_size - nhop) v *= win[j - (op.term_size - nhop) + nhop];
      const int pos = ((sin_curr + op.term_off + j) % cap + cap) % cap;
      sr[pos] += v;
    }
  }
}
