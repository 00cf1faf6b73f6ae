// Register-resident wavefront FFT for gfx950 (included by kernels.hip after its helpers).
//
// One wavefront transforms N = 2^LOGN complex points, N in [256, 4096], held P = N/64 per
// lane.  Element (lane + 64 m) lives in register m of lane `lane` BOTH on input and on
// output, so point-wise stages before and after a transform (windowing, log magnitude,
// liftering, spectral weighting) need no index permutation and no LDS at all.
//
// Decimation in frequency with the radix plan {R1, R2, ...} (product N, every R <= P):
//   pass j works on sub-transforms of length NJ = N / (R1..R(j-1)); a butterfly (c, b),
//   c = sub-transform, b in [0, NJ/Rj), takes elements b + (NJ/Rj) r, r < Rj, does an
//   Rj-point DFT in registers, multiplies output k by W_NJ^(b k) and hands it to
//   sub-transform c + CJ k (CJ = number of sub-transforms before the pass) at position b.
//   Butterfly beta = c (NJ/Rj) + b runs on lane beta % 64, slot beta / 64; its values sit
//   in registers slot + (P/Rj) r.
// Between passes the wavefront exchanges through LDS with stride NN + NN/Rn float2 per
// sub-transform (NN = next length, Rn = next radix): every read group is conflict free, the
// write groups of the LAST exchange (rows of 4 / 8 float2 at stride 5 / 9) are two-way --
// tools/fft_plan_sim.py models the index algebra and the gfx950 bank rules and reproduces the
// measured SQ_LDS_BANK_CONFLICT to the cycle (320 per frame pair of k_spgm_env_wf<11, 1>).
// A conflict-free layout exists (WF_SWZ below: packed rows, position XOR-ed by a function of
// the sub-transform index) and was measured: it costs more VALU than the conflicts cost LDS.
// A 2048-point transform is 3 register passes and 2 exchanges (128 LDS instructions per
// lane) instead of 6 LDS round trips of a radix-4 in-place FFT.
//
// Twiddles: W^(b k), k < R, are products of the seeds W^(b 2^i) (at most 4 factors); the
// seeds are evaluated once per kernel from exactly reduced integer phases (cs_turns).
// Inverse transforms call the same code with the real and imaginary arrays swapped
// (ifft(x) = swap(fft(swap(x))), unscaled).
#pragma once

// Build switches (tools/kbench.py ablations), both measured in round 6 on one box (profiles/r06_b_kbench_fft_layouts.txt):
//   WF_RD64  1: the exchange reads are kept as single ds_read_b64 (64 banks, two 32-lane groups, 2 LDS cycles each);
//            0 (default): the compiler pairs them into ds_read2_b64 (32 banks, 8 cycles per pair).  The single reads
//            halve the LDS read cycles -- and the kernels got SLOWER (k_spgm_env_wf 0.973 -> 1.035 ms, k_psd_frames_wf
//            0.189 -> 0.191): they are bound by VALU issue, not by the LDS, and the volatile loads cost scheduling freedom.
//   WF_SWZ   the exchange layout (WfEx below).  1 / 2 remove EVERY bank conflict (SQ_LDS_BANK_CONFLICT 32.77 M -> 0 per
//            launch of k_spgm_env_wf, as tools/fft_plan_sim.py predicts to the cycle) -- for address arithmetic that the
//            padded layout's immediate offsets do not need: k_psd_frames_wf +10 %, k_spgm_env_wf +-0, and with 2 the
//            hoisted XOR terms spill (2.08 ms).  0 (default): the padded strides of rounds 2 - 5, two-way conflicts on the
//            writes of the last exchange (17 % of the LDS cycles of a kernel whose LDS is busy a third of the time).
#ifndef WF_RD64
#define WF_RD64 0
#endif
typedef float wf_v2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) wf_v2 wf_lds_v2;
// one ds_read_b64 that the compiler will not pair with a neighbour into ds_read2_b64
DEV float2 lds_rd64(const float2* p) {
#if WF_RD64
  const wf_v2 v = *(volatile wf_lds_v2*)p; return make_float2(v.x, v.y);
#else
  return *p;
#endif
}

template <int LOGN> struct WfPlan;
template <> struct WfPlan<8>  { static constexpr int NP = 4; static constexpr int r(int j) { return 4; } };
template <> struct WfPlan<9>  { static constexpr int NP = 3; static constexpr int r(int j) { return 8; } };
template <> struct WfPlan<10> { static constexpr int NP = 3; static constexpr int r(int j) { return j < 2 ? 16 : 4; } };
template <> struct WfPlan<11> { static constexpr int NP = 3; static constexpr int r(int j) { return j < 2 ? 16 : 8; } };
template <> struct WfPlan<12> { static constexpr int NP = 3; static constexpr int r(int j) { return 16; } };

constexpr int wf_log2(int v) { return v <= 1 ? 0 : 1 + wf_log2(v >> 1); }
// sub-transform length before pass j
template <int LOGN> constexpr int wf_len(int j) { return j == 0 ? (1 << LOGN) : wf_len<LOGN>(j - 1) / WfPlan<LOGN>::r(j - 1); }
// seeds needed by pass j: log2(R) per distinct b (pass 0 has P/R slots with their own b)
template <int LOGN> constexpr int wf_nseed_pass(int j) {
  const int R = WfPlan<LOGN>::r(j), NJ = wf_len<LOGN>(j);
  if(NJ / R <= 1) return 0;
  const int slots = j == 0 ? ((1 << LOGN) / WAVE) / R : 1;
  return slots * wf_log2(R);
}
template <int LOGN> constexpr int wf_seed_base(int j) { return j == 0 ? 0 : wf_seed_base<LOGN>(j - 1) + wf_nseed_pass<LOGN>(j - 1); }
template <int LOGN> constexpr int wf_nseed() { return wf_seed_base<LOGN>(WfPlan<LOGN>::NP); }
#ifndef WF_SWZ
#define WF_SWZ 0                                     // 0: padded strides everywhere (rounds 2 - 5); 1: XOR-swizzled packed rows below 32 points; 2: every exchange
#endif
// layout of the exchange after pass j: element `pos` of sub-transform `cc` (length NN) -> float2 index.
// Packed rows (stride NN) with pos ^ f(cc): a read group (32 lanes) takes NBN consecutive positions of 32 / NBN
// consecutive sub-transforms; min(NN, 32) / NBN of them share a residue class of the 32 float2 banks, and f sends those to
// distinct NBN-blocks.  A write group (16 lanes) covers 16 consecutive float2 of one or more whole rows whatever f is.
template <int LOGN, int J> struct WfEx {
  static constexpr int RN = WfPlan<LOGN>::r(J + 1), NN = wf_len<LOGN>(J + 1), NBN = NN / RN;
  static constexpr bool SWZ = WF_SWZ == 2 || (WF_SWZ == 1 && NN < 32);
  static constexpr int ST = SWZ ? NN : NN + NN / RN;
  static constexpr int SH = (SWZ && NN < 32) ? wf_log2(32 / NN) : 0;
  static constexpr int FM = SWZ ? (NN < 32 ? NN : 32) / NBN : 1;
  DEV static int idx(int cc, int pos) {
    if constexpr (SWZ) return cc * NN + (pos ^ (((cc >> SH) & (FM - 1)) * NBN));
    else return cc * ST + pos;
  }
  static constexpr int elems = ((1 << LOGN) / NN) * ST;
};
// LDS float2 needed by the exchanges of one transform
template <int LOGN, int J = 0> constexpr int wf_lds_from() {
  if constexpr (J >= WfPlan<LOGN>::NP - 1) return 0;
  else return WfEx<LOGN, J>::elems > wf_lds_from<LOGN, J + 1>() ? WfEx<LOGN, J>::elems : wf_lds_from<LOGN, J + 1>();
}
template <int LOGN> constexpr int wf_lds_elems() { return wf_lds_from<LOGN, 0>(); }

template <int LOGN> struct WfTw { float c[wf_nseed<LOGN>()], s[wf_nseed<LOGN>()]; };

template <int LOGN, int J>
DEV void wf_seed_pass(WfTw<LOGN>& tw, int lane) {
  if constexpr (J < WfPlan<LOGN>::NP) {
    constexpr int R = WfPlan<LOGN>::r(J), NJ = wf_len<LOGN>(J), NB = NJ / R, LR = wf_log2(R);
    if constexpr (NB > 1) {
      constexpr int slots = J == 0 ? ((1 << LOGN) / WAVE) / R : 1;
#pragma unroll
      for(int s = 0; s < slots; s ++) {
        const int b = (lane + WAVE * s) & (NB - 1);
#pragma unroll
        for(int i = 0; i < LR; i ++) {
          const int ph = (b << i) & (NJ - 1);                        // exact phase index mod NJ
          float c, sn; cs_turns((double)ph * (1.0 / (double)NJ), & c, & sn);
          tw.c[wf_seed_base<LOGN>(J) + s * LR + i] = c;
          tw.s[wf_seed_base<LOGN>(J) + s * LR + i] = -sn;            // forward: e^{-j ...}
        }
      }
    }
    wf_seed_pass<LOGN, J + 1>(tw, lane);
  }
}
template <int LOGN> DEV void wf_init(WfTw<LOGN>& tw, int lane) { wf_seed_pass<LOGN, 0>(tw, lane); }

// ---------------------------------------------------------------- register DFTs (forward)
#define WF_SQH 0.70710678118654752f
template <int R> DEV void wf_dft(float (&a)[R], float (&b)[R]);

template <> DEV void wf_dft<2>(float (&a)[2], float (&b)[2]) {
  const float r0 = a[0] + a[1], i0 = b[0] + b[1], r1 = a[0] - a[1], i1 = b[0] - b[1];
  a[0] = r0; b[0] = i0; a[1] = r1; b[1] = i1;
}
DEV void wf_dft4v(float& a0, float& b0, float& a1, float& b1, float& a2, float& b2, float& a3, float& b3) {
  const float t0r = a0 + a2, t0i = b0 + b2, t1r = a0 - a2, t1i = b0 - b2;
  const float t2r = a1 + a3, t2i = b1 + b3;
  const float t3r = b1 - b3, t3i = a3 - a1;                          // (x1 - x3) * (-j)
  a0 = t0r + t2r; b0 = t0i + t2i;
  a1 = t1r + t3r; b1 = t1i + t3i;
  a2 = t0r - t2r; b2 = t0i - t2i;
  a3 = t1r - t3r; b3 = t1i - t3i;
}
template <> DEV void wf_dft<4>(float (&a)[4], float (&b)[4]) {
  wf_dft4v(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3]);
}
template <> DEV void wf_dft<8>(float (&a)[8], float (&b)[8]) {
  // even / odd 4-point transforms, then X[k] = E[k] + W8^k O[k], X[k+4] = E[k] - W8^k O[k]
  wf_dft4v(a[0], b[0], a[2], b[2], a[4], b[4], a[6], b[6]);          // E[0..3] in 0,2,4,6
  wf_dft4v(a[1], b[1], a[3], b[3], a[5], b[5], a[7], b[7]);          // O[0..3] in 1,3,5,7
  float er[4] = {a[0], a[2], a[4], a[6]}, ei[4] = {b[0], b[2], b[4], b[6]};
  float orr[4], oi[4];
  orr[0] = a[1]; oi[0] = b[1];
  orr[1] = (a[3] + b[3]) * WF_SQH; oi[1] = (b[3] - a[3]) * WF_SQH;   // * (1 - j)/sqrt2
  orr[2] = b[5]; oi[2] = -a[5];                                      // * (-j)
  orr[3] = (b[7] - a[7]) * WF_SQH; oi[3] = -(a[7] + b[7]) * WF_SQH;  // * (-1 - j)/sqrt2
#pragma unroll
  for(int k = 0; k < 4; k ++) {
    a[k] = er[k] + orr[k]; b[k] = ei[k] + oi[k];
    a[k + 4] = er[k] - orr[k]; b[k + 4] = ei[k] - oi[k];
  }
}
template <> DEV void wf_dft<16>(float (&a)[16], float (&b)[16]) {
  // n = 4 n1 + n2, k = k1 + 4 k2: 4-point transforms over n1, twiddle W16^(n2 k1), 4-point over n2
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f;   // cos, sin(pi/8)
#pragma unroll
  for(int n2 = 0; n2 < 4; n2 ++)
    wf_dft4v(a[n2], b[n2], a[4 + n2], b[4 + n2], a[8 + n2], b[8 + n2], a[12 + n2], b[12 + n2]);
  // after this a[4 k1 + n2] = A[n2][k1]; multiply by W16^(n2 k1) = (c, -s)
  auto rot = [&](int idx, float c, float s) {
    const float r = a[idx] * c + b[idx] * s, i = b[idx] * c - a[idx] * s; a[idx] = r; b[idx] = i; };
  rot(4 * 1 + 1, C1, S1);            // W16^1
  rot(4 * 1 + 2, WF_SQH, WF_SQH);    // W16^2
  rot(4 * 1 + 3, S1, C1);            // W16^3
  rot(4 * 2 + 1, WF_SQH, WF_SQH);    // W16^2
  { const float r = b[4 * 2 + 2], i = -a[4 * 2 + 2]; a[4 * 2 + 2] = r; b[4 * 2 + 2] = i; }   // W16^4 = -j
  rot(4 * 2 + 3, -WF_SQH, WF_SQH);   // W16^6
  rot(4 * 3 + 1, S1, C1);            // W16^3
  rot(4 * 3 + 2, -WF_SQH, WF_SQH);   // W16^6
  rot(4 * 3 + 3, -C1, -S1);          // W16^9
#pragma unroll
  for(int k1 = 0; k1 < 4; k1 ++)
    wf_dft4v(a[4 * k1], b[4 * k1], a[4 * k1 + 1], b[4 * k1 + 1], a[4 * k1 + 2], b[4 * k1 + 2],
      a[4 * k1 + 3], b[4 * k1 + 3]);
  // now a[4 k1 + k2] = X[k1 + 4 k2]: transpose the 4 x 4 index in registers
#pragma unroll
  for(int k1 = 0; k1 < 4; k1 ++)
#pragma unroll
    for(int k2 = k1 + 1; k2 < 4; k2 ++) {
      float t = a[4 * k1 + k2]; a[4 * k1 + k2] = a[4 * k2 + k1]; a[4 * k2 + k1] = t;
      t = b[4 * k1 + k2]; b[4 * k1 + k2] = b[4 * k2 + k1]; b[4 * k2 + k1] = t;
    }
}

// ---------------------------------------------------------------- one pass
template <int LOGN, int J, int P>
DEV void wf_pass(float (&xr)[P], float (&xi)[P], const WfTw<LOGN>& tw) {
  constexpr int R = WfPlan<LOGN>::r(J), NJ = wf_len<LOGN>(J), NB = NJ / R, S = P / R, LR = wf_log2(R);
  constexpr int slots = J == 0 ? S : 1;
  float wr[R], wi[R];
#pragma unroll
  for(int s = 0; s < S; s ++) {
    float a[R], b[R];
#pragma unroll
    for(int r = 0; r < R; r ++) { a[r] = xr[s + S * r]; b[r] = xi[s + S * r]; }
    wf_dft<R>(a, b);
    if constexpr (NB > 1) {
      if(s < slots) {                                // powers of this slot's twiddle from its seeds
        constexpr int base = wf_seed_base<LOGN>(J);
#pragma unroll
        for(int k = 1; k < R; k ++) {
          const int low = k & (-k);                  // lowest set bit: w^k = w^(k - low) * seed[log2 low]
          const int sd = base + s * LR + __builtin_ctz(low);
          if(k == low) {
            // opaque copy: keeps the 2 (R - 1) powers from being hoisted out of the caller's
            // frame loop, where they would stay live (~90 VGPRs for N = 2048) and force spills
            float sc = tw.c[sd], ss = tw.s[sd];
            asm volatile("" : "+v"(sc), "+v"(ss));
            wr[k] = sc; wi[k] = ss;
          } else {
            const float pr = wr[k - low], pi = wi[k - low];
            wr[k] = pr * wr[low] - pi * wi[low];
            wi[k] = pr * wi[low] + pi * wr[low];
          }
        }
      }
#pragma unroll
      for(int k = 1; k < R; k ++) {
        const float r = a[k] * wr[k] - b[k] * wi[k], i = a[k] * wi[k] + b[k] * wr[k];
        a[k] = r; b[k] = i;
      }
    }
#pragma unroll
    for(int k = 0; k < R; k ++) { xr[s + S * k] = a[k]; xi[s + S * k] = b[k]; }
  }
}

// ---------------------------------------------------------------- LDS exchange after pass J
template <int LOGN, int J, int P>
DEV void wf_exchange(float (&xr)[P], float (&xi)[P], float2* lds, int lane) {
  using EX = WfEx<LOGN, J>;
  constexpr int R = WfPlan<LOGN>::r(J), RN = WfPlan<LOGN>::r(J + 1);
  constexpr int NN = wf_len<LOGN>(J + 1), CJ = (1 << LOGN) / wf_len<LOGN>(J);
  constexpr int S = P / R, SN = P / RN, NBN = NN / RN;
#pragma unroll
  for(int s = 0; s < S; s ++) {
    const int beta = lane + WAVE * s;
    const int c = beta / NN, b = beta % NN;
#pragma unroll
    for(int k = 0; k < R; k ++)
      lds[EX::idx(c + CJ * k, b)] = make_float2(xr[s + S * k], xi[s + S * k]);
  }
  __syncthreads();
#pragma unroll
  for(int s2 = 0; s2 < SN; s2 ++) {
    const int beta2 = lane + WAVE * s2;
    const int c = beta2 / NBN, b2 = beta2 % NBN;
#pragma unroll
    for(int r2 = 0; r2 < RN; r2 ++) {
#if WF_RD64
      const wf_v2 v = *(volatile wf_lds_v2*)(lds + EX::idx(c, b2 + NBN * r2));
#else
      const float2 v = lds[EX::idx(c, b2 + NBN * r2)];
#endif
      xr[s2 + SN * r2] = v.x; xi[s2 + SN * r2] = v.y;
    }
  }
  __syncthreads();
}

template <int LOGN, int J, int P>
DEV void wf_run(float (&xr)[P], float (&xi)[P], const WfTw<LOGN>& tw, float2* lds, int lane) {
  wf_pass<LOGN, J, P>(xr, xi, tw);
  if constexpr (J + 1 < WfPlan<LOGN>::NP) {
    wf_exchange<LOGN, J, P>(xr, xi, lds, lane);
    wf_run<LOGN, J + 1, P>(xr, xi, tw, lds, lane);
  }
}

// forward, unnormalised: X[k] = sum_t x[t] e^{-2 pi j k t / N}; element lane + 64 m <-> register m.
// Inverse (unscaled): wave_fft<LOGN>(xi, xr, ...).  lds: wf_lds_elems<LOGN>() float2, private
// to the wavefront.
template <int LOGN>
DEV void wave_fft(float (&xr)[(1 << LOGN) / WAVE], float (&xi)[(1 << LOGN) / WAVE],
  const WfTw<LOGN>& tw, float2* lds, int lane) {
  wf_run<LOGN, 0, (1 << LOGN) / WAVE>(xr, xi, tw, lds, lane);
}

// Mirror helpers for the real-signal packing (two real frames per complex transform).
// Bin k = lane + 64 m; its mirror N - k sits in lane' = (64 - lane) % 64, register P - 1 - m
// (lane 0: own register (P - m) % P).  One cross-lane read per register, no LDS storage.
//
// wave_mirror_lo: xm[m] = value at the mirrored bin, for the bins of the lower half only
// (m < P/2 on every lane, m = P/2 meaningful on lane 0: the Nyquist bin).
template <int P>
DEV void wave_mirror_lo(const float (&x)[P], float (&xm)[P / 2 + 1], int lane) {
  const int pl = (WAVE - lane) & (WAVE - 1);
#pragma unroll
  for(int m = 0; m <= P / 2; m ++) {
    const float other = __shfl(x[P - 1 - m], pl, WAVE);
    xm[m] = lane == 0 ? x[(P - m) & (P - 1)] : other;
  }
}
// wave_reflect: fill the upper-half bins (k > N/2) of x with what the owner of the mirrored
// lower-half bin offers in v[m] (m < P/2).  Lane 0 keeps its x[P/2] (the Nyquist bin).
template <int P, int NV>
DEV void wave_reflect(const float (&v)[NV], float (&x)[P], int lane) {
  const int pl = (WAVE - lane) & (WAVE - 1);
#pragma unroll
  for(int M = P / 2; M < P; M ++) {
    const float other = __shfl(v[P - 1 - M], pl, WAVE);
    x[M] = lane == 0 ? (M > P / 2 ? v[P - M] : x[M]) : other;
  }
}
