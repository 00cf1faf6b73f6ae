// kbench_experiments.h -- bodies of the timing experiments of tools/kbench.py.  NOT part of the product: kernels.hip
// includes this file only under -DLLSM_KBENCH_EXPERIMENTS (which tools/kbench.py adds to its ablation builds under
// exp_build/); the product build has empty hooks in their place and refuses the switches below.
//   RT2_TIMING           thread 0 of workgroup 0 stamps the constant 100 MHz clock at the phase boundaries of the last
//                        llsmrt hop; rt.cpp prints the differences with LLSM_TIMING=1
//   IIR_FAKE_L2          every k_filtfilt job streams through the buffers of job 0 / 1, which stay in L2: what remains is
//                        the cost of the recursion without HBM traffic.  THE RESULTS ARE GARBAGE BY DESIGN.
//   IIR_GEN_EXPERIMENT   the Gaussian templates generated inside the first forward pass of the synthesis jobs instead of
//                        being read (k_white's arithmetic, not its seeds)
//   HT_TABLE_EXPERIMENT  k_harm_speech_tile takes the next k-step's twiddles from a table in global memory (zeros, L2-resident)
//                        instead of rotating them: what a per-F0 twiddle table would cost.  GARBAGE RESULTS.
#pragma once

#ifdef RT2_TIMING
__device__ unsigned long long g_rt2_ts[16];
#define RT2_T(i) do { if(blockIdx.x == 0 && threadIdx.x == 0) g_rt2_ts[i] = wall_clock64(); } while(0)
extern "C" void llsm_rt2_timing_fetch(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rt2_ts), sizeof(g_rt2_ts)); }
#else
#define RT2_T(i)
#endif

#ifdef IIR_FAKE_L2
#define IIR_EXP_JOB(job, jobs, j) \
  do { const FiltJob j0_ = (jobs)[(j) & 1]; if(j0_.n >= (job).n) { (job).src = j0_.src; (job).tmp = j0_.tmp; (job).dst = j0_.dst; (job).mid = j0_.mid; } } while(0)
#else
#define IIR_EXP_JOB(job, jobs, j)
#endif

#ifdef IIR_GEN_EXPERIMENT
template <class V>
__device__ __forceinline__ bool iir_exp_gen(bool fwd, bool square, const float* src, const float* gen_src, int idx0, V& q) {
  if(!(fwd && ! square && src == gen_src)) return false;
#pragma unroll
  for(int e = 0; e < 4; e ++) {
    float u1, u2; llsm_plan::rng_uniforms((unsigned long long)(size_t)src, (unsigned long long)(idx0 + e), & u1, & u2);
    q[e] = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
  }
  return true;
}
#define IIR_EXP_GEN(fwd, square, src, gen_src, idx0, q) iir_exp_gen(fwd, square, src, gen_src, idx0, q)
#else
#define IIR_EXP_GEN(fwd, square, src, gen_src, idx0, q) false
#endif

#ifdef HT_TABLE_EXPERIMENT
__device__ float2 g_ht_exp_tab[65536];                // 512 KB: the size of one F0's table (184 k-steps x 7 tiles x 64 lanes)
template <class T>
__device__ __forceinline__ bool ht_exp_twiddle(int nt, int ks, int tt, T& wr, T& wi) {
  const float2 t = g_ht_exp_tab[(((ks + 1) * nt + tt) * 64 + (threadIdx.x & 63)) & 65535];
  wr = t.x; wi = t.y;
  return true;
}
#define HT_EXP_TWIDDLE(NT, ks, tt, wr, wi) ht_exp_twiddle(NT, ks, tt, wr, wi)
#else
#define HT_EXP_TWIDDLE(NT, ks, tt, wr, wi) false
#endif
