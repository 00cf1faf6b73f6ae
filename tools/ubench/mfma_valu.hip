// Micro-benchmark: do f32 MFMA cycles and VALU cycles of ONE SIMD add or overlap on gfx950?
// Each loop iteration issues 14 independent v_mfma_f32_16x16x4_f32 (the shape of the harmonic-analysis
// inner loop) with V independent v_fma_f32 spread evenly between them, at W wavefronts per SIMD.
// Prints SIMD-cycles per iteration (2.4 GHz) beside the two models 32*14 + c*V (add) and max (overlap).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_valu tools/ubench/mfma_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// G = number of VALU groups per iteration: 14 = V / 14 after every MFMA, 2 = V / 2 after MFMA 7 and 14, 1 = all V after the 14th
template <int V, int G = 14>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
  f32x4 acc[14]; float x[28];
#pragma unroll
  for(int i = 0; i < 14; i ++) acc[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
  for(int i = 0; i < 28; i ++) x[i] = threadIdx.x * 1e-3f + i;
  float av = a + threadIdx.x * 1e-6f, bv = b;
  for(int it = 0; it < iters; it ++) {
#pragma unroll
    for(int m = 0; m < 14; m ++) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(av), "v"(bv));
      if((m + 1) % (14 / G) == 0) {
#pragma unroll
        for(int j = 0; j < V / G; j ++) {
          const int idx = (m * (V / G) + j) % 28;
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[idx]) : "v"(a), "v"(b));
        }
      }
    }
  }
  float s = 0;
#pragma unroll
  for(int i = 0; i < 14; i ++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for(int i = 0; i < 28; i ++) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int V, int G = 14> static void run(int wps) {
  const int blocks = 256 * 4 * wps, iters = 4000;
  float* out; hipMalloc(&out, (size_t)blocks * 64 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V, G>), dim3(blocks), dim3(64), 0, 0, out, 10, 1.0001f, 1e-7f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, G>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 1e-7f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * wps);     // SIMD cycles per iteration of one wavefront's work
  printf("V=%3d  G=%2d  W=%d  %8.3f ms  %7.1f cycles / iteration   (mfma alone 448; add-model %5.0f @2cyc %5.0f @4cyc)\n",
    V, G, wps, ms, cyc, 448.0 + 2.0 * V, 448.0 + 4.0 * V);
  hipFree(out);
}

int main() {
  for(int w = 1; w <= 4; w ++) {
    run<0>(w); run<14>(w); run<28>(w); run<42>(w); run<56>(w); run<112>(w);
    run<28, 7>(w); run<28, 2>(w); run<28, 1>(w); run<56, 7>(w); run<56, 2>(w); run<56, 1>(w); run<112, 2>(w); run<112, 1>(w);
  }
  return 0;
}
