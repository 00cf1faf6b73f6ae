// Micro-benchmark: latency of DEPENDENT instruction chains on gfx950 (one wavefront per SIMD, nothing to overlap with):
// what a latency-bound recursion (k_kalman: ~55 dependent instructions per frame) pays per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dep_latency tools/ubench/dep_latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
  float x = threadIdx.x * 1e-3f + 1.0f, y = x + 0.5f;
  float2v p = {x, y}; const float2v pa = {a, a}, pb = {b, b};
  for(int it = 0; it < iters; it ++) {
#pragma unroll
    for(int i = 0; i < 16; i ++) {
      if(MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
      else if(MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(pa), "v"(pb));
      else if(MODE == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
      else if(MODE == 3) { x = b / (x + a); }                         // IEEE division (div_scale / rcp / fma x 4 / div_fmas / div_fixup)
      else if(MODE == 4) { x = b * __builtin_amdgcn_rcpf(x + a); }    // reciprocal + multiply
      else if(MODE == 5) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(a), "v"(b)); }   // two chains
      else if(MODE == 6) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a)); }
      else if(MODE == 7) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(pa)); }
      else if(MODE == 8) { asm volatile("v_fma_f32 %0, %0, %1, %2\n s_nop 0" : "+v"(x) : "v"(a), "v"(b)); }
      else if(MODE == 9) { asm volatile("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(a), "v"(b) : "vcc"); }
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x + y + p.x + p.y;
}

template <int MODE> static void run(const char* name, int instr_per_rep, int blocks) {
  const int iters = 20000;
  float* out; (void)hipMalloc(&out, blocks * 64 * sizeof(float));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10, 1.0001f, 1e-7f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 1e-7f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double reps = (double)iters * 16;
  printf("%-26s %4d wavefronts: %8.3f ms  %7.1f cycles(2.4GHz) per repetition (%d instr) = %6.1f per instruction\n", name, blocks, ms,
    ms * 1e-3 * 2.4e9 / reps, instr_per_rep, ms * 1e-3 * 2.4e9 / reps / instr_per_rep);
  (void)hipFree(out);
}

int main() {
  for(int blocks : {256, 1024, 2048, 4096}) {
    run<0>("v_fma_f32 chain", 1, blocks);
    run<1>("v_pk_fma_f32 chain", 1, blocks);
    run<6>("v_add_f32 chain", 1, blocks);
    run<7>("v_pk_add_f32 chain", 1, blocks);
    run<2>("v_rcp_f32 chain", 1, blocks);
    run<3>("x = b / (x + a) IEEE", 11, blocks);
    run<4>("x = b * rcp(x + a)", 3, blocks);
    run<5>("two v_fma chains", 2, blocks);
    run<8>("v_fma + s_nop 0", 2, blocks);
    run<9>("v_cmp -> vcc -> cndmask", 2, blocks);
  }
  return 0;
}
