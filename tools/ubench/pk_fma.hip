// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_fma_f64 on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pk_fma tools/ubench/pk_fma.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  // 16 independent accumulator chains per lane so that latency is never the limit
  float x[16]; float2v p[8]; double d[8];
#pragma unroll
  for(int i = 0; i < 16; i ++) x[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
  for(int i = 0; i < 8; i ++) { p[i] = (float2v){x[2 * i], x[2 * i + 1]}; d[i] = x[i]; }
  const float2v pa = {a, a}, pb = {b, b};
  for(int it = 0; it < iters; it ++) {
    if(MODE == 0) {
#pragma unroll
      for(int i = 0; i < 16; i ++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    } else if(MODE == 1) {
#pragma unroll
      for(int i = 0; i < 8; i ++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
    } else if(MODE == 2) {
#pragma unroll
      for(int i = 0; i < 8; i ++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)a), "v"((double)b));
    } else if(MODE == 3) {
#pragma unroll
      for(int i = 0; i < 8; i ++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
    } else if(MODE == 4) {
#pragma unroll
      for(int i = 0; i < 8; i ++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
    } else if(MODE == 5) {
#pragma unroll
      for(int i = 0; i < 16; i ++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    } else if(MODE == 6) {                           // packed add, two different register pairs as sources (butterfly shape)
#pragma unroll
      for(int i = 0; i < 8; i ++) asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 3) & 7]));
    } else if(MODE == 7) {                           // packed fma with op_sel / neg modifiers (complex rotation second half)
#pragma unroll
      for(int i = 0; i < 8; i ++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(pb));
    } else if(MODE == 8) {                           // packed mul, second source a scalar-register pair
#pragma unroll
      for(int i = 0; i < 8; i ++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "s"(pa));
    } else if(MODE == 9) {                           // 2 : 1 mix of packed adds and scalar-form fmas
#pragma unroll
      for(int i = 0; i < 8; i ++) {
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0;
#pragma unroll
  for(int i = 0; i < 16; i ++) s += x[i];
#pragma unroll
  for(int i = 0; i < 8; i ++) s += p[i].x + p[i].y + (float)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> static double run(const char* name, int instr_per_iter, int flop_per_instr) {
  const int blocks = 256 * 8, iters = 20000;      // 8 wavefronts per SIMD
  float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 1e-7f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-7f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * instr_per_iter;   // wavefront-level instructions
  const double per_simd_cycle = winstr / (256.0 * 4) / (ms * 1e-3 * 2.4e9);
  printf("%-14s %8.3f ms  %6.3f wave-instr / SIMD / cycle(2.4GHz)  = %5.2f cycles per instr, %7.1f TFLOP/s\n", name, ms,
    per_simd_cycle, 1.0 / per_simd_cycle, winstr * 64 * flop_per_instr / (ms * 1e-3) / 1e12);
  hipFree(out);
  return ms;
}

int main() {
  run<0>("v_fma_f32", 16, 2);
  run<1>("v_pk_fma_f32", 8, 4);
  run<2>("v_fma_f64", 8, 2);
  run<3>("v_pk_add_f32", 8, 2);
  run<4>("v_pk_mul_f32", 8, 2);
  run<5>("v_add_f32", 16, 1);
  run<6>("v_pk_add 3reg", 8, 2);
  run<7>("v_pk_fma opsel", 8, 4);
  run<8>("v_pk_mul sgpr", 8, 2);
  run<9>("pk_add+fma mix", 16, 2);
  return 0;
}
