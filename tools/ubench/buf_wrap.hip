// Does a raw buffer load range-check (voffset + inst_offset) with 32-bit wrap-around?
// voffset = (lane - 64) * 4 (negative), inst_offset = 256 bytes: the sum addresses element `lane`.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 256 * 4, 0x00020000);
  const int voff = ((int)threadIdx.x - 64) * 4;
  float a, b;
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:256\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(voff), "s"(r) : "memory");
  b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff + 256, 0, 0));
  out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
}
int main() {
  float h[256], o[128]; for(int i = 0; i < 256; i ++) h[i] = i + 1;
  float *d, *dout; hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  printf("negative voffset + imm 256: lane0 %g lane1 %g lane63 %g (wrap works if 1 2 64)\n", o[0], o[1], o[63]);
  printf("single voffset            : lane0 %g lane1 %g lane63 %g\n", o[64], o[65], o[127]);
  return 0;
}
