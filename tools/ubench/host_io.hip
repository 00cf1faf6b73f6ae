// Micro-benchmark for the llsmrt hop boundary: how should ~230 KB of per-hop parameters reach a 64-workgroup kernel and
// ~115 KB of samples get back to the host, and how should the host learn that they are there?
//   path A  hipMemcpyAsync H2D, kernel (device -> device), hipMemcpyAsync D2H, hipStreamSynchronize       (what rt.cpp does)
//   path B  kernel reads the pinned block itself, writes device memory, hipMemcpyAsync D2H, synchronise
//   path C  kernel reads pinned, writes pinned (dword or dwordx4 stores), synchronise
//   path D  as C, but the host polls a pinned flag the last workgroup sets after a system-scope fence (no synchronise
//           on the critical path)
// Each kernel also dirties `ballast` bytes of device memory (the rings a hop writes), so that an end-of-kernel cache
// write-back has something to do.  Prints host-side microseconds per round trip (median of 400).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/host_io tools/ubench/host_io.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)

// S workgroups; workgroup s reads in_words/S words at in + s * (in_words/S), reduces them (so the reads cannot be
// dropped), writes out_words/S words at out + s * ..., dirties its share of the ballast, and (D) counts itself done.
template <int VEC>
__global__ __launch_bounds__(256) void k_io(const float* __restrict__ in, int in_words, float* __restrict__ out, int out_words,
  float* __restrict__ ballast, int ballast_words, unsigned* __restrict__ done_count, volatile unsigned* __restrict__ flag, unsigned token) {
  const int S = gridDim.x, s = blockIdx.x, tid = threadIdx.x;
  const int ni = in_words / S, no = out_words / S, nb = ballast_words / S;
  const float* pi = in + (size_t)s * ni;
  float acc = 0;
  for(int i = tid * 4; i < ni; i += 256 * 4) { float4 v = *(const float4*)(pi + i); acc += v.x + v.y + v.z + v.w; }
  __shared__ float red[256];
  red[tid] = acc; __syncthreads();
  for(int o = 128; o > 0; o >>= 1) { if(tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  const float sum = red[0];
  float* pb = ballast + (size_t)s * nb;
  for(int i = tid; i < nb; i += 256) pb[i] = sum + i;
  float* po = out + (size_t)s * no;
  if(VEC == 4) { for(int i = tid * 4; i < no; i += 256 * 4) *(float4*)(po + i) = make_float4(sum, sum + 1, sum + 2, (float)token); }
  else { for(int i = tid; i < no; i += 256) po[i] = (i & 3) == 3 ? (float)token : sum + i; }
  if(flag) {
    __threadfence_system();
    __syncthreads();
    if(tid == 0) {
      const unsigned prev = atomicAdd(done_count, 1u);
      if(prev == (unsigned)S - 1) { *done_count = 0; __threadfence_system(); __hip_atomic_store((unsigned*)flag, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
  }
}

static double median(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  const int S = 64, in_words = 64 * 900, out_words = 64 * 448, reps = 400;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *h_in, *h_out, *d_in, *d_out, *d_ballast; unsigned *d_count; unsigned* h_flag;
  CK(hipHostMalloc(&h_in, in_words * 4, hipHostMallocDefault)); CK(hipHostMalloc(&h_out, out_words * 4, hipHostMallocDefault));
  CK(hipHostMalloc(&h_flag, 64, hipHostMallocDefault));
  CK(hipMalloc(&d_in, in_words * 4)); CK(hipMalloc(&d_out, out_words * 4)); CK(hipMalloc(&d_ballast, 16 << 20)); CK(hipMalloc(&d_count, 4));
  CK(hipMemset(d_count, 0, 4));
  for(int i = 0; i < in_words; i ++) h_in[i] = 1e-3f * (i % 97);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for(int ballast_kb : {0, 1024, 8192}) {
    const int bw = ballast_kb * 256;
    printf("-- ballast %d KB dirtied per kernel, %d KB in, %d KB out, %d workgroups\n", ballast_kb, in_words * 4 / 1024, out_words * 4 / 1024, S);
    unsigned token = 1;
    std::vector<double> t;
    // A
    t.clear();
    for(int r = 0; r < reps; r ++, token ++) {
      auto t0 = now();
      CK(hipMemcpyAsync(d_in, h_in, in_words * 4, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_io<4>, dim3(S), dim3(256), 0, st, d_in, in_words, d_out, out_words, d_ballast, bw, d_count, (volatile unsigned*)nullptr, token);
      CK(hipMemcpyAsync(h_out, d_out, out_words * 4, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      t.push_back(us(t0, now()));
      if(h_out[3] != (float)token) { printf("A: wrong token\n"); return 1; }
    }
    printf("A  memcpy in, kernel, memcpy out, synchronise        %7.1f us\n", median(t));
    // B
    t.clear();
    for(int r = 0; r < reps; r ++, token ++) {
      auto t0 = now();
      hipLaunchKernelGGL(k_io<4>, dim3(S), dim3(256), 0, st, h_in, in_words, d_out, out_words, d_ballast, bw, d_count, (volatile unsigned*)nullptr, token);
      CK(hipMemcpyAsync(h_out, d_out, out_words * 4, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      t.push_back(us(t0, now()));
      if(h_out[3] != (float)token) { printf("B: wrong token\n"); return 1; }
    }
    printf("B  kernel reads pinned, memcpy out, synchronise      %7.1f us\n", median(t));
    // B2: memcpy in, kernel writes pinned, synchronise
    t.clear();
    for(int r = 0; r < reps; r ++, token ++) {
      auto t0 = now();
      CK(hipMemcpyAsync(d_in, h_in, in_words * 4, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_io<4>, dim3(S), dim3(256), 0, st, d_in, in_words, h_out, out_words, d_ballast, bw, d_count, (volatile unsigned*)nullptr, token);
      CK(hipStreamSynchronize(st));
      t.push_back(us(t0, now()));
      if(h_out[3] != (float)token) { printf("B2: wrong token\n"); return 1; }
    }
    printf("B2 memcpy in, kernel writes pinned x4, synchronise   %7.1f us\n", median(t));
    // C dword / dwordx4
    for(int vec : {1, 4}) {
      t.clear();
      for(int r = 0; r < reps; r ++, token ++) {
        auto t0 = now();
        if(vec == 4) hipLaunchKernelGGL(k_io<4>, dim3(S), dim3(256), 0, st, h_in, in_words, h_out, out_words, d_ballast, bw, d_count, (volatile unsigned*)nullptr, token);
        else hipLaunchKernelGGL(k_io<1>, dim3(S), dim3(256), 0, st, h_in, in_words, h_out, out_words, d_ballast, bw, d_count, (volatile unsigned*)nullptr, token);
        CK(hipStreamSynchronize(st));
        t.push_back(us(t0, now()));
        if(h_out[3] != (float)token) { printf("C: wrong token\n"); return 1; }
      }
      printf("C  kernel reads + writes pinned (x%d), synchronise    %7.1f us\n", vec, median(t));
    }
    // D: flag polling
    for(int vec : {1, 4}) {
      t.clear();
      std::vector<double> t2;
      for(int r = 0; r < reps; r ++, token ++) {
        auto t0 = now();
        if(vec == 4) hipLaunchKernelGGL(k_io<4>, dim3(S), dim3(256), 0, st, h_in, in_words, h_out, out_words, d_ballast, bw, d_count, (volatile unsigned*)h_flag, token);
        else hipLaunchKernelGGL(k_io<1>, dim3(S), dim3(256), 0, st, h_in, in_words, h_out, out_words, d_ballast, bw, d_count, (volatile unsigned*)h_flag, token);
        while(__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != token) { }
        auto t1 = now();
        t.push_back(us(t0, t1));
        bool ok = true;
        for(int i = 3; i < out_words; i += 4) if(h_out[i] != (float)token) { ok = false; break; }
        if(! ok) { printf("D: samples not all visible when the flag was\n"); return 1; }
        CK(hipStreamSynchronize(st));
        t2.push_back(us(t0, now()));
      }
      printf("D  kernel reads + writes pinned (x%d), host polls flag %7.1f us   (stream idle after %7.1f us)\n", vec, median(t), median(t2));
    }
  }
  return 0;
}
