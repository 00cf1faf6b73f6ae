// launch.h -- the launch macro of the kernel translation units: HIP-event profiling hooks around every launch
#pragma once
#include "kernels.h"
static inline void prof_begin(LaunchCtx* P, const char* name) {
  if(P -> prof_begin) P -> prof_begin(P -> prof_user, name, P -> stream);
}
static inline void prof_end(LaunchCtx* P) {
  if(P -> prof_end) P -> prof_end(P -> prof_user, P -> stream);
}
#define LAUNCH(name, kern, grid, block, lds, ...)                                    \
  do {                                                                               \
    prof_begin(P, name);                                                             \
    hipLaunchKernelGGL(kern, grid, block, lds, P -> stream, __VA_ARGS__);            \
    prof_end(P);                                                                     \
    hipError_t e_ = hipGetLastError();                                               \
    if(e_ != hipSuccess) return (int)e_;                                             \
  } while(0)
