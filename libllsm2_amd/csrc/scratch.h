// scratch.h -- device scratch of one API call on the default context (frameapi.cpp, coder.cpp): buffers are
// freed, after a stream synchronisation, when the object goes out of scope.
#ifndef LLSM_AMD_SCRATCH_H
#define LLSM_AMD_SCRATCH_H
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "batch.h"

namespace {
// device scratch of one call: freed on scope exit
struct Scratch {
  llsm_gpu_context* ctx = nullptr; LaunchCtx* P = nullptr; hipStream_t st = nullptr;
  std::vector<void*> blocks; bool bad = false;
  bool open() {
    ctx = llsm_default_context();
    if(! ctx) return false;
    if(hipSetDevice(llsm_engine_device(ctx)) != hipSuccess) { llsm_set_error("hipSetDevice failed"); return false; }
    P = llsm_engine_launch_ctx(ctx); st = P -> stream;
    return true;
  }
  template <class T> T* alloc(size_t n) {
    void* p = nullptr;
    if(hipMalloc(& p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { bad = true; llsm_set_error("per-frame API: hipMalloc failed"); return nullptr; }
    blocks.push_back(p);
    return (T*)p;
  }
  template <class T> T* up(const T* h, size_t n) {
    T* d = alloc<T>(n);
    if(d && n && hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, st) != hipSuccess) { bad = true; llsm_set_error("per-frame API: upload failed"); }
    return d;
  }
  template <class T> bool down(T* h, const T* d, size_t n) {
    if(bad) return false;
    if(n && hipMemcpyAsync(h, d, n * sizeof(T), hipMemcpyDeviceToHost, st) != hipSuccess) { bad = true; llsm_set_error("per-frame API: download failed"); }
    return ! bad;
  }
  bool sync() {
    if(hipStreamSynchronize(st) != hipSuccess) { bad = true; llsm_set_error("per-frame API: kernel failed"); }
    return ! bad;
  }
  bool run(int rc, const char* what) {
    if(rc != 0) { bad = true; llsm_set_error(std::string(what) + ": launch failed"); }
    return ! bad;
  }
  ~Scratch() { if(st) (void)hipStreamSynchronize(st); for(void* p : blocks) (void)hipFree(p); }
};
}  // namespace
#endif
