// engine.h -- internal glue shared by engine.cpp, capi.cpp and rt.cpp.
#ifndef LLSM_AMD_ENGINE_H
#define LLSM_AMD_ENGINE_H
#include <string>
#include "llsm_gpu.h"

void llsm_set_error(const std::string& msg);
// Process-wide default context used by the drop-in entry points
// (llsm_analyze, llsm_synthesize, llsm_create_rtsynth_buffer): device
// $LLSM_GPU_DEVICE (default 0), created on first use.  NULL when no GPU.
llsm_gpu_context* llsm_default_context(void);
unsigned long long llsm_next_seed(void);

// engine internals used by rt.cpp
struct LaunchCtx;
llsm_gpu_batch* llsm_engine_template_batch(llsm_gpu_context* ctx, const llsm_aoptions* opt,
  float fs, int S, int len, unsigned long long seed);
const float* llsm_engine_batch_colored(llsm_gpu_batch* b);
int llsm_engine_batch_nch_active(llsm_gpu_batch* b);
LaunchCtx* llsm_engine_launch_ctx(llsm_gpu_context* c);
int llsm_engine_device(llsm_gpu_context* c);
int llsm_engine_big_fft(llsm_gpu_context* c, int N, size_t elems);

int llsm_engine_batch_harmonics(llsm_gpu_batch* b, int refine_only);
int llsm_engine_chebyfilt(llsm_gpu_context* c, const float* d_src, int n, float c1, float c2, int square, float* d_dst);

unsigned long llsm_engine_config_epoch(void);          // bumped by every process-wide setting a batch bakes in at creation
int llsm_conv_hann_periodic(void);
int llsm_conv_filtfilt_pad(void);
int llsm_conv_lf_rd_clamp(void);

// l1.cpp
int llsm_l1_prefetch_rows(llsm_gpu_batch* b, const llsm_soptions* so);
int llsm_l1_synthesize_harmonics(llsm_gpu_batch* b, const llsm_soptions* so, const float* ynoise,
  float* ysin, float* yout);
int llsm_l1_prepare_batch(llsm_gpu_batch* b, llsm_chunk** src, int n_utt, const int* fo, int nspec);
int llsm_l1_writeback_hm(llsm_gpu_batch* b, llsm_chunk** src, int n_utt, const int* fo);
#endif
