// batch.h -- internal definitions shared by engine.cpp and l1.cpp: the device context, the
// device-resident batch and small allocation helpers.  Not installed.
#ifndef LLSM_AMD_BATCH_H
#define LLSM_AMD_BATCH_H
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "engine.h"
#include "kernels.h"
#include "llsm_gpu.h"

#define HIP_OK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if(e_ != hipSuccess) {                                                             \
      llsm_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));               \
      return -1;                                                                       \
    }                                                                                  \
  } while(0)

const float2* llsm_engine_twiddles(llsm_gpu_context* c, int* nmax);
hipError_t llsm_dev_malloc(void** p, size_t bytes);
void llsm_dev_free(void* p);

// ----------------------------------------------------------------- context
struct ProfPending { std::string name; hipEvent_t a, b; };
struct ProfEntry { double ms; int launches; };

struct llsm_gpu_context {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // second stream of the analysis: the Kalman smoother (HBM-bound) runs there beside the band filter (float64-bound);
  // forked and joined with the two events, created on first use
  hipStream_t aux = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  float2* tw = nullptr;
  FiltSectionD* sections = nullptr;          // Chebyshev block tables for one-shot filtering (llsm_engine_chebyfilt)
  int tw_nmax = 0;
  bool profiling = false;
  bool prof_skip = false;                     // the launch being bracketed is not the one llsm_gpu_profile_only named
  std::string prof_only;                      // empty: every kernel
  std::vector<ProfPending> pending;
  std::vector<hipEvent_t> pool;
  std::map<std::string, ProfEntry> prof;
  std::vector<std::string> prof_names;       // stable storage for get_profile
  LaunchCtx lc;
};


// grow-only page-locked host array (the subset of std::vector the layer-1 scheduler uses): its copies to and from the
// device are DMA transfers at link speed instead of the runtime's staged copies of pageable memory (~10 GB/s: the 16 MB
// of pulse tables of a 1024-utterance batch took 1.5 ms of every use_l1 synthesis).  Contents are not initialised.
template <class T> struct PinVec {
  T* p = nullptr; size_t n = 0, cap = 0; bool pinned = false;
  PinVec() = default;
  PinVec(const PinVec&) = delete; PinVec& operator=(const PinVec&) = delete;
  ~PinVec() { drop(); }
  void drop() { if(p) { if(pinned) (void)hipHostFree(p); else std::free(p); } p = nullptr; n = cap = 0; }
  // false: neither page-locked nor pageable memory could be had -- the array keeps its old size and contents, and the
  // caller reports the failure (a NULL block used to be copied into).  Portable: the block is page-locked
  // for EVERY device (the fan-out runs one scheduler per device), as PBuf's blocks are.
  bool resize(size_t count) {
    if(count > cap) {
      const size_t want = count + count / 4 + 16;
      T* q = nullptr; bool qp = true;
      if(hipHostMalloc((void**)& q, want * sizeof(T), hipHostMallocPortable) != hipSuccess) {   // (no device / no memory: pageable)
        (void)hipGetLastError(); q = (T*)std::malloc(want * sizeof(T)); qp = false;
      }
      if(! q) { llsm_set_error("host allocation failed (scheduler tables)"); return false; }
      if(p && n) std::memcpy(q, p, n * sizeof(T));
      const size_t keep = n;
      drop();
      p = q; cap = want; pinned = qp; n = keep;
    }
    n = count;
    return true;
  }
  T* data() { return p; } const T* data() const { return p; }
  size_t size() const { return n; } bool empty() const { return n == 0; }
  T& operator[](size_t i) { return p[i]; } const T& operator[](size_t i) const { return p[i]; }
};

template <class T> struct DevBuf {
  T* p = nullptr; size_t n = 0;
  int alloc(size_t count) {
    if(count <= n && p) return 0;
    if(p) { hipDeviceSynchronize(); llsm_dev_free(p); }   // regrow: earlier launches may still read it
    p = nullptr; n = 0;
    if(count == 0) return 0;
    hipError_t e = llsm_dev_malloc((void**)& p, count * sizeof(T));
    if(e != hipSuccess) { llsm_set_error(std::string("hipMalloc: ") + hipGetErrorString(e)); return -1; }
    n = count;
    return 0;
  }
  void release() { llsm_dev_free(p); p = nullptr; n = 0; }        // callers synchronise the stream first
};

struct llsm_gpu_batch {
  llsm_gpu_context* ctx;
  llsm_gpu_layout lay;
  llsm_aoptions opt; std::vector<float> chanfreq;
  float fs;
  float fnyq = 0;                         // LLSM_CONF_FNYQ of the parameters (axis of the PSD rows); default fs / 2
  std::vector<int> nx, nfrm, ny, x_off, frm_off, y_off;
  int max_nx = 0, max_ny = 0;
  float min_f0 = 0;                       // smallest voiced F0 seen by upload (0: unknown)
  bool f0_unknown = false;                // the F0 row's device address was handed out: min_f0 stays 0 until a FULL F0 upload (partial blob uploads keep it unknown)
  // plan constants (analysis, from opt.thop and fs)
  int nwin_sin, nwin_psd, nfft_psd, nfft_spgm, nspec;
  // user-visible flat arrays
  void* arr[LLSM_GPU_NARRAYS]; size_t arr_bytes[LLSM_GPU_NARRAYS];
  // The eleven parameter rows (F0 ... EENV_PHSE, kParamIds order) are pieces of ONE device block: a host that keeps its
  // staging rows at the same offsets moves them with one copy per direction (llsm_gpu_batch_transfer_params) -- small
  // device-to-host copies cost ~0.1 ms apiece whatever their size.  arr[] of those ids point into the block.
  void* pblock = nullptr; size_t pblock_bytes = 0; size_t pblock_off[11] = {0};
  DevBuf<float> packed;                     // the analysed rows as one record per frame (packed.h), formed on demand by llsm_gpu_batch_download_packed
  // index tables
  DevBuf<int> d_nx, d_nfrm, d_ny, d_x_off, d_frm_off, d_y_off, d_frm_utt;
  void* blob_stage = nullptr;                          // page-locked staging of llsm_gpu_batch_upload_blobs (64 MiB, on first use)
  DevBuf<int2> d_pairs; int npairs = 0;              // per-utterance frame pairs (kernels.h BatchDev::pairs)
  DevBuf<int2> d_hblocks; int nhblocks = 0;          // 16-aligned frame blocks per utterance (BatchDev::hblocks)
  // scratch
  DevBuf<float> ce, mid, iir_tmp, iir_edge[2], iir_seg[2], env, psd_log, pbuf;   // pbuf: Kalman forward checkpoints; iir_edge: scratch of the end jobs of fused band-pass jobs; iir_seg: of the time segments of few long signals
  DevBuf<int2> spgm_fix; DevBuf<int> spgm_fix_count;  // frame pairs whose spectrogram edge bins are recomputed exactly (k_spgm_env_wf FIX)
  DevBuf<float> colored, yexc, nframes;
  DevBuf<float2> env_cplx;                           // a_k e^{j phi_k} per (frame, channel, harmonic)
  DevBuf<int2> env_hits;                             // [max_ny][LLSM_EXC_HITS] envelope OLA plan
  DevBuf<int> env_over;                              // plan overflow flag
  DevBuf<int4> nf_units; int n_nf_units = 0, nf_halo = 0;   // work units of the fused noise filter + overlap-add
  DevBuf<int4> sin_units; int n_sin_units = 0, sin_halo = 0; // ... and of the fused harmonic frames + overlap-add
  DevBuf<int> live;
  DevBuf<float> win_sin, win_psd, win_env, win_filt;
  DevBuf<FiltSectionD> sections; DevBuf<FiltJob> jobs_ana, jobs_syn;
  int njobs_ana = 0, njobs_syn = 0, nch_active = 0;
  const void* key_ana[3] = {nullptr, nullptr, nullptr};   // scratch pointers the job tables embed
  const void* key_syn[3] = {nullptr, nullptr, nullptr};
  float inv_wpow = 0, norm_base = 0, norm_base_blackman = 0;
  DevBuf<int> nfft_u;
  // synthesis plan cache
  float syn_fs = 0; int nwin_env = 0, nwin_filt = 0, nfft_filt = 0; float inv_wsqr = 0;
  // layer 1 / pulse-by-pulse synthesis (l1.cpp)
  struct Effect { llsm_fgfm modifier = nullptr; void* info = nullptr; llsm_container* frame = nullptr; };
  int l1_nspec = 0;                       // 0: layer-1 arrays not allocated
  int maxnhar_conf = -1;                  // LLSM_CONF_MAXNHAR for llsm_frame_tolayer0, < 0: absent
  std::vector<Effect> effects;            // per frame (LLSM_FRAME_PBPEFF)
  std::vector<int> l1_had_hm;             // HAS_HM as uploaded by llsm_synthesize_batch
  DevBuf<double> l1_model_inv_t, l1_model_cumlog_t;     // Rd fit: 1 / M[c][j] and prefix sums of log M[c][j], transposed [j][c]
  DevBuf<float> l1_model_power, l1_model_param, l1_rd_raw, l1_cont, l1_f0_hm, l1_pulse_buf, l1_mixw, l1_hm_frames, l1_zero, l1_src_ampl;
  DevBuf<int> l1_prev, l1_next, l1_blk_off, l1_select;
  // rows the pulse scheduler reads on the host (l1.cpp), fetched before the noise branch is enqueued
  // rows the pulse scheduler reads, as ONE page-locked block [proj F doubles | f0 | rd | nvs | pbpsyn | has_hm, F values each]:
  // the device packs them into l1_proj (28 F bytes) and they come down in one copy (six copies of 0.8 MB cost ~0.1 ms apiece
  // while the device waits for the scheduler)
  struct L1Rows { PinVec<char> block; float *f0 = nullptr, *rd = nullptr; double* proj = nullptr; int *nvs = nullptr, *pbpsyn = nullptr, *has_hm = nullptr; bool valid = false; } l1_rows;
  PinVec<PbpJob> h_jobs; PinVec<PbpPulse> h_pulses; PinVec<PbpSeg> h_segs; PinVec<int2> h_blk;   // merged scheduler tables (host, page-locked)
  // per-utterance tables of the pulse scheduler, kept with the batch: cleared, not freed, so that a repeated call appends
  // into storage it already owns (4 vectors x 1024 utterances grown from empty were a third of the scheduler's time)
  struct L1UttPlan { std::vector<PbpJob> jobs; std::vector<PbpPulse> pulses; std::vector<PbpSeg> segs; std::vector<int2> blk;
                     std::vector<double> offsets; size_t pulse_total = 0; int size_max = 64; };
  std::vector<L1UttPlan> l1_plans;
  DevBuf<double> l1_alpha; DevBuf<float> l1_alpha_key;   // per-frame alpha cache of the LF model and its (Rd, F0) keys [2][F]
  DevBuf<double> l1_proj;                // next-cycle projection per frame (k_l1_projection), followed by the packed rows: 3.5 F doubles in all
  DevBuf<PbpJob> l1_jobs; DevBuf<PbpPulse> l1_pulses; DevBuf<PbpSeg> l1_segs; DevBuf<int2> l1_blk_jobs;
};


template <class T> inline int upload_arr(DevBuf<T>& d, const T* h, size_t n) {
  if(d.alloc(n)) return -1;
  if(n == 0) return 0;
  HIP_OK(hipMemcpy(d.p, h, n * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}
template <class T> inline int upload_vec(DevBuf<T>& d, const std::vector<T>& h) {
  if(d.alloc(h.size())) return -1;
  if(h.empty()) return 0;
  HIP_OK(hipMemcpy(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}


#endif
