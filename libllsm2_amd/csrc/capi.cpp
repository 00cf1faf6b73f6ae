// capi.cpp -- the reference's own entry points for the hot path, as thin
// wrappers over the batch engine: llsm_analyze (layer0.c:478-511),
// llsm_synthesize (layer0.c:636-664), their batched forms, and the
// llsm_chunk <-> flat-row converters (the AoS container tree of llsm.h is
// flattened to the SoA rows the kernels read).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "engine.h"
#include "llsm_gpu.h"
#include "model_internal.h"
#include "packed.h"
#include "plan.h"

namespace lp = llsm_plan;

// ---------------------------------------------------------- default context
static std::mutex g_ctx_mutex;
static llsm_gpu_context* g_ctx = nullptr;
static std::atomic<unsigned long long> g_seed(0x5EEDull);

llsm_gpu_context* llsm_default_context(void) {
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  if(g_ctx) return g_ctx;
  int dev = 0;
  const char* env = std::getenv("LLSM_GPU_DEVICE");
  if(env) dev = std::atoi(env);
  g_ctx = llsm_gpu_create_context(dev, nullptr);
  return g_ctx;
}
unsigned long long llsm_next_seed(void) { return g_seed.fetch_add(1); }
extern "C" void llsm_gpu_set_default_seed(unsigned long long seed) { g_seed.store(seed); }

// ------------------------------------------------------------ flat storage
namespace {
// grow-only page-locked host array: the staging buffers of a worker live as long as the process, so the
// copies to and from the device run at the PCIe rate instead of the pageable-memory rate
template <class T> struct PBuf {
  T* p = nullptr; size_t cap = 0, n = 0;
  T* data() { return p; }
  size_t size() const { return n; }
  void assign(size_t count, T v) {
    reserve(count);
    n = count;
    for(size_t i = 0; i < n; i ++) p[i] = v;
  }
  // room for `count` elements, contents unspecified (every caller overwrites them)
  void resize(size_t count) { reserve(count); n = count; }
  void reserve(size_t count) {
    if(count <= cap) return;
    release();
    const size_t want = count + count / 4 + 64;
    if(hipHostMalloc((void**)& p, want * sizeof(T), hipHostMallocPortable) == hipSuccess) { cap = want; pageable = false; }
    else { p = (T*)std::malloc(want * sizeof(T)); cap = p ? want : 0; pageable = true; }   // no device: plain memory (callers fail later)
  }
  void release() {                                      // by the allocator that made it
    if(p) { if(pageable) std::free(p); else (void)hipHostFree(p); }
    p = nullptr; cap = 0; pageable = false;
  }
  bool pageable = false;
  ~PBuf() { release(); }
};

// Staging rows of one block of utterances: ONE page-locked block laid out as the batch's device block of parameter rows
// (engine.cpp llsm_gpu_batch_params_layout), so that the eleven rows cross the link in one copy per direction.
struct FlatHost {
  int maxnhar = 0, maxnhar_e = 0, npsd = 0, nch = 0, F = 0;
  PBuf<char> block; size_t total = 0, off[11] = {0};
  float *f0 = nullptr, *ampl = nullptr, *phse = nullptr, *psd = nullptr, *psdres = nullptr, *edc = nullptr, *eamp = nullptr, *ephs = nullptr;
  int *nhar = nullptr, *nhar_e = nullptr, *has_psdres = nullptr;
  // room only: a download overwrites every element and llsm_chunk_to_flat writes every element of every row (defaults
  // included) -- filling 3 KB per frame first was a second pass over the staging memory of each block
  void layout(llsm_gpu_batch* b, const llsm_gpu_layout& L) {
    F = L.total_frames; maxnhar = L.maxnhar; maxnhar_e = L.maxnhar_e; npsd = L.npsd; nch = L.nchannel;
    int ids[11];
    llsm_gpu_batch_params_layout(b, & total, off, ids);  // ids: F0 NHAR AMPL PHSE PSD PSDRES HAS_PSDRES EDC NHAR_E EENV_AMPL EENV_PHSE
    block.resize(total ? total : 1);
    char* p = block.data();
    f0 = (float*)(p + off[0]); nhar = (int*)(p + off[1]); ampl = (float*)(p + off[2]); phse = (float*)(p + off[3]);
    psd = (float*)(p + off[4]); psdres = (float*)(p + off[5]); has_psdres = (int*)(p + off[6]); edc = (float*)(p + off[7]);
    nhar_e = (int*)(p + off[8]); eamp = (float*)(p + off[9]); ephs = (float*)(p + off[10]);
  }
  llsm_flat_params view() {
    llsm_flat_params v;
    v.maxnhar = maxnhar; v.maxnhar_e = maxnhar_e; v.npsd = npsd; v.nchannel = nch;
    v.f0 = f0; v.nhar = nhar; v.ampl = ampl; v.phse = phse;
    v.psd = psd; v.psdres = psdres; v.has_psdres = has_psdres;
    v.edc = edc; v.nhar_e = nhar_e; v.eenv_ampl = eamp; v.eenv_phse = ephs;
    return v;
  }
};

int chunk_nfrm(llsm_chunk* c) {
  int* n = (int*)llsm_container_get(c -> conf, LLSM_CONF_NFRM);
  return n ? *n : -1;
}
}  // namespace

// Row copies of llsm_chunk_to_flat.  The destination is staging memory: written once here, read once by the copy engine.
// Ordinary stores first READ every destination line into the cache (read for ownership) and write it back later --
// three DRAM transfers per line copied where two are needed, and with eight workers flattening at once the host's memory
// bandwidth is what llsm_synthesize_batch waits for.  Streaming (non-temporal) stores skip the read and keep the
// destination out of the cache; rows start 16-byte aligned whenever their width is a multiple of four values (the
// defaults), anything else takes memcpy.  Measured with 8 workers (profiles/r04_v_flat_nt.txt): llsm_synthesize_batch of
// 1024 utterances 46 - 50 -> 45 ms.
static inline void row_copy(FP_TYPE* dst, const FP_TYPE* src, size_t n) {
#if defined(__SSE2__)
  if(n >= 16 && ((uintptr_t)dst & 15) == 0) {
    size_t i = 0;
    for(; i + 4 <= n; i += 4) _mm_stream_ps(dst + i, _mm_loadu_ps(src + i));
    for(; i < n; i ++) dst[i] = src[i];
    return;
  }
#endif
  std::memcpy(dst, src, sizeof(FP_TYPE) * n);
}
static inline void row_fill(FP_TYPE* dst, FP_TYPE v, size_t n) {
  for(size_t i = 0; i < n; i ++) dst[i] = v;
}

// Frame i of `src` -> row frm_off + i.  Missing HM / eenv rows become nhar 0;
// harmonics beyond the flat row width are dropped (callers size the rows from
// the chunk, see scan_chunk below).
extern "C" int llsm_chunk_to_flat(llsm_chunk* src, llsm_flat_params* dst, int frm_off) {
  int nfrm = chunk_nfrm(src);
  if(nfrm < 0) return -1;
  const int me = dst -> maxnhar_e > 0 ? dst -> maxnhar_e : 1;
  for(int i = 0; i < nfrm; i ++) {
    llsm_container* fr = src -> frames[i];
    const size_t g = (size_t)frm_off + i;
    FP_TYPE* f0 = (FP_TYPE*)llsm_container_get(fr, LLSM_FRAME_F0);
    llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(fr, LLSM_FRAME_HM);
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(fr, LLSM_FRAME_NM);
    FP_TYPE* res = (FP_TYPE*)llsm_container_get(fr, LLSM_FRAME_PSDRES);
    dst -> f0[g] = f0 ? *f0 : 0;
    // rows are copied and padded in blocks (this loop is most of what llsm_synthesize_batch does on the host)
    const int mh = dst -> maxnhar, npsd = dst -> npsd, nch = dst -> nchannel;
    const int nh = hm ? (hm -> nhar < mh ? (hm -> nhar > 0 ? hm -> nhar : 0) : mh) : 0;
    dst -> nhar[g] = nh;
    FP_TYPE* ar = dst -> ampl + g * (size_t)mh; FP_TYPE* pr = dst -> phse + g * (size_t)mh;
    if(nh > 0) { row_copy(ar, hm -> ampl, (size_t)nh); row_copy(pr, hm -> phse, (size_t)nh); }
    if(mh > nh) { row_fill(ar + nh, 0, (size_t)(mh - nh)); row_fill(pr + nh, 0, (size_t)(mh - nh)); }
    int nhe = 0;
    if(nm) {
      FP_TYPE* ps = dst -> psd + g * (size_t)npsd;
      const int np = nm -> npsd < npsd ? (nm -> npsd > 0 ? nm -> npsd : 0) : npsd;
      if(np > 0) row_copy(ps, nm -> psd, (size_t)np);
      for(int j = np; j < npsd; j ++) ps[j] = (FP_TYPE)-120.0;
      for(int c = 0; c < nch; c ++) {
        const bool have = c < nm -> nchannel;
        dst -> edc[g * (size_t)nch + c] = have ? nm -> edc[c] : (FP_TYPE)1e-5;
        llsm_hmframe* e = have ? nm -> eenv[c] : NULL;
        int n = e ? (e -> nhar < dst -> maxnhar_e ? e -> nhar : dst -> maxnhar_e) : 0;
        if(n < 0) n = 0;
        if(n > nhe) nhe = n;
        FP_TYPE* ea = dst -> eenv_ampl + (g * (size_t)nch + c) * me; FP_TYPE* ep = dst -> eenv_phse + (g * (size_t)nch + c) * me;
        for(int k = 0; k < me; k ++) { ea[k] = k < n ? e -> ampl[k] : 0; ep[k] = k < n ? e -> phse[k] : 0; }
      }
    } else {                                            // no noise model on this frame: the rows' defaults (every element is written)
      row_fill(dst -> psd + g * (size_t)npsd, (FP_TYPE)-120.0, (size_t)npsd);
      for(int c = 0; c < nch; c ++) dst -> edc[g * (size_t)nch + c] = (FP_TYPE)1e-5;
      row_fill(dst -> eenv_ampl + g * (size_t)nch * me, 0, (size_t)nch * me);
      row_fill(dst -> eenv_phse + g * (size_t)nch * me, 0, (size_t)nch * me);
    }
    dst -> nhar_e[g] = nhe;
    dst -> has_psdres[g] = res != NULL;
    FP_TYPE* rr = dst -> psdres + g * (size_t)npsd;
    int nr = res ? llsm_fparray_length(res) : 0; if(nr > npsd) nr = npsd; if(nr < 0) nr = 0;
    if(nr > 0) row_copy(rr, res, (size_t)nr);
    if(npsd > nr) row_fill(rr + nr, 0, (size_t)(npsd - nr));
  }
#if defined(__SSE2__)
  _mm_sfence();                                        // the streaming stores are globally visible before the rows are handed on
#endif
  return 0;
}

// Row frm_off + i -> frame i of `dst` (as llsm_analyze leaves it:
// layer0.c:105-112 HM on voiced frames, :400-406 PSD + PSDRES on every frame,
// :448-458 edc on every frame and eenv on voiced frames).
extern "C" int llsm_flat_to_chunk(const llsm_flat_params* src, int frm_off, llsm_chunk* dst) {
  int nfrm = chunk_nfrm(dst);
  if(nfrm < 0) return -1;
  const int me = src -> maxnhar_e > 0 ? src -> maxnhar_e : 1;
  for(int i = 0; i < nfrm; i ++) {
    llsm_container* fr = dst -> frames[i];
    const size_t g = (size_t)frm_off + i;
    FP_TYPE* f0 = (FP_TYPE*)llsm_container_get(fr, LLSM_FRAME_F0);
    if(f0) *f0 = src -> f0[g];
    const bool voiced = src -> f0[g] != 0;
    if(voiced) {
      int nh = src -> nhar[g];
      llsm_hmframe* hm = llsm_create_hmframe(nh);
      std::memcpy(hm -> ampl, src -> ampl + g * src -> maxnhar, sizeof(FP_TYPE) * (size_t)nh);
      std::memcpy(hm -> phse, src -> phse + g * src -> maxnhar, sizeof(FP_TYPE) * (size_t)nh);
      llsm_container_attach_(fr, LLSM_FRAME_HM, hm,
        (llsm_fdestructor)llsm_delete_hmframe, (llsm_fcopy)llsm_copy_hmframe);
    }
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(fr, LLSM_FRAME_NM);
    if(nm) {
      const int np = nm -> npsd < src -> npsd ? nm -> npsd : src -> npsd;
      if(np > 0) std::memcpy(nm -> psd, src -> psd + g * (size_t)src -> npsd, sizeof(FP_TYPE) * (size_t)np);
      for(int c = 0; c < nm -> nchannel && c < src -> nchannel; c ++) {
        nm -> edc[c] = src -> edc[g * src -> nchannel + c];
        if(! voiced) continue;
        const int n = src -> nhar_e[g];
        const FP_TYPE* ea = src -> eenv_ampl + (g * (size_t)src -> nchannel + c) * me;
        const FP_TYPE* ep = src -> eenv_phse + (g * (size_t)src -> nchannel + c) * me;
        // what llsm_copy_hmframe_inplace does (grow the two arrays when they are too small, then fill), without the
        // temporary frame it would copy from: eight allocator calls per channel and frame became two
        llsm_hmframe* have = nm -> eenv[c];
        if(have -> nhar < n) {
          have -> ampl = (FP_TYPE*)llsm_model_regrow(have -> ampl, 0, sizeof(FP_TYPE) * (size_t)n);
          have -> phse = (FP_TYPE*)llsm_model_regrow(have -> phse, 0, sizeof(FP_TYPE) * (size_t)n);
        }
        for(int k = 0; k < n; k ++) { have -> ampl[k] = ea[k]; have -> phse[k] = ep[k]; }
        have -> nhar = n;
      }
    }
    if(src -> has_psdres[g]) {
      FP_TYPE* res = llsm_create_fparray(src -> npsd);
      std::memcpy(res, src -> psdres + g * src -> npsd, sizeof(FP_TYPE) * (size_t)src -> npsd);
      llsm_container_attach_(fr, LLSM_FRAME_PSDRES, res,
        (llsm_fdestructor)llsm_delete_fparray, (llsm_fcopy)llsm_copy_fparray);
    }
  }
  return 0;
}

// ------------------------------------------------------------- device fan-out
// llsm_analyze_batch / llsm_synthesize_batch cut their utterance list into blocks and hand them to a pool of
// workers -- LLSM_GPU_DEVICES devices (default 1: the default device; "all": every visible device) times
// LLSM_GPU_WORKERS workers per device (default 2) -- each with its own context (stream) and page-locked staging
// buffers.  Blocks are pulled from one queue, so devices balance themselves and, on one device, the upload /
// download of one worker overlaps the kernels of the other.  Utterances are independent units: no
// data-path collective, and a block's results do not depend on the worker that ran it (seeds follow the global
// utterance index).  llsm_gpu_set_fanout overrides the environment.
namespace {
// The batch object of a worker's last block, kept for the next one: creating and deleting a batch (two dozen device
// buffers from the caching pool, the layout / pair / unit / filter-job tables and their uploads) was 1.5 ms of the 7 - 8 ms
// a block of 32 utterances takes.  A batch is a function of (options, rates, utterance and frame counts) only -- the
// resident bench loop uses one for thousands of steps --, so the next block reuses it when those are equal (equal-length
// segments, the usual shape of batch jobs) and replaces it otherwise.  One for analysis, one for synthesis.
struct BatchKey {
  float thop = 0, lip = 0, rel = 0, fs = 0, fnyq = 0; int maxnhar = 0, me = 0, npsd = 0, nch = 0, refine = 0, method = 0;
  std::vector<float> chanfreq; std::vector<int> nx, nfrm;
  unsigned long epoch = 0;                              // llsm_engine_config_epoch(): conventions and plans baked in at creation
  bool operator==(const BatchKey& o) const {
    return epoch == o.epoch && thop == o.thop && lip == o.lip && rel == o.rel && fs == o.fs && fnyq == o.fnyq && maxnhar == o.maxnhar && me == o.me &&
      npsd == o.npsd && nch == o.nch && refine == o.refine && method == o.method && chanfreq == o.chanfreq && nx == o.nx && nfrm == o.nfrm;
  }
};
struct CachedBatch { llsm_gpu_batch* b = nullptr; BatchKey key; };
struct Worker {
  int device = 0; llsm_gpu_context* ctx = nullptr;
  bool busy = false;                                    // held by one call at a time (g_workers_mutex): the host may call from several threads
  FlatHost rows; PBuf<float> xf, ff, xres, ys, yn;
  PBuf<void*> ptab;                                     // page-locked pointer tables the pack / unpack / scatter kernels read
  CachedBatch cache[2];                                 // [0] analysis, [1] synthesis
};
static const bool g_batch_cache = [] { const char* e = std::getenv("LLSM_GPU_BATCH_CACHE"); return !(e && e[0] == '0'); }();
// the worker's batch for this block (slot 0 / 1); NULL on failure.  fnyq: 0 for analysis
static llsm_gpu_batch* worker_batch(Worker* w, int slot, const llsm_aoptions* ao, float fs, float fnyq, int n_utt, const int* nx, const int* nfrm) {
  BatchKey k;
  k.epoch = llsm_engine_config_epoch();
  k.thop = ao -> thop; k.lip = ao -> lip_radius; k.rel = ao -> rel_winsize; k.fs = fs; k.fnyq = fnyq; k.maxnhar = ao -> maxnhar;
  k.me = ao -> maxnhar_e; k.npsd = ao -> npsd; k.nch = ao -> nchannel; k.refine = ao -> f0_refine; k.method = ao -> hm_method;
  if(ao -> chanfreq && ao -> nchannel > 1) k.chanfreq.assign(ao -> chanfreq, ao -> chanfreq + (ao -> nchannel - 1));
  k.nx.assign(nx, nx + n_utt); k.nfrm.assign(nfrm, nfrm + n_utt);
  CachedBatch& c = w -> cache[slot];
  if(c.b && g_batch_cache && c.key == k) return c.b;
  if(c.b) { llsm_gpu_delete_batch(c.b); c.b = nullptr; }
  c.b = llsm_gpu_create_batch(w -> ctx, const_cast<llsm_aoptions*>(ao), fs, n_utt, nx, nfrm);
  c.key = std::move(k);
  return c.b;
}
// after a failed call, or where batches are not kept: the next block starts from a fresh one
static void worker_batch_drop(Worker* w, int slot) {
  CachedBatch& c = w -> cache[slot];
  if(c.b) { llsm_gpu_delete_batch(c.b); c.b = nullptr; }
}
// A kept batch pins its device memory to the worker until the next block of another shape (or
// llsm_gpu_release_cached_batches): fine for the blocks of a batch job (32 utterances of 1 s: ~50 MB of rows and
// waveforms), not for a one-off long utterance or a 1024-utterance block that up to 8 workers per device would each
// hold on to.  Batches whose user-visible arrays exceed $LLSM_GPU_BATCH_CACHE_MB (default 256) are not kept.
int env_int(const char* name, int dflt);
static bool worker_batch_small_enough(llsm_gpu_batch* b) {
  static const size_t cap = (size_t)env_int("LLSM_GPU_BATCH_CACHE_MB", 256) << 20;
  size_t tot = 0;
  for(int a = 0; a < LLSM_GPU_NARRAYS; a ++) tot += llsm_gpu_batch_array_bytes(b, a);
  return tot <= cap;
}
std::mutex g_workers_mutex;
bool g_default_ctx_taken = false;                       // llsm_default_context() already belongs to a worker (g_workers_mutex)
std::vector<Worker*> g_workers;                       // persistent: contexts and staging buffers are reused
int g_fan_devices = -1, g_fan_workers = -1, g_fan_block = -1;   // -1: environment / default

int default_workers() {
  const int hw = (int)std::thread::hardware_concurrency();
  return std::max(2, std::min(8, hw / 4));
}
int env_int(const char* name, int dflt) { const char* e = std::getenv(name); return e && *e ? std::atoi(e) : dflt; }
}  // namespace

// the batch objects idle workers keep for their next block (worker_batch above): released here
extern "C" void llsm_gpu_release_cached_batches(void) {
  std::lock_guard<std::mutex> lock(g_workers_mutex);
  for(Worker* w : g_workers) {
    if(w -> busy) continue;
    for(int k = 0; k < 2; k ++) worker_batch_drop(w, k);
  }
}

extern "C" int llsm_gpu_set_fanout(int n_devices, int workers_per_device, int block_utterances) {
  std::lock_guard<std::mutex> lock(g_workers_mutex);
  g_fan_devices = n_devices; g_fan_workers = workers_per_device; g_fan_block = block_utterances;
  for(Worker* w : g_workers)                            // another block size: the kept batches no longer match what comes next
    if(! w -> busy) for(int k = 0; k < 2; k ++) worker_batch_drop(w, k);
  return 0;
}

// contiguous blocks [u0, u1) of n utterances: the partition the workers pull from (exported for the CPU tests)
extern "C" int llsm_fanout_plan(int n_utt, int block, int* starts, int cap) {
  if(block < 1) block = 1;
  int nb = 0;
  for(int u = 0; u < n_utt; u += block) { if(starts && nb < cap) starts[nb] = u; nb ++; }
  return nb;
}

// Runs fn(worker, u0, u1) over every block.  `fake_workers` > 0: no device contexts (plumbing test hook).
static int fanout_run(int n_utt, const std::function<int(Worker*, int, int)>& fn, int fake_workers = 0, bool serial = false) {
  int ndev_req, nwork, block;
  {
    std::lock_guard<std::mutex> lock(g_workers_mutex);
    ndev_req = g_fan_devices >= 0 ? g_fan_devices : -2; // defaults: the object-model calls are bound by the HOST side (about 25 mallocs per frame of the reference's
    // container tree: 0.33 M frames/s with one worker on 1024 utterances, 1.8 M with 8, 2.1 M with 16 --
    // tools/bench_chunk_api.py), so a quarter of the host threads (2 .. 8) per device and small blocks
    nwork = g_fan_workers > 0 ? g_fan_workers : env_int("LLSM_GPU_WORKERS", default_workers());
    block = g_fan_block > 0 ? g_fan_block : env_int("LLSM_GPU_BLOCK", 32);   // 32: 2.44 M frames/s, 64: 2.26, 128: 2.24 (8 workers, tools/bench_chunk_api.py, r04_g)
  }
  int ndev = 1, first_dev = env_int("LLSM_GPU_DEVICE", 0);
  if(! fake_workers) {
    const int visible = llsm_gpu_device_count();
    if(ndev_req == -2) {
      const char* e = std::getenv("LLSM_GPU_DEVICES");
      ndev_req = ! e || ! *e ? 1 : (std::string(e) == "all" ? 0 : std::atoi(e));
    }
    ndev = ndev_req <= 0 ? visible : std::min(ndev_req, visible);
    if(ndev > 1) first_dev = 0;
    if(visible <= 0) ndev = 1;                          // fails loudly in the worker (no CPU fallback)
  }
  const int nblocks = llsm_fanout_plan(n_utt, block, nullptr, 0);
  int nthreads = fake_workers ? fake_workers : std::min(ndev * nwork, std::max(nblocks, 1));
  if(nthreads < 1 || serial) nthreads = 1;
  std::vector<Worker*> ws;
  {
    std::lock_guard<std::mutex> lock(g_workers_mutex);
    for(int t = 0; t < nthreads; t ++) {
      const int dev = first_dev + (fake_workers ? 0 : t % ndev);
      Worker* w = nullptr;
      for(Worker* c : g_workers) if(c -> device == dev && ! c -> busy) { w = c; break; }
      if(! w) { w = new Worker(); w -> device = dev; g_workers.push_back(w); }
      w -> busy = true;
      ws.push_back(w);
    }
  }
  struct Release {                                      // workers go back to the pool when this call is over
    std::vector<Worker*>& ws;
    ~Release() { std::lock_guard<std::mutex> lock(g_workers_mutex); for(Worker* w : ws) w -> busy = false; }
  } release{ws};
  std::atomic<int> next(0), failed(0);
  std::string first_error; std::mutex err_mutex;
  auto body = [&](Worker* w) {
    // threads this call spawns stage ~6 MB per utterance through page-locked blocks: on the device's own NUMA node
    // (engine.cpp llsm_gpu_bind_thread_to_device; the caller's own thread -- nthreads == 1 -- is never re-bound)
    if(! fake_workers && nthreads > 1) llsm_gpu_bind_thread_to_device(w -> device);
    if(! fake_workers && ! w -> ctx) {
      // the process-wide default context goes to ONE worker, for good: two concurrent batch calls each have their own
      // ws[0], and a context's stream, profiling vectors and lazy tables are not thread-safe
      bool take_default = false;
      if(w -> device == env_int("LLSM_GPU_DEVICE", 0)) {
        std::lock_guard<std::mutex> lock(g_workers_mutex);
        if(! g_default_ctx_taken) { g_default_ctx_taken = true; take_default = true; }
      }
      w -> ctx = take_default ? llsm_default_context() : llsm_gpu_create_context(w -> device, nullptr);
      if(! w -> ctx) { failed = 1; std::lock_guard<std::mutex> l(err_mutex); if(first_error.empty()) first_error = llsm_gpu_last_error(); return; }
    }
    for(;;) {
      const int bi = next.fetch_add(1);
      if(bi >= nblocks || failed.load()) break;
      const int u0 = bi * block, u1 = std::min(n_utt, u0 + block);
      if(fn(w, u0, u1)) { failed = 1; std::lock_guard<std::mutex> l(err_mutex); if(first_error.empty()) first_error = llsm_gpu_last_error(); break; }
    }
  };
  if(nthreads == 1) body(ws[0]);
  else {
    std::vector<std::thread> th;
    for(int t = 0; t < nthreads; t ++) th.emplace_back(body, ws[t]);
    for(auto& t : th) t.join();
  }
  if(failed.load()) { llsm_set_error(first_error.empty() ? "fan-out worker failed" : first_error); return -1; }
  return 0;
}

// plumbing test hook (tests/test_sharding.py): which worker ran which utterance, no device involved
extern "C" int llsm_fanout_selftest(int n_utt, int workers, int* owner) {
  std::vector<Worker*> seen; std::mutex m;
  for(int u = 0; u < n_utt; u ++) owner[u] = -1;
  return fanout_run(n_utt, [&](Worker* w, int u0, int u1) {
    int id;
    { std::lock_guard<std::mutex> l(m); auto it = std::find(seen.begin(), seen.end(), w); if(it == seen.end()) { seen.push_back(w); it = seen.end() - 1; } id = (int)(it - seen.begin()); }
    for(int u = u0; u < u1; u ++) { if(owner[u] != -1) return -1; owner[u] = id; }
    return 0;
  }, workers);
}

// ------------------------------------------------------------------ analyze
static int transfer_params(llsm_gpu_batch* b, FlatHost& h, int to_device) {
  return llsm_gpu_batch_transfer_params(b, to_device, h.block.data());
}
static int download_params(llsm_gpu_batch* b, FlatHost& h) { return transfer_params(b, h, 0); }
static int upload_params(llsm_gpu_batch* b, FlatHost& h) { return transfer_params(b, h, 1); }

// one block of utterances on one worker (its context, its staging buffers)
// slabs: the frames of each chunk carved out of one block (model.cpp "frame slabs") -- the additive batch call's default;
// $LLSM_PACKED_FRAMES: how llsm_analyze_batch / llsm_synthesize_batch move the parameter rows of slab frames.
//   0  eleven row arrays through page-locked staging, re-scattered / flattened frame by frame on the host (rounds 3 - 4)
//   1  one packed record per frame (csrc/packed.h); slabs and outputs in page-locked memory that the device reads and
//      writes ITSELF (k_pack_frames / k_unpack_frames / k_scatter_outputs over the link)
//   2  packed records, ordinary slabs: the block's records cross the link in ONE copy-engine transfer through the worker's
//      page-locked staging and the host moves each utterance's records with one contiguous copy (default)
static int packed_frames_mode() {
  static const int v = [] { const char* e = std::getenv("LLSM_PACKED_FRAMES"); return (e && *e) ? std::atoi(e) : 2; }();
  return v < 0 ? 0 : (v > 2 ? 2 : v);
}

// the drop-in llsm_analyze keeps the reference's "every pointer is its own heap block" unless $LLSM_FRAME_SLABS=1
static int analyze_block(bool slabs, Worker* w, llsm_aoptions* options, FP_TYPE** x, const int* nx, FP_TYPE fs, FP_TYPE** f0,
  const int* nfrm, int n_utt, llsm_chunk** results, FP_TYPE** x_ap) {
  static const bool timing = std::getenv("LLSM_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
    return std::chrono::duration<double, std::milli>(b2 - a).count(); };
  const auto t0 = now();
  llsm_gpu_batch* b = worker_batch(w, 0, options, fs, 0.0f, n_utt, nx, nfrm);
  if(! b) return -1;
  const auto t1 = now();
  llsm_gpu_layout L; llsm_gpu_batch_layout(b, & L);
  std::vector<int> xo(n_utt + 1), fo(n_utt + 1);
  llsm_gpu_batch_offsets(b, xo.data(), fo.data(), NULL);
  PBuf<float>& xf = w -> xf; PBuf<float>& ff = w -> ff;
  xf.resize((size_t)L.total_samples); ff.resize((size_t)L.total_frames);
  for(int u = 0; u < n_utt; u ++) {
    std::memcpy(xf.data() + xo[u], x[u], sizeof(float) * (size_t)nx[u]);
    std::memcpy(ff.data() + fo[u], f0[u], sizeof(float) * (size_t)nfrm[u]);
  }
  int rc;
  {
    const int ids[2] = {LLSM_GPU_X, LLSM_GPU_F0};
    void* host[2] = {xf.data(), ff.data()};
    const size_t bytes[2] = {xf.size() * sizeof(float), ff.size() * sizeof(float)};
    rc = llsm_gpu_batch_transfer_many(b, 1, 2, ids, host, bytes);
  }
  const auto t2 = now();
  if(! rc) rc = llsm_gpu_batch_analyze(b);
  // Heap frames (the drop-in llsm_analyze): their ~25 blocks per frame are allocated NOW, while the device computes --
  // every size follows from F0 and the options (plan.h), unless F0 refinement is about to change F0.  The values are
  // copied in below (llsm_frames_heap_fill, which rebuilds any frame whose analysed counts differ from the plan's).
  const bool prebuilt = ! slabs && ! rc && ! options -> f0_refine;
  std::vector<int> plan_nhar, plan_nhe;
  if(prebuilt) {
    plan_nhar.resize((size_t)L.total_frames); plan_nhe.resize((size_t)L.total_frames);
    for(int u = 0; u < n_utt; u ++)
      for(int i = 0; i < nfrm[u]; i ++) {
        const float f = f0[u][i];
        plan_nhar[fo[u] + i] = f > 0 ? llsm_plan::nhar(f, fs, L.maxnhar) : 0;
        plan_nhe[fo[u] + i] = f > 0 ? std::min(llsm_plan::nhar(f, fs, L.maxnhar_e), L.maxnhar_e) : 0;
      }
    llsm_flat_params pv; std::memset(& pv, 0, sizeof(pv));
    pv.maxnhar = L.maxnhar; pv.maxnhar_e = L.maxnhar_e; pv.npsd = L.npsd; pv.nchannel = L.nchannel;
    pv.f0 = ff.data(); pv.nhar = plan_nhar.data(); pv.nhar_e = plan_nhe.data();
    for(int u = 0; u < n_utt; u ++) {
      llsm_container* conf = llsm_aoptions_toconf(options, (FP_TYPE)(fs / 2.0));      // layer0.c:481-485
      *(int*)llsm_container_get(conf, LLSM_CONF_NFRM) = nfrm[u];
      llsm_chunk* ch = llsm_create_chunk(conf, 0);
      llsm_delete_container(conf);
      llsm_frames_heap_prealloc(& pv, fo[u], ch, nfrm[u]);
      results[u] = ch;                                   // (a failure below: analyze_batch_impl deletes what is in results)
    }
  }
  const auto t3 = now();
  // Packed path (slab frames; round 5): the device gathers each frame's rows into one record (csrc/packed.h); an utterance's
  // records reach that chunk's slab either written by the kernel itself (mode 1: page-locked slabs) or through one block
  // transfer and one contiguous copy (mode 2); the host lays the structs over them (model.cpp llsm_frames_packed_finish).
  // Falls back to the staged rows below when slabs are off, a slab cannot be had, or $LLSM_PACKED_FRAMES=0.
  const int packed_mode = packed_frames_mode();
  static std::once_flag hooks_once;
  std::call_once(hooks_once, [] {
    llsm_slab_set_pin_hooks([](size_t bytes) -> void* {
        void* p = nullptr;
        if(hipHostMalloc(& p, bytes, hipHostMallocPortable) == hipSuccess) return p;
        (void)hipGetLastError(); return nullptr; },
      [](void* p) { (void)hipHostFree(p); });
  });
  bool packed = slabs && packed_mode > 0 && ! rc;
  std::vector<void*> tok((size_t)n_utt, nullptr);
  w -> ptab.resize((size_t)n_utt);
  void** dstp = w -> ptab.data();
  for(int u = 0; u < n_utt; u ++) dstp[u] = nullptr;
  const LlsmPackedLayout PL = llsm_packed_layout(L.maxnhar, L.maxnhar_e, L.npsd, L.nchannel);
  if(packed) {
    for(int u = 0; u < n_utt && packed; u ++) {
      if(nfrm[u] <= 0) continue;
      dstp[u] = llsm_frames_packed_begin(nfrm[u], & PL, & tok[u], packed_mode == 1);
      if(! dstp[u]) packed = false;
    }
    if(packed && packed_mode == 1) rc = llsm_gpu_batch_download_packed(b, n_utt, dstp);      // the kernel writes the slabs itself
    else if(packed) {
      // one copy-engine transfer of the whole block's records into the worker's page-locked staging, then ONE contiguous
      // copy per utterance into its (ordinary) slab
      PBuf<char>& st = w -> rows.block;
      const size_t rec_bytes = (size_t)PL.words * sizeof(float);
      st.resize((size_t)L.total_frames * rec_bytes + 64);
      rc = llsm_gpu_batch_download_packed_block(b, st.data());
      if(! rc) for(int u = 0; u < n_utt; u ++)
        if(nfrm[u] > 0) std::memcpy(dstp[u], st.data() + (size_t)fo[u] * rec_bytes, (size_t)nfrm[u] * rec_bytes);
    }
    if(rc) packed = false;
    if(! packed) { for(int u = 0; u < n_utt; u ++) { llsm_frames_packed_abort(tok[u]); tok[u] = nullptr; } }
  }
  FlatHost& h = w -> rows;
  if(! rc && ! packed) {
    h.layout(b, L);
    rc = download_params(b, h);
  }
  const auto t4 = now();
  PBuf<float>& xres = w -> xres;
  if(! rc && x_ap) {
    xres.resize((size_t)L.total_samples);
    rc = llsm_gpu_batch_download(b, LLSM_GPU_XRES, xres.data(), xres.size() * sizeof(float));
  }
  if(rc || ! g_batch_cache || ! worker_batch_small_enough(b)) worker_batch_drop(w, 0);
  if(rc) { for(int u = 0; u < n_utt; u ++) llsm_frames_packed_abort(tok[u]); return -1; }
  llsm_flat_params v; std::memset(& v, 0, sizeof(v));
  if(! packed) v = h.view();
  for(int u = 0; u < n_utt; u ++) {
    if(prebuilt) {                                       // objects exist: values in
      llsm_frames_heap_fill(& v, fo[u], results[u], nfrm[u]);
      if(x_ap) {
        x_ap[u] = (FP_TYPE*)std::calloc(nx[u] > 0 ? nx[u] : 1, sizeof(FP_TYPE));
        std::memcpy(x_ap[u], xres.data() + xo[u], sizeof(float) * (size_t)nx[u]);
      }
      continue;
    }
    // layer0.c:481-485: conf from the options, NFRM filled in, frames pre-created
    llsm_container* conf = llsm_aoptions_toconf(options, (FP_TYPE)(fs / 2.0));
    *(int*)llsm_container_get(conf, LLSM_CONF_NFRM) = nfrm[u];
    llsm_chunk* ch = llsm_create_chunk(conf, 0);     // frames built at their final sizes below
    llsm_delete_container(conf);
    if(packed) {
      if(nfrm[u] > 0) llsm_frames_packed_finish(tok[u], & PL, ch, nfrm[u], options -> f0_refine ? f0[u] : NULL);
    } else {
      llsm_frames_from_flat_ex(& v, fo[u], ch, nfrm[u], slabs ? 1 : 0);
      if(options -> f0_refine)                       // llsm_analyze rewrites f0[] (dsputils.h:25)
        std::memcpy(f0[u], h.f0 + fo[u], sizeof(float) * (size_t)nfrm[u]);
    }
    results[u] = ch;
    if(x_ap) {
      x_ap[u] = (FP_TYPE*)std::calloc(nx[u] > 0 ? nx[u] : 1, sizeof(FP_TYPE));
      std::memcpy(x_ap[u], xres.data() + xo[u], sizeof(float) * (size_t)nx[u]);
    }
  }
  if(timing)
    std::fprintf(stderr, "[analyze_block %d utt] create batch %.3f, stage + upload %.3f, launch%s %.3f, wait + download %.3f, delete + objects %.3f ms\n",
      n_utt, ms(t0, t1), ms(t1, t2), prebuilt ? " + heap frames allocated beside the device" : "", ms(t2, t3), ms(t3, t4), ms(t4, now()));
  return 0;
}

static int analyze_batch_impl(bool slabs, llsm_aoptions* options, FP_TYPE** x, const int* nx,
  FP_TYPE fs, FP_TYPE** f0, const int* nfrm, int n_utt, llsm_chunk** results, FP_TYPE** x_ap) {
  for(int u = 0; u < n_utt; u ++) { results[u] = NULL; if(x_ap) x_ap[u] = NULL; }
  if(n_utt <= 0) return 0;
  const long long live0 = llsm_slab_live_bytes();
  const int rc = fanout_run(n_utt, [&](Worker* w, int u0, int u1) {
    return analyze_block(slabs, w, options, x + u0, nx + u0, fs, f0 + u0, nfrm + u0, u1 - u0, results + u0, x_ap ? x_ap + u0 : NULL);
  });
  if(slabs && ! rc && n_utt > 1) {                     // what this call added may stay mapped for the next one (model.cpp pool_cap)
    const long long added = llsm_slab_live_bytes() - live0;
    if(added > 0) llsm_slab_pool_hint((size_t)added);
  }
  if(rc) for(int u = 0; u < n_utt; u ++) {             // all or nothing, like a failed llsm_analyze
    if(results[u]) { llsm_delete_chunk(results[u]); results[u] = NULL; }
    if(x_ap && x_ap[u]) { std::free(x_ap[u]); x_ap[u] = NULL; }
  }
  return rc;
}

// $LLSM_FRAME_SLABS: 1 = slab frames from every entry point, 0 = never, unset = the additive batch call only
static int frame_slabs_env() {
  static const int v = [] { const char* e = std::getenv("LLSM_FRAME_SLABS"); return (e && *e) ? (e[0] == '0' ? 0 : 1) : -1; }();
  return v;
}

extern "C" int llsm_analyze_batch(llsm_aoptions* options, FP_TYPE** x, const int* nx,
  FP_TYPE fs, FP_TYPE** f0, const int* nfrm, int n_utt, llsm_chunk** results, FP_TYPE** x_ap) {
  return analyze_batch_impl(frame_slabs_env() != 0, options, x, nx, fs, f0, nfrm, n_utt, results, x_ap);
}

// The drop-in entry point (layer0.c:478-511).  Its frames are ordinary heap objects, as the reference's are: a host may
// free / realloc a member array of an analysed frame itself (SURVEY 8(b) "Ownership").  LLSM_FRAME_SLABS=1 opts into slabs.
extern "C" llsm_chunk* llsm_analyze(llsm_aoptions* options, FP_TYPE* x, int nx,
  FP_TYPE fs, FP_TYPE* f0, int nfrm, FP_TYPE** x_ap) {
  llsm_chunk* out = NULL;
  FP_TYPE* ap = NULL;
  if(analyze_batch_impl(frame_slabs_env() == 1, options, & x, & nx, fs, & f0, & nfrm, 1, & out, x_ap ? & ap : NULL)) return NULL;
  if(x_ap) *x_ap = ap;
  return out;
}

// --------------------------------------------------------------- synthesize
// layer0.c:525-533
static int synthesis_check_integrity(llsm_chunk* src) {
  if(! llsm_conf_checklayer0(src -> conf)) return 0;
  int nfrm = chunk_nfrm(src);
  for(int i = 0; i < nfrm; i ++)
    if(! llsm_frame_checklayer0(src -> frames[i]) && ! llsm_frame_checklayer1(src -> frames[i]))
      return 0;
  return 1;
}

// (the per-frame integrity walk of layer0.c:525-533 runs inside the blocks -- synthesize_block -- on the workers: as a serial
// pass over 204 800 cold frames in the calling thread it was a quarter of llsm_synthesize_batch; the confs are checked here)
static int synthesize_check(llsm_soptions* options, llsm_chunk** src, int n_utt) {
  for(int u = 0; u < n_utt; u ++)
    if(! src[u] || ! llsm_conf_checklayer0(src[u] -> conf)) {
      llsm_set_error("llsm_synthesize: chunk failed the layer-0 integrity check"); return -1;
    }
  // row widths from the chunks themselves; thop / channel plan from the first conf
  llsm_container* conf0 = src[0] -> conf;
  const FP_TYPE thop = *(FP_TYPE*)llsm_container_get(conf0, LLSM_CONF_THOP);
  const int npsd = *(int*)llsm_container_get(conf0, LLSM_CONF_NPSD);
  const int nch = *(int*)llsm_container_get(conf0, LLSM_CONF_NCHANNEL);
  const FP_TYPE fnyq = *(FP_TYPE*)llsm_container_get(conf0, LLSM_CONF_FNYQ);
  FP_TYPE* chanfreq = (FP_TYPE*)llsm_container_get(conf0, LLSM_CONF_CHANFREQ);
  // options->fs need not be 2 * FNYQ: the stored PSD lives on linspace(0, FNYQ, npsd) and is
  // interpolated onto the synthesis rate's bins (layer0.c:578, 606-607); everything else uses options->fs
  const int ncf = chanfreq ? llsm_fparray_length(chanfreq) : 0;
  if(nch > 1 && ncf < nch - 1) {
    llsm_set_error("llsm_synthesize: LLSM_CONF_CHANFREQ shorter than nchannel - 1"); return -1;
  }
  for(int u = 0; u < n_utt; u ++) {
    llsm_container* cf = src[u] -> conf;
    if(*(FP_TYPE*)llsm_container_get(cf, LLSM_CONF_THOP) != thop ||
       *(int*)llsm_container_get(cf, LLSM_CONF_NPSD) != npsd ||
       *(int*)llsm_container_get(cf, LLSM_CONF_NCHANNEL) != nch) {
      llsm_set_error("llsm_synthesize_batch: all chunks must share thop / npsd / nchannel"); return -1;
    }
    FP_TYPE* fq = (FP_TYPE*)llsm_container_get(cf, LLSM_CONF_FNYQ);
    FP_TYPE* cq = (FP_TYPE*)llsm_container_get(cf, LLSM_CONF_CHANFREQ);
    bool same = fq && *fq == fnyq && (cq != NULL) == (chanfreq != NULL) &&
      (! cq || llsm_fparray_length(cq) == ncf);
    for(int k = 0; same && k < ncf; k ++) same = cq[k] == chanfreq[k];
    if(! same) {
      llsm_set_error("llsm_synthesize_batch: all chunks must share LLSM_CONF_FNYQ and LLSM_CONF_CHANFREQ"); return -1;
    }
  }
  return 0;
}

// y = y_sin + y_noise is ONE float addition per sample (layer0.c:657-659; the device's k_synth_ola / k_ola_noise_mix /
// k_pbp_mix form it exactly so), so the sum never needs to cross the link: the host forms it from the two parts it has
// downloaded, bit for bit what the device holds (this file is built with -ffp-contract=off; tests assert the identity).
// 181 MB of 1 161 MB per 1 024 one-second utterances stay on the device.
extern "C" void llsm_gpu_sum_outputs(FP_TYPE* y, const FP_TYPE* y_sin, const FP_TYPE* y_noise, long long n) {
  for(long long i = 0; i < n; i ++) y[i] = y_sin[i] + y_noise[i];
}
// ... fused with the copy out of the staging rows
static void copy_parts_and_sum(const float* __restrict ys, const float* __restrict yn, float* __restrict oy,
  float* __restrict oys, float* __restrict oyn, size_t n) {
  for(size_t i = 0; i < n; i ++) { const float a = ys[i], b = yn[i]; oys[i] = a; oyn[i] = b; oy[i] = a + b; }
}

static int synthesize_block(bool pooled, Worker* w, llsm_soptions* options, llsm_chunk** src, int n_utt, unsigned long long seed,
  llsm_output** results) {
  static const bool timing = std::getenv("LLSM_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
    return std::chrono::duration<double, std::milli>(b2 - a).count(); };
  const auto t0 = now();
  llsm_container* conf0 = src[0] -> conf;
  const FP_TYPE thop = *(FP_TYPE*)llsm_container_get(conf0, LLSM_CONF_THOP);
  const int npsd = *(int*)llsm_container_get(conf0, LLSM_CONF_NPSD);
  const int nch = *(int*)llsm_container_get(conf0, LLSM_CONF_NCHANNEL);
  const FP_TYPE fnyq = *(FP_TYPE*)llsm_container_get(conf0, LLSM_CONF_FNYQ);
  FP_TYPE* chanfreq = (FP_TYPE*)llsm_container_get(conf0, LLSM_CONF_CHANFREQ);
  int maxnhar = 1, me = 0;
  std::vector<int> nfrm(n_utt), nx(n_utt, 0);
  // row widths = the largest harmonic counts among the frames.  This scan touches every frame's container, harmonic
  // model and envelope frames once more than the flatten below does (a fifth of its cache lines); it stays because a
  // host may have grown a frame beyond the conf's MAXNHAR (pitch shifting), and rows narrower than a frame would
  // truncate it silently.
  // Packed path (round 5): chunks whose frames still lie over the records llsm_analyze_batch put in their slabs are not
  // flattened frame by frame -- the records go back as they lie: one contiguous copy per utterance into the staging block
  // and one transfer (mode 2), or read in place by the device from page-locked slabs (mode 1).  The walk over the frames
  // only compares pointers and refreshes the counts in the records' headers (model.cpp llsm_chunk_packed_view).
  // All chunks of the block must qualify with one layout; otherwise the block takes the staged path below.
  const int packed_mode = packed_frames_mode();
  bool packed = pooled && packed_mode > 0 && ! options -> use_l1 && n_utt > 0;
  bool all_locked = true;                                // every chunk's records lie in page-locked memory (mode-1 slabs)
  LlsmPackedLayout PL; std::memset(& PL, 0, sizeof(PL));
  w -> ptab.resize((size_t)n_utt * 4);
  void** srctab = w -> ptab.data();                      // [n_utt] record blocks, then [3 n_utt] output arrays
  for(int u = 0; u < n_utt; u ++) nfrm[u] = chunk_nfrm(src[u]);
  for(int u = 0; u < n_utt && packed; u ++) {
    LlsmPackedLayout Lu; const void* rec = nullptr;
    srctab[u] = nullptr;
    if(nfrm[u] <= 0) { packed = false; break; }
    const int kind = llsm_chunk_packed_view(src[u], nfrm[u], & Lu, & rec);
    if(! kind) { packed = false; break; }
    all_locked = all_locked && kind == 2;
    if(u == 0) PL = Lu;
    else if(Lu.maxnhar != PL.maxnhar || Lu.maxnhar_e != PL.maxnhar_e || Lu.npsd != PL.npsd || Lu.nch != PL.nch) { packed = false; break; }
    srctab[u] = (void*)rec;
  }
  // Records wider than the synthesis batch can be (maxnhar is clamped to 2048 below, layer0.c:119, 130) would be parsed with
  // the wrong stride by k_unpack_frames: such chunks go down the row-by-row path, which clamps per frame.
  if(packed && (PL.npsd != npsd || PL.nch != nch || PL.maxnhar > 2048)) packed = false;
  if(packed) { maxnhar = PL.maxnhar; me = PL.maxnhar_e; }
  else                                                  // layer0.c:525-533 per frame (llsm_chunk_packed_view has checked the packed ones)
    for(int u = 0; u < n_utt; u ++)
      if(! synthesis_check_integrity(src[u])) { llsm_set_error("llsm_synthesize: chunk failed the layer-0 integrity check"); return -1; }
  for(int u = 0; u < n_utt && ! packed; u ++) {
    for(int i = 0; i < nfrm[u]; i ++) {
      llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(src[u] -> frames[i], LLSM_FRAME_HM);
      llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(src[u] -> frames[i], LLSM_FRAME_NM);
      if(hm && hm -> nhar > maxnhar) maxnhar = hm -> nhar;
      if(options -> use_l1) {                        // rows also hold VSPHSE and the HM rebuilt from it
        FP_TYPE* vs = (FP_TYPE*)llsm_container_get(src[u] -> frames[i], LLSM_FRAME_VSPHSE);
        if(vs && llsm_fparray_length(vs) > maxnhar) maxnhar = llsm_fparray_length(vs);
      }
      if(nm) for(int c = 0; c < nm -> nchannel; c ++)
        if(nm -> eenv[c] && nm -> eenv[c] -> nhar > me) me = nm -> eenv[c] -> nhar;
    }
  }
  if(maxnhar > 2048) maxnhar = 2048;               // layer0.c:119, 130
  llsm_aoptions ao; std::memset(& ao, 0, sizeof(ao));
  ao.thop = thop; ao.maxnhar = maxnhar; ao.maxnhar_e = me; ao.npsd = npsd; ao.nchannel = nch;
  ao.chanfreq = chanfreq; ao.lip_radius = 1.5f; ao.f0_refine = 0;
  int nspec_l1 = 0;
  if(options -> use_l1) {
    // layer0.c:168-169: FNYQ and LIPRADIUS from the conf; NSPEC sizes the vocal-tract rows
    FP_TYPE* lr = (FP_TYPE*)llsm_container_get(conf0, LLSM_CONF_LIPRADIUS);
    int* ns = (int*)llsm_container_get(conf0, LLSM_CONF_NSPEC);
    if(! lr || ! ns || *ns < 33 || ((*ns - 1) & (*ns - 2))) {
      llsm_set_error("llsm_synthesize: use_l1 needs LLSM_CONF_LIPRADIUS and LLSM_CONF_NSPEC (llsm_chunk_tolayer1)"); return -1;
    }
    ao.lip_radius = *lr; nspec_l1 = *ns;
  }
  ao.hm_method = LLSM_AOPTION_HMCZT; ao.rel_winsize = 4.0f;
  const auto t1 = now();
  if(options -> use_l1) worker_batch_drop(w, 1);        // a layer-1 batch carries state of its own: always a fresh one
  llsm_gpu_batch* b = worker_batch(w, 1, & ao, options -> fs, fnyq, n_utt, nx.data(), nfrm.data());
  if(! b) return -1;
  const auto t2 = now();
  llsm_gpu_batch_set_fnyq(b, fnyq);
  llsm_gpu_layout L; llsm_gpu_batch_layout(b, & L);
  std::vector<int> fo(n_utt + 1), yo(n_utt + 1);
  llsm_gpu_batch_offsets(b, NULL, fo.data(), yo.data());
  FlatHost& h = w -> rows;
  int rc = 0;
  if(! packed) {
    h.layout(b, L);
    llsm_flat_params v = h.view();
    for(int u = 0; u < n_utt; u ++) llsm_chunk_to_flat(src[u], & v, fo[u]);
  }
  const auto t3 = now();
  if(packed && all_locked && packed_mode == 1) rc = llsm_gpu_batch_upload_packed(b, n_utt, (const void* const*)srctab);   // read where they lie
  else if(packed) {
    // one contiguous copy per utterance into the page-locked staging, ONE copy-engine transfer, unpacked on the device
    PBuf<char>& st = w -> rows.block;
    const size_t rec_bytes = (size_t)PL.words * sizeof(float);
    st.resize((size_t)L.total_frames * rec_bytes + 64);
    for(int u = 0; u < n_utt; u ++) std::memcpy(st.data() + (size_t)fo[u] * rec_bytes, srctab[u], (size_t)nfrm[u] * rec_bytes);
    rc = llsm_gpu_batch_upload_packed_block(b, st.data());
  } else rc = upload_params(b, h);
  if(! rc && options -> use_l1) rc = llsm_l1_prepare_batch(b, src, n_utt, fo.data(), nspec_l1);
  const auto t4 = now();
  if(! rc) rc = llsm_gpu_batch_synthesize(b, options, seed, 0);
  const auto t5 = now();
  if(! rc && options -> use_l1) rc = llsm_l1_writeback_hm(b, src, n_utt, fo.data());
  // outputs: page-locked pooled blocks that the device writes itself (k_scatter_outputs) -- or, when those cannot be had,
  // the staged download and a copy per array below
  bool direct_out = pooled && packed_mode == 1 && ! rc;
  auto t5a = t5, t5b = t5;
  if(direct_out) {
    void** otab = srctab + n_utt;
    for(int u = 0; u < n_utt && direct_out; u ++) {
      const int ny = yo[u + 1] - yo[u];
      llsm_output* o = llsm_output_create_pooled(ny, options -> fs, 1);
      if(! o) { direct_out = false; break; }
      results[u] = o;
      otab[3 * u] = nullptr;                             // y: formed here from the two parts (k_scatter_outputs skips a NULL row)
      otab[3 * u + 1] = o -> y_sin; otab[3 * u + 2] = o -> y_noise;
    }
    t5a = now();
    if(direct_out && timing) { llsm_gpu_synchronize(w -> ctx); t5b = now(); }
    if(direct_out) rc = llsm_gpu_batch_download_outputs(b, n_utt, (float* const*)otab);
    if(direct_out && ! rc)
      for(int u = 0; u < n_utt; u ++) llsm_gpu_sum_outputs(results[u] -> y, results[u] -> y_sin, results[u] -> y_noise, results[u] -> ny);
    if(! direct_out || rc) for(int u = 0; u < n_utt; u ++) if(results[u]) { llsm_delete_output(results[u]); results[u] = NULL; }
  }
  PBuf<float>& ys = w -> ys; PBuf<float>& yn = w -> yn;
  if(! direct_out) { ys.resize((size_t)L.total_out); yn.resize((size_t)L.total_out); }
  if(! rc && ! direct_out) {                             // y_sin and y_noise only: y is their sum (copy_parts_and_sum)
    const int ids[2] = {LLSM_GPU_YSIN, LLSM_GPU_YNOISE};
    void* host[2] = {ys.data(), yn.data()};
    const size_t bytes[2] = {ys.size() * sizeof(float), yn.size() * sizeof(float)};
    rc = llsm_gpu_batch_transfer_many(b, 0, 2, ids, host, bytes);
  }
  const auto t6 = now();
  if(rc || options -> use_l1 || ! g_batch_cache || ! worker_batch_small_enough(b)) worker_batch_drop(w, 1);
  if(rc) return -1;
  if(timing)
    std::fprintf(stderr, "[synthesize_block %d utt] scan frames %.3f, create batch %.3f, flatten %.3f, upload rows %.3f, launch %.3f, wait + download %.3f ms"
      " (output blocks %.3f, rows in + kernels %.3f, samples out %.3f)\n",
      n_utt, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t5, t6), ms(t5, t5a), ms(t5a, t5b), ms(t5b, t6));
  for(int u = 0; u < n_utt && ! direct_out; u ++) {
    int ny = yo[u + 1] - yo[u];
    llsm_output* o;
    if(pooled) {                                        // the batch call: one pooled block per output (model.cpp)
      o = llsm_output_create_pooled(ny, options -> fs, 0);
      if(! o) { llsm_set_error("llsm_synthesize_batch: out of memory"); return -1; }
    } else {
      o = (llsm_output*)std::calloc(1, sizeof(llsm_output));
      o -> ny = ny; o -> fs = options -> fs;
      size_t bytes = sizeof(FP_TYPE) * (size_t)(ny > 0 ? ny : 1);
      // the drop-in llsm_synthesize: four heap blocks, as the reference's llsm_output (a host may keep an array and free
      // it itself); malloc, not calloc: every sample is written below
      o -> y = (FP_TYPE*)std::malloc(bytes); o -> y_sin = (FP_TYPE*)std::malloc(bytes);
      o -> y_noise = (FP_TYPE*)std::malloc(bytes);
      if(ny <= 0) { o -> y[0] = 0; o -> y_sin[0] = 0; o -> y_noise[0] = 0; }
    }
    copy_parts_and_sum(ys.data() + yo[u], yn.data() + yo[u], o -> y, o -> y_sin, o -> y_noise, (size_t)(ny > 0 ? ny : 0));
    results[u] = o;
  }
  return 0;
}

// $LLSM_OUTPUT_POOL: 1 = pooled outputs from every entry point, 0 = never, unset = the additive batch call only
static int output_pool_env() {
  static const int v = [] { const char* e = std::getenv("LLSM_OUTPUT_POOL"); return (e && *e) ? (e[0] == '0' ? 0 : 1) : -1; }();
  return v;
}
static int synthesize_batch_impl(bool pooled, llsm_soptions* options, llsm_chunk** src, int n_utt, llsm_output** results) {
  for(int u = 0; u < n_utt; u ++) results[u] = NULL;
  if(n_utt <= 0) return 0;
  if(synthesize_check(options, src, n_utt)) return -1;
  // one seed per call, as one llsm_synthesize draws from one rand() stream; utterance u of the call uses seed + u
  // whichever block / worker / device renders it
  const unsigned long long seed = llsm_next_seed();
  const long long live0 = llsm_output_live_bytes();
  const int rc = fanout_run(n_utt, [&](Worker* w, int u0, int u1) {
    return synthesize_block(pooled, w, options, src + u0, u1 - u0, seed + (unsigned long long)u0, results + u0);
  }, 0, options -> use_l1 != 0);                       // llsm_fgfm callbacks must arrive in frame / pulse order: one worker, blocks in order
  if(rc) for(int u = 0; u < n_utt; u ++) if(results[u]) { llsm_delete_output(results[u]); results[u] = NULL; }
  if(pooled && ! rc && n_utt > 1) {                    // what this call produced may stay mapped for the next one (model.cpp out_pool_cap)
    const long long added = llsm_output_live_bytes() - live0;
    if(added > 0) llsm_output_pool_hint((size_t)added);
  }
  return rc;
}
extern "C" int llsm_synthesize_batch(llsm_soptions* options, llsm_chunk** src, int n_utt,
  llsm_output** results) {
  return synthesize_batch_impl(output_pool_env() != 0, options, src, n_utt, results);
}

// The drop-in entry point (layer0.c:636-664): its output is four ordinary heap blocks, as the reference's.
extern "C" llsm_output* llsm_synthesize(llsm_soptions* options, llsm_chunk* src) {
  llsm_output* out = NULL;
  if(synthesize_batch_impl(output_pool_env() == 1, options, & src, 1, & out)) return NULL;
  return out;
}
