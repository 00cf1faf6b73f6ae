/*
 * llsm_gpu.h -- batch / device-resident entry points of libllsm2_amd.
 *
 * ADDITIVE to the reference API (SURVEY.md section 8b "Extra the replacement
 * needs"): the reference analyses one utterance per llsm_analyze call
 * (layer0.c:478); a GPU wants thousands of utterances per launch and wants
 * them to stay in HBM between analysis and synthesis.  llsm_analyze /
 * llsm_synthesize (llsm.h) are thin wrappers over a batch of one.
 *
 * Plain C ABI: pointers and sizes only.  All functions returning int return 0
 * on success and a negative value on failure; llsm_gpu_last_error() holds the
 * message.  Nothing here falls back to the CPU.
 */
#ifndef LLSM_AMD_LLSM_GPU_H
#define LLSM_AMD_LLSM_GPU_H

#include <stddef.h>
#include "llsm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct llsm_gpu_context llsm_gpu_context;
typedef struct llsm_gpu_batch   llsm_gpu_batch;

int         llsm_gpu_device_count(void);
const char* llsm_gpu_last_error(void);

/* `stream` is a hipStream_t handed over as void* (e.g. torch's current
 * stream) or NULL to let the context create its own. */
llsm_gpu_context* llsm_gpu_create_context(int device, void* stream);
void              llsm_gpu_delete_context(llsm_gpu_context* ctx);
void*             llsm_gpu_context_stream(llsm_gpu_context* ctx);
int               llsm_gpu_synchronize(llsm_gpu_context* ctx);

/* Conventions of the un-vendored ciglet primitives that the reference's own code cannot confirm (DESIGN.md
 * section 6).  Process-wide; read when contexts, batches and llsmrt buffers are created.
 *   "hann_periodic"       0 (default): overlap-add Hann windows use the symmetric definition (n - 1); 1: periodic (n)
 *   "moving_avg_half"     3 (default): moving_avg(x, n, 3) averages 7 taps; 1: 3 taps
 *   "filtfilt_pad"        15 (default, 3 x the 5 coefficients): samples of odd extension at both ends (1 .. 15)
 *   "interp1u_exclusive"  0 (default): interp1u's samples span [x0, x1]; 1: [x0, x1)
 *   "kalman_init"         0 (default): kalmanf1d starts from x0 = z0, P0 = R0; 1: prior (z0, R0) followed by the filter
 *                         update of the first frame as well (P0 = (1 - K)(R0 + Q0))
 *   "spec2env_lobe_1e6"   cig_spec2env's constant (layer 1: log envelope raised so that the lobe peaks of a flat harmonic
 *                         spectrum sit on it) in units of 1e-6; 133979 (default) = the calibrated 0.13397922601295542
 *   "lf_rd_clamp"         lfmodel_from_rd: 0 (default) Fant's Rd regression with the usual extension formulas outside
 *                         0.21 <= Rd <= 2.7; 1: Rd limited to the range the regression was fitted on (0.3 .. 2.7) first
 * The CPU oracle has the same switches (oracle.h o_set_convention); set returns 0 or -1, get the value or -1. */
int llsm_gpu_set_convention(const char* name, int value);
int llsm_gpu_get_convention(const char* name);

/* Device memory of deleted batches is kept in a per-device cache (bounded by
 * $LLSM_GPU_POOL_MB, default 8192; 0 disables it) so that per-utterance hosts do
 * not pay ~30 hipMalloc/hipFree pairs on every llsm_analyze / llsm_synthesize
 * call.  This returns the cached blocks of the current device to the driver. */
void llsm_gpu_release_cached_memory(void);

/* Per-kernel HIP-event timing of every launch made through the context
 * (used by bench.py for the roofline object).  Off by default. */
int llsm_gpu_set_profiling(llsm_gpu_context* ctx, int enabled);
/* Events around the launches of ONE kernel name only (NULL / "": every kernel again); llsm_gpu_set_profiling(ctx, 2) switches the
 * events on and keeps this filter, (ctx, 1) clears it.  Events are recorded on the stream of the launch, so the analysis keeps its
 * second stream while profiling. */
int llsm_gpu_profile_only(llsm_gpu_context* ctx, const char* kernel_name);
int llsm_gpu_reset_profile(llsm_gpu_context* ctx);
/* Fills up to `cap` entries; returns the number of distinct kernels. */
int llsm_gpu_get_profile(llsm_gpu_context* ctx, int cap, const char** names,
  double* total_ms, int* launches);

/* Diagnostic for the register-resident wavefront FFT every spectral kernel is
 * built on (csrc/wave_fft.h): `count` independent complex transforms of
 * 2^logn points (logn in [8, 12]), interleaved re/im float host buffers,
 * forward (e^{-j}, unnormalised) or inverse (unscaled).  Returns 0 on success. */
int llsm_gpu_fft_selftest(llsm_gpu_context* ctx, int logn, int count, int inverse,
  const float* in, float* out);

/* Shape of a batch: utterance u owns samples [x_off[u], x_off[u]+nx[u]),
 * frames [frm_off[u], frm_off[u]+nfrm[u]) and output samples
 * [y_off[u], y_off[u]+ny[u]) of the flat arrays below. */
typedef struct {
  int n_utt;
  int total_samples;   /* sum nx   */
  int total_frames;    /* sum nfrm */
  int total_out;       /* sum ny, ny = round((nfrm+1)*thop*fs) (layer0.c:643) */
  int maxnhar, maxnhar_e, npsd, nchannel;
  int ntemplate_ext;   /* per (utterance, channel) white-noise template length */
} llsm_gpu_layout;

/* Flat arrays of a batch (row-major; F = total_frames). */
enum {
  LLSM_GPU_X = 0,        /* float [total_samples]        input waveform          */
  LLSM_GPU_F0,           /* float [F]                    (refined) F0            */
  LLSM_GPU_NHAR,         /* int   [F]                                            */
  LLSM_GPU_AMPL,         /* float [F][maxnhar]                                   */
  LLSM_GPU_PHSE,         /* float [F][maxnhar]                                   */
  LLSM_GPU_PSD,          /* float [F][npsd]              dB                      */
  LLSM_GPU_PSDRES,       /* float [F][npsd]              LLSM_FRAME_PSDRES       */
  LLSM_GPU_EDC,          /* float [F][nchannel]                                  */
  LLSM_GPU_NHAR_E,       /* int   [F]                                            */
  LLSM_GPU_EENV_AMPL,    /* float [F][nchannel][maxnhar_e]                       */
  LLSM_GPU_EENV_PHSE,    /* float [F][nchannel][maxnhar_e]                       */
  LLSM_GPU_XRES,         /* float [total_samples]        x - harmonic resynthesis*/
  LLSM_GPU_Y,            /* float [total_out]                                    */
  LLSM_GPU_YSIN,         /* float [total_out]                                    */
  LLSM_GPU_YNOISE,       /* float [total_out]                                    */
  LLSM_GPU_WHITE,        /* float [n_utt][nchannel][ntemplate_ext] Gaussian templates */
  LLSM_GPU_HAS_PSDRES,   /* int   [F]                    frame carries PSDRES    */
  /* layer-1 members, present after llsm_gpu_batch_enable_layer1 (nspec = nfft / 2 + 1) */
  LLSM_GPU_RD,           /* float [F]                    LLSM_FRAME_RD           */
  LLSM_GPU_VTMAGN,       /* float [F][nspec]             LLSM_FRAME_VTMAGN, dB   */
  LLSM_GPU_VSPHSE,       /* float [F][maxnhar]           LLSM_FRAME_VSPHSE       */
  LLSM_GPU_NVSPHSE,      /* int   [F]                    length of VSPHSE; 0: no layer-1 members */
  LLSM_GPU_PBPSYN,       /* int   [F]                    LLSM_FRAME_PBPSYN       */
  LLSM_GPU_HAS_HM,       /* int   [F]                    AMPL / PHSE / NHAR rows valid (LLSM_FRAME_HM present) */
  LLSM_GPU_NARRAYS
};

/* options->thop, fs fix every window / FFT size of the batch.  nx, nfrm:
 * n_utt entries each. */
llsm_gpu_batch* llsm_gpu_create_batch(llsm_gpu_context* ctx,
  const llsm_aoptions* options, FP_TYPE fs, int n_utt, const int* nx,
  const int* nfrm);
void llsm_gpu_delete_batch(llsm_gpu_batch* b);
int  llsm_gpu_batch_layout(llsm_gpu_batch* b, llsm_gpu_layout* dst);
/* offsets: n_utt+1 entries each (may be NULL) */
int  llsm_gpu_batch_offsets(llsm_gpu_batch* b, int* x_off, int* frm_off, int* y_off);

/* Frequency axis of the PSD / PSDRES rows: they span linspace(0, fnyq, npsd) (LLSM_CONF_FNYQ,
 * layer0.c:578).  Defaults to fs / 2 (what llsm_analyze stores); a batch that synthesises parameters
 * analysed at another sampling rate sets the analysis Nyquist here (layer0.c:606-607 interpolates the
 * stored PSD onto the synthesis rate's bins). */
int  llsm_gpu_batch_set_fnyq(llsm_gpu_batch* b, FP_TYPE fnyq);

/* y = y_sin + y_noise on the HOST, bit for bit what the device holds in its y row (one float addition per sample,
 * layer0.c:657-659): a host that has downloaded the two parts forms the sum here instead of moving a third waveform
 * over the link (llsm_synthesize{,_batch} do so internally). */
void llsm_gpu_sum_outputs(FP_TYPE* y, const FP_TYPE* y_sin, const FP_TYPE* y_noise, long long n);

/* Diagnosis only: an intermediate plane of the batch's last analysis, [total_frames][nfft_psd / 2 + 1] float32 --
 * which = 0: the log envelope behind the Kalman process variance (layer0.c:339-343), 1: the log periodogram of the
 * residual (layer0.c:354-360).  dst == NULL: only the size.  Returns the number of floats, -1 on error. */
long long llsm_gpu_batch_debug_plane(llsm_gpu_batch* b, int which, float* dst, long long cap);

/* Page-locked host buffers for the copies below (optional: any host pointer works, but
 * pageable memory is staged by the runtime and reaches a fraction of the PCIe rate). */
void* llsm_gpu_alloc_host(size_t bytes);
void  llsm_gpu_free_host(void* p);
/* Host-side placement.  llsm_gpu_device_numa_node: the NUMA node the device hangs off (sysfs `numa_node` of its PCI
 * function; -1 when the platform does not say).  llsm_gpu_bind_thread_to_device: binds the CALLING thread to that node's
 * CPUs (intersected with the CPUs it may already use) so that the staging copies into / out of page-locked blocks and the
 * blocks it allocates stay on the socket the PCIe link is on; returns the number of CPUs in the new mask, 0 if nothing was
 * changed (node unknown, no overlap, or $LLSM_GPU_NUMA_BIND=0).  The workers of the in-process fan-out call it on
 * themselves; a host that fills its own buffers may call it from the filling thread. */
int   llsm_gpu_device_numa_node(int device);
int   llsm_gpu_bind_thread_to_device(int device);

/* host <-> device copies of one flat array (whole array, host pointer) */
int   llsm_gpu_batch_upload(llsm_gpu_batch* b, int array_id, const void* src, size_t bytes);
int   llsm_gpu_batch_download(llsm_gpu_batch* b, int array_id, void* dst, size_t bytes);
/* The analysed rows FRAME-major: one record of llsm_gpu_batch_packed_words() 4-byte words per frame (layout: csrc/packed.h --
 * f0, nhar, nhar_e, has_psdres, then ampl | phse | psd | fparray header | psdres | edc | eenv_ampl | eenv_phse, every piece on
 * a 16-byte boundary), written by a kernel straight into PAGE-LOCKED host blocks, one per utterance: dst[u] receives the
 * nfrm[u] records of utterance u (dst itself page-locked: the kernel reads the table).  What llsm_analyze_batch uses to land
 * an utterance's frames in its chunk's slab without a staging copy; upload_packed is the other direction (asynchronous: the
 * synthesis launches follow on the stream); download_outputs writes y / y_sin / y_noise of utterance u to tab[3 u + 0 / 1 / 2]
 * (page-locked arrays of ny[u] samples; NULL entries skipped). */
int   llsm_gpu_batch_packed_words(llsm_gpu_batch* b);
int   llsm_gpu_batch_download_packed(llsm_gpu_batch* b, int n_utt, void* const* dst);
int   llsm_gpu_batch_upload_packed(llsm_gpu_batch* b, int n_utt, const void* const* src);
int   llsm_gpu_batch_download_outputs(llsm_gpu_batch* b, int n_utt, float* const* tab);
/* the records of the whole batch as ONE block through the copy engine (host: page-locked, total_frames x words x 4 bytes,
 * utterance u at frm_off[u] x words); upload is asynchronous */
int   llsm_gpu_batch_download_packed_block(llsm_gpu_batch* b, void* host);
int   llsm_gpu_batch_upload_packed_block(llsm_gpu_batch* b, const void* host);
/* several arrays in one call: all copies enqueued, the stream waited for once (to_device != 0: upload; same checks as the
 * single-array calls; the host buffers must stay valid until the call returns) */
int   llsm_gpu_batch_transfer_many(llsm_gpu_batch* b, int to_device, int n, const int* array_ids, void* const* host, const size_t* bytes);
/* The eleven parameter rows of a batch (LLSM_GPU_F0, NHAR, AMPL, PHSE, PSD, PSDRES, HAS_PSDRES, EDC, NHAR_E, EENV_AMPL,
 * EENV_PHSE) are pieces of ONE device block: params_layout reports its size, the byte offset of each piece and the array
 * id behind it (any output may be NULL); transfer_params moves the whole block in one copy to / from a host buffer laid out
 * with the same offsets (to_device != 0: upload).  One copy instead of eleven: a small device-to-host copy costs ~0.1 ms
 * whatever its size. */
int   llsm_gpu_batch_params_layout(llsm_gpu_batch* b, size_t* total_bytes, size_t* offsets11, int* array_ids11);
int   llsm_gpu_batch_transfer_params(llsm_gpu_batch* b, int to_device, void* host_block);
/* device address of a flat array (stays valid until the batch is deleted) */
void* llsm_gpu_batch_device_ptr(llsm_gpu_batch* b, int array_id);
size_t llsm_gpu_batch_array_bytes(llsm_gpu_batch* b, int array_id);

/* Enqueue layer-0 analysis of every utterance of the batch (inputs: X, F0;
 * outputs: F0 (refined), NHAR .. EENV_PHSE, XRES).  Asynchronous. */
int llsm_gpu_batch_analyze(llsm_gpu_batch* b);
/* Enqueue layer-0 synthesis from the parameter arrays currently resident in
 * the batch (outputs: Y, YSIN, YNOISE).  seed drives the counter-based
 * Gaussian generator; use_injected_white != 0 makes it read LLSM_GPU_WHITE
 * instead.  Asynchronous. */
int llsm_gpu_batch_synthesize(llsm_gpu_batch* b, const llsm_soptions* options,
  unsigned long long seed, int use_injected_white);

/* ---- layer 1 (source-filter model) on a device-resident batch; replaces layer1.c:129-195 ----
 * enable_layer1   allocates the layer-1 arrays (nfft: size of the vocal-tract response, power of two)
 * tolayer1        llsm_chunk_tolayer1 over every utterance: RD on every frame, VTMAGN / VSPHSE on voiced
 *                 frames (inputs: F0, NHAR, AMPL, PHSE; lip radius from the batch options)
 * tolayer0        llsm_frame_tolayer0 over every frame with layer-1 members (only_missing != 0: only frames
 *                 whose HAS_HM flag is 0); writes NHAR / AMPL / PHSE and sets HAS_HM
 * set_maxnhar_conf  LLSM_CONF_MAXNHAR as llsm_frame_tolayer0 reads it (layer1.c:166-167); < 0: absent
 * set_pbpeffect   LLSM_FRAME_PBPEFF of one frame: `modifier` is called on the host, in frame / pulse order,
 *                 while llsm_gpu_batch_synthesize (use_l1 = 1) schedules the pulses (layer0.c:208-217)
 * llsm_gpu_batch_synthesize with options->use_l1 = 1 then renders layer0.c:148-287 on the device. */
int llsm_gpu_batch_enable_layer1(llsm_gpu_batch* b, int nfft);
int llsm_gpu_batch_tolayer1(llsm_gpu_batch* b, int nfft);
int llsm_gpu_batch_tolayer0(llsm_gpu_batch* b, int only_missing);
int llsm_gpu_batch_set_maxnhar_conf(llsm_gpu_batch* b, int maxnhar_conf);
int llsm_gpu_batch_set_pbpeffect(llsm_gpu_batch* b, int frame, llsm_fgfm modifier, void* info,
  llsm_container* src_frame);

/* chunk <-> flat layer-1 rows (same row indexing as llsm_flat_params) */
typedef struct {
  int nspec, maxnhar;
  FP_TYPE* rd; int* has_rd; FP_TYPE* vtmagn; FP_TYPE* vsphse; int* nvsphse; int* pbpsyn; int* has_hm;
} llsm_flat_l1;
int llsm_chunk_to_flat_l1(llsm_chunk* src, llsm_flat_l1* dst, int frm_off);
int llsm_flat_l1_to_chunk(const llsm_flat_l1* src, int frm_off, llsm_chunk* dst);

/* Convenience wrappers in the reference's own object model: n_utt independent
 * llsm_analyze / llsm_synthesize calls fused into one batch. Arrays of
 * n_utt pointers; results[] receives caller-owned objects. */
int llsm_analyze_batch(llsm_aoptions* options, FP_TYPE** x, const int* nx,
  FP_TYPE fs, FP_TYPE** f0, const int* nfrm, int n_utt, llsm_chunk** results,
  FP_TYPE** x_ap);
int llsm_synthesize_batch(llsm_soptions* options, llsm_chunk** src, int n_utt,
  llsm_output** results);

/* Device fan-out of the two wrappers above: the utterance list is cut into blocks of `block_utterances` that a pool of
 * n_devices x workers_per_device workers pulls from one queue (one context / stream and page-locked staging per worker;
 * on one device the transfers of one worker overlap the kernels of the other; across devices the queue balances the
 * load).  No data-path collective (utterances are independent); results do not depend on the placement.
 * Defaults: $LLSM_GPU_DEVICES (1; "all" = every visible device), $LLSM_GPU_WORKERS (a quarter of the host threads, 2 .. 8:
 * these calls are bound by building the reference's container trees on the host), $LLSM_GPU_BLOCK (32);
 * arguments <= 0 keep the default.  use_l1 synthesis runs its blocks in order on one worker (host-ordered callbacks).
 * llsm_fanout_plan returns the number of blocks (and their first utterances); llsm_fanout_selftest runs the queue
 * with `workers` device-less workers and reports which one took each utterance (CPU test of the plumbing). */
int llsm_gpu_set_fanout(int n_devices, int workers_per_device, int block_utterances);
int llsm_fanout_plan(int n_utt, int block, int* starts, int cap);
int llsm_fanout_selftest(int n_utt, int workers, int* owner);

/* chunk <-> flat rows: frame i of `src` into row frm_off+i of host-side flat
 * arrays laid out like the batch (used by the wrappers above and by tests). */
typedef struct {
  int maxnhar, maxnhar_e, npsd, nchannel;
  FP_TYPE* f0; int* nhar; FP_TYPE* ampl; FP_TYPE* phse;
  FP_TYPE* psd; FP_TYPE* psdres; int* has_psdres; FP_TYPE* edc; int* nhar_e;
  FP_TYPE* eenv_ampl; FP_TYPE* eenv_phse;
} llsm_flat_params;
int llsm_chunk_to_flat(llsm_chunk* src, llsm_flat_params* dst, int frm_off);
/* Frame slabs.  The frames of a chunk that llsm_analyze_batch returns are carved out of ONE block per chunk
 * (the reference: about 25 heap blocks per frame), with the reference's own destructors and copy constructors attached:
 * llsm_container_attach / remove / copy, llsm_copy_*_inplace, llsm_delete_container on single frames and
 * llsm_delete_chunk behave as container.c / frame.c specify (copies are ordinary heap objects; the block goes when its
 * last object is deleted).  The one thing a host must not do is pass a member ARRAY of such a frame (hm->ampl,
 * nm->psd ...) to free / realloc itself.  The drop-in llsm_analyze returns ordinary heap frames -- the reference's
 * ownership rule -- unless $LLSM_FRAME_SLABS=1; $LLSM_FRAME_SLABS=0 switches slabs off everywhere.  Released blocks are
 * kept for the next chunk up to $LLSM_SLAB_POOL_MB if that is set; otherwise up to 64 MB, raised (never above
 * $LLSM_SLAB_POOL_MAX_MB, default 1024) to the slab volume the largest llsm_analyze_batch call so far produced -- what the
 * host itself had live a moment ago -- until llsm_slab_trim.
 *   llsm_slab_stats   live slabs, their bytes, bytes kept in the pool (any pointer may be NULL)
 *   llsm_slab_trim    hands the pooled blocks back to the allocator (those of the pooled outputs below as well)
 * Outputs.  llsm_synthesize returns the reference's four heap blocks (struct, y, y_sin, y_noise).  llsm_synthesize_batch
 * builds each llsm_output as ONE block -- struct and the three arrays -- taken from a pool of released blocks (three
 * 177 KB arrays per utterance were three fresh mappings and as many unmappings: more time than the synthesis itself);
 * llsm_delete_output recognises such an output and returns its block to the pool, which is kept up to
 * $LLSM_OUTPUT_POOL_MB if set, else up to the volume the largest batch call so far produced (32 MB ... 1 GB).  The arrays
 * of such an output must not be passed to free() one by one.  $LLSM_OUTPUT_POOL=0 / 1: never / from llsm_synthesize too. */
void llsm_slab_stats(long long* live_slabs, long long* live_bytes, long long* pooled_bytes);
void llsm_slab_trim(void);
/* n chunks at once: llsm_delete_chunk (llsm.h) on each; entries are set to NULL.  A chunk whose frames still lie untouched in
 * the slab llsm_analyze_batch built them in drops all its references in ONE decrement (nothing deleted, attached or regrown
 * since, every container and member pointer still inside the slab: values written through the structs do not count);
 * otherwise one walk of range checks per frame (objects a host attached itself go through their own destructors). */
void llsm_delete_chunks(llsm_chunk** chunks, int n);
/* Batch objects between calls (round 4).  llsm_analyze / llsm_synthesize and their *_batch forms run on persistent workers
 * (one context, stream and page-locked staging each); a worker also keeps the device batch of its last block -- buffers,
 * layout, filter-job and unit tables -- and reuses it when the next block has the same options, rates, utterance and frame
 * counts (equal-length segments: the usual shape of batch jobs), instead of building and tearing one down per block.
 * Results do not depend on it.  A differently shaped block replaces the kept batch; llsm_gpu_release_cached_batches()
 * releases the batches of idle workers (their device memory); $LLSM_GPU_BATCH_CACHE=0 switches the reuse off. */
void llsm_gpu_release_cached_batches(void);
int llsm_flat_to_chunk(const llsm_flat_params* src, int frm_off, llsm_chunk* dst);

/* Flat wire format of a layer-0 chunk (csrc/wire.cpp): ONE contiguous, position-independent
 * blob with the conf scalars and the flat rows above -- for caching analysed utterances on
 * disk, for sending them between ranks, or for uploading into a batch without building the
 * container tree.  The reference has no serialisation (container.c, frame.c keep ~25 heap
 * blocks per frame).  Little-endian, versioned ("LLSM2L0", version 2; version 1 is still read); row widths are the
 * largest harmonic counts present in the chunk.  Host-only.
 *   llsm_chunk_blob_size  bytes llsm_chunk_to_blob will write (0 on a chunk without conf)
 *   llsm_chunk_to_blob    returns the bytes written, or -1
 *   llsm_blob_view        validates an untrusted blob and points `view` INTO it (no copy);
 *                         0 on success; thop / fnyq / nfrm are optional outputs.  The blob's
 *                         address must be a multiple of 8 (rejected otherwise)
 *   llsm_blob_to_chunk    rebuilds a caller-owned chunk (llsm_delete_chunk), NULL if malformed
 *   llsm_blob_view_l1     the layer-1 rows of a blob (version 2 blobs of chunks that went through
 *                         llsm_chunk_tolayer1: RD, VTMAGN, VSPHSE, PBPSYN, which frames still hold an HM); view->nspec == 0
 *                         when there are none.  LLSM_FRAME_PBPEFF (a host callback) is never carried.
 *   llsm_gpu_batch_upload_blob  the rows of a blob straight into utterance `utt` of a batch (frame counts and row
 *                         widths must fit; enables layer 1 on the batch when the blob carries it) -- no container tree */
size_t      llsm_chunk_blob_size(llsm_chunk* src);
long long   llsm_chunk_to_blob(llsm_chunk* src, void* dst, size_t capacity);
int         llsm_blob_view(const void* blob, size_t bytes, llsm_flat_params* view, int* nfrm,
  FP_TYPE* thop, FP_TYPE* fnyq);
llsm_chunk* llsm_blob_to_chunk(const void* blob, size_t bytes);
int         llsm_blob_view_l1(const void* blob, size_t bytes, llsm_flat_l1* view);
int         llsm_gpu_batch_upload_blob(llsm_gpu_batch* b, int utt, const void* blob, size_t bytes);
/* utterances [utt0, utt0 + n) from blobs[0 .. n): rows gathered in page-locked staging, one copy per array and group */
int         llsm_gpu_batch_upload_blobs(llsm_gpu_batch* b, int utt0, int n, const void* const* blobs, const size_t* bytes);

/* ---- llsmrt stream groups (BASELINE.json config 4: many concurrent streams per GPU) ----
 * The reference's llsmrt buffer is one stream (llsmrt.h:33-54).  A group advances n_streams
 * independent streams (own noise templates, own rings, own output) by one hop per
 * llsm_rtsynth_group_feed with ONE kernel sequence; llsm_create_rtsynth_buffer is a group of
 * one.  Seeds: stream s draws from default seed + s.  fetch is a bulk, non-blocking pull of
 * up to max_samples samples of one stream (dst_p / dst_ap may be NULL) and returns the count. */
typedef void llsm_rtsynth_group;
llsm_rtsynth_group* llsm_create_rtsynth_group(llsm_soptions* options, llsm_container* conf,
  int capacity_samples, int n_streams);
void llsm_delete_rtsynth_group(llsm_rtsynth_group* g);
int  llsm_rtsynth_group_getlatency(llsm_rtsynth_group* g);
int  llsm_rtsynth_group_numoutput(llsm_rtsynth_group* g, int stream);
void llsm_rtsynth_group_feed(llsm_rtsynth_group* g, llsm_container** frames);
/* n_hops hops in one call (frames[k * n_streams + s]: stream s, hop k): a producer that is ahead of its consumer -- the
 * reference's feed only blocks on a FULL ring, llsmrt.c:489-493 -- packs and enqueues hop k + 1 while hop k is on the
 * device; on return the samples of every hop are in the rings, bit-identical to n_hops single feeds.
 * Like n_hops calls of feed it BLOCKS while a ring is full: a caller that is also the only consumer must keep
 * n_hops x (hop + 1) samples within the free room of the rings (capacity_samples - numoutput), or pull from another thread --
 * exactly the reference's rule for a producer that runs ahead (llsmrt.c:489-493). */
void llsm_rtsynth_group_feed_many(llsm_rtsynth_group* g, llsm_container** frames, int n_hops);
/* One hop of a buffer / group as ONE device submission: the copy-in, the launches and the copy-out of a feed are
 * stream-captured and replayed through an executable hipGraph that is updated in place every hop.  on = 1 / 0 switches
 * it for the process (default: $LLSM_RT_GRAPH, else off -- see DESIGN.md section 8 for the measurement), on < 0 only
 * queries; returns the previous setting.  llsm_gpu_rt_graph_hops: hops submitted that way so far. */
int       llsm_gpu_rt_graph(int on);
/* Shared-F0 tiles: frames of one utterance that carry a bit-identical F0 (fixed-F0 material, flat stretches of an F0
 * track) are analysed 16 at a time as the rows of one matrix product whose twiddles and window are formed once per
 * tile (k_harm_speech_tile).  Which frames qualify depends only on the utterance's own F0
 * row, never on the rest of the batch.  on = 1 / 0 switches the tile kernels for the process (default:
 * $LLSM_GPU_F0_TILES, else on), on < 0 only queries; returns the previous setting.  With 0 every frame takes the
 * per-frame kernels (results agree to float32 rounding; tests/test_gpu_tiles.py). */
int       llsm_gpu_shared_f0_tiles(int on);
/* Shared phasor tables of the harmonic resynthesis (k_synth_ola4): the frames one workgroup walks that carry the
 * F0 bits of the group's first voiced frame read the row / column phasors of every k-step from one LDS table instead
 * of rotating and re-seeding them per frame.  The table holds exactly the values the per-frame recurrences produce,
 * so x_res, y_sin and y are bit-identical on and off (tests/test_gpu_synth_tables.py).  on = 1 / 0 switches it for
 * the process (default: $LLSM_GPU_SYNTH_TABLES, else on), on < 0 only queries; returns the previous setting. */
int       llsm_gpu_synth_tables(int on);
/* Pulse-by-pulse synthesis (layer 1): a pulse group is real, so its inverse transform runs as ONE complex transform of
 * half its size (k_pbp_pulse, round 4: half the LDS, twice the pulse groups in flight) instead of the full-size
 * transform of the Hermitian-completed spectrum.  on = 1 / 0 switches it for the process (default on), on < 0 only
 * queries; returns the previous setting.  The samples agree to float32 rounding (tests/test_gpu_l1.py). */
int       llsm_gpu_pbp_real_ifft(int on);
/* The Kalman smoother of an analysis on a second stream beside the band filter and the envelope analysis (it needs
 * nothing they produce and is bound by HBM where they are bound by arithmetic); joined before the call returns its
 * work to the context's stream, so callers see one stream.  on = 1 / 0 for the process (default: $LLSM_GPU_OVERLAP,
 * else on), on < 0 only queries; returns the previous setting.  Results do not depend on it. */
int       llsm_gpu_analysis_overlap(int on);
long long llsm_gpu_rt_graph_hops(void);
/* Kernel launches per hop of a buffer / group.  0: five single-purpose launches.  1: two -- envelope frames beside the
 * harmonic frame, ring adds and excitation in the first; noise filter (four wavefronts per pair of streams), noise ring
 * and the hop's output samples in the second; a pulse-by-pulse buffer's dual-buffer bookkeeping rides in the first.
 * 2: one -- a workgroup of 512 threads takes a pair of streams through both (transforms of at most 2048 points).
 * 3 (default): one, with the hop's temporaries in LDS and every ring cell read and written once (k_rt_hop2; windows of at
 * most 1024 samples, transforms of at most 2048 points, else as 2).  A pulse-by-pulse hop on which pulse groups are
 * due adds the pulse kernel in front.  Sets the mode for the process (default: $LLSM_RT_FUSED, else 3), on < 0 only
 * queries; returns the previous setting.  1, 2 and 3 give bit-identical samples; 0 differs from them by the float32
 * rounding of the noise part. */
/* Pipelined feeds (round 4).  A feed is synchronous by default, as llsmrt.c's is: when it returns, the hop's samples are in
 * the rings.  With on = 1 a feed returns as soon as the hop is enqueued; its samples are appended when the NEXT feed
 * starts, when a fetch or a numoutput call finds the stream's ring empty (it then waits for the hop in flight), or on
 * clear -- so a consumer that pulls blocks while numoutput allows sees them one hop later (one hop of extra latency), a
 * consumer that drains to zero after every feed gets the reference's behaviour without the overlap, and the host side
 * of a hop (pulls, packing the next frames) runs beside the device instead of after it.  The samples are the same.  Hops that write rebuilt harmonic
 * models back onto the caller's frames stay synchronous.  Default: $LLSM_RT_PIPELINE, else off; on < 0 only queries;
 * returns the previous setting. */
int       llsm_gpu_rt_pipeline(int on);
int       llsm_gpu_rt_fused(int on);
/* The kernels of a (one- or two-launch) hop read the hop's parameter rows from the pinned host block and write
 * the hop's samples into the pinned host block themselves, instead of a copy launch before and after them (the rows
 * hold nfft harmonic slots of which a frame uses a few hundred; each copy was a dependent launch about as long as one of
 * the kernels).  Same kernels, same arithmetic: the samples are bit-identical.  on = 1 / 0 switches it for the process
 * (default: $LLSM_RT_DIRECT, else on), on < 0 only queries; returns the previous setting.  Hops of a pulse-by-pulse buffer
 * that rebuild harmonic rows on the device and hops replayed as a graph keep the copy in. */
int       llsm_gpu_rt_direct(int on);
int  llsm_rtsynth_group_fetch(llsm_rtsynth_group* g, int stream, FP_TYPE* dst_p, FP_TYPE* dst_ap,
  int max_samples);
/* every stream at once (a mixer's pull): row s of dst_p / dst_ap -- [n_streams][max_samples], either may be NULL --
 * receives up to max_samples samples of stream s; counts (n_streams ints, may be NULL) gets the samples per stream;
 * returns the smallest count. */
int  llsm_rtsynth_group_fetch_all(llsm_rtsynth_group* g, FP_TYPE* dst_p, FP_TYPE* dst_ap, int max_samples,
  int* counts);

/* Seed used by llsm_synthesize / llsm_create_rtsynth_buffer (the reference
 * draws from libc rand(), dsputils.c:357; here every call advances a
 * process-wide counter starting from this seed). */
void llsm_gpu_set_default_seed(unsigned long long seed);

/* Index plan exported for tests: same float32
 * evaluation the kernels use. which: 0 center(i) 1 nwin_sin 2 nwin_env
 * 3 nwin_filt 4 nwin_psd 5 ny(i=nfrm) 6 hwin(f0) 7 nhar(f0,i=maxnhar)
 * 8 env_ola(i,j) 9 dcwin(f0) 10 spgmwin(f0,i=nwin_psd) */
int llsm_gpu_plan_index(int which, int i, int j, FP_TYPE f0, FP_TYPE thop,
  FP_TYPE fs, FP_TYPE rel_winsize);

#ifdef __cplusplus
}
#endif
#endif
