/*
 * llsm.h -- public C interface of libllsm2_amd, the MI355X-native drop-in for
 * the layer-0 analysis / synthesis path of Sleepwalking/libllsm2 (2.1.0).
 *
 * This header is written from scratch.  It declares, with identical names,
 * argument meaning, struct layouts and ownership rules, the symbols that a C
 * host linked against the reference's libllsm2.a uses for this path; each
 * block cites the reference interface it replaces (file:line in the reference
 * tree).  The numeric work behind llsm_analyze / llsm_synthesize runs in HIP
 * kernels on gfx950; the data model (containers, frames, chunks) is host code.
 *
 * FP_TYPE is float (the reference's default, makefile:20); a double build of
 * this library does not exist.
 */
#ifndef LLSM_AMD_LLSM_H
#define LLSM_AMD_LLSM_H

#ifndef FP_TYPE
#define FP_TYPE float
#endif
/* A host built with the reference's -DFP_TYPE=double (makefile:20, 31) must not link against this float library by
 * accident: every array would be read with the wrong element size.  Refused at compile time (C99 and C++ alike). */
typedef char llsm_amd_FP_TYPE_must_be_float[(sizeof(FP_TYPE) == sizeof(float)) ? 1 : -1];

#ifdef __cplusplus
extern "C" {
#endif

/* replaces llsm.h:25-28 */
#define LLSM_VERSION_STRING   "2.1.0"
#define LLSM_VERSION_MAJOR    2
#define LLSM_VERSION_MINOR    1
#define LLSM_VERSION_REVISION 0

/* member destructor / copy-constructor hooks (replaces llsm.h:31, 34) */
typedef void  (*llsm_fdestructor)(void*);
typedef void* (*llsm_fcopy)(void*);

/* ---- boxed scalars and length-prefixed arrays (replaces llsm.h:38-47;
 * container.c:24-71: the length of an fparray lives in the int stored
 * immediately before the returned pointer) ---- */
FP_TYPE* llsm_create_fp(FP_TYPE x);
int*     llsm_create_int(int x);
FP_TYPE* llsm_create_fparray(int size);
FP_TYPE* llsm_copy_fp(FP_TYPE* src);
int*     llsm_copy_int(int* src);
FP_TYPE* llsm_copy_fparray(FP_TYPE* src);
void     llsm_delete_fp(FP_TYPE* dst);
void     llsm_delete_int(int* dst);
void     llsm_delete_fparray(FP_TYPE* dst);
int      llsm_fparray_length(FP_TYPE* src);

/* ---- generic index -> object container (replaces llsm.h:54-92).
 * Users read the fields directly (test/test-structs.c:21-44), so the layout
 * is part of the ABI. ---- */
typedef struct {
  void** members;
  llsm_fdestructor* destructors;
  llsm_fcopy* copyctors;
  int nmember;
} llsm_container;

llsm_container* llsm_create_container(int nmember);
llsm_container* llsm_copy_container(llsm_container* src);
void  llsm_copy_container_inplace(llsm_container* dst, llsm_container* src);
void  llsm_delete_container(llsm_container* dst);
void* llsm_container_get(llsm_container* src, int index);
/* Attaching takes ownership iff dtor != NULL; copyctor == NULL means the
 * member is shared (shallow) by llsm_copy_container; an index beyond the
 * current size grows the container. */
#define llsm_container_attach(dst, index, ptr, dtor, copyctor) \
  llsm_container_attach_(dst, index, ptr, (llsm_fdestructor)dtor, (llsm_fcopy)copyctor)
void  llsm_container_attach_(llsm_container* dst, int index, void* ptr,
  llsm_fdestructor dtor, llsm_fcopy copyctor);
void  llsm_container_remove(llsm_container* dst, int index);

/* ---- member indices of a frame container (replaces llsm.h:98-108) ---- */
#define LLSM_FRAME_F0        0   /* FP_TYPE            */
#define LLSM_FRAME_HM        1   /* llsm_hmframe       */
#define LLSM_FRAME_NM        2   /* llsm_nmframe       */
#define LLSM_FRAME_PSDRES    3   /* fparray, residual PSD */
#define LLSM_FRAME_PBPEFF    8   /* llsm_pbpeffect     */
#define LLSM_FRAME_PBPSYN    9   /* int                */
#define LLSM_FRAME_RD       10   /* FP_TYPE            */
#define LLSM_FRAME_VTMAGN   11   /* fparray, dB        */
#define LLSM_FRAME_VSPHSE   12   /* fparray            */

/* ---- member indices of a configuration container (replaces llsm.h:115-128) */
#define LLSM_CONF_NFRM       0   /* int      */
#define LLSM_CONF_THOP       1   /* FP_TYPE, seconds */
#define LLSM_CONF_MAXNHAR    2   /* int      */
#define LLSM_CONF_MAXNHAR_E  3   /* int      */
#define LLSM_CONF_NPSD       4   /* int      */
#define LLSM_CONF_NOSWARP    5   /* deprecated */
#define LLSM_CONF_FNYQ       6   /* FP_TYPE, Hz */
#define LLSM_CONF_NCHANNEL   7   /* int      */
#define LLSM_CONF_CHANFREQ   8   /* fparray, Hz */
#define LLSM_CONF_NSPEC     10   /* int      */
#define LLSM_CONF_LIPRADIUS 11   /* FP_TYPE, cm */

/* ---- harmonic model frame (replaces llsm.h:134-151) ---- */
typedef struct {
  FP_TYPE* ampl;
  FP_TYPE* phse;
  int      nhar;
} llsm_hmframe;

llsm_hmframe* llsm_create_hmframe(int nhar);
llsm_hmframe* llsm_copy_hmframe(llsm_hmframe* src);
void llsm_copy_hmframe_inplace(llsm_hmframe* dst, llsm_hmframe* src);
void llsm_delete_hmframe(llsm_hmframe* dst);
void llsm_hmframe_phaseshift(llsm_hmframe* dst, FP_TYPE theta);
FP_TYPE* llsm_hmframe_harpsd(llsm_hmframe* src, int db_scale);

/* ---- noise model frame (replaces llsm.h:157-174) ---- */
typedef struct {
  llsm_hmframe** eenv;
  FP_TYPE* edc;
  FP_TYPE* psd;
  int npsd;
  int nchannel;
} llsm_nmframe;

llsm_nmframe* llsm_create_nmframe(int nchannel, int nhar_e, int npsd);
llsm_nmframe* llsm_copy_nmframe(llsm_nmframe* src);
void llsm_copy_nmframe_inplace(llsm_nmframe* dst, llsm_nmframe* src);
void llsm_delete_nmframe(llsm_nmframe* dst);

/* ---- glottal-flow hooks (replaces llsm.h:181-208).  Carried for ABI
 * completeness; pulse-by-pulse synthesis is outside this library's path. ---- */
typedef struct {
  FP_TYPE Fa;
  FP_TYPE Rk;
  FP_TYPE Rg;
  FP_TYPE T0;
  FP_TYPE Ee;
} llsm_gfm;
typedef void (*llsm_fgfm)(llsm_gfm* dst, FP_TYPE* delta_t, void* info,
  llsm_container* src_frame);
typedef struct {
  llsm_fgfm modifier;
  void* info;
} llsm_pbpeffect;
llsm_pbpeffect* llsm_create_pbpeffect(llsm_fgfm modifier, void* info);
llsm_pbpeffect* llsm_copy_pbpeffect(llsm_pbpeffect* src);
void llsm_delete_pbpeffect(llsm_pbpeffect* dst);

/* ---- frames (replaces llsm.h:217-243; layer-1 converters are not part of
 * this library) ---- */
llsm_container* llsm_create_frame(int nhar, int nchannel, int nhar_e, int npsd);
void llsm_frame_phaseshift(llsm_container* dst, FP_TYPE theta);
void llsm_frame_phasesync_rps(llsm_container* dst, int layer1_based);
/* frame.c:180-213 (needs the deprecated LLSM_CONF_NOSWARP; NULL without it, as the reference) */
FP_TYPE* llsm_frame_compute_snr(llsm_container* src, llsm_container* conf, int as_aperiodicity);
int  llsm_frame_checklayer0(llsm_container* src);
int  llsm_frame_checklayer1(llsm_container* src);
int  llsm_conf_checklayer0(llsm_container* src);
int  llsm_conf_checklayer1(llsm_container* src);                 /* replaces llsm.h:243 */
/* layer-1 (source-filter) conversion; replaces llsm.h:221, 324-327 (layer1.c:129-195).
 * llsm_chunk_tolayer1 attaches LLSM_CONF_NSPEC, LLSM_FRAME_RD on every frame and LLSM_FRAME_VTMAGN /
 * LLSM_FRAME_VSPHSE on voiced frames; llsm_frame_tolayer0 / llsm_chunk_tolayer0 rebuild LLSM_FRAME_HM. */
void llsm_frame_tolayer0(llsm_container* dst, llsm_container* conf);

/* ---- synthesis result (replaces llsm.h:246-255) ---- */
typedef struct {
  int ny;
  FP_TYPE fs;
  FP_TYPE* y;
  FP_TYPE* y_sin;
  FP_TYPE* y_noise;
} llsm_output;
void llsm_delete_output(llsm_output* dst);

/* ---- analysis options (replaces llsm.h:260-283; defaults layer0.c:27-43) */
typedef struct {
  FP_TYPE thop;
  int maxnhar;
  int maxnhar_e;
  int npsd;
  int nchannel;
  FP_TYPE* chanfreq;
  FP_TYPE lip_radius;
  int f0_refine;
  int hm_method;
  FP_TYPE rel_winsize;
} llsm_aoptions;

llsm_aoptions*  llsm_create_aoptions(void);
void            llsm_delete_aoptions(llsm_aoptions* dst);
llsm_container* llsm_aoptions_toconf(llsm_aoptions* src, FP_TYPE fnyq);

#define LLSM_AOPTION_HMPP  0
#define LLSM_AOPTION_HMCZT 1

/* ---- synthesis options (replaces llsm.h:290-304; defaults layer0.c:78-87) */
typedef struct {
  FP_TYPE fs;
  int use_iczt;
  int use_l1;
  FP_TYPE iczt_param_a;
  FP_TYPE iczt_param_b;
} llsm_soptions;

llsm_soptions* llsm_create_soptions(FP_TYPE fs);
void           llsm_delete_soptions(llsm_soptions* dst);

/* ---- chunks (replaces llsm.h:310-333) ---- */
typedef struct {
  llsm_container* conf;
  llsm_container** frames;
} llsm_chunk;

llsm_chunk* llsm_create_chunk(llsm_container* conf, int init_frames);
llsm_chunk* llsm_copy_chunk(llsm_chunk* src);
void        llsm_delete_chunk(llsm_chunk* dst);
void        llsm_chunk_phasesync_rps(llsm_chunk* dst, int layer1_based);
void        llsm_chunk_phasepropagate(llsm_chunk* dst, int sign);
FP_TYPE*    llsm_chunk_getf0(llsm_chunk* src, int* dst_nfrm);
void        llsm_chunk_tolayer1(llsm_chunk* dst, int nfft);    /* replaces llsm.h:324-325 */
void        llsm_chunk_tolayer0(llsm_chunk* dst);              /* replaces llsm.h:326-327 */

/* ---- THE HOT PATH (replaces llsm.h:336-339; layer0.c:478-511, 636-664).
 * Same contract as the reference: llsm_analyze returns a caller-owned chunk
 * (llsm_delete_chunk), rewrites f0[] when options->f0_refine, and hands back
 * the aperiodic residual in *x_ap (caller frees) when x_ap != NULL;
 * llsm_synthesize returns NULL when the chunk fails the layer-0 integrity
 * check.  Additionally NULL is returned when no gfx950 device / HIP runtime is
 * usable -- there is no CPU fallback (llsm_gpu_last_error() says why). ---- */
llsm_chunk*  llsm_analyze(llsm_aoptions* options, FP_TYPE* x, int nx,
  FP_TYPE fs, FP_TYPE* f0, int nfrm, FP_TYPE** x_ap);
llsm_output* llsm_synthesize(llsm_soptions* options, llsm_chunk* src);

/* ---- frame coder (replaces llsm.h:346-362; coder.c:44-292): frames <-> vectors of order_spec + order_bap + 3
 * values [voicing, f0, Rd, spectrum points on a mel axis, band aperiodicities].  Needs a layer-1 conf
 * (LLSM_CONF_NSPEC, LLSM_CONF_LIPRADIUS) and, for voiced frames, LLSM_FRAME_RD / LLSM_FRAME_VTMAGN. ---- */
typedef void llsm_coder;
llsm_coder* llsm_create_coder(llsm_container* conf, int order_spec, int order_bap);
void llsm_delete_coder(llsm_coder* dst);
FP_TYPE* llsm_coder_encode(llsm_coder* c, llsm_container* src);                   /* malloc'ed vector */
llsm_container* llsm_coder_decode_layer1(llsm_coder* c, FP_TYPE* src);
llsm_container* llsm_coder_decode_layer0(llsm_coder* c, FP_TYPE* src);
/* additive: many frames per launch (one frame per call costs a device round trip) */
int llsm_coder_dimension(llsm_coder* c);
int llsm_coder_encode_frames(llsm_coder* c, llsm_container** frames, int n, FP_TYPE* dst /* [n][dimension] */);
int llsm_coder_decode_frames(llsm_coder* c, const FP_TYPE* src, int n, int use_layer1, llsm_container** out);

#ifdef __cplusplus
}
#endif
#endif
