// model_internal.h -- what model.cpp shares with the rest of the library beyond the public headers
#ifndef LLSM_AMD_MODEL_INTERNAL_H
#define LLSM_AMD_MODEL_INTERNAL_H
#include <stddef.h>
#include "llsm.h"
#include "llsm_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif
/* realloc for a member array of a frame that may live in a frame slab (model.cpp): never hands slab memory to the
 * allocator; keep_bytes of the old contents are carried over */
void* llsm_model_regrow(void* p, size_t keep_bytes, size_t new_bytes);
/* dst->frames[0 .. nfrm) of an analysed utterance from the flat rows frm_off .. (model.cpp): in one slab per chunk
 * (llsm_frames_from_flat; llsm_analyze_batch) or, use_slabs = 0, as ordinary heap objects (the drop-in llsm_analyze) */
void llsm_frames_from_flat(const llsm_flat_params* src, int frm_off, llsm_chunk* dst, int nfrm);
void llsm_frames_from_flat_ex(const llsm_flat_params* src, int frm_off, llsm_chunk* dst, int nfrm, int use_slabs);
// heap frames allocated from the plan's sizes (src: f0 / nhar / nhar_e rows + scalar sizes only), then filled from the analysed rows
void llsm_frames_heap_prealloc(const llsm_flat_params* src, int frm_off, llsm_chunk* dst, int nfrm);
int llsm_frames_heap_fill(const llsm_flat_params* src, int frm_off, llsm_chunk* dst, int nfrm);
/* frames laid over packed records that the device copied into a registered slab (model.cpp; capi.cpp analyze_block) */
struct LlsmPackedLayout;
void llsm_slab_set_pin_hooks(void* (*alloc_locked)(size_t), void (*free_locked)(void*));
void* llsm_frames_packed_begin(int nfrm, const struct LlsmPackedLayout* L, void** token, int page_locked);
void llsm_frames_packed_abort(void* token);
void llsm_frames_packed_finish(void* token, const struct LlsmPackedLayout* L, llsm_chunk* dst, int nfrm, FP_TYPE* f0_out);
int llsm_chunk_packed_view(llsm_chunk* src, int nfrm, struct LlsmPackedLayout* L, const void** records);
/* slab pool: bytes of live slabs; a call that produced `bytes` of slabs lets the pool keep that much (model.cpp pool_cap) */
/* pooled outputs (model.cpp): struct + three arrays of ny samples in one block; llsm_delete_output knows them */
llsm_output* llsm_output_create_pooled(int ny, FP_TYPE fs, int page_locked);
long long llsm_output_live_bytes(void);
void llsm_output_pool_hint(size_t bytes);
void llsm_output_pool_trim(void);
long long llsm_slab_live_bytes(void);
void llsm_slab_pool_hint(size_t bytes);
#ifdef __cplusplus
}
#endif
#endif
