/*
 * buffer.h -- ring / dual / object-ring buffers of libllsm2_amd (header-only C99).
 *
 * Written from scratch; replaces the reference's installed buffer.h (buffer.h:32-273): same type
 * names, field order (users read .data / .capacity / .curr directly), function names and semantics:
 *   - a ring's `curr` is the write head; samples are addressed by NEGATIVE lags relative to it
 *   - appendchunk(n, src) == forward(n) + writechunk(-n, n, src); appendblank(n) zero-fills
 *   - a dual buffer keeps the past (data_bkwd) and the future (data_frwd) of the same time axis:
 *     offsets < 0 address the past, offsets >= 0 the future; forward(n) retires n future samples
 *   - an object ring owns its entries through the destructor given at creation
 * The llsmrt path of this library keeps its audio rings on the device (csrc/rt.cpp, k_rt_*); these
 * host-side buffers are the same contract for host code, as in the reference (llsmrt.c uses them).
 */
#ifndef LLSM_AMD_BUFFER_H
#define LLSM_AMD_BUFFER_H

#include <assert.h>
#include <stdlib.h>
#include <string.h>
#include "llsm.h"

/* slot of lag `lag` (any sign) relative to head `curr` in a buffer of `cap` slots */
static inline int llsm_buffer_slot_(int curr, int lag, int cap) {
  int s = (curr + lag) % cap;
  return s < 0 ? s + cap : s;
}

/* ---- llsm_ringbuffer (replaces buffer.h:32-138) ---- */
typedef struct {
  FP_TYPE* data;
  int capacity;
  int curr;
} llsm_ringbuffer;

static inline llsm_ringbuffer* llsm_create_ringbuffer(int capacity) {
  assert(capacity > 0);
  llsm_ringbuffer* r = (llsm_ringbuffer*)malloc(sizeof(llsm_ringbuffer));
  r -> data = (FP_TYPE*)calloc((size_t)capacity, sizeof(FP_TYPE));
  r -> capacity = capacity; r -> curr = 0;
  return r;
}
static inline void llsm_delete_ringbuffer(llsm_ringbuffer* dst) {
  if(dst == NULL) return;
  free(dst -> data); free(dst);
}
static inline FP_TYPE llsm_ringbuffer_read(llsm_ringbuffer* src, int idx) {
  assert(idx < 0 && idx >= -src -> capacity);
  return src -> data[llsm_buffer_slot_(src -> curr, idx, src -> capacity)];
}
static inline void llsm_ringbuffer_write(llsm_ringbuffer* dst, int idx, FP_TYPE x) {
  assert(idx < 0 && idx >= -dst -> capacity);
  dst -> data[llsm_buffer_slot_(dst -> curr, idx, dst -> capacity)] = x;
}
static inline void llsm_ringbuffer_forward(llsm_ringbuffer* dst, int size) {
  dst -> curr = llsm_buffer_slot_(dst -> curr, size, dst -> capacity);
}
static inline void llsm_ringbuffer_append(llsm_ringbuffer* dst, FP_TYPE x) {
  dst -> data[dst -> curr] = x;
  llsm_ringbuffer_forward(dst, 1);
}
static inline void llsm_ringbuffer_readchunk(llsm_ringbuffer* src, int lag, int size, FP_TYPE* dst) {
  assert(size > 0 && lag + size <= 0 && lag > -src -> capacity);
  for(int i = 0; i < size; i ++) dst[i] = src -> data[llsm_buffer_slot_(src -> curr, lag + i, src -> capacity)];
}
static inline void llsm_ringbuffer_writechunk(llsm_ringbuffer* dst, int lag, int size, FP_TYPE* src) {
  assert(size > 0 && lag + size <= 0 && lag >= -dst -> capacity);
  for(int i = 0; i < size; i ++) dst -> data[llsm_buffer_slot_(dst -> curr, lag + i, dst -> capacity)] = src[i];
}
static inline void llsm_ringbuffer_addchunk(llsm_ringbuffer* dst, int lag, int size, FP_TYPE* src) {
  assert(size > 0 && lag + size <= 0 && lag >= -dst -> capacity);
  for(int i = 0; i < size; i ++) dst -> data[llsm_buffer_slot_(dst -> curr, lag + i, dst -> capacity)] += src[i];
}
static inline void llsm_ringbuffer_appendchunk(llsm_ringbuffer* dst, int size, FP_TYPE* src) {
  assert(size > 0 && size <= dst -> capacity);
  llsm_ringbuffer_forward(dst, size);
  llsm_ringbuffer_writechunk(dst, -size, size, src);
}
static inline void llsm_ringbuffer_appendblank(llsm_ringbuffer* dst, int size) {
  assert(size > 0 && size <= dst -> capacity);
  llsm_ringbuffer_forward(dst, size);
  for(int i = 0; i < size; i ++) dst -> data[llsm_buffer_slot_(dst -> curr, i - size, dst -> capacity)] = 0;
}

/* ---- llsm_dualbuffer (replaces buffer.h:140-217) ---- */
typedef struct {
  FP_TYPE* data_frwd;
  FP_TYPE* data_bkwd;
  int capacity;
  int curr;
} llsm_dualbuffer;

static inline llsm_dualbuffer* llsm_create_dualbuffer(int capacity) {
  assert(capacity > 0);
  llsm_dualbuffer* d = (llsm_dualbuffer*)malloc(sizeof(llsm_dualbuffer));
  d -> data_frwd = (FP_TYPE*)calloc((size_t)capacity, sizeof(FP_TYPE));
  d -> data_bkwd = (FP_TYPE*)calloc((size_t)capacity, sizeof(FP_TYPE));
  d -> capacity = capacity; d -> curr = 0;
  return d;
}
static inline void llsm_delete_dualbuffer(llsm_dualbuffer* dst) {
  if(dst == NULL) return;
  free(dst -> data_frwd); free(dst -> data_bkwd); free(dst);
}
/* the half that holds offset + i: the past for negative positions, the future otherwise */
static inline FP_TYPE* llsm_dualbuffer_half_(llsm_dualbuffer* b, int pos) {
  return pos < 0 ? b -> data_bkwd : b -> data_frwd;
}
static inline void llsm_dualbuffer_readchunk(llsm_dualbuffer* src, int offset, int size, FP_TYPE* dst) {
  assert(size > 0 && size < src -> capacity);
  for(int i = 0; i < size; i ++)
    dst[i] = llsm_dualbuffer_half_(src, offset + i)[llsm_buffer_slot_(src -> curr, offset + i, src -> capacity)];
}
static inline void llsm_dualbuffer_forward(llsm_dualbuffer* dst, int size) {
  for(int i = 0; i < size; i ++) {
    dst -> data_bkwd[dst -> curr] = dst -> data_frwd[dst -> curr];
    dst -> data_frwd[dst -> curr] = 0;
    dst -> curr = llsm_buffer_slot_(dst -> curr, 1, dst -> capacity);
  }
}
static inline void llsm_dualbuffer_addchunk(llsm_dualbuffer* dst, int offset, int size, FP_TYPE* src) {
  assert(size > 0 && size < dst -> capacity);
  for(int i = 0; i < size; i ++)
    llsm_dualbuffer_half_(dst, offset + i)[llsm_buffer_slot_(dst -> curr, offset + i, dst -> capacity)] += src[i];
}

/* ---- llsm_vringbuffer: ring of owned objects (replaces buffer.h:219-273) ---- */
typedef struct {
  void** data;
  int capacity;
  int curr;
  llsm_fdestructor destructor;
} llsm_vringbuffer;

static inline llsm_vringbuffer* llsm_create_vringbuffer(int capacity, llsm_fdestructor destructor) {
  llsm_vringbuffer* r = (llsm_vringbuffer*)malloc(sizeof(llsm_vringbuffer));
  r -> data = (void**)calloc((size_t)capacity, sizeof(void*));
  r -> capacity = capacity; r -> curr = 0; r -> destructor = destructor;
  return r;
}
static inline void llsm_vringbuffer_replace_(llsm_vringbuffer* r, int slot, void* x) {
  if(r -> data[slot] != NULL) r -> destructor(r -> data[slot]);
  r -> data[slot] = x;
}
static inline void llsm_delete_vringbuffer(llsm_vringbuffer* dst) {
  if(dst == NULL) return;
  for(int i = 0; i < dst -> capacity; i ++) llsm_vringbuffer_replace_(dst, i, NULL);
  free(dst -> data); free(dst);
}
static inline void* llsm_vringbuffer_read(llsm_vringbuffer* src, int idx) {
  assert(idx < 0 && idx >= -src -> capacity);
  return src -> data[llsm_buffer_slot_(src -> curr, idx, src -> capacity)];
}
static inline void llsm_vringbuffer_write(llsm_vringbuffer* dst, int idx, void* x) {
  assert(idx < 0 && idx >= -dst -> capacity);
  llsm_vringbuffer_replace_(dst, llsm_buffer_slot_(dst -> curr, idx, dst -> capacity), x);
}
static inline void llsm_vringbuffer_append(llsm_vringbuffer* dst, void* x) {
  llsm_vringbuffer_replace_(dst, dst -> curr, x);
  dst -> curr = llsm_buffer_slot_(dst -> curr, 1, dst -> capacity);
}

#endif
