/* C99 host against include/buffer.h: the ring-buffer known-answer test of the reference
 * (test/test-structs.c:168-214: three pairs of equivalent realisations, exact equality), plus the dual
 * buffer and the object ring.  CPU only; built and run by tests/test_c_host.py with gcc -std=c99. */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include "buffer.h"

static int freed = 0;
static void count_free(void* p) { freed ++; free(p); }

int main(void) {
  llsm_ringbuffer* rb = llsm_create_ringbuffer(4096);
  FP_TYPE x[100], y[200];
  srand(1);
  for(int i = 0; i < 100; i ++) x[i] = (FP_TYPE)rand() / RAND_MAX - 0.5;
  for(int i = 0; i < 100; i ++) {
    if(i % 2 == 0) llsm_ringbuffer_appendchunk(rb, 100, x);
    else { llsm_ringbuffer_forward(rb, 100); llsm_ringbuffer_writechunk(rb, -100, 100, x); }
    if(i % 3 == 0) llsm_ringbuffer_appendblank(rb, 100);
    else for(int j = 0; j < 100; j ++) llsm_ringbuffer_append(rb, 0);
    if(i % 4 == 0) for(int j = 0; j < 200; j ++) y[j] = llsm_ringbuffer_read(rb, j - 200);
    else llsm_ringbuffer_readchunk(rb, -200, 200, y);
    for(int j = 0; j < 100; j ++) assert(y[j] == x[j]);
    for(int j = 100; j < 200; j ++) assert(y[j] == 0);
    if(i <= 5) continue;
    llsm_ringbuffer_readchunk(rb, -1300, 200, y);
    for(int j = 0; j < 100; j ++) assert(y[j] == 0);
    for(int j = 100; j < 200; j ++) assert(y[j] == x[j - 100]);
  }
  /* addchunk accumulates; write overwrites */
  llsm_ringbuffer_addchunk(rb, -200, 100, x);
  assert(llsm_ringbuffer_read(rb, -200) == x[0] + x[0]);
  llsm_ringbuffer_write(rb, -1, 7);
  assert(llsm_ringbuffer_read(rb, -1) == 7);
  llsm_delete_ringbuffer(rb);

  /* dual buffer: overlap-add into the future, retire into the past (llsmrt.c:380-384, 128) */
  llsm_dualbuffer* db = llsm_create_dualbuffer(64);
  FP_TYPE p[8] = {1, 2, 3, 4, 5, 6, 7, 8}, q[8];
  llsm_dualbuffer_addchunk(db, 2, 8, p);          /* future samples 2..9 */
  llsm_dualbuffer_addchunk(db, 6, 4, p);          /* overlap on 6..9 */
  llsm_dualbuffer_forward(db, 5);                 /* samples 0..4 are now the past: -5..-1 */
  llsm_dualbuffer_readchunk(db, -3, 8, q);        /* -3..4 == old 2..9 */
  FP_TYPE want[8] = {1, 2, 3, 4, 5 + 1, 6 + 2, 7 + 3, 8 + 4};
  for(int i = 0; i < 8; i ++) assert(q[i] == want[i]);
  llsm_dualbuffer_addchunk(db, -2, 4, p);         /* straddles the present: 2 past + 2 future samples */
  llsm_dualbuffer_readchunk(db, -2, 4, q);
  assert(q[0] == 2 + 1 && q[1] == 3 + 2 && q[2] == 4 + 3 && q[3] == 6 + 4);
  for(int k = 0; k < 200; k ++) llsm_dualbuffer_forward(db, 1);   /* wraps; everything retires and is cleared */
  llsm_dualbuffer_readchunk(db, 0, 8, q);
  for(int i = 0; i < 8; i ++) assert(q[i] == 0);
  llsm_delete_dualbuffer(db);

  /* object ring: entries are destroyed when overwritten and at deletion */
  llsm_vringbuffer* vr = llsm_create_vringbuffer(4, count_free);
  for(int i = 0; i < 6; i ++) { int* v = malloc(sizeof(int)); *v = i; llsm_vringbuffer_append(vr, v); }
  assert(freed == 2 && *(int*)llsm_vringbuffer_read(vr, -1) == 5 && *(int*)llsm_vringbuffer_read(vr, -4) == 2);
  int* v = malloc(sizeof(int)); *v = 42; llsm_vringbuffer_write(vr, -2, v);
  assert(freed == 3 && *(int*)llsm_vringbuffer_read(vr, -2) == 42);
  llsm_delete_vringbuffer(vr);
  assert(freed == 7);
  printf("buffer.h: ring / dual / object-ring KATs ok\n");
  return 0;
}
