// rt.cpp -- placeholder until the streaming kernels land.
#include "engine.h"
#include "llsmrt.h"

extern "C" {
llsm_rtsynth_buffer* llsm_create_rtsynth_buffer(llsm_soptions*, llsm_container*, int) {
  llsm_set_error("llsmrt: streaming synthesis not built yet"); return nullptr;
}
void llsm_delete_rtsynth_buffer(llsm_rtsynth_buffer*) {}
int  llsm_rtsynth_buffer_getlatency(llsm_rtsynth_buffer*) { return 0; }
int  llsm_rtsynth_buffer_numoutput(llsm_rtsynth_buffer*) { return 0; }
void llsm_rtsynth_buffer_feed(llsm_rtsynth_buffer*, llsm_container*) {}
int  llsm_rtsynth_buffer_fetch(llsm_rtsynth_buffer*, FP_TYPE*) { return 0; }
int  llsm_rtsynth_buffer_fetch_decomposed(llsm_rtsynth_buffer*, FP_TYPE*, FP_TYPE*) { return 0; }
void llsm_rtsynth_buffer_clear(llsm_rtsynth_buffer*) {}
}
