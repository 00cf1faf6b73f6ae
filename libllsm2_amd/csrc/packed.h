// packed.h -- frame-major layout of the analysed rows of one frame ("packed frame"), shared by the device kernel that
// forms it (kernels.hip k_pack_frames), the engine call that ships it (engine.cpp llsm_gpu_batch_download_packed) and the
// host code that lays the reference's frame objects over it (model.cpp llsm_frames_packed_finish).
// Offsets in 4-byte words from the start of a frame's record; every piece starts on a 16-byte boundary.
//   [0] f0   [1] nhar   [2] nhar_e   [3] has_psdres
//   ampl[maxnhar] | phse[maxnhar] | psd[npsd] | 3 words of padding, int length = npsd | psdres[npsd]      (an fparray:
//   the length sits in the int right before the data, container.c:35-46) | edc[nch] | eenv_ampl[nch][me] | eenv_phse[nch][me]
#ifndef LLSM_AMD_PACKED_H
#define LLSM_AMD_PACKED_H
struct LlsmPackedLayout {
  int maxnhar, maxnhar_e, npsd, nch, me;   // me = max(maxnhar_e, 1): the width of an envelope row
  int o_ampl, o_phse, o_psd, o_reshdr, o_psdres, o_edc, o_eamp, o_ephs, words;
};
static inline int llsm_packed_up4(int w) { return (w + 3) & ~3; }
static inline LlsmPackedLayout llsm_packed_layout(int maxnhar, int maxnhar_e, int npsd, int nch) {
  LlsmPackedLayout L;
  L.maxnhar = maxnhar; L.maxnhar_e = maxnhar_e; L.npsd = npsd; L.nch = nch; L.me = maxnhar_e > 0 ? maxnhar_e : 1;
  int at = 4;
  L.o_ampl = at; at += llsm_packed_up4(maxnhar > 0 ? maxnhar : 1);
  L.o_phse = at; at += llsm_packed_up4(maxnhar > 0 ? maxnhar : 1);
  L.o_psd = at; at += llsm_packed_up4(npsd > 0 ? npsd : 1);
  L.o_reshdr = at; at += 4;
  L.o_psdres = at; at += llsm_packed_up4(npsd > 0 ? npsd : 1);
  L.o_edc = at; at += llsm_packed_up4(nch > 0 ? nch : 1);
  L.o_eamp = at; at += llsm_packed_up4((nch > 0 ? nch : 1) * L.me);
  L.o_ephs = at; at += llsm_packed_up4((nch > 0 ? nch : 1) * L.me);
  L.words = at;
  return L;
}
#endif
