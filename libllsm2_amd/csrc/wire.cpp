// wire.cpp -- flat wire format of a layer-0 chunk.
//
// The reference has no serialisation: a chunk is a tree of ~25 heap blocks per frame
// (container.c:158-195, frame.c:137-150).  The blob below is ONE contiguous,
// position-independent block holding the conf scalars and the same struct-of-arrays rows the
// device batch uses (llsm_flat_params), so an analysed utterance can be cached on disk, sent
// between ranks, or uploaded into a batch without building the container tree at all.
//
//   header (little-endian, 8-byte aligned sections)
//     char     magic[8]   "LLSM2L0\0"
//     uint32   version    2   (version 1 = the same without the layer-1 arrays; still read)
//     uint32   header_bytes
//     int32    nfrm, maxnhar, maxnhar_e, npsd, nchannel, nchanfreq
//     float    thop, fnyq, lip_radius; int32 nspec (0: no layer-1 members; version 1: reserved)
//     uint64   total_bytes
//     uint64   offset[LLSM_BLOB_NARRAYS]   (from the start of the blob)
//   arrays: chanfreq[nchanfreq], f0[F], nhar[F], ampl[F][maxnhar], phse[F][maxnhar],
//           psd[F][npsd], psdres[F][npsd], has_psdres[F], edc[F][nchannel], nhar_e[F],
//           eenv_ampl[F][nchannel][max(maxnhar_e,1)], eenv_phse[...]
//   version 2, when nspec > 0 (LLSM_CONF_NSPEC present): rd[F], has_rd[F], vtmagn[F][nspec], vsphse[F][maxnhar],
//           nvsphse[F], pbpsyn[F], has_hm[F]   (llsm_flat_l1; LLSM_FRAME_PBPEFF is a host function pointer and
//           is not carried)
// Row widths are the largest nhar / envelope nhar present in the chunk, not the analysis
// maxima, so a blob is as small as its content.  Host-only code (no device access).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

#include "engine.h"
#include "llsm_gpu.h"

namespace {
enum { A_CHANFREQ, A_F0, A_NHAR, A_AMPL, A_PHSE, A_PSD, A_PSDRES, A_HASRES, A_EDC, A_NHAR_E, A_EAMP, A_EPHS, NARR1,
       A_RD = NARR1, A_HASRD, A_VTMAGN, A_VSPHSE, A_NVS, A_PBPSYN, A_HASHM, NARR };

struct Header {
  char magic[8];
  uint32_t version, header_bytes;
  int32_t nfrm, maxnhar, maxnhar_e, npsd, nchannel, nchanfreq;
  float thop, fnyq, lip_radius; int32_t nspec;
  uint64_t total_bytes;
  uint64_t offset[NARR];                               // version 1 blobs carry the first NARR1 entries only
};
size_t header_bytes_of(uint32_t version) { return sizeof(Header) - (version == 1 ? sizeof(uint64_t) * (NARR - NARR1) : 0); }
const char MAGIC[8] = {'L', 'L', 'S', 'M', '2', 'L', '0', '\0'};

size_t pad8(size_t n) { return (n + 7) & ~(size_t)7; }

struct Shape { int nfrm, maxnhar, me, npsd, nch, ncf, nspec = 0; float thop, fnyq, lip; const FP_TYPE* chanfreq; };

bool shape_of(llsm_chunk* c, Shape& s) {
  if(! c || ! c -> conf) return false;
  int* nfrm = (int*)llsm_container_get(c -> conf, LLSM_CONF_NFRM);
  int* npsd = (int*)llsm_container_get(c -> conf, LLSM_CONF_NPSD);
  int* nch = (int*)llsm_container_get(c -> conf, LLSM_CONF_NCHANNEL);
  FP_TYPE* thop = (FP_TYPE*)llsm_container_get(c -> conf, LLSM_CONF_THOP);
  FP_TYPE* fnyq = (FP_TYPE*)llsm_container_get(c -> conf, LLSM_CONF_FNYQ);
  FP_TYPE* lip = (FP_TYPE*)llsm_container_get(c -> conf, LLSM_CONF_LIPRADIUS);
  FP_TYPE* cf = (FP_TYPE*)llsm_container_get(c -> conf, LLSM_CONF_CHANFREQ);
  if(! nfrm || ! npsd || ! nch || ! thop || ! fnyq) return false;
  s.nfrm = *nfrm; s.npsd = *npsd; s.nch = *nch; s.thop = *thop; s.fnyq = *fnyq;
  s.lip = lip ? *lip : (FP_TYPE)1.5;
  s.chanfreq = cf; s.ncf = cf ? llsm_fparray_length(cf) : 0;
  s.maxnhar = 0; s.me = 0;
  int* nspec = (int*)llsm_container_get(c -> conf, LLSM_CONF_NSPEC);
  s.nspec = nspec && *nspec > 0 ? *nspec : 0;
  for(int i = 0; i < s.nfrm; i ++) {
    FP_TYPE* vs = (FP_TYPE*)llsm_container_get(c -> frames[i], LLSM_FRAME_VSPHSE);
    if(s.nspec && vs && llsm_fparray_length(vs) > s.maxnhar) s.maxnhar = llsm_fparray_length(vs);
    llsm_hmframe* hm = (llsm_hmframe*)llsm_container_get(c -> frames[i], LLSM_FRAME_HM);
    llsm_nmframe* nm = (llsm_nmframe*)llsm_container_get(c -> frames[i], LLSM_FRAME_NM);
    if(hm && hm -> nhar > s.maxnhar) s.maxnhar = hm -> nhar;
    if(nm) for(int k = 0; k < nm -> nchannel; k ++)
      if(nm -> eenv[k] && nm -> eenv[k] -> nhar > s.me) s.me = nm -> eenv[k] -> nhar;
  }
  return s.nfrm >= 0 && s.npsd > 0 && s.nch > 0;
}

// byte sizes of the arrays, in blob order
void array_bytes(const Shape& s, size_t* b) {
  const size_t F = (size_t)s.nfrm, me = (size_t)(s.me > 0 ? s.me : 1);
  b[A_CHANFREQ] = sizeof(float) * (size_t)s.ncf;
  b[A_F0] = sizeof(float) * F; b[A_NHAR] = sizeof(int32_t) * F;
  b[A_AMPL] = b[A_PHSE] = sizeof(float) * F * (size_t)s.maxnhar;
  b[A_PSD] = b[A_PSDRES] = sizeof(float) * F * (size_t)s.npsd;
  b[A_HASRES] = sizeof(int32_t) * F;
  b[A_EDC] = sizeof(float) * F * (size_t)s.nch;
  b[A_NHAR_E] = sizeof(int32_t) * F;
  b[A_EAMP] = b[A_EPHS] = sizeof(float) * F * (size_t)s.nch * me;
  const size_t L1 = s.nspec > 0 ? 1 : 0;
  b[A_RD] = sizeof(float) * F * L1; b[A_HASRD] = b[A_NVS] = b[A_PBPSYN] = b[A_HASHM] = sizeof(int32_t) * F * L1;
  b[A_VTMAGN] = sizeof(float) * F * (size_t)s.nspec; b[A_VSPHSE] = sizeof(float) * F * (size_t)s.maxnhar * L1;
}

size_t layout(const Shape& s, uint64_t* off, uint32_t version = 2) {
  size_t b[NARR]; array_bytes(s, b);
  size_t at = pad8(header_bytes_of(version));
  const int narr = version == 1 ? NARR1 : NARR;
  for(int i = 0; i < narr; i ++) { if(off) off[i] = at; at += pad8(b[i]); }
  return at;
}

llsm_flat_l1 l1_view_of(const Header& h, unsigned char* base) {
  llsm_flat_l1 v; std::memset(& v, 0, sizeof(v));
  v.nspec = h.version >= 2 ? h.nspec : 0; v.maxnhar = h.maxnhar;
  if(v.nspec <= 0) return v;
  v.rd = (FP_TYPE*)(base + h.offset[A_RD]); v.has_rd = (int*)(base + h.offset[A_HASRD]);
  v.vtmagn = (FP_TYPE*)(base + h.offset[A_VTMAGN]); v.vsphse = (FP_TYPE*)(base + h.offset[A_VSPHSE]);
  v.nvsphse = (int*)(base + h.offset[A_NVS]); v.pbpsyn = (int*)(base + h.offset[A_PBPSYN]); v.has_hm = (int*)(base + h.offset[A_HASHM]);
  return v;
}

llsm_flat_params view_of(const Header& h, unsigned char* base) {
  llsm_flat_params v;
  v.maxnhar = h.maxnhar; v.maxnhar_e = h.maxnhar_e; v.npsd = h.npsd; v.nchannel = h.nchannel;
  v.f0 = (FP_TYPE*)(base + h.offset[A_F0]); v.nhar = (int*)(base + h.offset[A_NHAR]);
  v.ampl = (FP_TYPE*)(base + h.offset[A_AMPL]); v.phse = (FP_TYPE*)(base + h.offset[A_PHSE]);
  v.psd = (FP_TYPE*)(base + h.offset[A_PSD]); v.psdres = (FP_TYPE*)(base + h.offset[A_PSDRES]);
  v.has_psdres = (int*)(base + h.offset[A_HASRES]); v.edc = (FP_TYPE*)(base + h.offset[A_EDC]);
  v.nhar_e = (int*)(base + h.offset[A_NHAR_E]);
  v.eenv_ampl = (FP_TYPE*)(base + h.offset[A_EAMP]); v.eenv_phse = (FP_TYPE*)(base + h.offset[A_EPHS]);
  return v;
}

// every field of a header read from untrusted bytes is checked before any pointer is formed
bool header_ok(const Header& h, size_t bytes);
bool read_header(const void* blob, size_t bytes, Header& h) {
  std::memset(& h, 0, sizeof(h));
  if(bytes < header_bytes_of(1)) return false;
  std::memcpy(& h, blob, header_bytes_of(1));
  if(h.version == 2) { if(bytes < sizeof(Header)) return false; std::memcpy(& h, blob, sizeof(Header)); }
  return header_ok(h, bytes);
}
bool header_ok(const Header& h, size_t bytes) {
  if(std::memcmp(h.magic, MAGIC, 8) != 0 || (h.version != 1 && h.version != 2)) return false;
  if(h.header_bytes != header_bytes_of(h.version) || h.total_bytes != bytes) return false;
  if(h.version == 2 && (h.nspec < 0 || h.nspec > 65537)) return false;
  if(h.nfrm < 0 || h.nfrm > (1 << 24) || h.maxnhar < 0 || h.maxnhar > 65536) return false;
  if(h.maxnhar_e < 0 || h.maxnhar_e > 4096 || h.npsd <= 0 || h.npsd > 65536) return false;
  if(h.nchannel <= 0 || h.nchannel > 4096 || h.nchanfreq < 0 || h.nchanfreq > 4096) return false;
  Shape s; s.nfrm = h.nfrm; s.maxnhar = h.maxnhar; s.me = h.maxnhar_e; s.npsd = h.npsd;
  s.nch = h.nchannel; s.ncf = h.nchanfreq; s.nspec = h.version == 2 ? h.nspec : 0;
  uint64_t off[NARR];
  if(layout(s, off, h.version) != bytes) return false;
  for(int i = 0; i < (h.version == 1 ? NARR1 : NARR); i ++) if(off[i] != h.offset[i]) return false;
  return true;
}
}  // namespace

extern "C" size_t llsm_chunk_blob_size(llsm_chunk* src) {
  Shape s;
  if(! shape_of(src, s)) { llsm_set_error("llsm_chunk_blob_size: chunk without NFRM/NPSD/NCHANNEL/THOP/FNYQ"); return 0; }
  return layout(s, nullptr);
}

extern "C" long long llsm_chunk_to_blob(llsm_chunk* src, void* dst, size_t capacity) {
  Shape s;
  if(! shape_of(src, s)) { llsm_set_error("llsm_chunk_to_blob: chunk without NFRM/NPSD/NCHANNEL/THOP/FNYQ"); return -1; }
  Header h; std::memset(& h, 0, sizeof(h));
  const size_t total = layout(s, h.offset);
  if(! dst || capacity < total) { llsm_set_error("llsm_chunk_to_blob: destination too small"); return -1; }
  std::memcpy(h.magic, MAGIC, 8);
  h.version = 2; h.header_bytes = (uint32_t)header_bytes_of(2); h.nspec = s.nspec;
  h.nfrm = s.nfrm; h.maxnhar = s.maxnhar; h.maxnhar_e = s.me; h.npsd = s.npsd; h.nchannel = s.nch;
  h.nchanfreq = s.ncf; h.thop = s.thop; h.fnyq = s.fnyq; h.lip_radius = s.lip; h.total_bytes = total;
  unsigned char* base = (unsigned char*)dst;
  std::memset(base, 0, total);
  std::memcpy(base, & h, sizeof(h));
  if(s.ncf > 0) std::memcpy(base + h.offset[A_CHANFREQ], s.chanfreq, sizeof(float) * (size_t)s.ncf);
  llsm_flat_params v = view_of(h, base);
  if(llsm_chunk_to_flat(src, & v, 0)) { llsm_set_error("llsm_chunk_to_blob: malformed chunk"); return -1; }
  if(s.nspec > 0) {
    llsm_flat_l1 q = l1_view_of(h, base);
    if(llsm_chunk_to_flat_l1(src, & q, 0)) { llsm_set_error("llsm_chunk_to_blob: malformed chunk"); return -1; }
  }
  return (long long)total;
}

extern "C" int llsm_blob_view(const void* blob, size_t bytes, llsm_flat_params* view, int* nfrm,
  FP_TYPE* thop, FP_TYPE* fnyq) {
  if(! blob || bytes < header_bytes_of(1) || ! view) { llsm_set_error("llsm_blob_view: truncated blob"); return -1; }
  // the view forms float* / int* into the blob: offsets are 8-aligned relative to its start, so the
  // start itself must be (a blob at an odd offset inside a network / file buffer must be copied first)
  if(((uintptr_t)blob & 7u) != 0) { llsm_set_error("llsm_blob_view: blob address must be 8-byte aligned"); return -1; }
  Header h;
  if(! read_header(blob, bytes, h)) { llsm_set_error("llsm_blob_view: not a version-1 / version-2 LLSM2L0 blob of this size"); return -1; }
  *view = view_of(h, (unsigned char*)blob);
  if(nfrm) *nfrm = h.nfrm;
  if(thop) *thop = h.thop;
  if(fnyq) *fnyq = h.fnyq;
  // rows are untrusted too: harmonic counts must fit their row widths
  for(int i = 0; i < h.nfrm; i ++)
    if(view -> nhar[i] < 0 || view -> nhar[i] > h.maxnhar || view -> nhar_e[i] < 0 ||
       view -> nhar_e[i] > (h.maxnhar_e > 0 ? h.maxnhar_e : 0)) {
      llsm_set_error("llsm_blob_view: harmonic count outside its row"); return -1;
    }
  if(h.version == 2 && h.nspec > 0) {
    const llsm_flat_l1 q = l1_view_of(h, (unsigned char*)blob);
    for(int i = 0; i < h.nfrm; i ++)
      if(q.nvsphse[i] < 0 || q.nvsphse[i] > h.maxnhar) { llsm_set_error("llsm_blob_view: VSPHSE length outside its row"); return -1; }
  }
  return 0;
}

// layer-1 rows of a (validated) blob: view -> nspec == 0 when the blob carries none
extern "C" int llsm_blob_view_l1(const void* blob, size_t bytes, llsm_flat_l1* view) {
  llsm_flat_params v;
  if(! view || llsm_blob_view(blob, bytes, & v, nullptr, nullptr, nullptr)) return -1;
  Header h; read_header(blob, bytes, h);
  *view = l1_view_of(h, (unsigned char*)blob);
  return 0;
}

extern "C" llsm_chunk* llsm_blob_to_chunk(const void* blob, size_t bytes) {
  llsm_flat_params v; int nfrm = 0;
  if(llsm_blob_view(blob, bytes, & v, & nfrm, nullptr, nullptr)) return nullptr;
  Header h; read_header(blob, bytes, h);
  llsm_aoptions ao; std::memset(& ao, 0, sizeof(ao));
  ao.thop = h.thop; ao.maxnhar = h.maxnhar; ao.maxnhar_e = h.maxnhar_e; ao.npsd = h.npsd;
  ao.nchannel = h.nchannel; ao.lip_radius = h.lip_radius;
  // the conf stores nchannel - 1 band edges (layer0.c:63-66); a blob with fewer gets zeros
  FP_TYPE* cf = (FP_TYPE*)std::calloc((size_t)(h.nchannel > 1 ? h.nchannel - 1 : 1), sizeof(FP_TYPE));
  const FP_TYPE* src_cf = (const FP_TYPE*)((const unsigned char*)blob + h.offset[A_CHANFREQ]);
  for(int i = 0; i < h.nchannel - 1 && i < h.nchanfreq; i ++) cf[i] = src_cf[i];
  ao.chanfreq = cf;
  llsm_container* conf = llsm_aoptions_toconf(& ao, h.fnyq);
  std::free(cf);
  *(int*)llsm_container_get(conf, LLSM_CONF_NFRM) = nfrm;
  llsm_chunk* ch = llsm_create_chunk(conf, 1);
  llsm_delete_container(conf);
  if(! ch) { llsm_set_error("llsm_blob_to_chunk: out of memory"); return nullptr; }
  if(llsm_flat_to_chunk(& v, 0, ch)) { llsm_delete_chunk(ch); return nullptr; }
  if(h.version == 2 && h.nspec > 0) {                   // layer-1 members: LLSM_CONF_NSPEC, RD / VTMAGN / VSPHSE / PBPSYN; HM only where it was
    llsm_container_attach_(ch -> conf, LLSM_CONF_NSPEC, llsm_create_int(h.nspec), (llsm_fdestructor)llsm_delete_int, (llsm_fcopy)llsm_copy_int);
    const llsm_flat_l1 q = l1_view_of(h, (unsigned char*)blob);
    if(llsm_flat_l1_to_chunk(& q, 0, ch)) { llsm_delete_chunk(ch); return nullptr; }
    for(int i = 0; i < nfrm; i ++) if(! q.has_hm[i]) llsm_container_attach_(ch -> frames[i], LLSM_FRAME_HM, NULL, NULL, NULL);
  }
  return ch;
}
