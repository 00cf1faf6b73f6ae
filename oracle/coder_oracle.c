/*
 * coder_oracle.c -- CPU restatement of libllsm2's frame coder (coder.c:44-292).  TEST INFRASTRUCTURE ONLY; "parity unpinned" (oracle.h).
 *   coder.c:44-77    llsm_create_coder            -> o_coder_create
 *   coder.c:88-168   llsm_coder_encode            -> o_coder_encode
 *   coder.c:170-286  llsm_coder_decode_layer0/1   -> o_coder_decode
 * ciglet / Ooura primitives, OUR definitions:
 *   ddct(n, -1, a)   C[k] = sum_j a[j] cos(pi (j + 1/2) k / n)   (DCT-II, unnormalised; Ooura fft4g "case 2")
 *   ddct(n, +1, a)   C[k] = sum_j a[j] cos(pi j (k + 1/2) / n)   (its inverse up to a[0] / 2 and 2 / n)
 *   freq2mel(f) = 1127.01048 ln(1 + f / 700), mel2freq its inverse (HTK mel)
 *   exp_2 / log_2    exact exp / log
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define DB2LOG(x) ((x) * 2.3025851 / 20.0)
#define LOG2DB(x) ((x) / 2.3025851 * 20.0)
#define IN2LOG(x) ((x) * 2.3025851 / 10.0)
#define LOG2IN(x) ((x) / 2.3025851 * 10.0)

static void ddct(int n, int isgn, fp* a) {
  /* cos(pi m / (2 n)), m < 4 n, then (2 j + 1) k mod 4 n indexes it; zero inputs are skipped */
  double* tab = malloc(sizeof(double) * 4 * n);
  double* out = calloc(n, sizeof(double));
  for(int m = 0; m < 4 * n; m ++) tab[m] = cos(M_PI * m / (2.0 * n));
  for(int j = 0; j < n; j ++) {
    if(a[j] == 0) continue;
    for(int k = 0; k < n; k ++) {
      long long m = isgn < 0 ? (long long)(2 * j + 1) * k : (long long)j * (2 * k + 1);
      out[k] += (double)a[j] * tab[m % (4 * n)];
    }
  }
  for(int k = 0; k < n; k ++) a[k] = (fp)out[k];
  free(out); free(tab);
}
static double freq2mel(double f) { return 1127.01048 * log(1.0 + f / 700.0); }
static double mel2freq(double m) { return 700.0 * (exp(m / 1127.01048) - 1.0); }

struct o_coder {
  int order_spec, order_bap, nfullspec, nchannel, nhar_e, npsd;
  fp fnyq, liprad;
  fp* psdaxis; fp* melaxis; fp* faxis; fp* apaxis;
};

o_coder* o_coder_create(fp fnyq, int nchannel, int nhar_e, int npsd, int nspec, fp liprad, int order_spec, int order_bap) {
  o_coder* c = calloc(1, sizeof(o_coder));
  c -> order_spec = order_spec; c -> order_bap = order_bap; c -> nfullspec = (nspec - 1) * 2;
  c -> nchannel = nchannel; c -> nhar_e = nhar_e; c -> npsd = npsd; c -> fnyq = fnyq; c -> liprad = liprad;
  c -> faxis = calloc(c -> nfullspec, sizeof(fp));
  for(int i = 0; i < c -> nfullspec; i ++) c -> faxis[i] = (fp)(fnyq * 2 * i / c -> nfullspec);
  c -> psdaxis = calloc(npsd, sizeof(fp));
  for(int i = 0; i < npsd; i ++) c -> psdaxis[i] = (fp)((double)fnyq * i / (npsd - 1));
  double mel_ceil = freq2mel(fnyq), mel_floor = freq2mel(50);
  c -> melaxis = calloc(nspec, sizeof(fp));
  for(int i = 0; i < nspec; i ++) c -> melaxis[i] = (fp)mel2freq(mel_floor + (mel_ceil - mel_floor) * i / nspec);
  c -> apaxis = calloc(order_bap + 1, sizeof(fp));
  for(int i = 0; i <= order_bap; i ++) c -> apaxis[i] = (fp)((double)fnyq * i / order_bap);
  return c;
}
void o_coder_delete(o_coder* c) {
  if(! c) return;
  free(c -> psdaxis); free(c -> melaxis); free(c -> faxis); free(c -> apaxis); free(c);
}

/* enc: order_spec + order_bap + 3 values.  psd: npsd dB values; vtmagn: ns dB values (voiced frames). */
void o_coder_encode(const o_coder* c, fp f0, fp rd, const fp* psd, const fp* vtmagn, fp* enc) {
  int ns = c -> nfullspec / 2 + 1;
  memset(enc, 0, sizeof(fp) * (c -> order_spec + c -> order_bap + 3));
  enc[0] = f0 > 0; enc[1] = f0;
  fp* spec_psd = malloc(sizeof(fp) * ns);
  o_interp1(c -> psdaxis, psd, c -> npsd, c -> faxis, ns, spec_psd);
  for(int j = 0; j < ns; j ++) spec_psd[j] = (fp)exp(IN2LOG((double)spec_psd[j]));
  if(f0 > 0) {
    enc[2] = rd;
    o_lfmodel gfm = o_lfmodel_from_rd(rd, (fp)(1.0 / f0), 1.0);
    fp* lfmagn = malloc(sizeof(fp) * ns); fp lfmagnf0 = 0;
    o_lfmodel_spectrum(gfm, c -> faxis, ns, lfmagn, NULL);
    o_lfmodel_spectrum(gfm, & f0, 1, & lfmagnf0, NULL);
    fp* spec_env = calloc(ns, sizeof(fp));
    for(int j = 1; j < ns; j ++)
      spec_env[j] = (fp)(exp(DB2LOG((double)vtmagn[j])) * lfmagn[j] / lfmagnf0 * f0 / c -> faxis[j]);
    spec_env[0] = spec_env[1];
    o_lipfilter(c -> liprad, c -> fnyq / ns, ns, spec_env, NULL, 0);
    for(int j = 1; j < ns; j ++) spec_env[j] *= (fp)(spec_env[j] * 44100 / 4 / f0);
    for(int j = 0; j < ns; j ++) spec_psd[j] += spec_env[j];
    for(int j = 0; j < c -> order_bap; j ++) {
      int n0 = j * (ns - 1) / c -> order_bap, n1 = (j + 1) * (ns - 1) / c -> order_bap;
      fp apsum = 0;
      for(int k = n0; k < n1; k ++) apsum += 1 - spec_env[k] / spec_psd[k];
      enc[3 + c -> order_spec + j] = apsum / (n1 - n0);
    }
    free(lfmagn); free(spec_env);
  } else for(int j = 0; j < c -> order_bap; j ++) enc[3 + c -> order_spec + j] = 1.0;
  for(int j = 0; j < ns; j ++) spec_psd[j] = (fp)(log((double)spec_psd[j]) * 0.5);
  fp* mel_psd = malloc(sizeof(fp) * ns);
  o_interp1(c -> faxis, spec_psd, ns, c -> melaxis, ns, mel_psd);
  ddct(ns - 1, -1, mel_psd);
  mel_psd[0] *= 0.5;
  ddct(c -> order_spec, 1, mel_psd);
  for(int j = 0; j < c -> order_spec; j ++) enc[3 + j] = (fp)(mel_psd[j] * 2.0 / (ns - 1));
  free(spec_psd); free(mel_psd);
}

/* The aperiodicity per bin as the decoder forms it (coder.c:196-209: band values interpolated over apaxis, low-frequency
 * post-processing on voiced frames) -- the tests read the decoder's conditioning off it: the harmonic part is
 * sqrt(psd (1 - ap)), so 1 / (1 - ap) is the amplification of a rounding error in ap. */
void o_coder_aperiodicity(const o_coder* c, const fp* src, fp* full_ap) {
  int ns = c -> nfullspec / 2 + 1;
  int voicing = src[0] > 0.5;
  const fp* src_bap = src + 3 + c -> order_spec;
  fp* bap_pad = calloc(c -> order_bap + 1, sizeof(fp));
  for(int j = 0; j < c -> order_bap; j ++) bap_pad[j + 1] = src_bap[j];
  bap_pad[0] = voicing ? 0 : 1;
  o_interp1(c -> apaxis, bap_pad, c -> order_bap + 1, c -> faxis, ns, full_ap);
  for(int j = 0; j < ns && voicing; j ++) {
    fp fj = j * c -> fnyq / ns;
    if(fj < 500) full_ap[j] = (fp)1e-3;
    else if(fj < 2000) full_ap[j] = (fp)(1e-3 + (full_ap[j] - 1e-3) * (fj - 500) / 1500);
  }
  free(bap_pad);
}

/* Decoded frame: f0 (0 when unvoiced), rd, nhar, psd[npsd]; layer 1: vtmagn[ns], vsphse[nhar];
 * layer 0: ampl[nhar], phse[nhar].  Arrays must hold maxnhar / ns values. */
void o_coder_decode(const o_coder* c, const fp* src, int use_layer1, fp* f0_out, fp* rd_out, int* nhar_out, fp* psd_out,
  fp* vtmagn, fp* vsphse, fp* ampl_out, fp* phse_out, int maxnhar) {
  int ns = c -> nfullspec / 2 + 1;
  int voicing = src[0] > 0.5;
  fp f0 = src[1] > 20.0 ? src[1] : (fp)20.0;
  fp rd = src[2] < 0.02 ? (fp)0.02 : (src[2] > 3.0 ? (fp)3.0 : src[2]);
  int nhar = voicing ? (int)(c -> fnyq / f0) : 0;
  if(nhar > maxnhar) nhar = maxnhar;
  *f0_out = f0 * voicing; *rd_out = rd; *nhar_out = nhar;
  const fp* src_spec = src + 3; const fp* src_bap = src + 3 + c -> order_spec;
  fp* mel_psd = calloc(ns, sizeof(fp)); fp* bap_pad = calloc(c -> order_bap + 1, sizeof(fp));
  for(int j = 0; j < c -> order_spec; j ++) mel_psd[j] = (fp)(src_spec[j] * 0.5 * (ns - 1) * 2.0 / c -> order_spec);
  ddct(c -> order_spec, -1, mel_psd);
  mel_psd[0] *= 0.5;
  ddct(ns - 1, 1, mel_psd);
  for(int j = 0; j < ns - 1; j ++) mel_psd[j] *= (fp)(2.0 / (ns - 1));
  mel_psd[ns - 1] = mel_psd[ns - 2];
  for(int j = 0; j < c -> order_bap; j ++) bap_pad[j + 1] = src_bap[j];
  bap_pad[0] = voicing ? 0 : 1;
  fp* full_psd = malloc(sizeof(fp) * ns); fp* full_ap = malloc(sizeof(fp) * ns);
  o_interp1(c -> melaxis, mel_psd, ns, c -> faxis, ns, full_psd);
  o_interp1(c -> apaxis, bap_pad, c -> order_bap + 1, c -> faxis, ns, full_ap);
  for(int j = 0; j < ns; j ++) {
    if(voicing) {
      fp fj = j * c -> fnyq / ns;
      if(fj < 500) full_ap[j] = (fp)1e-3;
      else if(fj < 2000) full_ap[j] = (fp)(1e-3 + (full_ap[j] - 1e-3) * (fj - 500) / 1500);
    }
    full_psd[j] = (fp)exp(2.0 * (double)full_psd[j]);
    fp sum_psd = full_psd[j], per_psd = sum_psd * ((fp)1.0 - full_ap[j]);
    full_psd[j] = (fp)sqrt((double)(per_psd * f0 * 4 / 44100));
    full_ap[j] = sum_psd * full_ap[j];
  }
  fp* full_spec = full_psd; fp* full_noise = full_ap;
  o_interp1(c -> faxis, full_noise, ns, c -> psdaxis, c -> npsd, psd_out);
  for(int j = 0; j < c -> npsd; j ++) psd_out[j] = (fp)LOG2IN(log((double)psd_out[j]));
  if(nhar > 0 && use_layer1) {
    o_lfmodel gfm = o_lfmodel_from_rd(rd, (fp)(1.0 / f0), 1.0);
    fp* lfmagn = malloc(sizeof(fp) * ns); fp lfmagnf0 = 0;
    o_lfmodel_spectrum(gfm, c -> faxis, ns, lfmagn, NULL);
    o_lfmodel_spectrum(gfm, & f0, 1, & lfmagnf0, NULL);
    o_lipfilter(c -> liprad, c -> fnyq / ns, ns, full_spec, NULL, 1);
    for(int j = 1; j < ns; j ++)
      full_spec[j] = (fp)LOG2DB(log((double)(full_spec[j] * c -> faxis[j] / f0 * lfmagnf0 / lfmagn[j])));
    full_spec[0] = full_spec[1];
    for(int j = 0; j < ns; j ++) vtmagn[j] = full_spec[j];
    fp* harfreq = malloc(sizeof(fp) * (nhar + 1));
    for(int i = 0; i <= nhar; i ++) harfreq[i] = (fp)((double)(nhar * f0) * i / nhar);
    o_lfmodel_spectrum(gfm, harfreq + 1, nhar, NULL, vsphse);
    free(harfreq); free(lfmagn);
  }
  if(nhar > 0 && ! use_layer1) {
    fp* harfreq = malloc(sizeof(fp) * (nhar + 1));
    for(int i = 0; i <= nhar; i ++) harfreq[i] = (fp)((double)(nhar * f0) * i / nhar);
    fp* ampl = malloc(sizeof(fp) * nhar); fp* vsp = calloc(nhar, sizeof(fp)); fp* lfm = malloc(sizeof(fp) * nhar);
    fp* vtphse = malloc(sizeof(fp) * nhar);
    o_interp1(c -> faxis, full_spec, ns, harfreq + 1, nhar, ampl);
    for(int i = 0; i < nhar; i ++) ampl_out[i] = ampl[i];
    o_lipfilter(c -> liprad, f0, nhar, ampl, NULL, 1);
    o_lfmodel gfm = o_lfmodel_from_rd(rd, (fp)(1.0 / f0), 1.0);
    o_lfmodel_spectrum(gfm, harfreq + 1, nhar, lfm, vsp);
    for(int i = 0; i < nhar; i ++) ampl[i] /= (fp)(lfm[i] / (i + 1.0) / lfm[0]);
    o_harmonic_minphase(ampl, nhar, vtphse);
    o_lipfilter(c -> liprad, f0, nhar, NULL, vtphse, 0);
    for(int i = 0; i < nhar; i ++) phse_out[i] = vtphse[i] + vsp[i];
    free(harfreq); free(ampl); free(vsp); free(lfm); free(vtphse);
  }
  free(mel_psd); free(bap_pad); free(full_spec); free(full_noise);
}
