"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module -- and only as the checker / reported CPU baseline.  See oracle.h
for the "parity unpinned" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    """Compile liboracle_f32.so / liboracle_f64.so with gcc (make -C oracle)."""
    need = force or not all(
        os.path.exists(os.path.join(_HERE, f"liboracle_{s}.so")) for s in ("f32", "f64"))
    if not need:
        srcs = [os.path.join(_HERE, f) for f in
                ("dsp_core.c", "llsm_oracle.c", "rt_oracle.c", "l1_oracle.c", "coder_oracle.c", "oracle.h")]
        newest = max(os.path.getmtime(s) for s in srcs)
        need = any(os.path.getmtime(os.path.join(_HERE, f"liboracle_{s}.so")) < newest
                   for s in ("f32", "f64"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])


def _mk_structs(fpt):
    class AOptions(C.Structure):
        _fields_ = [("thop", fpt), ("maxnhar", C.c_int), ("maxnhar_e", C.c_int),
                    ("npsd", C.c_int), ("nchannel", C.c_int), ("chanfreq", fpt * 8),
                    ("lip_radius", fpt), ("f0_refine", C.c_int), ("hm_method", C.c_int),
                    ("rel_winsize", fpt)]

    class SOptions(C.Structure):
        _fields_ = [("fs", fpt), ("use_iczt", C.c_int), ("use_l1", C.c_int),
                    ("iczt_param_a", fpt), ("iczt_param_b", fpt)]

    P = C.POINTER(fpt)
    PI = C.POINTER(C.c_int)

    class CParams(C.Structure):
        _fields_ = [("nfrm", C.c_int), ("maxnhar", C.c_int), ("maxnhar_e", C.c_int),
                    ("npsd", C.c_int), ("nchannel", C.c_int),
                    ("thop", fpt), ("fnyq", fpt), ("chanfreq", fpt * 8),
                    ("f0", P), ("nhar", PI), ("ampl", P), ("phse", P),
                    ("psd", P), ("psdres", P), ("edc", P), ("nhar_e", PI),
                    ("eenv_ampl", P), ("eenv_phse", P)]
    return AOptions, SOptions, CParams


class Params:
    """Flat SoA layer-0 parameter set of one utterance (numpy-owned)."""
    FIELDS = ("f0", "nhar", "ampl", "phse", "psd", "psdres", "edc", "nhar_e",
              "eenv_ampl", "eenv_phse")

    def __init__(self, nfrm, maxnhar, maxnhar_e, npsd, nchannel, thop, fnyq, chanfreq, dtype):
        self.nfrm, self.maxnhar, self.maxnhar_e = nfrm, maxnhar, maxnhar_e
        self.npsd, self.nchannel = npsd, nchannel
        self.thop, self.fnyq, self.chanfreq = thop, fnyq, list(chanfreq)
        self.dtype = np.dtype(dtype)
        d = self.dtype
        self.f0 = np.zeros(nfrm, d)
        self.nhar = np.zeros(nfrm, np.int32)
        self.ampl = np.zeros((nfrm, maxnhar), d)
        self.phse = np.zeros((nfrm, maxnhar), d)
        self.psd = np.full((nfrm, npsd), -120.0, d)
        self.psdres = np.zeros((nfrm, npsd), d)
        self.edc = np.full((nfrm, nchannel), 1e-5, d)
        self.nhar_e = np.zeros(nfrm, np.int32)
        self.eenv_ampl = np.zeros((nfrm, nchannel, maxnhar_e), d)
        self.eenv_phse = np.zeros((nfrm, nchannel, maxnhar_e), d)

    def astype(self, dtype):
        q = Params(self.nfrm, self.maxnhar, self.maxnhar_e, self.npsd, self.nchannel,
                   self.thop, self.fnyq, self.chanfreq, dtype)
        for f in self.FIELDS:
            a = getattr(self, f)
            setattr(q, f, np.ascontiguousarray(a.astype(q.dtype if a.dtype.kind == "f" else a.dtype)))
        return q

    def copy(self):
        return self.astype(self.dtype)


class Oracle:
    def __init__(self, dtype=np.float64):
        build()
        self.dtype = np.dtype(dtype)
        suffix = "f32" if self.dtype == np.float32 else "f64"
        self.fpt = C.c_float if suffix == "f32" else C.c_double
        self.lib = C.CDLL(os.path.join(_HERE, f"liboracle_{suffix}.so"))
        self.AOptions, self.SOptions, self.CParams = _mk_structs(self.fpt)
        L = self.lib
        for n in ("o_idx_center", "o_idx_nwin_sin", "o_idx_nwin_env", "o_idx_nwin_filt",
                  "o_idx_nwin_psd", "o_idx_ny", "o_idx_hwin", "o_idx_nhar", "o_idx_env_ola",
                  "o_idx_dcwin", "o_idx_spgmwin", "o_nextpow2"):
            getattr(L, n).restype = C.c_int
        L.o_idx_center.argtypes = [C.c_int, C.c_float, C.c_float]
        L.o_idx_nwin_sin.argtypes = L.o_idx_nwin_env.argtypes = [C.c_float, C.c_float]
        L.o_idx_nwin_filt.argtypes = L.o_idx_nwin_psd.argtypes = [C.c_float, C.c_float]
        L.o_idx_ny.argtypes = [C.c_int, C.c_float, C.c_float]
        L.o_idx_hwin.argtypes = [C.c_float, C.c_float, C.c_float]
        L.o_idx_nhar.argtypes = [C.c_float, C.c_float, C.c_int]
        L.o_idx_env_ola.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float]
        L.o_idx_dcwin.argtypes = [C.c_float, C.c_float, C.c_float]
        L.o_idx_spgmwin.argtypes = [C.c_float, C.c_float, C.c_int]
        L.o_nextpow2.argtypes = [C.c_double]
        L.o_idx_rawfrac.restype = C.c_float
        L.o_idx_rawfrac.argtypes = [C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int)]
        L.o_wrap.restype = self.fpt
        L.o_wrap.argtypes = [self.fpt]
        L.o_rng_normal.restype = self.fpt
        L.o_rng_normal.argtypes = [C.c_ulonglong, C.c_ulonglong]
        L.o_rt_create.restype = C.c_void_p
        L.o_synthesize.restype = C.c_int

    def set_convention(self, name, value):
        """same names / values as llsm_gpu_set_convention (llsm_gpu.h)"""
        assert self.lib.o_set_convention(name.encode(), C.c_int(int(value))) == 0, name

    # ---- helpers ----
    def arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    def p(self, a):
        return a.ctypes.data_as(C.POINTER(self.fpt)) if a is not None else None

    @staticmethod
    def pi(a):
        return a.ctypes.data_as(C.POINTER(C.c_int))

    def f(self, v):
        return self.fpt(v)

    def cparams(self, pr):
        assert pr.dtype == self.dtype
        c = self.CParams()
        c.nfrm, c.maxnhar, c.maxnhar_e, c.npsd, c.nchannel = (
            pr.nfrm, pr.maxnhar, pr.maxnhar_e, pr.npsd, pr.nchannel)
        c.thop, c.fnyq = pr.thop, pr.fnyq
        for i, v in enumerate(pr.chanfreq):
            c.chanfreq[i] = v
        c.f0, c.nhar = self.p(pr.f0), self.pi(pr.nhar)
        c.ampl, c.phse = self.p(pr.ampl), self.p(pr.phse)
        c.psd, c.psdres = self.p(pr.psd), self.p(pr.psdres)
        c.edc, c.nhar_e = self.p(pr.edc), self.pi(pr.nhar_e)
        c.eenv_ampl, c.eenv_phse = self.p(pr.eenv_ampl), self.p(pr.eenv_phse)
        return c

    def aoptions(self, **kw):
        o = self.AOptions()
        self.lib.o_default_aoptions(C.byref(o))
        for k, v in kw.items():
            if k == "chanfreq":
                for i, x in enumerate(v):
                    o.chanfreq[i] = x
            else:
                setattr(o, k, v)
        return o

    def soptions(self, fs, **kw):
        o = self.SOptions()
        self.lib.o_default_soptions(C.byref(o), self.f(fs))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    # ---- primitives ----
    def fft(self, re, im, inverse=False):
        re, im = self.arr(re).copy(), self.arr(im).copy()
        self.lib.o_fft(self.p(re), self.p(im), C.c_int(len(re)), C.c_int(int(inverse)))
        return re, im

    def hanning(self, n):
        w = np.zeros(n, self.dtype); self.lib.o_hanning(self.p(w), C.c_int(n)); return w

    def blackman(self, n):
        w = np.zeros(n, self.dtype); self.lib.o_blackman(self.p(w), C.c_int(n)); return w

    def fetch_frame(self, x, center, nf):
        x = self.arr(x); out = np.zeros(nf, self.dtype)
        self.lib.o_fetch_frame(self.p(x), C.c_int(len(x)), C.c_int(center), C.c_int(nf), self.p(out))
        return out

    def czt(self, x, omega0, nout, bluestein=False):
        x = self.arr(x); yr = np.zeros(nout, self.dtype); yi = np.zeros(nout, self.dtype)
        self.lib.o_set_czt_mode(C.c_int(int(bluestein)))
        self.lib.o_czt(self.p(x), C.c_int(len(x)), self.f(omega0), C.c_int(nout), self.p(yr), self.p(yi))
        self.lib.o_set_czt_mode(C.c_int(0))
        return yr, yi

    def iczt(self, xr, xi, omega0, n, bluestein=False):
        xr, xi = self.arr(xr), self.arr(xi); y = np.zeros(n, self.dtype)
        self.lib.o_set_czt_mode(C.c_int(int(bluestein)))
        self.lib.o_iczt(self.p(xr), self.p(xi), C.c_int(len(xr)), self.f(omega0), C.c_int(n), self.p(y))
        self.lib.o_set_czt_mode(C.c_int(0))
        return y

    def gensins(self, freq, ampl, phse, fs, n):
        freq, ampl, phse = self.arr(freq), self.arr(ampl), self.arr(phse)
        y = np.zeros(n, self.dtype)
        self.lib.o_gensins(self.p(freq), self.p(ampl), self.p(phse), C.c_int(len(freq)),
                           self.f(fs), C.c_int(n), self.p(y))
        return y

    def interp1(self, xi, yi, xq):
        xi, yi, xq = self.arr(xi), self.arr(yi), self.arr(xq); yq = np.zeros(len(xq), self.dtype)
        self.lib.o_interp1(self.p(xi), self.p(yi), C.c_int(len(xi)), self.p(xq), C.c_int(len(xq)), self.p(yq))
        return yq

    def interp1u(self, x0, x1, yi, xq):
        yi, xq = self.arr(yi), self.arr(xq); yq = np.zeros(len(xq), self.dtype)
        self.lib.o_interp1u(self.f(x0), self.f(x1), self.p(yi), C.c_int(len(yi)), self.p(xq),
                            C.c_int(len(xq)), self.p(yq))
        return yq

    def moving_avg(self, x, h):
        x = self.arr(x); y = np.zeros(len(x), self.dtype)
        self.lib.o_moving_avg(self.p(x), C.c_int(len(x)), C.c_int(h), self.p(y)); return y

    def kalman(self, z, Q, R):
        z, Q, R = self.arr(z), self.arr(Q), self.arr(R); n = len(z)
        P = np.zeros(n, self.dtype); y = np.zeros(n, self.dtype); s = np.zeros(n, self.dtype)
        self.lib.o_kalmanf1d(self.p(z), self.p(Q), self.p(R), C.c_int(n), self.p(P), self.p(y))
        self.lib.o_kalmans1d(self.p(y), self.p(P), self.p(Q), C.c_int(n), self.p(s))
        return y, P, s

    def cheby1(self, order, rp, wn, highpass):
        b = np.zeros(order + 1); a = np.zeros(order + 1)
        self.lib.o_cheby1(C.c_int(order), C.c_double(rp), C.c_double(wn), C.c_int(int(highpass)),
                          b.ctypes.data_as(C.POINTER(C.c_double)), a.ctypes.data_as(C.POINTER(C.c_double)))
        return b, a

    def get_chebyshev_filter(self, cutoff, highpass):
        a = np.zeros(5, self.dtype); b = np.zeros(5, self.dtype)
        self.lib.o_get_chebyshev_filter(self.f(cutoff), C.c_int(int(highpass)), self.p(a), self.p(b))
        return b, a

    def filtfilt(self, b, a, x):
        b, a, x = self.arr(b), self.arr(a), self.arr(x); y = np.zeros(len(x), self.dtype)
        self.lib.o_filtfilt(self.p(b), C.c_int(len(b)), self.p(a), C.c_int(len(a)), self.p(x),
                            C.c_int(len(x)), self.p(y))
        return y

    def chebyfilt(self, x, c1, c2):
        x = self.arr(x); y = np.zeros(len(x), self.dtype)
        self.lib.o_chebyfilt(self.p(x), C.c_int(len(x)), self.f(c1), self.f(c2), self.p(y)); return y

    def spec2env(self, S, nfft, f0):
        S = self.arr(S); env = np.zeros(nfft // 2 + 1, self.dtype)
        self.lib.o_spec2env(self.p(S), C.c_int(nfft), self.f(f0), self.p(env)); return env

    def stft_frame(self, x, center, winsize, nfft, blackman):
        x = self.arr(x); ns = nfft // 2 + 1
        m = np.zeros(ns, self.dtype); ph = np.zeros(ns, self.dtype); ws = self.fpt(0)
        self.lib.o_stft_frame(self.p(x), C.c_int(len(x)), C.c_int(center), C.c_int(winsize),
                              C.c_int(nfft), C.c_int(int(blackman)), self.p(m), self.p(ph), C.byref(ws))
        return m, ph, ws.value

    def rng_normal(self, seed, n, start=0):
        return np.array([self.lib.o_rng_normal(seed, start + i) for i in range(n)], self.dtype)

    # ---- llsm building blocks ----
    def harmonic_czt(self, x, f0, fs, nhar):
        x = self.arr(x); a = np.zeros(nhar, self.dtype); p = np.zeros(nhar, self.dtype)
        self.lib.o_harmonic_czt(self.p(x), C.c_int(len(x)), self.f(f0), self.f(fs), C.c_int(nhar),
                                self.p(a), self.p(p))
        return a, p

    def harmonic_analysis(self, x, fs, f0, thop, rel_winsize, maxnhar, method):
        x, f0 = self.arr(x), self.arr(f0); nfrm = len(f0)
        nhar = np.zeros(nfrm, np.int32)
        a = np.zeros((nfrm, maxnhar), self.dtype); p = np.zeros((nfrm, maxnhar), self.dtype)
        self.lib.o_harmonic_analysis(self.p(x), C.c_int(len(x)), self.f(fs), self.p(f0), C.c_int(nfrm),
                                     self.f(thop), self.f(rel_winsize), C.c_int(maxnhar), C.c_int(method),
                                     C.c_int(maxnhar), self.pi(nhar), self.p(a), self.p(p))
        return nhar, a, p

    def estimate_psd(self, x, nfft):
        x = self.arr(x); psd = np.zeros(nfft // 2 + 1, self.dtype)
        self.lib.o_estimate_psd(self.p(x), C.c_int(len(x)), C.c_int(nfft), self.p(psd)); return psd

    def synth_frame(self, ampl, phse, f0, nx, method="bank", bluestein=False):
        ampl, phse = self.arr(ampl), self.arr(phse); y = np.zeros(nx, self.dtype)
        fn = self.lib.o_synth_harmonic_frame if method == "bank" else self.lib.o_synth_harmonic_frame_iczt
        self.lib.o_set_czt_mode(C.c_int(int(bluestein)))
        fn(self.p(ampl), self.p(phse), C.c_int(len(ampl)), self.f(f0), C.c_int(nx), self.p(y))
        self.lib.o_set_czt_mode(C.c_int(0))
        return y

    def synth_frame_auto_choice(self, sopt, nhar, nx):
        a = np.zeros(max(nhar, 1), self.dtype); y = np.zeros(nx, self.dtype)
        return self.lib.o_synth_harmonic_frame_auto(C.byref(sopt), self.p(a), self.p(a), C.c_int(nhar),
                                                    self.f(0.01), C.c_int(nx), self.p(y))

    def bandlimited_noise(self, nx, fmin, fmax, seed, white=None):
        y = np.zeros(nx, self.dtype)
        w = self.arr(white) if white is not None else None
        self.lib.o_generate_bandlimited_noise(C.c_int(nx), self.f(fmin), self.f(fmax),
                                              C.c_ulonglong(seed), self.p(w), self.p(y))
        return y

    def refine_f0(self, x, fs, f0, thop):
        x = self.arr(x); f0 = self.arr(f0).copy()
        self.lib.o_refine_f0(self.p(x), C.c_int(len(x)), self.f(fs), self.p(f0), C.c_int(len(f0)), self.f(thop))
        return f0

    # ---- entry points ----
    def analyze(self, aopt, x, fs, f0, want_res=False, bluestein=False):
        x = self.arr(x); f0 = self.arr(f0).copy(); nfrm = len(f0)
        pr = Params(nfrm, aopt.maxnhar, aopt.maxnhar_e, aopt.npsd, aopt.nchannel,
                    float(aopt.thop), fs / 2.0, [aopt.chanfreq[i] for i in range(aopt.nchannel - 1)],
                    self.dtype)
        cp = self.cparams(pr)
        res = np.zeros(len(x), self.dtype) if want_res else None
        self.lib.o_set_czt_mode(C.c_int(int(bluestein)))
        self.lib.o_analyze(C.byref(aopt), self.p(x), C.c_int(len(x)), self.f(fs), self.p(f0),
                           C.c_int(nfrm), C.byref(cp), self.p(res))
        self.lib.o_set_czt_mode(C.c_int(0))
        pr.f0_refined = f0
        return (pr, res) if want_res else pr

    def ny(self, nfrm, thop, fs):
        return self.lib.o_idx_ny(nfrm, thop, fs)

    def synthesize(self, sopt, pr, seed=0, white=None, bluestein=False):
        cp = self.cparams(pr)
        ny = self.ny(pr.nfrm, pr.thop, float(sopt.fs))
        y = np.zeros(ny, self.dtype); ys = np.zeros(ny, self.dtype); yn = np.zeros(ny, self.dtype)
        w = self.arr(white) if white is not None else None
        self.lib.o_set_czt_mode(C.c_int(int(bluestein)))
        self.lib.o_synthesize(C.byref(sopt), C.byref(cp), C.c_ulonglong(seed), self.p(w),
                              self.p(y), self.p(ys), self.p(yn))
        self.lib.o_set_czt_mode(C.c_int(0))
        return y, ys, yn

    def phasepropagate(self, pr, sign):
        cp = self.cparams(pr); self.lib.o_chunk_phasepropagate(C.byref(cp), C.c_int(sign))

    def phasesync_rps(self, pr):
        cp = self.cparams(pr); self.lib.o_chunk_phasesync_rps(C.byref(cp))

    # ---- llsmrt ----
    def rt_run(self, sopt, pr, capacity=4096, seed=0):
        """Feed every frame, drain after each feed (single-thread pattern of
        test/test-llsmrt.c:129-145); returns (y_p, y_ap, latency)."""
        cp = self.cparams(pr)
        h = C.c_void_p(self.lib.o_rt_create(C.byref(sopt), C.byref(cp), C.c_int(capacity), C.c_ulonglong(seed)))
        lat = self.lib.o_rt_latency(h)
        yp, yap = [], []
        a, b = self.fpt(0), self.fpt(0)
        for i in range(pr.nfrm):
            self.lib.o_rt_feed(h, C.byref(cp), C.c_int(i))
            while self.lib.o_rt_fetch(h, C.byref(a), C.byref(b)):
                yp.append(a.value); yap.append(b.value)
        self.lib.o_rt_delete(h)
        return np.array(yp, self.dtype), np.array(yap, self.dtype), lat


# ---------------------------------------------------------------- layer 1 / PbP (l1_oracle.c)
class L1Params:
    """Flat layer-1 members of one utterance (numpy-owned), beside Params."""

    def __init__(self, nfrm, nspec, maxnhar, lip_radius, dtype):
        d = np.dtype(dtype)
        self.nfrm, self.nspec, self.maxnhar, self.lip_radius, self.dtype = nfrm, nspec, maxnhar, lip_radius, d
        self.rd = np.zeros(nfrm, d)
        self.vtmagn = np.zeros((nfrm, nspec), d)
        self.vsphse = np.zeros((nfrm, maxnhar), d)
        self.nvsphse = np.zeros(nfrm, np.int32)
        self.has_l1 = np.zeros(nfrm, np.int32)
        self.has_hm = np.ones(nfrm, np.int32)
        self.pbpsyn = np.zeros(nfrm, np.int32)
        self.has_eff = np.zeros(nfrm, np.int32)
        self.dbg = None

    def copy(self):
        q = L1Params(self.nfrm, self.nspec, self.maxnhar, self.lip_radius, self.dtype)
        for f in ("rd", "vtmagn", "vsphse", "nvsphse", "has_l1", "has_hm", "pbpsyn", "has_eff"):
            setattr(q, f, getattr(self, f).copy())
        return q


def _l1_struct(fpt):
    P, PI = C.POINTER(fpt), C.POINTER(C.c_int)

    class CL1(C.Structure):
        _fields_ = [("nfrm", C.c_int), ("nspec", C.c_int), ("maxnhar", C.c_int), ("lip_radius", fpt),
                    ("rd", P), ("vtmagn", P), ("vsphse", P), ("nvsphse", PI), ("has_l1", PI), ("has_hm", PI),
                    ("pbpsyn", PI), ("has_eff", PI), ("dbg_y_hm", P), ("dbg_y_pbp", P), ("dbg_y_mix", P)]

    class LF(C.Structure):
        _fields_ = [("T0", fpt), ("te", fpt), ("tp", fpt), ("ta", fpt), ("Ee", fpt)]

    class GFM(C.Structure):
        _fields_ = [("Fa", fpt), ("Rk", fpt), ("Rg", fpt), ("T0", fpt), ("Ee", fpt)]
    return CL1, LF, GFM


def _l1_init(self):
    if hasattr(self, "CL1"):
        return
    self.CL1, self.LF, self.GFM = _l1_struct(self.fpt)
    L = self.lib
    L.o_lfmodel_from_rd.restype = self.LF
    L.o_lfmodel_from_rd.argtypes = [self.fpt, self.fpt, self.fpt]
    L.o_lfmodel_spectrum.argtypes = [self.LF, C.POINTER(self.fpt), C.c_int, C.POINTER(self.fpt), C.POINTER(self.fpt)]
    L.o_lfmodel_waveform.argtypes = [self.LF, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
    L.o_glottal_create.restype = C.c_void_p
    L.o_glottal_fit.restype = self.fpt
    L.o_glottal_fit.argtypes = [C.POINTER(self.fpt), C.c_int, C.c_void_p]
    L.o_glottal_delete.argtypes = [C.c_void_p]
    L.o_pulse_projection.restype = self.fpt
    L.o_pulse_projection.argtypes = [self.fpt] * 5
    L.o_minphase_fftsize.restype = C.c_int
    self.FGFM = C.CFUNCTYPE(None, C.POINTER(self.GFM), C.POINTER(self.fpt), C.c_void_p, C.c_int)


def _cl1(self, q, ny=0, debug=False):
    _l1_init(self)
    c = self.CL1()
    c.nfrm, c.nspec, c.maxnhar, c.lip_radius = q.nfrm, q.nspec, q.maxnhar, q.lip_radius
    c.rd, c.vtmagn, c.vsphse = self.p(q.rd), self.p(q.vtmagn), self.p(q.vsphse)
    c.nvsphse, c.has_l1, c.has_hm = self.pi(q.nvsphse), self.pi(q.has_l1), self.pi(q.has_hm)
    c.pbpsyn, c.has_eff = self.pi(q.pbpsyn), self.pi(q.has_eff)
    if debug:
        q.dbg = {k: np.zeros(ny, self.dtype) for k in ("hm", "pbp", "mix")}
        c.dbg_y_hm, c.dbg_y_pbp, c.dbg_y_mix = self.p(q.dbg["hm"]), self.p(q.dbg["pbp"]), self.p(q.dbg["mix"])
    return c


def _lfmodel_from_rd(self, rd, T0, Ee=1.0):
    _l1_init(self)
    return self.lib.o_lfmodel_from_rd(self.f(rd), self.f(T0), self.f(Ee))


def _lfmodel_spectrum(self, lf, freq):
    _l1_init(self)
    freq = self.arr(freq); m = np.zeros(len(freq), self.dtype); ph = np.zeros(len(freq), self.dtype)
    self.lib.o_lfmodel_spectrum(lf, self.p(freq), C.c_int(len(freq)), self.p(m), self.p(ph))
    return m, ph


def _lfmodel_waveform(self, lf, t):
    _l1_init(self)
    t = np.ascontiguousarray(t, np.float64); out = np.zeros(len(t))
    PD = C.POINTER(C.c_double)
    self.lib.o_lfmodel_waveform(lf, t.ctypes.data_as(PD), C.c_int(len(t)), out.ctypes.data_as(PD))
    return out


def _glottal_fit_many(self, ampls, param, nhar_cache):
    _l1_init(self)
    param = self.arr(param)
    g = C.c_void_p(self.lib.o_glottal_create(self.p(param), C.c_int(len(param)), C.c_int(nhar_cache)))
    out = []
    for a in ampls:
        a = self.arr(a)
        out.append(self.lib.o_glottal_fit(self.p(a), C.c_int(len(a)), g))
    self.lib.o_glottal_delete(g)
    return np.array(out)


def _vec_fn(name, nout_of):
    def f(self, a, *args):
        _l1_init(self)
        a = self.arr(a); out = np.zeros(nout_of(len(a), *args), self.dtype)
        getattr(self.lib, name)(self.p(a), C.c_int(len(a)), *[C.c_int(x) if isinstance(x, int) else self.f(x) for x in args], self.p(out))
        return out
    return f


def _harmonic_minphase(self, ampl):
    _l1_init(self)
    a = self.arr(ampl); out = np.zeros(len(a), self.dtype)
    self.lib.o_harmonic_minphase(self.p(a), C.c_int(len(a)), self.p(out)); return out


def _harmonic_envelope(self, ampl, f0n, nfft):
    _l1_init(self)
    a = self.arr(ampl); out = np.zeros(nfft // 2 + 1, self.dtype)
    self.lib.o_harmonic_envelope(self.p(a), C.c_int(len(a)), self.f(f0n), C.c_int(nfft), self.p(out)); return out


def _minphase(self, logmag, nfft):
    _l1_init(self)
    a = self.arr(logmag); out = np.zeros(nfft // 2 + 1, self.dtype)
    self.lib.o_minphase(self.p(a), C.c_int(nfft), self.p(out)); return out


def _lipfilter(self, radius, f0, ampl, phse, inverse):
    _l1_init(self)
    a, ph = self.arr(ampl).copy(), self.arr(phse).copy()
    self.lib.o_lipfilter(self.f(radius), self.f(f0), C.c_int(len(a)), self.p(a), self.p(ph), C.c_int(int(inverse)))
    return a, ph


def _smoothing_filter(self, x, order):
    _l1_init(self)
    x = self.arr(x); y = np.zeros(len(x), self.dtype)
    self.lib.o_smoothing_filter(self.p(x), C.c_int(len(x)), C.c_int(order), self.p(y)); return y


def _interp_in_blank(self, x, blank=0.0):
    _l1_init(self)
    x = self.arr(x); y = np.zeros(len(x), self.dtype)
    self.lib.o_interp_in_blank(self.p(x), C.c_int(len(x)), self.f(blank), self.p(y)); return y


def _chunk_tolayer1(self, pr, nfft, lip_radius=1.5):
    """llsm_chunk_tolayer1 (layer1.c:129-149) -> L1Params"""
    _l1_init(self)
    q = L1Params(pr.nfrm, nfft // 2 + 1, pr.maxnhar, lip_radius, self.dtype)
    cp, cq = self.cparams(pr), _cl1(self, q)
    self.lib.o_chunk_tolayer1(C.byref(cp), C.byref(cq), C.c_int(nfft))
    return q


def _chunk_tolayer0(self, pr, q, maxnhar_conf=-1):
    _l1_init(self)
    cp, cq = self.cparams(pr), _cl1(self, q)
    self.lib.o_chunk_tolayer0(C.byref(cp), C.byref(cq), C.c_int(maxnhar_conf))


def _l1_phasepropagate(self, pr, q, sign):
    """llsm_chunk_phasepropagate on HM, eenv and VSPHSE (layer0.c:694-706, frame.c:152-166)"""
    _l1_init(self)
    self.phasepropagate(pr, sign)
    cq = _cl1(self, q)
    delta = np.cumsum(pr.f0.astype(self.dtype)) * self.dtype.type(pr.thop) * sign * 2.0 * np.pi
    for i in range(pr.nfrm):
        self.lib.o_l1_phaseshift(C.byref(cq), C.c_int(i), self.f(float(delta[i])))


def _l1_phasesync_rps(self, pr, q, layer1_based):
    """llsm_chunk_phasesync_rps (frame.c:168-178): shift every member by -reference phase"""
    _l1_init(self)
    cq = _cl1(self, q)
    me = max(pr.maxnhar_e, 1)
    for i in range(pr.nfrm):
        ref = 0.0
        if layer1_based and q.has_l1[i] and q.nvsphse[i] > 0:
            ref = float(q.vsphse[i, 0])
        elif q.has_hm[i] and pr.nhar[i] > 0:
            ref = float(pr.phse[i, 0])
        th = -ref
        k = np.arange(1, pr.maxnhar + 1)
        if q.has_hm[i]:
            n = pr.nhar[i]
            pr.phse[i, :n] = [self.lib.o_wrap(self.f(float(v))) for v in pr.phse[i, :n] + th * k[:n]]
        for c in range(pr.nchannel):
            n = pr.nhar_e[i]
            if n:
                pr.eenv_phse[i, c, :n] = [self.lib.o_wrap(self.f(float(v))) for v in pr.eenv_phse[i, c, :n] + th * k[:n]]
        self.lib.o_l1_phaseshift(C.byref(cq), C.c_int(i), self.f(th))


def _synthesize_l1(self, sopt, pr, q, seed=0, white=None, maxnhar_conf=-1, effect=None, debug=False):
    """llsm_synthesize with use_l1 = 1 (layer0.c:636-664, 148-287).  effect(gfm, frame) -> delta_t mutates gfm."""
    _l1_init(self)
    cp = self.cparams(pr)
    ny = self.ny(pr.nfrm, pr.thop, float(sopt.fs))
    cq = _cl1(self, q, ny, debug)
    y = np.zeros(ny, self.dtype); ys = np.zeros(ny, self.dtype); yn = np.zeros(ny, self.dtype)
    w = self.arr(white) if white is not None else None
    if effect is not None:
        def tramp(g, dt, info, frame):
            dt[0] = effect(g.contents, frame)
        cb = self.FGFM(tramp)
    else:
        cb = C.cast(None, self.FGFM)
    self.lib.o_synthesize_l1.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, self.FGFM, C.c_void_p,
                                         C.c_ulonglong, C.POINTER(self.fpt), C.POINTER(self.fpt), C.POINTER(self.fpt),
                                         C.POINTER(self.fpt)]
    self.lib.o_synthesize_l1(C.cast(C.byref(sopt), C.c_void_p), C.cast(C.byref(cp), C.c_void_p), C.cast(C.byref(cq), C.c_void_p),
                             maxnhar_conf, cb, None, seed, self.p(w), self.p(y), self.p(ys), self.p(yn))
    return y, ys, yn


for _n, _f in dict(lfmodel_from_rd=_lfmodel_from_rd, lfmodel_spectrum=_lfmodel_spectrum, lfmodel_waveform=_lfmodel_waveform,
                   glottal_fit_many=_glottal_fit_many, harmonic_minphase=_harmonic_minphase,
                   harmonic_envelope=_harmonic_envelope, minphase=_minphase, lipfilter=_lipfilter,
                   smoothing_filter=_smoothing_filter, interp_in_blank=_interp_in_blank,
                   chunk_tolayer1=_chunk_tolayer1, chunk_tolayer0=_chunk_tolayer0,
                   l1_phasepropagate=_l1_phasepropagate, l1_phasesync_rps=_l1_phasesync_rps,
                   synthesize_l1=_synthesize_l1).items():
    setattr(Oracle, _n, _f)


def _rt_run_l1(self, sopt, pr, q, capacity=4096, seed=0, maxnhar_conf=-1, effect=None):
    """llsmrt with options.use_l1 = 1: feed every frame (llsmrt.c:295-420 path), drain after each feed;
    returns (y_p, y_ap, latency).  pr / q are modified like the reference modifies the frames (HM rebuilt)."""
    _l1_init(self)
    cp = self.cparams(pr); cq = _cl1(self, q)
    h = C.c_void_p(self.lib.o_rt_create(C.byref(sopt), C.byref(cp), C.c_int(capacity), C.c_ulonglong(seed)))
    lat = self.lib.o_rt_latency(h)
    if effect is not None:
        def tramp(g, dt, info, frame):
            dt[0] = effect(g.contents, frame)
        cb = self.FGFM(tramp)
    else:
        cb = C.cast(None, self.FGFM)
    self.lib.o_rt_feed_l1.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, self.FGFM, C.c_void_p]
    yp, yap = [], []
    a, b = self.fpt(0), self.fpt(0)
    for i in range(pr.nfrm):
        self.lib.o_rt_feed_l1(h, C.cast(C.byref(cp), C.c_void_p), C.cast(C.byref(cq), C.c_void_p), i, maxnhar_conf, cb, None)
        while self.lib.o_rt_fetch(h, C.byref(a), C.byref(b)):
            yp.append(a.value); yap.append(b.value)
    self.lib.o_rt_delete(h)
    return np.array(yp, self.dtype), np.array(yap, self.dtype), lat


Oracle.rt_run_l1 = _rt_run_l1


# ---------------------------------------------------------------- frame coder (coder_oracle.c)
def _coder_encode_chunk(self, pr, q, order_spec, order_bap):
    """llsm_coder_encode on every frame (coder.c:88-168) -> [nfrm][order_spec + order_bap + 3]"""
    _l1_init(self)
    L = self.lib
    L.o_coder_create.restype = C.c_void_p
    L.o_coder_create.argtypes = [self.fpt, C.c_int, C.c_int, C.c_int, C.c_int, self.fpt, C.c_int, C.c_int]
    c = C.c_void_p(L.o_coder_create(pr.fnyq, pr.nchannel, pr.maxnhar_e, pr.npsd, q.nspec, q.lip_radius, order_spec, order_bap))
    dim = order_spec + order_bap + 3
    enc = np.zeros((pr.nfrm, dim), self.dtype)
    L.o_coder_encode.argtypes = [C.c_void_p, self.fpt, self.fpt, C.POINTER(self.fpt), C.POINTER(self.fpt), C.POINTER(self.fpt)]
    for i in range(pr.nfrm):
        row = np.zeros(dim, self.dtype)
        L.o_coder_encode(c, float(pr.f0[i]), float(q.rd[i]), self.p(np.ascontiguousarray(pr.psd[i])),
                         self.p(np.ascontiguousarray(q.vtmagn[i])), self.p(row))
        enc[i] = row
    L.o_coder_delete.argtypes = [C.c_void_p]
    L.o_coder_delete(c)
    return enc


def _coder_decode_chunk(self, enc, use_layer1, pr_like, nspec, lip_radius, order_spec, order_bap, maxnhar):
    """llsm_coder_decode_layer{0,1} on every vector (coder.c:170-292) -> (Params, L1Params)"""
    _l1_init(self)
    L = self.lib
    L.o_coder_create.restype = C.c_void_p
    L.o_coder_create.argtypes = [self.fpt, C.c_int, C.c_int, C.c_int, C.c_int, self.fpt, C.c_int, C.c_int]
    c = C.c_void_p(L.o_coder_create(pr_like.fnyq, pr_like.nchannel, pr_like.maxnhar_e, pr_like.npsd, nspec, lip_radius, order_spec, order_bap))
    nfrm = len(enc)
    pr = Params(nfrm, maxnhar, pr_like.maxnhar_e, pr_like.npsd, pr_like.nchannel, pr_like.thop, pr_like.fnyq, pr_like.chanfreq, self.dtype)
    q = L1Params(nfrm, nspec, maxnhar, lip_radius, self.dtype)
    P = C.POINTER(self.fpt)
    L.o_coder_decode.argtypes = [C.c_void_p, P, C.c_int, P, P, C.POINTER(C.c_int), P, P, P, P, P, C.c_int]
    for i in range(nfrm):
        f0 = self.fpt(0); rd = self.fpt(0); nh = C.c_int(0)
        row = np.ascontiguousarray(enc[i], self.dtype)
        psd = np.zeros(pr.npsd, self.dtype); vt = np.zeros(nspec, self.dtype); vs = np.zeros(maxnhar, self.dtype)
        a = np.zeros(maxnhar, self.dtype); ph = np.zeros(maxnhar, self.dtype)
        L.o_coder_decode(c, self.p(row), int(use_layer1), C.byref(f0), C.byref(rd), C.byref(nh), self.p(psd), self.p(vt), self.p(vs),
                         self.p(a), self.p(ph), maxnhar)
        pr.f0[i] = f0.value; q.rd[i] = rd.value; pr.psd[i] = psd; pr.psdres[i] = 0
        n = nh.value
        if use_layer1:
            pr.nhar[i] = 0; q.has_hm[i] = 0 if n > 0 else 1
            if n > 0:
                q.vtmagn[i] = vt; q.vsphse[i, :n] = vs[:n]; q.nvsphse[i] = n; q.has_l1[i] = 1
        else:
            pr.nhar[i] = n; pr.ampl[i, :n] = a[:n]; pr.phse[i, :n] = ph[:n]
        pr.nhar_e[i] = pr_like.maxnhar_e if f0.value > 0 else 0    # llsm_create_frame: eenv of nhar_e zero harmonics
    L.o_coder_delete.argtypes = [C.c_void_p]
    L.o_coder_delete(c)
    return pr, q


Oracle.coder_encode_chunk = _coder_encode_chunk
def _coder_aperiodicity_chunk(self, enc, pr_like, nspec, lip_radius, order_spec, order_bap):
    """[nfrm][nspec]: the aperiodicity per bin as the decoder forms it (o_coder_aperiodicity)."""
    L = self.lib
    L.o_coder_create.restype = C.c_void_p
    L.o_coder_create.argtypes = [self.fpt, C.c_int, C.c_int, C.c_int, C.c_int, self.fpt, C.c_int, C.c_int]
    c = C.c_void_p(L.o_coder_create(pr_like.fnyq, pr_like.nchannel, pr_like.maxnhar_e, pr_like.npsd, nspec, lip_radius, order_spec, order_bap))
    P = C.POINTER(self.fpt)
    L.o_coder_aperiodicity.argtypes = [C.c_void_p, P, P]
    enc = np.ascontiguousarray(enc, self.dtype)
    out = np.zeros((enc.shape[0], nspec), self.dtype)
    for i in range(enc.shape[0]):
        row = np.ascontiguousarray(enc[i]); ap = np.zeros(nspec, self.dtype)
        L.o_coder_aperiodicity(c, self.p(row), self.p(ap)); out[i] = ap
    L.o_coder_delete.argtypes = [C.c_void_p]
    L.o_coder_delete(c)
    return out


Oracle.coder_decode_chunk = _coder_decode_chunk
Oracle.coder_aperiodicity_chunk = _coder_aperiodicity_chunk
