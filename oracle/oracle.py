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
                ("dsp_core.c", "llsm_oracle.c", "rt_oracle.c", "l1_oracle.c", "oracle.h")]
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
