"""-m gpu: the installed per-frame API (include/dsputils.h, include/llsmutils.h) through the C-ABI vs the
float64 oracle's restatement of the same reference functions, on identical inputs."""
import ctypes as C

import numpy as np
import pytest

import libllsm2_amd as llsm
from conftest import FS, make_speechlike, make_utterance, wrap
from gpu_common import rel_rms, report

pytestmark = pytest.mark.gpu
P = llsm.P_fp
PI = llsm.P_int


class LF(C.Structure):
    _fields_ = [("T0", C.c_float), ("te", C.c_float), ("tp", C.c_float), ("ta", C.c_float), ("Ee", C.c_float)]


@pytest.fixture(scope="module")
def L():
    L = llsm.load()
    fp = C.c_float
    L.llsm_harmonic_czt.argtypes = [P, C.c_int, fp, fp, C.c_int, P, P]
    for n in ("llsm_synthesize_harmonic_frame", "llsm_synthesize_harmonic_frame_iczt"):
        getattr(L, n).restype = P; getattr(L, n).argtypes = [P, P, C.c_int, fp, C.c_int]
    L.llsm_estimate_psd.argtypes = [P, C.c_int, C.c_int, P]
    L.llsm_compute_spectrogram.argtypes = [P, C.c_int, PI, PI, C.c_int, C.c_int, C.c_char_p, C.POINTER(P), C.POINTER(P)]
    L.llsm_compute_dc.argtypes = [P, C.c_int, PI, PI, C.c_int, P]
    L.llsm_subband_energy.restype = P; L.llsm_subband_energy.argtypes = [P, C.c_int, fp, fp]
    L.llsm_harmonic_analysis.argtypes = [P, C.c_int, fp, P, C.c_int, fp, fp, C.c_int, C.c_int, PI, C.POINTER(P), C.POINTER(P)]
    L.llsm_refine_f0.argtypes = [P, C.c_int, fp, P, C.c_int, fp]
    L.llsm_generate_white_noise.restype = P; L.llsm_generate_white_noise.argtypes = [C.c_int]
    L.llsm_generate_bandlimited_noise.restype = P; L.llsm_generate_bandlimited_noise.argtypes = [C.c_int, fp, fp]
    L.llsm_harmonic_minphase.restype = P; L.llsm_harmonic_minphase.argtypes = [P, C.c_int]
    for n in ("llsm_harmonic_spectrum", "llsm_harmonic_envelope"):
        getattr(L, n).restype = P; getattr(L, n).argtypes = [P, C.c_int, fp, C.c_int]
    L.llsm_create_cached_glottal_model.restype = C.c_void_p; L.llsm_create_cached_glottal_model.argtypes = [P, C.c_int, C.c_int]
    L.llsm_delete_cached_glottal_model.argtypes = [C.c_void_p]
    L.llsm_spectral_glottal_fitting.restype = fp; L.llsm_spectral_glottal_fitting.argtypes = [P, C.c_int, C.c_void_p]
    L.llsm_lfmodel_from_rd.restype = LF; L.llsm_lfmodel_from_rd.argtypes = [fp, fp, fp]
    L.llsm_lfmodel_spectrum.restype = P; L.llsm_lfmodel_spectrum.argtypes = [LF, P, C.c_int, P]
    L.llsm_make_filtered_pulse.restype = P
    L.llsm_make_filtered_pulse.argtypes = [C.POINTER(llsm.Container), C.POINTER(LF), P, C.c_int, C.c_int, C.c_int, fp, fp, fp]
    L.llsm_harmonic_peakpicking.argtypes = [P, P, C.c_int, fp, C.c_int, fp, P, P]
    L.llsm_lipfilter.argtypes = [fp, fp, C.c_int, P, P, C.c_int]
    L.llsm_smoothing_filter.restype = P; L.llsm_smoothing_filter.argtypes = [P, C.c_int, C.c_int]
    L.llsm_spectrum_from_envelope.restype = P; L.llsm_spectrum_from_envelope.argtypes = [P, P, C.c_int, C.c_int, fp]
    import ctypes.util
    L._free = C.CDLL(ctypes.util.find_library("c")).free
    L._free.argtypes = [C.c_void_p]
    return L


def f32(a):
    return np.ascontiguousarray(a, np.float32)


def take(L, ptr, n):
    a = np.ctypeslib.as_array(ptr, (n,)).copy()
    L._free(C.cast(ptr, C.c_void_p))
    return a


def test_harmonic_czt_and_frame_synthesis(L, o64):
    x = make_utterance(4, 137.0, nx=4000)
    n, f0, nh = 1288, 137.0, 60
    fr = f32(x[1000:1000 + n])
    a = np.zeros(nh, np.float32); p = np.zeros(nh, np.float32)
    L.llsm_harmonic_czt(fr.ctypes.data_as(P), n, f0, FS, nh, a.ctypes.data_as(P), p.ctypes.data_as(P))
    ao, po = o64.harmonic_czt(fr, f0, FS, nh)
    big = ao > 1e-4 * ao.max()
    m = dict(ampl_rel=float((np.abs(a - ao)[big] / ao[big]).max()), phse=float(np.abs(wrap(p - po))[big].max()))
    assert m["ampl_rel"] < 1e-4 and m["phse"] < 1e-3, m
    # both frame synthesisers vs the oracle's recurrent bank (test/test-harmonic.c:32-48 through the product)
    rng = np.random.default_rng(2)
    am, ph = f32(rng.standard_normal(100)), f32(rng.standard_normal(100) * 100)
    yo = o64.synth_frame(am, ph, float(np.float32(0.01)), 1024)       # the C entry point takes f0 as float
    for fn in (L.llsm_synthesize_harmonic_frame, L.llsm_synthesize_harmonic_frame_iczt):
        y = take(L, fn(am.ctypes.data_as(P), ph.ctypes.data_as(P), 100, 0.01, 1024), 1024)
        e = rel_rms(y, yo); m[fn.__name__] = e
        assert e < 1e-5, (fn.__name__, e)
    report("frameapi_czt_synth", m)


def test_spectrogram_psd_dc_subband(L, o64):
    x, _ = make_speechlike(7, nx=12000)
    x = f32(x)
    centers = np.array([0, 700, 3000, 6000, 11990], np.int32); wins = np.array([882, 1470, 2500, 441, 1024], np.int32)
    nfft, ns = 2048, 1025
    for wt, bl in ((b"hanning", 0), (b"blackman", 1)):
        spec = np.zeros((5, ns), np.float32); ph = np.zeros((5, ns), np.float32)
        rows = (P * 5)(*[spec[i].ctypes.data_as(P) for i in range(5)]); prow = (P * 5)(*[ph[i].ctypes.data_as(P) for i in range(5)])
        L.llsm_compute_spectrogram(x.ctypes.data_as(P), len(x), centers.ctypes.data_as(PI), wins.ctypes.data_as(PI), 5, nfft, wt, rows, prow)
        for i in range(5):
            mo, po, ws = o64.stft_frame(x, int(centers[i]), int(wins[i]), nfft, bl)
            w1024 = (o64.blackman(1024) if bl else o64.hanning(1024)).sum()
            mo = mo * (1024.0 / (0.5 * w1024) / wins[i])
            assert np.abs(spec[i] - mo).max() < 2e-5 * mo.max(), (wt, i)
            strong = mo > 1e-3 * mo.max()
            assert np.abs(wrap(ph[i] - po))[strong].max() < 2e-3, (wt, i)
    psd = np.zeros(513, np.float32)
    fr = f32(x[2000:2882])
    L.llsm_estimate_psd(fr.ctypes.data_as(P), 882, 1024, psd.ctypes.data_as(P))
    assert np.abs(psd - o64.estimate_psd(fr, 1024)).max() < 1e-5 * psd.max()
    dc = np.zeros(5, np.float32)
    L.llsm_compute_dc(x.ctypes.data_as(P), len(x), centers.ctypes.data_as(PI), wins.ctypes.data_as(PI), 5, dc.ctypes.data_as(P))
    ref = [o64.fetch_frame(x, int(c), int(w)).mean() for c, w in zip(centers, wins)]
    assert np.abs(dc - ref).max() < 1e-6
    for lo, hi in ((0.0, 2000 / FS), (2000 / FS, 4000 / FS), (8000 / FS, 0.5)):
        y = take(L, L.llsm_subband_energy(x.ctypes.data_as(P), len(x), lo, hi), len(x))
        yo = o64.chebyfilt(x, lo, hi) ** 2
        assert rel_rms(y, yo) < 1e-5, (lo, hi, rel_rms(y, yo))


def test_harmonic_analysis_and_refine(L, o64):
    x, f0 = make_speechlike(9, nx=16000)
    x = f32(x); f0 = f32(f0); nfrm = len(f0)
    for method in (llsm.HMCZT, llsm.HMPP):
        nhar = np.zeros(nfrm, np.int32); pa = (P * nfrm)(); pp = (P * nfrm)()
        L.llsm_harmonic_analysis(x.ctypes.data_as(P), len(x), FS, f0.ctypes.data_as(P), nfrm, 0.005, 4.0, 80, method,
                                 nhar.ctypes.data_as(PI), pa, pp)
        no, ao, po = o64.harmonic_analysis(x, FS, f0, 0.005, 4.0, 80, method)
        assert np.array_equal(nhar, no)
        worst = 0.0
        for i in range(nfrm):
            if f0[i] == 0:
                assert not bool(pa[i]) and not bool(pp[i]); continue
            a = take(L, pa[i], int(nhar[i])); p = take(L, pp[i], int(nhar[i]))
            worst = max(worst, np.abs(a - ao[i, :nhar[i]]).max() / ao[i].max())
        assert worst < (1e-5 if method == llsm.HMCZT else 5e-5), (method, worst)
    xs = f32(make_utterance(21, 200.0, nx=20000, sigma=0.002)); f0r = np.full(90, 203.0, np.float32)
    ref = o64.refine_f0(xs, FS, f0r, 0.005)
    L.llsm_refine_f0(xs.ctypes.data_as(P), len(xs), FS, f0r.ctypes.data_as(P), 90, 0.005)
    assert np.abs(f0r[5:85] - ref[5:85]).max() < 2e-2 and np.abs(f0r[5:85] - 200.0).max() < 0.5


def test_noise_generators(L):
    w = take(L, L.llsm_generate_white_noise(50000), 50000)
    assert abs(w.mean()) < 0.02 and abs(w.std() - 1.0) < 0.02 and np.array_equal(w[:20000], w[20000:40000])
    w2 = take(L, L.llsm_generate_white_noise(1000), 1000)
    assert not np.array_equal(w[:1000], w2)                     # every call advances the seed (libc rand() in the reference)
    nx = 60000
    y = take(L, L.llsm_generate_bandlimited_noise(nx, 2000 / FS, 4000 / FS), nx)
    S = np.abs(np.fft.rfft(y * np.hanning(nx))) ** 2
    f = np.fft.rfftfreq(nx, 1 / FS)
    inband = S[(f > 2300) & (f < 3700)].mean(); out_lo = S[f < 1500].mean(); out_hi = S[f > 5000].mean()
    assert inband > 300 * out_lo and inband > 300 * out_hi and 0.05 < y.std() < 1.0


def test_layer1_frame_helpers(L, o64):
    rng = np.random.default_rng(5)
    nh = 73
    a = f32(np.exp(-np.arange(nh) / 25.0) * (1 + 0.3 * rng.standard_normal(nh)) ** 2 + 1e-3)
    mp = take(L, L.llsm_harmonic_minphase(a.ctypes.data_as(P), nh), nh)
    assert np.abs(wrap(mp - o64.harmonic_minphase(a))).max() < 2e-4
    f0n = 150.0 / FS
    env = take(L, L.llsm_harmonic_envelope(a.ctypes.data_as(P), nh, f0n, 2048), 1025)
    assert np.abs(env - o64.harmonic_envelope(a, f0n, 2048)).max() < 0.002
    _l = o64.lib; from oracle.oracle import _l1_init; _l1_init(o64)
    X = take(L, L.llsm_harmonic_spectrum(a.ctypes.data_as(P), nh, f0n, 2048), 1025)
    Xo = np.zeros(1025); ad = a.astype(np.float64)
    _l.o_harmonic_spectrum(o64.p(ad), C.c_int(nh), o64.f(f0n), C.c_int(2048), o64.p(Xo))
    assert np.abs(X - Xo).max() < 1e-5 * Xo.max()
    # test/test-dsputils.c:135-166 through the product: Rd fit < 0.02 over 500 spectra
    par = f32(np.linspace(0.02, 3.0, 64))
    g = L.llsm_create_cached_glottal_model(par.ctypes.data_as(P), 64, 20)
    assert g
    freq = f32(200.0 * (np.arange(20) + 1)); worst = 0.0
    for k in range(500):
        tp = 0.3 + (2.5 - 0.3) / 500.0 * k
        lf = L.llsm_lfmodel_from_rd(tp, 1 / 200.0, 1.0)
        am = take(L, L.llsm_lfmodel_spectrum(lf, freq.ctypes.data_as(P), 20, None), 20)
        am = f32(am / (np.arange(20) + 1) * rng.uniform(0.01, 5.0))
        worst = max(worst, abs(L.llsm_spectral_glottal_fitting(am.ctypes.data_as(P), 20, g) - tp))
    L.llsm_delete_cached_glottal_model(g)
    report("frameapi_glottal_fit", dict(worst_abs_err=worst))
    assert worst < 0.02


def test_make_filtered_pulse(L, o64):
    from test_gpu_l1 import l1_chunk_from_oracle, q32
    from gpu_common import oracle_analyze
    x, f0 = make_speechlike(1, nx=12000)
    ao = llsm.make_aoptions(f0_refine=0)
    pr, _ = oracle_analyze(o64, ao, FS, x, f0)
    pr = pr.astype(np.float32).astype(np.float64)
    q = q32(o64.chunk_tolayer1(pr, 2048))
    ch = l1_chunk_from_oracle(L, ao, pr, q, FS)
    i = int(np.flatnonzero(f0 > 0)[25]); fi = float(pr.f0[i])
    lf = L.llsm_lfmodel_from_rd(float(q.rd[i]), 1.0 / fi, 1.0)
    srcs = (LF * 3)(lf, lf, lf); offs = f32([0.3, 0.3 + FS / fi, 0.3 + 2 * FS / fi])
    y = take(L, L.llsm_make_filtered_pulse(ch.contents.frames[i], srcs, offs.ctypes.data_as(P), 3, 200, 2048, FS / 2, 1.5, FS), 2048)
    from oracle.oracle import _l1_init
    _l1_init(o64)
    lfo = o64.lfmodel_from_rd(float(q.rd[i]), 1.0 / fi)
    so = (o64.LF * 3)(lfo, lfo, lfo); oo = offs.astype(np.float64); yo = np.zeros(2048)
    n = int(q.nvsphse[i])
    o64.lib.o_make_filtered_pulse(o64.f(float(q.rd[i])), o64.f(fi), o64.p(np.ascontiguousarray(q.vtmagn[i])), C.c_int(q.nspec),
                                  o64.p(np.ascontiguousarray(q.vsphse[i, :n])), C.c_int(n), so, o64.p(oo), C.c_int(3), C.c_int(200),
                                  C.c_int(2048), o64.f(FS / 2), o64.f(1.5), o64.f(FS), o64.p(yo))
    e = rel_rms(y, yo)
    report("frameapi_filtered_pulse", dict(rel_rms=e, rms=float(np.sqrt(np.mean(yo ** 2)))))
    assert e < 1e-5, e
    L.llsm_delete_chunk(ch)


def test_host_helpers_match_oracle(L, o64):
    a = f32([1.0, 0.5, 0.25]); p = f32([0.1, 0.2, 0.3])
    ao, po = o64.lipfilter(1.5, 200.0, a, p, False)
    L.llsm_lipfilter(1.5, 200.0, 3, a.ctypes.data_as(P), p.ctypes.data_as(P), 0)
    assert np.allclose(a, ao, rtol=1e-6) and np.allclose(p, po, atol=1e-6)
    r = f32(np.random.default_rng(0).uniform(0.5, 1.5, 50))
    y = take(L, L.llsm_smoothing_filter(r.ctypes.data_as(P), 50, 4), 50)
    assert np.allclose(y, o64.smoothing_filter(r, 4), atol=1e-6)


def test_frame_compute_snr(L):
    """llsm_frame_compute_snr (frame.c:180-213): NULL on a 2.1 conf (no LLSM_CONF_NOSWARP), and with the deprecated
    member attached the SNR in dB and the aperiodicity 1 / (1 + snr) on npsd warped points, consistent with each other
    and with the pieces it is made of (llsm_harmonic_envelope, llsm_warp_frequency, llsm_spectral_mean)."""
    CONF_NOSWARP = 5
    x, f0 = make_speechlike(17, nx=12000)
    f0 = f0.astype(np.float32)
    ao = llsm.make_aoptions(f0_refine=0)
    L.llsm_analyze.restype = C.POINTER(llsm.Chunk)
    ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f0.ctypes.data_as(llsm.P_fp), len(f0), None)
    assert ch
    i = int(np.flatnonzero(f0 > 0)[len(np.flatnonzero(f0 > 0)) // 2])
    fr, conf = ch.contents.frames[i], ch.contents.conf
    L.llsm_frame_compute_snr.restype = llsm.P_fp
    L.llsm_frame_compute_snr.argtypes = [C.POINTER(llsm.Container), C.POINTER(llsm.Container), C.c_int]
    assert not L.llsm_frame_compute_snr(fr, conf, 0)                     # no NOSWARP on a 2.1 conf
    L.llsm_container_attach_(conf, CONF_NOSWARP, C.cast(L.llsm_create_fp(15000.0), C.c_void_p),
                             C.cast(L.llsm_delete_fp, C.c_void_p), C.cast(L.llsm_copy_fp, C.c_void_p))
    nm = C.cast(L.llsm_container_get(fr, llsm.FRAME_NM), C.POINTER(llsm.NMFrame)).contents
    hm = C.cast(L.llsm_container_get(fr, llsm.FRAME_HM), C.POINTER(llsm.HMFrame)).contents
    npsd = nm.npsd
    p_snr = L.llsm_frame_compute_snr(fr, conf, 0); p_ap = L.llsm_frame_compute_snr(fr, conf, 1)
    assert p_snr and p_ap
    snr = np.ctypeslib.as_array(p_snr, (npsd,)).copy(); ap = np.ctypeslib.as_array(p_ap, (npsd,)).copy()
    psd = np.ctypeslib.as_array(nm.psd, (npsd,)).copy()
    # (above the last harmonic the envelope underflows: snr -> -inf, aperiodicity -> 1, as in the reference)
    ok = np.isfinite(snr)
    assert ok.sum() > npsd // 4 and np.all((ap > 0) & (ap <= 1)) and np.all(ap[~ok] == 1.0)
    assert np.allclose(ap[ok], 1.0 / (1.0 + 10.0 ** (snr[ok] / 10.0)), rtol=2e-4, atol=1e-7)
    # rebuilt from the public pieces
    nfft = max(64, int(2 ** (np.ceil(np.log2(hm.nhar)) + 2)))
    L.llsm_harmonic_envelope.restype = llsm.P_fp
    L.llsm_harmonic_envelope.argtypes = [llsm.P_fp, C.c_int, C.c_float, C.c_int]
    L.llsm_warp_frequency.restype = llsm.P_fp; L.llsm_warp_frequency.argtypes = [C.c_float, C.c_float, C.c_int, C.c_float]
    L.llsm_spectral_mean.restype = llsm.P_fp
    L.llsm_spectral_mean.argtypes = [llsm.P_fp, C.c_int, C.c_float, llsm.P_fp, C.c_int]
    env = L.llsm_harmonic_envelope(hm.ampl, hm.nhar, C.c_float(float(f0[i]) / (FS / 2) / 2.0), nfft)
    e = np.ctypeslib.as_array(env, (nfft // 2 + 1,)).astype(np.float32)
    var = ((10.0 ** (e.astype(np.float64) / 20.0)).astype(np.float32) ** 2 * np.float32(0.5)).astype(np.float32)
    axis = L.llsm_warp_frequency(0.0, FS / 2, npsd, 15000.0)
    mean = L.llsm_spectral_mean(var.ctypes.data_as(llsm.P_fp), nfft // 2 + 1, FS / 2, axis, npsd)
    with np.errstate(divide="ignore"):
        ref = 10.0 * np.log10(np.ctypeslib.as_array(mean, (npsd,)).astype(np.float64)) - psd
    assert np.abs(snr[ok] - ref[ok]).max() < 1e-3, np.abs(snr[ok] - ref[ok]).max()
    L.llsm_delete_chunk(ch)
