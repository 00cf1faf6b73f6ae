"""CPU tests of the layer-1 / pulse-by-pulse oracle (oracle/l1_oracle.c) and of the product's host-side
LF model (csrc/lfmodel.h): own definitions of the ciglet LF model checked against numerical integration,
the reference's re-statable known-answer tests for this part (test/test-dsputils.c:135-166 Rd fit < 0.02;
test/test-layer1-anasynth.c:26-56 acceptance on arctic_a0001), and self-consistency of the restated code
(layer 0 -> 1 -> 0 round trip; pulse-by-pulse output ~ harmonic-model output)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import FS, make_speechlike, wrap
from test_host_logic import hooks  # noqa: F401  (fixture: builds tests/_host_hooks.so)
from verify_utils import GOLDEN, assert_reference_acceptance, read_wav, spectral_distribution_stats


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("rd", [0.3, 0.7, 1.0, 1.6, 2.2, 2.7, 2.95])
def test_lf_spectrum_is_the_fourier_transform_of_the_waveform(o64, rd):
    T0 = 1 / 180.0
    lf = o64.lfmodel_from_rd(rd, T0)
    t = np.linspace(0, T0, 400001)
    w = o64.lfmodel_waveform(lf, t)
    assert abs(np.trapezoid(w, t)) < 1e-9                       # zero net flow
    assert abs(w.min() + 1.0) < 0.03                            # E(Te) = -Ee
    f = np.array([0.0, 180.0, 360.0, 900.0, 2500.0, 7000.0, 15000.0])
    m, ph = o64.lfmodel_spectrum(lf, f)
    num = np.array([np.trapezoid(w * np.exp(-2j * np.pi * ff * t), t) for ff in f])
    assert np.abs(m - np.abs(num)).max() < 1e-8 * np.abs(num).max() + 1e-13
    assert np.abs(wrap(ph[1:] - np.angle(num[1:]))).max() < 1e-5


def test_product_lfmodel_header_matches_oracle(o64, hooks):
    hooks.hook_lf_spectrum.argtypes = [C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    f = np.array([0.0, 110.0, 220.0, 1000.0, 5000.0, 21000.0])
    for rd in (0.05, 0.25, 0.9, 1.7, 2.69, 2.71, 3.0):
        for f0 in (80.0, 220.0, 440.0):
            m = np.zeros(len(f)); ph = np.zeros(len(f)); par = np.zeros(5)
            hooks.hook_lf_spectrum(rd, 1 / f0, f.ctypes.data, len(f), m.ctypes.data, ph.ctypes.data, par.ctypes.data)
            lf = o64.lfmodel_from_rd(rd, 1 / f0)
            assert abs(par[0] - lf.te) < 1e-14 and abs(par[1] - lf.tp) < 1e-14 and abs(par[2] - lf.ta) < 1e-14
            mo, pho = o64.lfmodel_spectrum(lf, f)
            assert np.abs(m - mo).max() <= 1e-9 * mo.max(), (rd, f0)
            assert np.abs(wrap(ph[1:] - pho[1:])).max() < 1e-7, (rd, f0)


def test_glottal_fit_kat(o64):
    """test/test-dsputils.c:135-166: 500 LF spectra with Rd in [0.3, 2.5), random gain, 20 harmonics:
    |estimate - truth| < 0.02."""
    nhar, f0 = 20, 200.0
    freq = f0 * (np.arange(nhar) + 1)
    rng = np.random.default_rng(1)
    amps, truth = [], []
    for k in range(500):
        tp = 0.3 + (2.5 - 0.3) / 500.0 * k
        m, _ = o64.lfmodel_spectrum(o64.lfmodel_from_rd(tp, 1 / f0), freq)
        amps.append(m / (np.arange(nhar) + 1) * rng.uniform(0.01, 5.0)); truth.append(tp)
    est = o64.glottal_fit_many(amps, np.linspace(0.02, 3.0, 64), nhar)
    assert np.abs(est - np.array(truth)).max() < 0.02


def test_minphase_against_numpy(o64):
    rng = np.random.default_rng(3)
    nfft = 256
    # a log-magnitude response of a stable minimum-phase system: poles inside the unit circle
    poles = 0.9 * np.exp(1j * np.array([0.3, 1.1, 2.0])); poles = np.r_[poles, poles.conj()]
    a = np.poly(poles).real
    H = 1.0 / np.fft.rfft(a, nfft)
    ph = o64.minphase(np.log(np.abs(H)), nfft)
    assert np.abs(wrap(ph - np.angle(H))).max() < 2e-3


def test_smoothing_and_blank_interpolation(o64):
    x = np.array([0, 0, 1.0, 0, 0, 4.0, 5.0, 0, 0])
    assert np.allclose(o64.interp_in_blank(x), [1, 1, 1, 2, 3, 4, 5, 5, 5])
    assert np.allclose(o64.interp_in_blank(np.zeros(4)), 0)
    r = np.random.default_rng(0).uniform(0.5, 1.5, 50)
    y = o64.smoothing_filter(r, 4)
    assert np.allclose(y[:2], r[:4].mean()) and np.allclose(y[-2:], r[-4:].mean())
    c = np.full(30, 0.8)
    assert np.allclose(o64.smoothing_filter(c, 4), 0.8)
    assert np.array_equal(o64.smoothing_filter(r[:3], 4), r[:3])       # shorter than the order: copied


@pytest.fixture(scope="module")
def speech(o64):
    x, f0 = make_speechlike(1, nx=30000)
    pr = o64.analyze(o64.aoptions(f0_refine=0), x, FS, f0)
    q = o64.chunk_tolayer1(pr, 2048)
    return x, f0, pr, q


def test_layer1_round_trip(o64, speech):
    """layer 0 -> layer 1 -> layer 0 returns the harmonic amplitudes (the envelope passes through them)
    and the phases (vocal-tract minimum phase + source phase)."""
    x, f0, pr, q = speech
    assert np.all(q.rd[f0 > 0] > 0.1) and np.all(q.rd < 3.0)
    assert np.array_equal(q.nvsphse, np.where(f0 > 0, pr.nhar, 0))
    p2 = pr.copy(); p2.nhar[:] = 0; p2.ampl[:] = 0; p2.phse[:] = 0
    q2 = q.copy(); q2.has_hm[:] = 0
    o64.chunk_tolayer0(p2, q2)
    assert np.array_equal(p2.nhar, pr.nhar)
    d, e = [], []
    for i in np.flatnonzero(f0 > 0):
        n = min(pr.nhar[i], 40)
        d.append(20 * np.log10(p2.ampl[i, :n] / pr.ampl[i, :n]))
        e.append(wrap(p2.phse[i, :n] - pr.phse[i, :n]))
    d, e = np.concatenate(d), np.concatenate(e)
    assert abs(d.mean()) < 0.2 and d.std() < 0.6 and np.percentile(np.abs(d), 99) < 2.5, (d.mean(), d.std())
    assert np.percentile(np.abs(e), 99) < 0.15, np.percentile(np.abs(e), 99)


def test_pbp_matches_harmonic_model_on_steady_voicing(o64, speech):
    """The reference builds the pulses so that 'the pulse-by-pulse synthesized speech matches the result from
    harmonic models' (llsmutils.c:70-73): with PBPSYN on everywhere the PbP branch must carry the same
    signal as the layer-0 resynthesis (this pins the sign / origin conventions of our LF model jointly)."""
    x, f0, pr, q = speech
    so = o64.soptions(FS)
    q3 = q.copy(); q3.pbpsyn[:] = 1; q3.has_hm[:] = 0
    y, ys, yn = o64.synthesize_l1(so, pr.copy(), q3, seed=1, debug=True)
    y0, ys0, yn0 = o64.synthesize(so, pr, seed=1)
    pbp, mix = q3.dbg["pbp"], q3.dbg["mix"]
    sl = slice(6000, 12000)                                     # inside the first voiced stretch
    assert mix[sl].min() > 0.999
    c = np.corrcoef(pbp[sl], ys0[sl])[0, 1]
    g = np.sqrt(np.mean(pbp[sl] ** 2) / np.mean(ys0[sl] ** 2))
    assert c > 0.97 and 0.85 < g < 1.2, (c, g)
    assert np.array_equal(yn, yn0)                               # the noise part does not depend on use_l1


def test_pbp_effect_callback_order_and_mixing(o64, speech):
    """Alternating PBPSYN (test-layer1-anasynth.c:34-39 pattern) with HM dropped: cross-fade weights stay in
    [0, 1 + rate], callbacks arrive in frame order, once per pulse."""
    x, f0, pr, q = speech
    q4 = q.copy(); q4.has_hm[:] = 0
    idx = np.arange(pr.nfrm)
    q4.pbpsyn[:] = (idx % 40 > 20).astype(np.int32)
    q4.has_eff[:] = q4.pbpsyn
    calls = []

    def effect(g, frame):
        calls.append(frame)
        g.Rk *= 1.05
        return 0.0
    y, ys, yn = o64.synthesize_l1(o64.soptions(FS), pr.copy(), q4, seed=2, effect=effect, debug=True)
    assert calls == sorted(calls) and len(calls) > 20
    assert set(calls) <= set(np.flatnonzero(q4.pbpsyn).tolist())
    mix = q4.dbg["mix"]
    assert mix.min() > -0.01 and mix.max() < 1.01 and np.any(mix > 0.99) and np.any(mix < 0.01)   # no clamp in layer0.c:242-256
    assert np.all(np.isfinite(ys)) and np.sqrt(np.mean(ys[6000:12000] ** 2)) > 0.05


def test_layer1_anasynth_acceptance_on_arctic(o64):
    """test/test-layer1-anasynth.c:26-56 restated on the oracle: layer 1 from nfft 2048, RPS (layer-1 based),
    HM dropped, PBPSYN on frames with i % 100 > 50, phase propagation, use_l1 synthesis; acceptance vs the
    input (KLD, spectral correlation) and spectral agreement with the layer-0 reconstruction."""
    x, fs = read_wav(os.path.join(GOLDEN, "arctic_a0001.wav"))
    f0 = np.load(os.path.join(GOLDEN, "arctic_a0001_f0_hop128.npy"))
    ao = o64.aoptions(thop=128.0 / fs, f0_refine=0)
    pr = o64.analyze(ao, x, fs, f0)
    so = o64.soptions(fs)
    y0, _, _ = o64.synthesize(so, pr, seed=4)
    q = o64.chunk_tolayer1(pr, 2048)
    o64.l1_phasesync_rps(pr, q, 1)
    q.has_hm[:] = 0
    q.pbpsyn[:] = (np.arange(pr.nfrm) % 100 > 50).astype(np.int32)
    o64.l1_phasepropagate(pr, q, 1)
    y1, ys1, yn1 = o64.synthesize_l1(so, pr, q, seed=4)
    assert_reference_acceptance(x, y1, "oracle layer-1 anasynth vs input")
    cc, k0, k1 = spectral_distribution_stats(y0, y1)
    assert cc > 0.95 and k0 < 0.05 and k1 < 0.05, (cc, k0, k1)
