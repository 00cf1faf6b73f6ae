"""The reference's own known-answer / acceptance tests, re-stated against the
CPU oracle (the strongest pin available: ciglet is absent, the reference has no
golden vectors -- SURVEY.md section 8c)."""
import os

import numpy as np
import pytest

from verify_utils import GOLDEN, assert_reference_acceptance, empirical_kld, read_wav
from conftest import wrap


def chirp_signal():
    # test/test-dsputils.c:44-74
    nx, fs, thop = 100000, 20000.0, 0.005
    nfrm = int(np.floor(nx / fs / thop))
    center = np.array([int(np.floor(i * thop * fs + 0.5)) for i in range(nfrm)])
    rate = center / nx
    ampl0_truth = rate.copy()
    f0 = (100 + 100 * rate).astype(np.float32)
    i = np.arange(nx)
    f0_inst = 100 + 100.0 * i / nx
    ampl0 = i / nx
    phase = np.cumsum(f0_inst / fs * 2.0 * 3.1415927)
    x = ampl0 * np.sin(phase) + 0.5 * np.sin(2 * phase) + 0.25 * np.sin(3 * phase)
    return x.astype(np.float32), fs, thop, f0, ampl0_truth


@pytest.mark.parametrize("method", [0, 1])
def test_chirp_harmonic_analysis(o64, method):
    """test/test-dsputils.c:44-133: thresholds 0.01 (ampl) / 0.1 rad (phase)."""
    x, fs, thop, f0, truth = chirp_signal()
    nfrm = len(f0)
    nhar, ampl, phse = o64.harmonic_analysis(x, fs, f0, thop, 4.0, 3, method)
    assert np.all(nhar == 3)
    for h, tr in ((0, truth), (1, np.full(nfrm, 0.5)), (2, np.full(nfrm, 0.25))):
        err = np.zeros(nfrm)
        err[5:nfrm - 5] = (ampl[:, h] - tr)[5:nfrm - 5]
        assert abs(err.mean()) < 0.01 and abs(err.std()) < 0.01, (h, err.mean(), err.std())
    perr = np.zeros(nfrm - 1)
    for i in range(5, nfrm - 5):
        inc = f0[i] * 2.0 * 3.1415927 * thop
        perr[i - 1] = wrap(phse[i, 0] - (phse[i - 1, 0] + inc))
    assert abs(perr.mean()) < 0.1 and abs(perr.std()) < 0.1, (perr.mean(), perr.std())


def test_iczt_equals_sinusoid_bank(o64, o32):
    """test/test-harmonic.c:32-48 (prints dB only; we require <= -60 dB)."""
    rng = np.random.default_rng(1)
    ampl = rng.standard_normal(100); phse = rng.standard_normal(100) * 10
    for o, lim in ((o64, -200), (o32, -60)):
        for blu in (False, True):
            y1 = o.synth_frame(ampl, phse, 0.01, 1024, "iczt", bluestein=blu)
            y2 = o.synth_frame(ampl, phse, 0.01, 1024, "bank")
            assert 20 * np.log10(np.std(y1 - y2) / np.std(y2)) < lim


def test_auto_switch_rule(o64):
    """llsmutils.c:45-58 / layer0.c:84-85: ICZT iff log(nx)*0.275 < log(nhar)-2.26."""
    so = o64.soptions(44100.0)
    assert o64.synth_frame_auto_choice(so, 100, 442) == 1
    assert o64.synth_frame_auto_choice(so, 4, 441) == 0
    assert o64.synth_frame_auto_choice(so, 51, 442) == 0
    assert o64.synth_frame_auto_choice(so, 52, 442) == 1
    so.use_iczt = 0
    assert o64.synth_frame_auto_choice(so, 100, 442) == 0


def arctic():
    x, fs = read_wav(os.path.join(GOLDEN, "arctic_a0001.wav"))
    f0 = np.load(os.path.join(GOLDEN, "arctic_a0001_f0_hop128.npy"))
    return x, fs, f0


def test_all_unvoiced_noninteger_hop(o32):
    """test/test-layer0-edgecase.c:10-29: hop 100.5, all unvoiced, y_sin == 0."""
    x, fs, _ = arctic()
    x = x[:40000]
    nhop = 100.5
    nfrm = int(len(x) / nhop)
    f0 = np.zeros(nfrm, np.float32)
    ao = o32.aoptions(thop=nhop / fs)
    pr = o32.analyze(ao, x, fs, f0)
    y, ys, yn = o32.synthesize(o32.soptions(fs), pr, seed=3)
    assert np.all(ys == 0)
    assert np.all(np.isfinite(y)) and np.sqrt(np.mean(yn ** 2)) > 1e-4
    assert np.all(pr.nhar == 0) and np.all(pr.nhar_e == 0)


@pytest.mark.parametrize("method", [0, 1])
def test_config1_anasynth_acceptance(o32, method):
    """test/test-layer0-anasynth.c:29-74 with the committed F0 track."""
    x, fs, f0 = arctic()
    ao = o32.aoptions(thop=128.0 / fs, npsd=128, maxnhar=400, maxnhar_e=5,
                      hm_method=method, f0_refine=0)
    pr = o32.analyze(ao, x, fs, f0, bluestein=True)
    so = o32.soptions(fs)
    y, ys, yn = o32.synthesize(so, pr, seed=11, bluestein=True)
    assert_reference_acceptance(x, y, f"anasynth method={method}")
    o32.phasesync_rps(pr)
    o32.phasepropagate(pr, 1)
    y2, _, _ = o32.synthesize(so, pr, seed=11, bluestein=True)
    assert_reference_acceptance(x, y2, f"anasynth+rps method={method}")


def test_rt_equals_offline(o32):
    """test/test-llsmrt.c:161-164 (harmonic-model path): RT output vs offline."""
    x, fs, f0 = arctic()
    x = x[:60000]; nfrm = 60000 // 128; f0 = f0[:nfrm]
    ao = o32.aoptions(thop=128.0 / fs, npsd=128, maxnhar=400, maxnhar_e=5, f0_refine=0)
    pr = o32.analyze(ao, x, fs, f0, bluestein=True)
    so = o32.soptions(fs)
    y, ys, yn = o32.synthesize(so, pr, seed=5, bluestein=True)
    yp, yap, lat = o32.rt_run(so, pr, capacity=4096, seed=5)
    assert lat == 128 + 256                     # curr_nhop + nfft/2 (nfft 512 at hop 128), llsmrt.c:568-571
    # deterministic part: sample-exact up to float rounding after latency alignment
    n = min(len(yp) - lat, len(ys)) - 600
    d = yp[lat:lat + n] - ys[:n]
    assert np.sqrt(np.mean(d[600:] ** 2)) < 1e-4 * np.sqrt(np.mean(ys[600:n] ** 2)) + 1e-7
    assert_reference_acceptance(y[:n], (yp + yap)[lat:lat + n], "rt vs offline")


def test_empirical_kld_selfcheck():
    """test/verify-utils.h:32-69."""
    rng = np.random.default_rng(0)
    x = rng.normal(1.0, 1.0, 100000)
    k1 = empirical_kld(x, rng.normal(1.0, 1.0, 50000))
    k2 = empirical_kld(x, rng.normal(0.0, 1.0, 50000))
    k3 = empirical_kld(x, rng.normal(1.0, np.sqrt(3.0), 50000))
    assert abs(k1) < 0.02 and k2 > k1 and k3 > k1
    assert abs(k2 - 0.5) < 0.05
