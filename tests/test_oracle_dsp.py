"""Pins every ciglet-contract primitive of the oracle against numpy/scipy
(the oracle is 'parity unpinned' w.r.t. the real reference binary: ciglet is
absent; see oracle/oracle.h)."""
import numpy as np
import pytest
import scipy.signal as ss


def test_fft_matches_numpy(o64, o32):
    rng = np.random.default_rng(1)
    for n in (8, 512, 1024, 2048, 4096):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        re, im = o64.fft(x.real, x.imag)
        assert np.abs(re + 1j * im - np.fft.fft(x)).max() < 1e-10
        re, im = o64.fft(x.real, x.imag, inverse=True)
        assert np.abs(re + 1j * im - np.fft.ifft(x)).max() < 1e-12
        re, im = o32.fft(x.real, x.imag)
        assert np.abs(re + 1j * im - np.fft.fft(x)).max() < 2e-3


def test_windows_symmetric(o64):
    for n in (441, 442, 882, 1470):
        assert np.abs(o64.hanning(n) - ss.get_window("hann", n, fftbins=False)).max() < 1e-14
        assert np.abs(o64.blackman(n) - np.blackman(n)).max() < 1e-14


def test_fetch_frame_zero_pads(o64):
    x = np.arange(1, 11.0)
    f = o64.fetch_frame(x, 2, 8)     # x[2-4 .. 2+3]
    assert np.array_equal(f, [0, 0, 1, 2, 3, 4, 5, 6])
    f = o64.fetch_frame(x, 9, 6)     # x[6..11]
    assert np.array_equal(f, [7, 8, 9, 10, 0, 0])
    f = o64.fetch_frame(x, 4, 5)     # odd size: x[4-2 ..]
    assert np.array_equal(f, [3, 4, 5, 6, 7])


def test_cheby1_matches_scipy(o64):
    for i in range(48):
        wn = (i + 1) * 0.02
        for hp in (0, 1):
            b, a = o64.cheby1(4, 0.5, wn, hp)
            B, A = ss.cheby1(4, 0.5, wn, "high" if hp else "low")
            assert np.abs(b - B).max() < 1e-12 and np.abs(a - A).max() < 1e-12


def test_chebyshev_table_rows(o64):
    # dsputils.c:31-32: 2/4/8 kHz @ 44.1 kHz -> rows 4 / 8 / 17
    for hz, row in ((2000, 4), (4000, 8), (8000, 17)):
        b, a = o64.get_chebyshev_filter(hz / 44100.0, 0)
        B, A = ss.cheby1(4, 0.5, (row + 1) * 0.02, "low")
        assert np.abs(b - B).max() < 1e-12 and np.abs(a - A).max() < 1e-12
    b, a = o64.get_chebyshev_filter(0.5, 1)    # clamps to row 47
    B, A = ss.cheby1(4, 0.5, 0.96, "high")
    assert np.abs(a - A).max() < 1e-12


def test_chebyshev_table_vs_reference_header(o64):
    """Spot check against the reference's own table when the tree is mounted."""
    import os, re
    path = "/root/reference/filter-coef.h"
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted")
    txt = open(path).read()
    tabs = {}
    for name in ("cheby_l_a", "cheby_l_b", "cheby_h_a", "cheby_h_b"):
        m = re.search(name + r"\[240\]\s*=\s*\{([^}]*)\}", txt)
        tabs[name] = np.array([float(v) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]).reshape(48, 5)
    for i in range(48):
        bl, al = o64.cheby1(4, 0.5, (i + 1) * 0.02, 0)
        bh, ah = o64.cheby1(4, 0.5, (i + 1) * 0.02, 1)
        assert np.abs(al - tabs["cheby_l_a"][i]).max() < 2e-9
        assert np.abs(bl - tabs["cheby_l_b"][i]).max() < 2e-9
        assert np.abs(ah - tabs["cheby_h_a"][i]).max() < 2e-9
        assert np.abs(bh - tabs["cheby_h_b"][i]).max() < 2e-9


def test_filtfilt_matches_scipy(o64):
    rng = np.random.default_rng(2)
    x = rng.standard_normal(20128)
    for wn, bt in ((0.1, "low"), (0.18, "high"), (0.36, "low"), (0.36, "high")):
        b, a = ss.cheby1(4, 0.5, wn, bt)
        assert np.abs(o64.filtfilt(b, a, x) - ss.filtfilt(b, a, x)).max() < 1e-9


def test_chebyfilt_bandpass_is_high_then_low(o64):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(4000)
    bh, ah = ss.cheby1(4, 0.5, 0.10, "high")
    bl, al = ss.cheby1(4, 0.5, 0.18, "low")
    ref = ss.filtfilt(bl, al, ss.filtfilt(bh, ah, x))
    assert np.abs(o64.chebyfilt(x, 2000 / 44100.0, 4000 / 44100.0) - ref).max() < 1e-9


def test_czt_direct_and_bluestein(o64):
    rng = np.random.default_rng(4)
    n, w0 = 1470, 2 * np.pi * 120 / 44100
    x = rng.standard_normal(n)
    t = np.arange(n)
    ref = np.array([np.sum(x * np.exp(-1j * w0 * k * t)) for k in range(101)])
    yr, yi = o64.czt(x, w0, 101)
    assert np.abs(yr + 1j * yi - ref).max() < 1e-9
    yr, yi = o64.czt(x, w0, 101, bluestein=True)
    assert np.abs(yr + 1j * yi - ref).max() < 1e-8


def test_interp_and_moving_avg(o64):
    rng = np.random.default_rng(5)
    xi = np.linspace(0, 22050, 256); yi = rng.standard_normal(256)
    xq = np.arange(512) * 22050.0 / 512
    assert np.abs(o64.interp1(xi, yi, xq) - np.interp(xq, xi, yi)).max() < 1e-10
    y513 = rng.standard_normal(513)
    xq = np.linspace(0, 22050, 256)
    assert np.abs(o64.interp1u(0, 22050, y513, xq) - np.interp(xq, np.linspace(0, 22050, 513), y513)).max() < 1e-9
    x = rng.standard_normal(50)
    ref = np.array([x[max(0, i - 3): min(50, i + 4)].mean() for i in range(50)])
    assert np.abs(o64.moving_avg(x, 3) - ref).max() < 1e-12


def test_kalman_smoother_reduces_variance(o64):
    rng = np.random.default_rng(6)
    n = 200
    truth = np.cumsum(rng.standard_normal(n) * 0.05)
    z = truth + rng.standard_normal(n) * 1.28
    y, P, s = o64.kalman(z, np.full(n, 0.0025), np.full(n, np.pi ** 2 / 6))
    assert np.mean((s - truth) ** 2) < np.mean((y - truth) ** 2) < np.mean((z - truth) ** 2)
    assert np.all(P > 0)


def test_spec2env_is_smooth_log_envelope(o64):
    nfft, f0 = 2048, 120 / 44100.0
    k = np.arange(nfft // 2 + 1)
    env_true = np.exp(-k / 300.0)
    S = env_true * (1.05 + np.cos(2 * np.pi * k / (f0 * nfft)))     # harmonic ripple
    env = o64.spec2env(S, nfft, f0)
    # ripple removed: residual roughness much smaller than the input's
    rough_in = np.std(np.diff(np.log(S + 1e-10), 2)[50:900])
    rough_out = np.std(np.diff(env, 2)[50:900])
    assert rough_out < 0.1 * rough_in


def test_rng_is_standard_normal(o64, o32):
    z = o64.rng_normal(1234, 50000)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    assert abs(((z - z.mean()) ** 4).mean() / z.var() ** 2 - 3) < 0.15
    assert np.abs(o32.rng_normal(1234, 100) - z[:100]).max() < 1e-5


def test_wrap(o64):
    for v in (-7.0, -np.pi, 0.0, 3.0, np.pi, 6.28, 100.0):
        w = o64.lib.o_wrap(v)
        assert -np.pi < w <= np.pi + 1e-15
        assert abs(np.angle(np.exp(1j * (w - v)))) < 1e-12


def test_index_plan_benchmark_values(o64):
    L = o64.lib
    # SURVEY section 8 header: derived numbers at 44.1 kHz / 5 ms
    assert L.o_idx_nwin_sin(0.005, 44100.0) == 442
    assert L.o_idx_nwin_env(0.005, 44100.0) == 441
    assert L.o_idx_nwin_filt(0.005, 44100.0) == 441
    assert L.o_idx_nwin_psd(0.005, 44100.0) == 882
    assert L.o_idx_ny(200, 0.005, 44100.0) == 44321
    assert [L.o_idx_hwin(f, 44100.0, 4.0) for f in (80, 120, 200, 400)] == [2206, 1470, 882, 442]
    assert [L.o_idx_nhar(f, 44100.0, 100) for f in (80, 120, 200, 400)] == [100, 100, 100, 55]
    assert L.o_nextpow2(441 * 1.2 + 32) == 1024 and L.o_nextpow2(0.03 * 44100) == 2048
