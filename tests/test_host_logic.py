"""CPU checks of host-side numeric helpers the product ships (cheby.h): the
regenerated Chebyshev table, its steady-state initial conditions, and the
block decomposition the wave-parallel IIR kernel uses -- emulated on the host
by tests/host_hooks.cpp and compared with scipy.signal.filtfilt."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.signal as ss

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def hooks():
    so = os.path.join(HERE, "_host_hooks.so")
    src = os.path.join(HERE, "host_hooks.cpp")
    hdrs = [os.path.join(ROOT, "libllsm2_amd", "csrc", h) for h in ("cheby.h", "lfmodel.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in hdrs]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "libllsm2_amd", "csrc"), "-o", so, src])
    return C.CDLL(so)


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_sections_match_scipy(hooks):
    for row in range(48):
        for hp in (0, 1):
            b = np.zeros(5); a = np.zeros(5); zi = np.zeros(4)
            hooks.hook_section(row, hp, dp(b), dp(a), dp(zi))
            B, A = ss.cheby1(4, 0.5, (row + 1) * 0.02, "high" if hp else "low")
            assert np.abs(b - B).max() < 1e-12 and np.abs(a - A).max() < 1e-12
            assert np.abs(zi - ss.lfilter_zi(B, A)).max() < 1e-9 * max(1, np.abs(ss.lfilter_zi(B, A)).max())


def test_row_selection(hooks):
    hooks.hook_row_of.argtypes = [C.c_float]
    assert [hooks.hook_row_of(f / 44100.0) for f in (2000, 4000, 8000)] == [4, 8, 17]
    assert hooks.hook_row_of(0.5) == 47 and hooks.hook_row_of(0.0001) == 0


def test_block_filtfilt_equals_scipy(hooks):
    rng = np.random.default_rng(0)
    for n in (17, 100, 2047, 2048, 2049, 20128, 44100):
        x = rng.standard_normal(n)
        for row, hp in ((4, 0), (4, 1), (8, 0), (17, 1), (47, 0), (0, 0), (0, 1)):
            y = np.zeros(n)
            hooks.hook_block_filtfilt(row, hp, dp(x), n, dp(y))
            B, A = ss.cheby1(4, 0.5, (row + 1) * 0.02, "high" if hp else "low")
            ref = ss.filtfilt(B, A, x, padlen=min(15, n - 1))
            tol = 1e-7 if row == 0 else 2e-9          # row 0 (Wn = 0.02) is ill-conditioned even in float64
            assert np.abs(y - ref).max() < tol * max(1.0, np.abs(ref).max()), (n, row, hp)


def test_overlap_add_halo_covers_every_overlapping_frame():
    """The fused overlap-add kernels (k_synth_ola, k_noise_filter_ola) start a unit `halo` frames
    before its first frame i0, halo = floor((W + 1) / hop) with W the frame length (engine.cpp).
    Invariant they rely on: no frame further back reaches the first sample of frame i0, i.e.
    center(i0) - center(i0 - k) >= W for every k > halo, with the product's own (float32,
    reference-order) frame centres (plan.h via llsm_gpu_plan_index).  Also: centres never decrease."""
    import libllsm2_amd as llsm
    L = llsm.load()
    L.llsm_gpu_plan_index.restype = C.c_int
    L.llsm_gpu_plan_index.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
    center = lambda i, thop, fs: L.llsm_gpu_plan_index(0, i, 0, 0.0, thop, fs, 4.0)
    for fs, thop in ((8000.0, 0.005), (16000.0, 0.005), (22050.0, 128.0 / 22050.0), (44100.0, 0.005),
                     (44100.0, 0.010), (48000.0, 0.005), (44100.0, 0.0029), (96000.0, 0.005)):
        hop = float(np.float32(thop)) * fs
        nwin_sin = L.llsm_gpu_plan_index(1, 0, 0, 0.0, thop, fs, 4.0)
        nwin_filt = L.llsm_gpu_plan_index(3, 0, 0, 0.0, thop, fs, 4.0)
        nfft = 1
        while nfft < nwin_filt * 1.2 + 32:
            nfft *= 2
        c = np.array([center(i, thop, fs) for i in range(0, 3000)])
        assert np.all(np.diff(c) >= 0), (fs, thop)
        for W in (nwin_sin, nfft):
            halo = int(np.floor((W + 1) / max(hop, 1.0)))
            for k in (halo + 1, halo + 2):
                assert np.all(c[k:] - c[:-k] >= W), (fs, thop, W, halo, k)
