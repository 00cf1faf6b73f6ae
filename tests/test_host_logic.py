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
    hdr = os.path.join(ROOT, "libllsm2_amd", "csrc", "cheby.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
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
