import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def o64():
    from oracle.oracle import Oracle
    return Oracle(np.float64)


@pytest.fixture(scope="session")
def o32():
    from oracle.oracle import Oracle
    return Oracle(np.float32)


FS = 44100.0


def make_utterance(u, F0, nx=44100, fs=FS, sigma=0.01):
    """Synthetic utterance of BASELINE.md section 3 (deterministic per u)."""
    rng = np.random.default_rng(20260927 + u)
    K = min(int(fs / 2 / F0), 100)
    n = np.arange(nx)
    phi = rng.uniform(-np.pi, np.pi, K)
    x = np.zeros(nx)
    for k in range(1, K + 1):
        x += (0.3 / k) * np.cos(2 * np.pi * k * F0 * n / fs + phi[k - 1])
    x += sigma * rng.standard_normal(nx)
    return x.astype(np.float32)


def make_speechlike(u, nx=30000, fs=FS, thop=0.005):
    """Voiced/unvoiced utterance with a moving F0 track: returns (x, f0[nfrm])."""
    rng = np.random.default_rng(777 + u)
    nfrm = int(nx / fs / thop)
    t = np.arange(nfrm) * thop
    f0 = 140.0 + 50.0 * np.sin(2 * np.pi * 1.3 * t + u) + 15.0 * np.sin(2 * np.pi * 4.1 * t)
    voiced = np.ones(nfrm, bool)
    voiced[: 6] = False
    voiced[nfrm // 2 - 8: nfrm // 2 + 6] = False
    voiced[-5:] = False
    f0 = np.where(voiced, f0, 0.0)
    # sample-rate F0 and voicing
    ts = np.arange(nx) / fs
    f0s = np.interp(ts, t, np.where(voiced, f0, 140.0))
    vs = np.interp(ts, t, voiced.astype(float))
    phase = 2 * np.pi * np.cumsum(f0s) / fs
    x = np.zeros(nx)
    for k in range(1, 40):
        a = 0.25 / k ** 1.2 * (1.0 + 0.3 * np.sin(2 * np.pi * 0.7 * ts + k))
        x += a * np.cos(k * phase + 0.37 * k * k)
    x *= vs
    noise = rng.standard_normal(nx)
    x += (0.004 + 0.03 * (1 - vs)) * noise
    return x.astype(np.float32), f0.astype(np.float32)


def wrap(a):
    return np.angle(np.exp(1j * np.asarray(a, dtype=np.float64)))
