"""Python restatement of the reference's acceptance oracles
(test/verify-utils.h:8-171): Perez-Cruz nearest-neighbour KL estimator,
waveform-distribution and STFT-magnitude-distribution checks."""
import os
import wave

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_wav(path):
    w = wave.open(path)
    assert w.getnchannels() == 1 and w.getsampwidth() == 2
    x = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
    return x, float(w.getframerate())


def empirical_kld(x, y):                       # verify-utils.h:8-29
    xs = np.sort(np.asarray(x, np.float64)); ys = np.sort(np.asarray(y, np.float64))
    nx, ny = len(xs), len(ys)
    ys[0] = xs[0]; ys[-1] = xs[-1]
    yi = np.minimum(np.searchsorted(ys, xs, side="left"), ny - 1)
    yi = np.maximum(yi, 1)
    xi = np.maximum(np.arange(nx), 1)
    dx = np.maximum(xs[xi] - xs[xi - 1], 1e-10)
    dy = np.maximum(ys[yi] - ys[yi - 1], 1e-10)
    return float(np.sum(np.log(ny * dy / nx / dx)) / nx - 1.0)


def _dither(x, rng):                           # verify-utils.h:71-74
    return x + rng.standard_normal(len(x)) * 1e-4


def data_distribution_klds(x, y, seed=0):      # verify-utils.h:76-110
    rng = np.random.default_rng(seed)
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    out = [empirical_kld(_dither(x, rng), _dither(y, rng))]
    dx, dy = np.diff(x), np.diff(y)
    out.append(empirical_kld(_dither(np.r_[x[0], dx], rng), _dither(np.r_[y[0], dy], rng)))
    ddx, ddy = np.diff(x, 2), np.diff(y, 2)
    out.append(empirical_kld(_dither(np.r_[x[0], ddx, dx[-1]], rng), _dither(np.r_[y[0], ddy, dy[-1]], rng)))
    return out


def stft_mag(x, nhop=512, hop_fc=4):
    nfft = nhop * hop_fc
    nfrm = len(x) // nhop
    w = np.hanning(nfft + 1)[:-1]
    xp = np.concatenate([np.zeros(nfft // 2), np.asarray(x, np.float64), np.zeros(nfft)])
    frames = np.stack([xp[i * nhop: i * nhop + nfft] * w for i in range(nfrm)])
    return np.abs(np.fft.rfft(frames, axis=1)) / (w.sum() / 2)


def spectral_distribution_stats(x, y, seed=0):  # verify-utils.h:121-171
    rng = np.random.default_rng(seed)
    X, Y = stft_mag(x), stft_mag(y)
    m = min(len(X), len(Y))
    cc = float(np.corrcoef(X[:m].ravel(), Y[:m].ravel())[0, 1])
    k0 = empirical_kld(_dither(X.ravel(), rng), _dither(Y.ravel(), rng))
    k1 = empirical_kld(_dither(np.diff(X, axis=0).ravel(), rng), _dither(np.diff(Y, axis=0).ravel(), rng))
    return cc, k0, k1


def assert_reference_acceptance(x, y, tag=""):
    klds = data_distribution_klds(x, y)
    cc, k0, k1 = spectral_distribution_stats(x, y)
    msg = f"{tag}: waveform KLD {klds}, spectral corr {cc:.4f}, spectral KLD {k0:.4f} {k1:.4f}"
    assert all(k < 0.05 for k in klds), msg     # verify-utils.h:88,97,107
    assert cc > 0.95, msg                       # :140
    assert k0 < 0.05 and k1 < 0.05, msg         # :149,167
    return msg
