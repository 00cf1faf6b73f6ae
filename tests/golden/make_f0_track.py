"""Generates arctic_a0001_f0_hop128.npy: a substitute F0 track for the
reference's config-1 fixture (test/arctic_a0001.wav, CMU ARCTIC, copied here as
a data fixture).  The reference's tests obtain F0 from libpyin (hop 128,
50-500 Hz; test/test-layer0-anasynth.c:21-27), which is unavailable; this is
our own YIN-style estimator.  F0 is an INPUT of the path under test, so any
plausible track serves; it is committed so every run sees the same one.

Run:  python tests/golden/make_f0_track.py
"""
import os
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def read_wav(path):
    w = wave.open(path)
    assert w.getnchannels() == 1 and w.getsampwidth() == 2
    x = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float64) / 32768.0
    return x, w.getframerate()


def yin_track(x, fs, nhop=128, fmin=50.0, fmax=500.0, thr=0.15):
    nfrm = len(x) // nhop
    lmin, lmax = int(fs / fmax), int(fs / fmin)
    W = lmax + 200
    xp = np.concatenate([np.zeros(W), x, np.zeros(2 * W)])
    f0 = np.zeros(nfrm)
    rms_all = np.sqrt(np.mean(x ** 2))
    for i in range(nfrm):
        c = i * nhop + W
        seg = xp[c - W // 2: c - W // 2 + W + lmax]
        a = seg[:W]
        if np.sqrt(np.mean(a ** 2)) < 0.05 * rms_all:
            continue
        d = np.array([np.sum((a - seg[l: l + W]) ** 2) for l in range(lmax + 1)])
        cm = np.ones(lmax + 1)
        cs = np.cumsum(d[1:])
        cm[1:] = d[1:] * np.arange(1, lmax + 1) / np.maximum(cs, 1e-12)
        cand = np.where(cm[lmin:lmax] < thr)[0]
        if len(cand) == 0:
            continue
        l = cand[0] + lmin
        while l + 1 < lmax and cm[l + 1] < cm[l]:
            l += 1
        y0, y1, y2 = cm[l - 1], cm[l], cm[l + 1]
        den = y0 - 2 * y1 + y2
        off = 0.5 * (y0 - y2) / den if abs(den) > 1e-12 else 0.0
        f0[i] = fs / (l + off)
    # median-of-5 on voiced runs, drop isolated voiced frames
    out = f0.copy()
    for i in range(2, nfrm - 2):
        w = f0[i - 2: i + 3]
        if f0[i] > 0 and np.count_nonzero(w) >= 4:
            out[i] = np.median(w[w > 0])
        elif f0[i] > 0 and np.count_nonzero(w) <= 2:
            out[i] = 0
    return out.astype(np.float32)


if __name__ == "__main__":
    # arctic_a0001: config 1; are-you-ready: config 5 (test/test-pbpeffects.c reads it; also a data fixture)
    for name in ("arctic_a0001", "are-you-ready"):
        x, fs = read_wav(os.path.join(HERE, name + ".wav"))
        f0 = yin_track(x, fs)
        np.save(os.path.join(HERE, name + "_f0_hop128.npy"), f0)
        v = f0[f0 > 0]
        print(name, "frames", len(f0), "voiced", len(v), "median F0", np.median(v), "range", v.min(), v.max())
