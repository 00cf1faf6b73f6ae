"""CPU tests of the coder oracle (oracle/coder_oracle.c; coder.c:44-292): encode -> decode keeps the spectral
envelope within the resolution of the code, and the reference's own acceptance (test/test-coder.c:31-51:
analyze -> layer 1 -> encode(64, 5) -> decode layer 0 / layer 1 -> synthesize, waveform-distribution KLD < 0.05
against the input) holds on the oracle."""
import os

import numpy as np
import pytest

from conftest import FS, make_speechlike
from verify_utils import GOLDEN, data_distribution_klds, read_wav


def test_encode_decode_round_trip(o64):
    x, f0 = make_speechlike(1, nx=20000)
    pr = o64.analyze(o64.aoptions(f0_refine=0), x, FS, f0)
    q = o64.chunk_tolayer1(pr, 2048)
    enc = o64.coder_encode_chunk(pr, q, 64, 5)
    assert enc.shape == (pr.nfrm, 72)
    v = f0 > 0
    assert np.array_equal(enc[:, 0], v.astype(float)) and np.allclose(enc[:, 1], f0) and np.allclose(enc[v, 2], q.rd[v])
    assert np.all(enc[~v, 67:] == 1.0) and np.all((enc[v, 67:] > 0) & (enc[v, 67:] <= 1.0))
    p0, q0 = o64.coder_decode_chunk(enc, False, pr, 1025, 1.5, 64, 5, 441)
    p1, q1 = o64.coder_decode_chunk(enc, True, pr, 1025, 1.5, 64, 5, 441)
    assert np.array_equal(p0.f0, f0.astype(np.float64)) and np.array_equal(p1.f0, p0.f0)
    i = int(np.flatnonzero(v)[30])
    assert p0.nhar[i] == int((FS / 2) / f0[i]) and q1.nvsphse[i] == p0.nhar[i] and p1.nhar[i] == 0
    # the decoded harmonic amplitudes follow the analysed ones (64 mel-spaced points: a few dB)
    n = 25
    d = 20 * np.log10(p0.ampl[i, :n] / pr.ampl[i, :n])
    assert abs(np.median(d)) < 3.0 and np.percentile(np.abs(d), 80) < 6.0, d      # weak harmonics under the noise are not kept
    # layer-1 decode -> layer 0 gives the same amplitudes as the direct layer-0 decode
    p1c = p1.copy(); o64.chunk_tolayer0(p1c, q1)
    m = min(p1c.nhar[i], p0.nhar[i], 60)
    d2 = 20 * np.log10(p1c.ampl[i, :m] / p0.ampl[i, :m])
    assert np.abs(d2).max() < 1.5, d2
    # noise PSD survives within a few dB where it matters
    dn = p0.psd[i] - pr.psd[i]
    assert abs(np.median(dn)) < 3.0


def test_coder_acceptance_on_arctic(o64):
    x, fs = read_wav(os.path.join(GOLDEN, "arctic_a0001.wav"))
    f0 = np.load(os.path.join(GOLDEN, "arctic_a0001_f0_hop128.npy"))
    pr = o64.analyze(o64.aoptions(thop=128.0 / fs, f0_refine=0), x, fs, f0)
    q = o64.chunk_tolayer1(pr, 2048)
    enc = o64.coder_encode_chunk(pr, q, 64, 5)
    so = o64.soptions(fs)
    mh = int((fs / 2) / 20.0) + 1
    for use_l1 in (False, True):
        p, ql = o64.coder_decode_chunk(enc, use_l1, pr, 1025, 1.5, 64, 5, mh)
        if use_l1:
            o64.chunk_tolayer0(p, ql)
        o64.phasepropagate(p, 1)
        y, ys, yn = o64.synthesize(so, p, seed=2)
        klds = data_distribution_klds(x, y)
        assert all(k < 0.05 for k in klds), (use_l1, klds)          # test-coder.c:48-49, verify-utils.h:88-107


def test_decoder_aperiodicity_rows(o64):
    """o_coder_aperiodicity (the conditioning the GPU coder test reads): band values interpolated over
    linspace(0, FNYQ, order_bap + 1) with 0 (voiced) / 1 (unvoiced) in front, 1e-3 below 500 Hz and a ramp to 2 kHz on
    voiced frames (coder.c:196-209)."""
    x, f0 = make_speechlike(1, nx=12000)
    pr = o64.analyze(o64.aoptions(f0_refine=0), x, FS, f0)
    ns, osp, obap = 1025, 64, 4                                  # knots on bins 0, 256, 512, 768, 1024
    enc = np.zeros((3, 3 + osp + obap))
    enc[0, :3] = (1, 200.0, 1.0); enc[0, 3 + osp:] = (0.2, 0.5, 0.9, 1.0)
    enc[1, :3] = (0, 0.0, 1.0);   enc[1, 3 + osp:] = (1.0, 1.0, 1.0, 1.0)
    enc[2, :3] = (1, 100.0, 0.5); enc[2, 3 + osp:] = (1.0, 1.0, 0.3, 0.6)
    ap = o64.coder_aperiodicity_chunk(enc, pr, ns, 1.5, osp, obap)
    fj = np.arange(ns) * (FS / 2) / ns                           # the post-processing's own axis (j fnyq / ns)
    assert np.all(ap[1] == 1.0)                                  # unvoiced: 1 in front of ones
    for r, bap in ((0, enc[0, 3 + osp:]), (2, enc[2, 3 + osp:])):
        knots = np.r_[0.0, bap]
        lin = np.interp(np.arange(ns) / (ns - 1.0) * obap, np.arange(obap + 1), knots)
        hi = fj >= 2000
        assert np.allclose(ap[r, hi], lin[hi], atol=1e-12)
        assert np.all(ap[r, fj < 500] == 1e-3)
        mid = (fj >= 500) & (fj < 2000)
        assert np.allclose(ap[r, mid], 1e-3 + (lin[mid] - 1e-3) * (fj[mid] - 500) / 1500, atol=1e-12)
    assert ap[0, 1024] == 1.0 and ap[2, 512] == 1.0 and abs(ap[2, 256] - 1.0) < 1e-12
