"""-m gpu: the reference's config-1 acceptance on the GPU path, and
size-independent properties at BASELINE.json's full batch size (config 2)."""
import ctypes as C
import os

import numpy as np
import pytest

import libllsm2_amd as llsm
from conftest import FS, make_utterance
from gpu_common import (analysis_metrics, aopt_kwargs, assert_contract, gpu_analyze, Yard, oracle32_metrics, oracle_analyze, rel_rms,
                        report)
from verify_utils import GOLDEN, assert_reference_acceptance, read_wav

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = llsm.Context(0)
    yield c
    c.close()


def test_config1_arctic_anasynth_acceptance_and_parity(ctx, o64):
    """BASELINE.json configs[0]: test/test-layer0-anasynth.c:29-74 ('czt' run) through the
    drop-in entry points, F0 from the committed track; the reference's own KLD < 0.05 /
    corr > 0.95 thresholds, before and after phasesync_rps + phasepropagate; plus parameter
    parity with the oracle on the same input."""
    L = llsm.load()
    x, fs = read_wav(os.path.join(GOLDEN, "arctic_a0001.wav"))
    f0 = np.load(os.path.join(GOLDEN, "arctic_a0001_f0_hop128.npy"))
    ao = llsm.make_aoptions(thop=128.0 / fs, npsd=128, maxnhar=400, maxnhar_e=5, f0_refine=0, hm_method=llsm.HMCZT)
    so = llsm.make_soptions(fs)
    f0c = f0.copy()
    ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), fs, f0c.ctypes.data_as(llsm.P_fp), len(f0c), None)
    assert bool(ch), L.llsm_gpu_last_error()
    out = L.llsm_synthesize(C.byref(so), ch)
    assert bool(out)
    y = np.ctypeslib.as_array(out.contents.y, (out.contents.ny,)).copy()
    L.llsm_delete_output(out)
    msg1 = assert_reference_acceptance(x, y, "GPU anasynth")
    L.llsm_chunk_phasesync_rps(ch, 0)
    L.llsm_chunk_phasepropagate(ch, 1)
    out = L.llsm_synthesize(C.byref(so), ch)
    y2 = np.ctypeslib.as_array(out.contents.y, (out.contents.ny,)).copy()
    L.llsm_delete_output(out); L.llsm_delete_chunk(ch)
    msg2 = assert_reference_acceptance(x, y2, "GPU anasynth + rps/propagate")
    # parameter parity on a 1.5 s excerpt (oracle cost)
    n = 66150; nf = n // 128
    b, g, xres = gpu_analyze(ctx, ao, fs, [x[:n]], [f0[:nf]])
    b.close()
    pr, xr = oracle_analyze(o64, ao, fs, x[:n], f0[:nf])
    m = analysis_metrics(g, slice(0, nf), pr, xres, xr)
    report("config1_arctic", dict(m, acceptance=msg1, acceptance_rps=msg2))
    assert_contract(m, Yard(aopt_kwargs(ao), x[:n], fs, f0[:nf]), "config1_arctic")


def test_full_batch_properties(ctx, o64):
    """Config 2 at full size (1024 x 1 s, fixed 120 Hz): (i) identical utterances give
    bit-identical rows wherever they sit in the batch; (ii) spot utterances match the
    oracle; (iii) analysis -> synthesis preserves the harmonic part and the noise level;
    (iv) analysis is homogeneous of degree one (x -> 2x doubles amplitudes, PSD +6.02 dB)."""
    U, nx, nfrm = 1024, 44100, 200
    base = [make_utterance(u, 120.0) for u in range(4)]
    xs = [base[u % 4] if u % 4 else base[0] for u in range(U)]
    xs[7] = (2.0 * base[3]).astype(np.float32)                     # utterance 3 doubled
    f0 = np.full(nfrm, 120.0, np.float32)
    ao = llsm.make_aoptions(f0_refine=0)
    b = llsm.Batch(ctx, ao, FS, [nx] * U, [nfrm] * U)
    b.upload(llsm.A_X, np.concatenate(xs)); b.upload(llsm.A_F0, np.tile(f0, U))
    b.analyze()
    b.synthesize(llsm.make_soptions(FS), seed=99)
    ctx.sync()
    g = b.download_params()
    ys, yn = b.download(llsm.A_YSIN), b.download(llsm.A_YNOISE)
    rows = lambda k, u: g[k][u * nfrm:(u + 1) * nfrm]
    # (i) placement independence
    for k in (llsm.A_AMPL, llsm.A_PHSE, llsm.A_PSD, llsm.A_PSDRES, llsm.A_EDC, llsm.A_EENV_AMPL, llsm.A_EENV_PHSE):
        for u in (4, 512, 1020):
            assert np.array_equal(rows(k, 0), rows(k, u)), (k, u)
        assert np.array_equal(rows(k, 1), rows(k, 1021)), k
    ny = b.y_off[1]
    assert np.array_equal(ys[:ny], ys[4 * ny:5 * ny])               # deterministic part only (noise seeds differ)
    # (ii) oracle spot check
    pr, _ = oracle_analyze(o64, ao, FS, base[1], f0)
    m = analysis_metrics(g, slice(nfrm, 2 * nfrm), pr, np.zeros(0), np.zeros(0))
    m["xres_rel_rms"] = 0.0                                         # (no residual passed: rows only)
    assert_contract(m, Yard(aopt_kwargs(ao), base[1], FS, f0), "config2 spot")
    # (iii) round trip
    x0 = base[0]
    core = slice(2000, 42000)
    assert rel_rms(ys[:ny][core], x0[core]) < 0.06                 # harmonic part carries the signal (sigma = 0.01 noise)
    res_rms = np.sqrt(np.mean((x0[core] - ys[:ny][core]) ** 2))
    assert 0.8 < np.sqrt(np.mean(yn[:ny][core] ** 2)) / res_rms < 1.25
    # (iv) homogeneity
    a1, a2 = rows(llsm.A_AMPL, 3), rows(llsm.A_AMPL, 7)
    assert np.abs(a2 - 2 * a1).max() <= 2e-6 * a1.max()
    assert np.abs(wrapdiff(rows(llsm.A_PHSE, 7), rows(llsm.A_PHSE, 3))).max() < 1e-5
    assert np.abs(rows(llsm.A_PSD, 7) - rows(llsm.A_PSD, 3) - 20 * np.log10(2.0)).max() < 0.02
    report("full_batch", dict(spot=m, round_trip_rel=rel_rms(ys[:ny][core], x0[core])))
    b.close()


def wrapdiff(a, b):
    return np.angle(np.exp(1j * (a.astype(np.float64) - b.astype(np.float64))))


def test_create_batch_rejects_bad_arguments(ctx):
    """The additive C entry points validate what they are handed and report through
    llsm_gpu_last_error instead of reading out of bounds."""
    ao = llsm.make_aoptions(f0_refine=0)
    for nx, nfrm, what in (([-1], [10], "negative"), ([1000], [-3], "negative")):
        with pytest.raises(llsm.LlsmError, match=what):
            llsm.Batch(ctx, ao, FS, nx, nfrm)
    with pytest.raises(llsm.LlsmError, match="positive"):
        llsm.Batch(ctx, ao, 0.0, [1000], [4])
    bad = llsm.make_aoptions(f0_refine=0, nchannel=9, chanfreq=[float(1000 * (i + 1)) for i in range(8)])
    with pytest.raises(llsm.LlsmError, match="nchannel"):
        llsm.Batch(ctx, bad, FS, [1000], [4])
    L = llsm.load()
    assert L.llsm_gpu_batch_analyze(None) == -1
    assert b"no batch" in L.llsm_gpu_last_error()
