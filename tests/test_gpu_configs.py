"""-m gpu parity over a matrix of configurations (sampling rate 8 ... 96 kHz, hop, channel count, envelope
harmonics): every kernel variant the launchers can pick -- register-FFT sizes 256 ... 2048 and
the in-place LDS fallbacks, the <4,4> / <4,8> / <8,8> envelope templates, one and two IIR
sections per band -- against the float64 oracle, with the tolerances of test_gpu_parity.py."""
import numpy as np
import pytest

import libllsm2_amd as llsm
from conftest import make_speechlike, make_utterance
from gpu_common import (analysis_metrics, aopt_kwargs, assert_contract, gpu_analyze, Yard, oracle32_metrics, params_to_gpu_rows,
                        rel_rms, report)
from test_gpu_parity import SYN_TOL

pytestmark = pytest.mark.gpu

CONFIGS = {
    # id: (fs, thop, analysis options)
    "8k_2ch": (8000.0, 0.005, dict(nchannel=2, chanfreq=[1500.0], maxnhar=60)),
    "16k_3ch": (16000.0, 0.005, dict(nchannel=3, chanfreq=[1000.0, 3000.0])),
    "22k_hop128": (22050.0, 128.0 / 22050.0, dict(npsd=128, maxnhar=200, maxnhar_e=5)),
    "44k_hop10ms": (44100.0, 0.010, dict()),
    "48k": (48000.0, 0.005, dict()),
    "96k": (96000.0, 0.005, dict(maxnhar=200)),     # 4096-point spectrogram (LDS kernel), 2048-point fused noise filter
    # 25 ms hop at 96 kHz: the PSD window of 4 hops needs a 16384-point transform (beyond the LDS: global-scratch
    # kernel, round 4; refused until then), the noise filter an 8192-point one (128 KB of LDS)
    "96k_hop25ms": (96000.0, 0.025, dict()),
    "44k_me8_2ch": (44100.0, 0.005, dict(nchannel=2, chanfreq=[3000.0], maxnhar_e=8)),
    "44k_6ch": (44100.0, 0.005, dict(nchannel=6, chanfreq=[1000.0, 2000.0, 4000.0, 6000.0, 10000.0],
                                      maxnhar_e=5)),
    "44k_no_envelope_harmonics": (44100.0, 0.005, dict(maxnhar_e=0)),
    "44k_1ch": (44100.0, 0.005, dict(nchannel=1, chanfreq=[], maxnhar_e=3, npsd=64)),
}


@pytest.fixture(scope="module")
def ctx():
    c = llsm.Context(0)
    yield c
    c.close()


def _run_parity(ctx, o64, cid, fs, thop, kw, x, f0, oracle_out=None, quiet=False):
    """oracle_out: (Params, residual, (y, y_sin, y_noise)) computed elsewhere (tools/fuzz_soak.py runs the oracle in worker
    processes); returns the metrics."""
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)
    okw = aopt_kwargs(ao)
    if "chanfreq" in kw:
        okw["chanfreq"] = kw["chanfreq"]
    if oracle_out is None:
        pr, xr = o64.analyze(o64.aoptions(**okw), x, fs, f0, want_res=True)
    else:
        pr, xr = oracle_out[0], oracle_out[1]

    # analysis on the GPU vs the oracle
    b, g, xres = gpu_analyze(ctx, ao, fs, [x], [f0])
    try:
        m = analysis_metrics(g, slice(0, len(f0)), pr, xres, xr)
    finally:
        b.close()

    # synthesis on the GPU from the oracle's (float32-rounded) parameters vs the oracle
    b = llsm.Batch(ctx, ao, fs, [0], [len(f0)])
    try:
        b.upload_params(params_to_gpu_rows(pr))
        b.synthesize(llsm.make_soptions(fs), seed=5)
        ctx.sync()
        y, ys, yn = b.download(llsm.A_Y), b.download(llsm.A_YSIN), b.download(llsm.A_YNOISE)
    finally:
        b.close()
    if oracle_out is None:
        p32 = pr.astype(np.float32).astype(np.float64)
        yo, yso, yno = o64.synthesize(o64.soptions(fs), p32, seed=5)
    else:
        yo, yso, yno = oracle_out[2]
    m.update(ysin_rel_rms=rel_rms(ys, yso), ynoise_rel_rms=rel_rms(yn, yno), y_rel_rms=rel_rms(y, yo))
    try:
        assert_contract(m, Yard(okw, x, fs, f0), cid)
    finally:
        if not quiet:
            report("config_" + cid, m)
    assert len(yo) == len(y)
    for k in ("ysin_rel_rms", "ynoise_rel_rms", "y_rel_rms"):
        assert m[k] <= SYN_TOL, (cid, k, m[k])
    return m


@pytest.mark.parametrize("cid", sorted(CONFIGS))
def test_config_matrix_parity(ctx, o64, cid):
    fs, thop, kw = CONFIGS[cid]
    x, f0 = make_speechlike(11, nx=int((0.4 if thop < 0.02 else 1.2) * fs), fs=fs, thop=thop)
    _run_parity(ctx, o64, cid, fs, thop, kw, x, f0.astype(np.float32))


@pytest.mark.parametrize("f0_hz", [30.0, 52.0, 61.0, 950.0, 3000.0])
def test_extreme_f0_parity(ctx, o64, f0_hz):
    """Very low F0: the 3-period spectrogram window exceeds the 2048-point transform (time-aliased
    staging path of k_spgm_env) and the harmonic windows are at their largest; very high F0: the
    smallest windows and a handful of harmonics."""
    fs, thop = 44100.0, 0.005
    nx = int(0.5 * fs)
    x = make_utterance(21, f0_hz, nx=nx, fs=fs)
    f0 = np.full(int(nx / fs / thop), f0_hz, np.float32)
    _run_parity(ctx, o64, "f0_%d" % int(f0_hz), fs, thop, dict(), x, f0)


def _fuzz_case(seed):
    """A random but reproducible configuration: sampling rate, hop (integer and fractional sample counts), band plan,
    PSD size, harmonic limits, utterance length and voicing pattern."""
    r = np.random.default_rng(9000 + seed)
    fs = float(r.choice([8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000]))
    thop = float(r.choice([0.004, 0.005, 0.0075, 0.01, 128.0 / fs, 200.5 / fs, 77.25 / fs]))
    nch = int(r.integers(1, 6))
    top = 0.45 * fs
    edges = np.sort(r.uniform(0.03 * fs, top, size=nch - 1)).round(0)
    # band edges at least 0.02 x Nyquist apart (rows of the Chebyshev table) and distinct
    for k in range(1, len(edges)):
        edges[k] = max(edges[k], edges[k - 1] + 0.03 * fs)
    edges = [float(e) for e in edges if e < 0.49 * fs]
    kw = dict(nchannel=len(edges) + 1, chanfreq=edges, npsd=int(r.choice([32, 64, 128, 129, 200, 256])),
              maxnhar=int(r.choice([20, 60, 100, 160])), maxnhar_e=int(r.integers(0, 7)))
    nx = int(r.uniform(0.18, 0.45) * fs)
    return fs, thop, kw, nx


def test_long_utterance_parity(ctx, o64):
    """One 12 s utterance (2400 frames, 529 k samples): the block IIR walks 345 tiles per pass, the Kalman smoother
    2400 frames per bin, the overlap-add units span the whole utterance; and the tails of the PSD error
    distribution get 12 x more draws than in the short cases."""
    fs, thop = 44100.0, 0.005
    x, f0 = make_speechlike(5, nx=int(12 * fs), fs=fs, thop=thop)
    _run_parity(ctx, o64, "long_12s", fs, thop, dict(), x, f0.astype(np.float32))


@pytest.mark.parametrize("seed", range(48))
def test_random_configurations_parity(ctx, o64, seed):
    """Seeded fuzz over the configuration space (the matrix above is hand-picked; this one is not): every case goes
    through analysis and synthesis against the float64 oracle with the same tolerances."""
    fs, thop, kw, nx = _fuzz_case(seed)
    x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop)
    _run_parity(ctx, o64, "fuzz_%02d" % seed, fs, thop, kw, x, f0.astype(np.float32))


def other_rate_case(ctx, o64, seed):
    """Parameters of a random configuration synthesised at ANOTHER sampling rate (layer0.c:535-634 with options->fs != 2 FNYQ:
    band plan and window sizes follow the synthesis rate, the PSD rows are interpolated from the analysis axis)."""
    from gpu_common import oracle_analyze, params_to_gpu_rows, rel_rms
    from test_gpu_parity import SYN_TOL
    fs, thop, kw, nx = _fuzz_case(seed)
    r = np.random.default_rng(4000 + seed)
    fs2 = float(r.choice([8000, 16000, 22050, 32000, 44100, 48000, 96000]))
    if fs2 == fs:
        fs2 = fs * 1.5
    x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop); f0 = f0.astype(np.float32)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)
    pr, _ = oracle_analyze(o64, ao, fs, x, f0)
    p32 = pr.astype(np.float32).astype(np.float64)
    yo, yso, yno = o64.synthesize(o64.soptions(fs2), p32, seed=5)
    b = llsm.Batch(ctx, ao, fs2, [0], [len(f0)])
    assert b.L.llsm_gpu_batch_set_fnyq(b.h, fs / 2) == 0
    b.upload_params(params_to_gpu_rows(pr))
    b.synthesize(llsm.make_soptions(fs2), seed=5); ctx.sync()
    y, ys, yn = b.download(llsm.A_Y), b.download(llsm.A_YSIN), b.download(llsm.A_YNOISE); b.close()
    assert len(y) == len(yo), (len(y), len(yo))
    m = dict(fs=fs, fs_syn=fs2, ysin=rel_rms(ys, yso), ynoise=rel_rms(yn, yno), y=rel_rms(y, yo))
    for k in ("ysin", "ynoise", "y"):
        assert m[k] <= SYN_TOL, m
    return m


@pytest.mark.parametrize("seed", range(8))
def test_random_configurations_other_rate(ctx, o64, seed):
    report("fuzz_other_rate_%d" % seed, other_rate_case(ctx, o64, seed))
