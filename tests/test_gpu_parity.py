"""-m gpu parity tests: HIP path (through the C-ABI) vs the float64 CPU oracle on
identical seeded inputs.  Tolerances are the parity statement of SURVEY.md
section 8(d) / BASELINE.md section 3; every run also writes its measured errors
to gpurun_out/parity_*.json."""
import os

import numpy as np
import pytest

import libllsm2_amd as llsm
from conftest import FS, make_speechlike, make_utterance, wrap
from gpu_common import (analysis_metrics, aopt_kwargs, assert_contract, assert_hmpp_contract, gpu_analyze, Yard, oracle32_metrics, oracle_analyze,
                        params_to_gpu_rows, rel_rms, report)

pytestmark = pytest.mark.gpu

# ---- tolerances (float32 HIP path vs float64 oracle): gpu_common.CONTRACT / CONDITIONED ----
# Harmonics: |a e^{j phi} - oracle's| <= 1e-5 of the largest amplitude for EVERY harmonic, plus SURVEY 8(d)'s relative
# 1e-4 / 1e-3 rad above -40 dB.  PSD / PSDRES / band energies: SURVEY 8(d)'s 0.05 dB / 1e-4, or -- where the float32
# CONDITIONING of the algorithm itself exceeds that (a log of a periodogram bin at a Rayleigh null) -- at most half the
# distance of the float32 build of the oracle (the reference's own FP_TYPE = float arithmetic) from the float64 build.
# No distribution statements, no tiers: tests/test_gpu_regressions.py holds every input that ever exceeded a bound.
SYN_TOL = 1e-4          # relative RMS of y_sin / y_noise / y


@pytest.fixture(scope="module")
def ctx():
    c = llsm.Context(0)
    yield c
    c.close()


def small_inputs():
    xs, f0s = [], []
    xs.append(make_utterance(0, 120.0, nx=22050)); f0s.append(np.full(100, 120.0, np.float32))
    x, f0 = make_speechlike(1, nx=30000); xs.append(x); f0s.append(f0)
    xs.append(make_utterance(2, 311.0, nx=15000)); f0s.append(np.full(68, 311.0, np.float32))
    x, _ = make_speechlike(3, nx=9000); xs.append(x); f0s.append(np.zeros(int(9000 / FS / 0.005), np.float32))
    return xs, f0s


def test_analysis_parity_small_batch(ctx, o64):
    xs, f0s = small_inputs()
    ao = llsm.make_aoptions(f0_refine=0)
    b, g, xres = gpu_analyze(ctx, ao, FS, xs, f0s)
    rep = {}
    try:
        for u, (x, f0) in enumerate(zip(xs, f0s)):
            pr, xr = oracle_analyze(o64, ao, FS, x, f0)
            sl = slice(b.frm_off[u], b.frm_off[u + 1])
            m = analysis_metrics(g, sl, pr, xres[b.x_off[u]:b.x_off[u + 1]], xr)
            rep[f"utt{u}"] = m
        report("analysis_small", rep)
        for u, m in rep.items():
            i = int(u[3:])
            assert_contract(m, Yard(aopt_kwargs(ao), xs[i], FS, f0s[i]), u)
    finally:
        b.close()


def test_synthesis_parity_small_batch(ctx, o64):
    """Oracle-analysed parameters (identical on both sides) -> synthesis on the
    GPU vs the oracle, same counter-RNG seed."""
    xs, f0s = small_inputs()
    ao = llsm.make_aoptions(f0_refine=0)
    prs = [oracle_analyze(o64, ao, FS, x, f0)[0] for x, f0 in zip(xs, f0s)]
    b = llsm.Batch(ctx, ao, FS, [0] * len(xs), [len(f) for f in f0s])
    rows = [params_to_gpu_rows(p) for p in prs]
    b.upload_params({k: np.concatenate([r[k] for r in rows]) for k in rows[0]})
    so = llsm.make_soptions(FS)
    seed = 77
    b.synthesize(so, seed=seed)
    ctx.sync()
    y, ys, yn = b.download(llsm.A_Y), b.download(llsm.A_YSIN), b.download(llsm.A_YNOISE)
    rep = {}
    try:
        for u, pr in enumerate(prs):
            # the oracle synthesises from the same float32-rounded parameters
            p32 = pr.astype(np.float32).astype(np.float64)
            yo, yso, yno = o64.synthesize(o64.soptions(FS), p32, seed=seed + u)
            sl = slice(b.y_off[u], b.y_off[u + 1])
            assert len(yo) == b.y_off[u + 1] - b.y_off[u]
            rep[f"utt{u}"] = dict(ysin_rel_rms=rel_rms(ys[sl], yso), ynoise_rel_rms=rel_rms(yn[sl], yno),
                                  y_rel_rms=rel_rms(y[sl], yo), ysin_abs_max=float(np.abs(ys[sl] - yso).max()),
                                  ynoise_abs_max=float(np.abs(yn[sl] - yno).max()),
                                  ysin_rms=float(np.sqrt(np.mean(yso ** 2))), ynoise_rms=float(np.sqrt(np.mean(yno ** 2))))
        report("synthesis_small", rep)
        for u, m in rep.items():
            if m["ysin_rms"] > 0:
                assert m["ysin_rel_rms"] <= SYN_TOL, (u, m)
            else:
                assert m["ysin_abs_max"] == 0, (u, m)
            assert m["ynoise_rel_rms"] <= SYN_TOL and m["y_rel_rms"] <= SYN_TOL, (u, m)
    finally:
        b.close()


def test_injected_white_templates(ctx, o64):
    """Noise-template injection: same Gaussian templates on both sides."""
    x = make_utterance(5, 150.0, nx=12000); f0 = np.full(54, 150.0, np.float32)
    ao = llsm.make_aoptions(f0_refine=0)
    pr = oracle_analyze(o64, ao, FS, x, f0)[0]
    b = llsm.Batch(ctx, ao, FS, [0], [54])
    try:
        b.upload_params(params_to_gpu_rows(pr))
        rng = np.random.default_rng(9)
        ny = b.y_off[1]
        ntpl = min(20000, ny) + 128
        white = np.zeros((1, 4, b.layout.ntemplate_ext), np.float32)
        white[0, :, :ntpl] = rng.standard_normal((4, ntpl)).astype(np.float32)
        b.upload(llsm.A_WHITE, white)
        b.synthesize(llsm.make_soptions(FS), seed=0, injected_white=True)
        ctx.sync()
        yn = b.download(llsm.A_YNOISE)
        p32 = pr.astype(np.float32).astype(np.float64)
        _, _, yno = o64.synthesize(o64.soptions(FS), p32, seed=0, white=white[0, :, :ntpl].astype(np.float64))
        assert rel_rms(yn, yno) <= SYN_TOL
    finally:
        b.close()


def test_all_unvoiced_noninteger_hop(ctx):
    """test/test-layer0-edgecase.c:10-29 through the drop-in entry points."""
    import ctypes as C
    L = llsm.load()
    x, _ = make_speechlike(7, nx=40000)
    nhop = 100.5
    nfrm = int(len(x) / nhop)
    f0 = np.zeros(nfrm, np.float32)
    ao = llsm.make_aoptions(thop=nhop / FS)
    ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f0.ctypes.data_as(llsm.P_fp), nfrm, None)
    assert bool(ch), L.llsm_gpu_last_error()
    so = llsm.make_soptions(FS)
    out = L.llsm_synthesize(C.byref(so), ch)
    assert bool(out), L.llsm_gpu_last_error()
    ny = out.contents.ny
    ys = np.ctypeslib.as_array(out.contents.y_sin, (ny,))
    y = np.ctypeslib.as_array(out.contents.y, (ny,))
    assert np.all(ys == 0) and np.all(np.isfinite(y)) and np.sqrt(np.mean(y ** 2)) > 1e-4
    L.llsm_delete_output(out); L.llsm_delete_chunk(ch)


def test_dropin_matches_batch_path(ctx, o64):
    """llsm_analyze / llsm_synthesize (chunk objects) == the batch arrays."""
    import ctypes as C
    L = llsm.load()
    x, f0 = make_speechlike(11, nx=20000)
    ao = llsm.make_aoptions(f0_refine=0)
    b, g, xres = gpu_analyze(ctx, ao, FS, [x], [f0])
    b.close()
    f0c = f0.copy()
    xap = llsm.P_fp()
    ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f0c.ctypes.data_as(llsm.P_fp), len(f0c), C.byref(xap))
    assert bool(ch), L.llsm_gpu_last_error()
    assert np.array_equal(np.ctypeslib.as_array(xap, (len(x),)), xres)
    nfrm = len(f0)
    for i in (0, 10, nfrm // 2, nfrm - 1):
        fr = ch.contents.frames[i]
        hm = C.cast(L.llsm_container_get(fr, llsm.FRAME_HM), C.POINTER(llsm.HMFrame)).contents
        nm = C.cast(L.llsm_container_get(fr, llsm.FRAME_NM), C.POINTER(llsm.NMFrame)).contents
        res = C.cast(L.llsm_container_get(fr, llsm.FRAME_PSDRES), llsm.P_fp)
        assert hm.nhar == g[llsm.A_NHAR][i]
        if hm.nhar:
            assert np.array_equal(np.ctypeslib.as_array(hm.ampl, (hm.nhar,)), g[llsm.A_AMPL][i, :hm.nhar])
            assert np.array_equal(np.ctypeslib.as_array(hm.phse, (hm.nhar,)), g[llsm.A_PHSE][i, :hm.nhar])
        assert np.array_equal(np.ctypeslib.as_array(nm.psd, (nm.npsd,)), g[llsm.A_PSD][i])
        assert np.array_equal(np.ctypeslib.as_array(nm.edc, (nm.nchannel,)), g[llsm.A_EDC][i])
        assert L.llsm_fparray_length(res) == 256
        assert np.array_equal(np.ctypeslib.as_array(res, (256,)), g[llsm.A_PSDRES][i])
    # synthesis through the object model, fixed seed -> reproducible and sane
    so = llsm.make_soptions(FS)
    L.llsm_gpu_set_default_seed(1234)
    o1 = L.llsm_synthesize(C.byref(so), ch)
    L.llsm_gpu_set_default_seed(1234)
    o2 = L.llsm_synthesize(C.byref(so), ch)
    assert bool(o1) and bool(o2)
    ny = o1.contents.ny
    y1 = np.ctypeslib.as_array(o1.contents.y, (ny,)).copy(); y2 = np.ctypeslib.as_array(o2.contents.y, (ny,))
    assert np.array_equal(y1, y2)
    assert rel_rms(y1[2000:len(x) - 2000], x[2000:len(x) - 2000]) < 0.5
    # y never crosses the link (round 6): llsm_synthesize downloads y_sin and y_noise and forms y = y_sin + y_noise on the host
    # (layer0.c:657-659: one float addition per sample) -- bit for bit the row the DEVICE forms for the same chunk and seed
    ys1 = np.ctypeslib.as_array(o1.contents.y_sin, (ny,)); yn1 = np.ctypeslib.as_array(o1.contents.y_noise, (ny,))
    assert np.array_equal(y1, ys1 + yn1)
    b2 = llsm.Batch(ctx, ao, FS, [len(x)], [nfrm])
    b2.upload_params(g); b2.synthesize(so, seed=1234); ctx.sync()
    assert np.array_equal(b2.download(llsm.A_YSIN), ys1) and np.array_equal(b2.download(llsm.A_YNOISE), yn1)
    assert np.array_equal(b2.download(llsm.A_Y), y1)
    ysum = np.empty(ny, np.float32)
    L.llsm_gpu_sum_outputs(ysum.ctypes.data, ys1.ctypes.data, yn1.ctypes.data, ny)
    assert np.array_equal(ysum, y1)
    b2.close()
    # NULL on a chunk that fails the integrity check (layer0.c:637): voiced frame without HM
    L.llsm_container_remove(ch.contents.frames[10], llsm.FRAME_NM)
    assert not bool(L.llsm_synthesize(C.byref(so), ch))
    L.llsm_delete_output(o1); L.llsm_delete_output(o2); L.llsm_delete_chunk(ch)


def test_f0_refine_matches_oracle_and_mutates_f0(ctx, o64):
    import ctypes as C
    L = llsm.load()
    x = make_utterance(21, 200.0, nx=20000, sigma=0.002)
    f0 = np.full(90, 203.0, np.float32)                 # 1.5 % off
    ref = o64.refine_f0(x, FS, f0, 0.005)
    ao = llsm.make_aoptions(f0_refine=1)
    f0c = f0.copy()
    ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f0c.ctypes.data_as(llsm.P_fp), len(f0c), None)
    assert bool(ch), L.llsm_gpu_last_error()
    L.llsm_delete_chunk(ch)
    mid = slice(5, 85)
    assert np.abs(f0c[mid] - 200.0).max() < 0.5            # pulled onto the true F0
    assert np.abs(f0c[mid] - ref[mid]).max() < 2e-2         # and equal to the oracle's estimate


def test_empty_and_ragged_batches(ctx):
    ao = llsm.make_aoptions(f0_refine=0)
    b = llsm.Batch(ctx, ao, FS, [], [])
    b.analyze(); b.synthesize(llsm.make_soptions(FS)); ctx.sync(); b.close()
    # ragged: utterances of very different lengths, one with zero frames, one tiny
    xs = [make_utterance(30, 100.0, nx=5000), make_utterance(32, 140.0, nx=300), make_utterance(31, 250.0, nx=11111)]
    f0s = [np.full(22, 100.0, np.float32), np.zeros(0, np.float32), np.full(50, 250.0, np.float32)]
    b, g, xres = gpu_analyze(ctx, ao, FS, xs, f0s)
    b.synthesize(llsm.make_soptions(FS), seed=3); ctx.sync()
    y = b.download(llsm.A_Y)
    assert np.all(np.isfinite(y)) and np.all(np.isfinite(g[llsm.A_PSD]))
    # batch invariance: utterance 2 alone gives bit-identical rows
    b2, g2, xres2 = gpu_analyze(ctx, ao, FS, [xs[2]], [f0s[2]])
    sl = slice(b.frm_off[2], b.frm_off[3])
    for k in (llsm.A_AMPL, llsm.A_PHSE, llsm.A_PSD, llsm.A_PSDRES, llsm.A_EDC, llsm.A_EENV_AMPL):
        assert np.array_equal(g[k][sl], g2[k]), k
    assert np.array_equal(xres[b.x_off[2]:b.x_off[3]], xres2)
    # the frameless utterance: nothing to subtract, nothing to synthesise
    assert np.array_equal(xres[b.x_off[1]:b.x_off[2]], xs[1])
    assert not np.any(y[b.y_off[1]:b.y_off[2]])
    b.close(); b2.close()


def test_overlap_add_units_do_not_change_results(ctx, monkeypatch):
    """The fused overlap-add kernels cut every utterance into units of frames (one wavefront each,
    halo frames recomputed).  The cut must be invisible: bit-identical residual and waveforms for
    the default unit length and for very short units (many boundaries, units shorter than the halo)."""
    x, f0 = make_speechlike(5, nx=33000)
    xs = [x, make_utterance(6, 180.0, nx=9000)]
    f0s = [f0, np.full(38, 180.0, np.float32)]
    ao = llsm.make_aoptions(f0_refine=0)

    def run():
        b, g, xres = gpu_analyze(ctx, ao, FS, xs, f0s)
        b.synthesize(llsm.make_soptions(FS), seed=11); ctx.sync()
        out = [xres] + [b.download(a) for a in (llsm.A_YSIN, llsm.A_YNOISE, llsm.A_Y)]
        b.close()
        return out

    ref = run()
    for sin_unit, noise_unit in (("1", "2"), ("3", "4"), ("7", "10")):
        monkeypatch.setenv("LLSM_GPU_SIN_UNIT", sin_unit)
        monkeypatch.setenv("LLSM_GPU_NOISE_UNIT", noise_unit)
        got = run()
        for name, a, r in zip(("xres", "ysin", "ynoise", "y"), got, ref):
            assert np.array_equal(a, r), (name, sin_unit, noise_unit, float(np.abs(a - r).max()))


def test_hmpp_peak_picking_parity(ctx, o64):
    """LLSM_AOPTION_HMPP (dsputils.c:196-213, 126-143) vs the oracle, plus the reference's own
    chirp KAT thresholds for this method (test-dsputils.c:44-133) on the GPU result."""
    xs, f0s = small_inputs()
    xs, f0s = xs[:3], f0s[:3]
    ao = llsm.make_aoptions(f0_refine=0, hm_method=llsm.HMPP)
    b, g, xres = gpu_analyze(ctx, ao, FS, xs, f0s)
    rep = {}
    for u, (x, f0) in enumerate(zip(xs, f0s)):
        pr, xr = oracle_analyze(o64, ao, FS, x, f0)
        sl = slice(b.frm_off[u], b.frm_off[u + 1])
        rep[f"utt{u}"] = analysis_metrics(g, sl, pr, xres[b.x_off[u]:b.x_off[u + 1]], xr)
    b.close()
    report("analysis_hmpp", rep)
    for u, m in rep.items():
        # gpu_common.HMPP_CONTRACT: SURVEY 8(d) above -40 dB (measured 2.3e-4 rad); the every-harmonic bound relative to
        # the float32 oracle's own distance from float64 (peak-picked phases interpolate WRAPPED bin phases,
        # dsputils.c:140-141: on a weak harmonic a float32 difference in the peak position meets a slope of pi per bin)
        i = int(u[3:])
        assert_hmpp_contract(m, Yard(aopt_kwargs(ao), xs[i], FS, f0s[i]), u)
    # chirp KAT on the GPU
    from test_oracle_kat import chirp_signal
    x, fs, thop, f0, truth = chirp_signal()
    ao = llsm.make_aoptions(f0_refine=0, hm_method=llsm.HMPP, thop=thop, maxnhar=3)
    b, g, _ = gpu_analyze(ctx, ao, fs, [x], [f0])
    b.close()
    nfrm = len(f0)
    ampl, phse = g[llsm.A_AMPL].astype(np.float64), g[llsm.A_PHSE].astype(np.float64)
    for h, tr in ((0, truth), (1, np.full(nfrm, 0.5)), (2, np.full(nfrm, 0.25))):
        err = np.zeros(nfrm); err[5:nfrm - 5] = (ampl[:, h] - tr)[5:nfrm - 5]
        assert abs(err.mean()) < 0.01 and abs(err.std()) < 0.01, (h, err.mean(), err.std())
    perr = np.zeros(nfrm - 1)
    for i in range(5, nfrm - 5):
        perr[i - 1] = wrap(phse[i, 0] - (phse[i - 1, 0] + f0[i] * 2.0 * 3.1415927 * thop))
    assert abs(perr.mean()) < 0.1 and abs(perr.std()) < 0.1


def test_chirp_kat_czt_on_gpu(ctx):
    """test-dsputils.c:44-133 with LLSM_AOPTION_HMCZT on the GPU path."""
    from test_oracle_kat import chirp_signal
    x, fs, thop, f0, truth = chirp_signal()
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, maxnhar=3)
    b, g, _ = gpu_analyze(ctx, ao, fs, [x], [f0])
    b.close()
    nfrm = len(f0)
    ampl, phse = g[llsm.A_AMPL].astype(np.float64), g[llsm.A_PHSE].astype(np.float64)
    assert np.all(g[llsm.A_NHAR] == 3)
    for h, tr in ((0, truth), (1, np.full(nfrm, 0.5)), (2, np.full(nfrm, 0.25))):
        err = np.zeros(nfrm); err[5:nfrm - 5] = (ampl[:, h] - tr)[5:nfrm - 5]
        assert abs(err.mean()) < 0.01 and abs(err.std()) < 0.01, (h, err.mean(), err.std())
    perr = np.zeros(nfrm - 1)
    for i in range(5, nfrm - 5):
        perr[i - 1] = wrap(phse[i, 0] - (phse[i - 1, 0] + f0[i] * 2.0 * 3.1415927 * thop))
    assert abs(perr.mean()) < 0.1 and abs(perr.std()) < 0.1


def test_unsupported_configurations_fail_loudly(ctx):
    ao = llsm.make_aoptions(f0_refine=0)
    b = llsm.Batch(ctx, ao, FS, [1000], [4])
    with pytest.raises(llsm.LlsmError):
        b.synthesize(llsm.make_soptions(FS, use_l1=1))
    b.close()
    # HMPP needs a transform of the four-period window: any size up to 2^17 points is computed (F0 >= 1.35 Hz at
    # 44.1 kHz; beyond the LDS on global scratch since round 4); below that the batch is refused with the F0 and the
    # window length in the text -- not rows of nhar = 0 (dsputils.c:318-326 has no limit; the CZT analysis has none)
    x = make_utterance(3, 20.0, nx=30000)
    f0 = np.full(30, 1.2, np.float32)
    b = llsm.Batch(ctx, llsm.make_aoptions(f0_refine=0, hm_method=llsm.HMPP), FS, [len(x)], [len(f0)])
    b.upload(llsm.A_X, x); b.upload(llsm.A_F0, f0)
    with pytest.raises(llsm.LlsmError, match="131072"):
        b.analyze()
    b.close()
    f0 = np.full(30, 20.0, np.float32)
    b = llsm.Batch(ctx, llsm.make_aoptions(f0_refine=0), FS, [len(x)], [len(f0)])      # 20 Hz through the CZT
    b.upload(llsm.A_X, x); b.upload(llsm.A_F0, f0)
    b.analyze(); ctx.sync()
    assert int(b.download(llsm.A_NHAR).min()) == 100
    b.close()


@pytest.mark.parametrize("f0_hz", [15.0, 20.0])
def test_hmpp_below_the_lds_transform(ctx, o64, f0_hz):
    """HMPP at 15 / 20 Hz, 44.1 kHz: windows of 11760 / 8820 samples -> 16384-point transforms (dsputils.c:318-326),
    more than the LDS holds: k_harm_pp_big on global scratch.  Mixed with an utterance at 120 Hz in the same batch
    (its frames stay with the LDS kernel).  Refused until round 4."""
    x0 = make_utterance(43, f0_hz, nx=40000)
    f00 = np.full(int(len(x0) / FS / 0.005), f0_hz, np.float32)
    x1 = make_utterance(44, 120.0, nx=20000)
    f01 = np.full(int(len(x1) / FS / 0.005), 120.0, np.float32)
    ao = llsm.make_aoptions(f0_refine=0, hm_method=llsm.HMPP)
    b, g, xres = gpu_analyze(ctx, ao, FS, [x0, x1], [f00, f01])
    rep = {}
    for u, (x, f0) in enumerate(((x0, f00), (x1, f01))):
        pr, xr = oracle_analyze(o64, ao, FS, x, f0)
        sl = slice(b.frm_off[u], b.frm_off[u + 1])
        m = analysis_metrics(g, sl, pr, xres[b.x_off[u]:b.x_off[u + 1]], xr)
        rep[f"utt{u}"] = m
        assert int(pr.nhar.max()) == 100
        assert_hmpp_contract(m, Yard(aopt_kwargs(ao), x, FS, f0), f"utt{u}")
    b.close()
    report("analysis_hmpp_f0_%d" % int(f0_hz), rep)


@pytest.mark.parametrize("method", ["HMCZT", "HMPP"])
def test_rows_do_not_depend_on_a_known_lowest_f0(ctx, method):
    """whoever takes the F0 row's device address may rewrite it, so llsm_gpu_batch_device_ptr(F0) marks the
    batch's lowest F0 as unknown and every F0-sized provision falls back to its maximum (window table of the shared-F0
    tiles: 48 KB; peak-picking transform: 2^17 points on global scratch).  The rows must be the SAME bits as with the
    lowest F0 known from the upload: which kernel a frame takes never depends on that value."""
    xs, f0s = small_inputs()
    xs, f0s = xs[:3], f0s[:3]
    ao = llsm.make_aoptions(f0_refine=0, hm_method=getattr(llsm, method))
    rows = []
    for unknown in (False, True):
        b = llsm.Batch(ctx, ao, FS, [len(x) for x in xs], [len(f) for f in f0s])
        b.upload(llsm.A_X, np.concatenate(xs)); b.upload(llsm.A_F0, np.concatenate(f0s))
        if unknown:
            assert b.device_ptr(llsm.A_F0)
        b.analyze(); ctx.sync()
        rows.append((b.download_params(), b.download(llsm.A_XRES)))
        b.close()
    (g0, x0), (g1, x1) = rows
    assert np.array_equal(x0, x1)
    for k in g0:
        assert np.array_equal(g0[k], g1[k]), k


def test_hmpp_low_f0_uses_the_8192_point_transform(ctx, o64):
    """HMPP at F0 = 30 Hz, 44.1 kHz: the four-period window is 5880 samples, llsm_get_fftsize gives 8192 points
    (128 KB of LDS for k_harm_pp).  Until round 2 such frames came back without harmonics."""
    f0_hz = 30.0
    x = make_utterance(41, f0_hz, nx=30000)
    f0 = np.full(int(len(x) / FS / 0.005), f0_hz, np.float32)
    ao = llsm.make_aoptions(f0_refine=0, hm_method=llsm.HMPP)
    b, g, xres = gpu_analyze(ctx, ao, FS, [x], [f0])
    pr, xr = oracle_analyze(o64, ao, FS, x, f0)
    m = analysis_metrics(g, slice(0, len(f0)), pr, xres, xr)
    b.close()
    report("analysis_hmpp_f0_30", m)
    assert int(pr.nhar.max()) == 100
    assert_hmpp_contract(m, Yard(aopt_kwargs(ao), x, FS, f0), "f0_30")


@pytest.mark.parametrize("case", ["tiled_1s_and_ragged", "short_hop_8k", "eight_envelope_harmonics", "two_channels"])
def test_excitation_by_template_position_equals_the_per_sample_kernel(ctx, monkeypatch, case):
    """k_excite_env4 (round 5: a thread owns four residues of the template's tiling period and walks the tiles; templates
    loaded once, phasors by rotation) against k_excite_env (one thread per output sample, plan.h stretch_index): same
    (frame, offset) table, same accumulation order -- the noise part may differ by the float32 rounding of at most three
    phasor rotations per sample.  Utterances longer than the template (tiles and cross-fades: 2.2 and 3.4 tiles), exactly
    at its length, shorter than it, not a multiple of four samples, and empty; hops shorter than the staged range."""
    fs, thop, kw = FS, 0.005, dict()
    if case == "short_hop_8k":
        fs, thop, kw = 8000.0, 0.004, dict(nchannel=2, chanfreq=[1500.0], maxnhar=40)       # 32-sample hop: the HBM path
    elif case == "eight_envelope_harmonics":
        kw = dict(maxnhar_e=8, nchannel=2, chanfreq=[3000.0])
    elif case == "two_channels":
        fs, thop, kw = 16000.0, 200.5 / 16000.0, dict(nchannel=2, chanfreq=[3000.0], maxnhar_e=3)
    lens = [int(1.0 * fs), int(1.55 * fs), 20000, 19999, 20131, 7001, 0] if case == "tiled_1s_and_ragged" else [int(3.1 * fs), 5003]
    xs, f0s = [], []
    for k, nx in enumerate(lens):
        if nx == 0:
            xs.append(np.zeros(0, np.float32)); f0s.append(np.zeros(0, np.float32)); continue
        x, f0 = make_speechlike(70 + k, nx=nx, fs=fs, thop=thop)
        xs.append(x); f0s.append(f0.astype(np.float32))
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)

    def run():
        b, g, xres = gpu_analyze(ctx, ao, fs, xs, f0s)
        b.synthesize(llsm.make_soptions(fs), seed=21); ctx.sync()
        yn = b.download(llsm.A_YNOISE); off = b.y_off.copy()
        b.close()
        return yn, off

    monkeypatch.setenv("LLSM_GPU_EXCITE4", "0")
    ref, off = run()
    monkeypatch.setenv("LLSM_GPU_EXCITE4", "1")
    got, _ = run()
    assert np.all(np.isfinite(got)) and len(got) == len(ref)
    rep = {}
    for u in range(len(lens)):
        a, r = got[off[u]:off[u + 1]], ref[off[u]:off[u + 1]]
        rms = float(np.sqrt(np.mean(r.astype(np.float64) ** 2))) if len(r) else 0.0
        if rms == 0:                                      # (too short for a noise frame: silence on both sides)
            assert not np.any(a)
            continue
        rep[f"utt{u}"] = dict(ny=int(len(r)), rel_rms=rel_rms(a, r), abs_max_over_rms=float(np.abs(a - r).max() / rms))
        assert rep[f"utt{u}"]["rel_rms"] <= 2e-6 and rep[f"utt{u}"]["abs_max_over_rms"] <= 5e-5, (case, u, rep[f"utt{u}"])
    report("excite4_vs_per_sample_" + case, rep)
