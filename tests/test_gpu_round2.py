"""-m gpu: BASELINE.json configs[2] (per-GPU shard of the 80-400 Hz sweep at size) and configs[3]
(64 concurrent llsmrt streams, 256-sample pulls) on the harmonic-model path; synthesis at a
sampling rate other than the analysis rate (layer0.c:578, 606-607); degenerate batches
(frames over empty signals, frameless utterances); conf mismatches inside a batch."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import libllsm2_amd as llsm
from conftest import FS, make_speechlike, make_utterance
from gpu_common import (analysis_metrics, aopt_kwargs, assert_contract, gpu_analyze, Yard, oracle32_metrics, oracle_analyze,
                        params_to_gpu_rows, rel_rms, report)
from test_gpu_parity import SYN_TOL
from test_gpu_rt import chunk_from_oracle, rt_run

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = llsm.Context(0)
    yield c
    c.close()


def test_config3_sweep_shard_at_size(ctx, o64):
    """One GPU's share of config 3: 1024 one-second utterances whose F0 sweeps 80 -> 400 Hz
    logarithmically (bench.py --workload sweep builds exactly these).  Spot utterances at six F0s
    against the oracle (analysis parameters and waveforms); the whole batch through
    size-independent properties: nhar follows the index plan, the resynthesis carries the
    harmonic part (x - y_sin is the noise floor), outputs finite."""
    sys.path.insert(0, ROOT)
    import bench
    from libllsm2_amd.sharding import sweep_f0
    U, nx, nfrm = 1024, bench.NX, bench.NFRM
    f0_of = lambda u: sweep_f0(u, U)
    # the signals of bench.make_batch_inputs (same seeds, same formula), built with numpy so that the test
    # does not depend on torch's device initialisation
    x = np.empty((U, nx), np.float32)
    n = np.arange(nx, dtype=np.float64)

    def one(u):
        K, phi, noise = bench.synth_phases_noise(u, f0_of(u))
        k = np.arange(1, K + 1, dtype=np.float64)[:, None]
        x[u] = ((0.3 / k) * np.cos(2 * np.pi * k * f0_of(u) * n[None, :] / FS + phi[:, None])).sum(0) + 0.01 * noise
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:     # numpy releases the GIL in cos / sum
        list(ex.map(one, range(U)))
    f0s = np.asarray([np.float32(f0_of(u)) for u in range(U)], np.float32)
    ao = llsm.make_aoptions(f0_refine=0)
    b = llsm.Batch(ctx, ao, FS, [nx] * U, [nfrm] * U)
    b.upload(llsm.A_X, x.reshape(-1)); b.upload(llsm.A_F0, np.repeat(f0s, nfrm))
    b.analyze()
    b.synthesize(llsm.make_soptions(FS), seed=31)
    ctx.sync()
    g = b.download_params()
    xres = b.download(llsm.A_XRES).reshape(U, nx)
    ys = b.download(llsm.A_YSIN); yn = b.download(llsm.A_YNOISE); y = b.download(llsm.A_Y)
    ny = b.y_off[1]
    L = llsm.load()
    # whole batch: plan and energy properties
    nh_plan = np.asarray([L.llsm_gpu_plan_index(7, 100, 0, float(f), 0.005, FS, 4.0) for f in f0s])
    assert np.array_equal(g[llsm.A_NHAR].reshape(U, nfrm), np.repeat(nh_plan[:, None], nfrm, 1))
    assert np.all(np.isfinite(y)) and np.all(np.isfinite(g[llsm.A_PSD])) and np.all(np.isfinite(g[llsm.A_AMPL]))
    mid = slice(3000, 41000)
    res_rms = np.sqrt(np.mean(xres[:, mid] ** 2, axis=1))
    assert res_rms.max() < 0.035 and res_rms.min() > 0.005, (res_rms.min(), res_rms.max())   # sigma = 0.01 noise floor + estimation noise (oracle: 0.0136 at 120 Hz, 0.0275 at 400 Hz)
    ysu = ys.reshape(U, ny)[:, mid]
    d = np.sqrt(np.mean((x[:, mid] - ysu) ** 2, axis=1))
    assert d.max() < 0.035, d.max()
    # spot parity against the oracle at six F0s across the sweep
    rep = {}
    for u in (0, 200, 411, 640, 850, 1023):
        f0 = np.full(nfrm, f0s[u], np.float32)
        pr, xr = oracle_analyze(o64, ao, FS, x[u], f0)
        m = analysis_metrics(g, slice(u * nfrm, (u + 1) * nfrm), pr, xres[u], xr)
        # synthesis of THIS utterance's GPU parameters by the oracle (same seed convention: utterance index
        # enters the counter RNG through its template rows, so compare the deterministic part only)
        from oracle.oracle import Params
        q = Params(pr.nfrm, pr.maxnhar, pr.maxnhar_e, pr.npsd, pr.nchannel, pr.thop, pr.fnyq, pr.chanfreq, np.float64)
        sl = slice(u * nfrm, (u + 1) * nfrm)
        q.f0[:] = g[llsm.A_F0][sl]; q.nhar[:] = g[llsm.A_NHAR][sl]; q.ampl[:] = g[llsm.A_AMPL][sl]; q.phse[:] = g[llsm.A_PHSE][sl]
        q.psd[:] = g[llsm.A_PSD][sl]; q.psdres[:] = g[llsm.A_PSDRES][sl]; q.edc[:] = g[llsm.A_EDC][sl]
        q.nhar_e[:] = g[llsm.A_NHAR_E][sl]
        q.eenv_ampl[:] = g[llsm.A_EENV_AMPL][sl].reshape(q.eenv_ampl.shape); q.eenv_phse[:] = g[llsm.A_EENV_PHSE][sl].reshape(q.eenv_phse.shape)
        yo, yso, yno = o64.synthesize(o64.soptions(FS), q, seed=31)
        m["ysin_rel_rms"] = rel_rms(ys[b.y_off[u]:b.y_off[u + 1]], yso)
        m["f0"] = float(f0s[u])
        m["_f32"] = Yard(aopt_kwargs(ao), x[u], FS, f0)
        rep[f"utt{u}"] = m
    f32 = {k: m.pop("_f32") for k, m in rep.items()}
    report("config3_sweep_shard", rep)
    b.close()
    for k, m in rep.items():
        assert_contract(m, f32[k], k)
        assert m["ysin_rel_rms"] <= SYN_TOL, (k, m["ysin_rel_rms"])


def test_config4_group_of_64_streams(ctx):
    """BASELINE.json configs[3] shape (harmonic-model path): 64 lock-stepped streams, one hop per feed,
    consumer pulls 256 samples per stream -- each stream equals a single-stream buffer with its seed."""
    L = llsm.load()
    S, thop, ndist = 64, 0.005, 8
    ao = llsm.make_aoptions(f0_refine=0, thop=thop)
    chunks, nfrm = [], None
    for s in range(ndist):
        x, _ = make_speechlike(40 + s, nx=12000)
        n = int(len(x) / FS / thop)
        t = np.arange(n) * thop
        f0 = (110 + 25 * s + 30 * np.sin(2 * np.pi * 1.1 * t + s)).astype(np.float32)
        f0[:2] = 0
        if s == 5:
            f0[n // 2:] = 0
        ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f0.ctypes.data_as(llsm.P_fp), n, None)
        assert bool(ch), L.llsm_gpu_last_error()
        chunks.append(ch); nfrm = n
    so = llsm.make_soptions(FS)
    seed = 4200
    singles = []
    for s in range(S):
        L.llsm_gpu_set_default_seed(seed + s)
        yp, yap, lat = rt_run(L, so, chunks[s % ndist], nfrm)
        singles.append((yp, yap))
    L.llsm_gpu_set_default_seed(seed)
    g = L.llsm_create_rtsynth_group(C.byref(so), chunks[0].contents.conf, 4096, S)
    assert g, L.llsm_gpu_last_error()
    assert L.llsm_rtsynth_group_getlatency(g) == lat
    outp = [[] for _ in range(S)]; outap = [[] for _ in range(S)]
    bp = np.zeros(256, np.float32); bap = np.zeros(256, np.float32)
    FrameArr = C.POINTER(llsm.Container) * S
    for i in range(nfrm):
        fr = FrameArr(*[chunks[s % ndist].contents.frames[i] for s in range(S)])
        L.llsm_rtsynth_group_feed(g, fr)
        for s in range(S):
            while L.llsm_rtsynth_group_numoutput(g, s) >= 256 or (i == nfrm - 1 and L.llsm_rtsynth_group_numoutput(g, s) > 0):
                n = L.llsm_rtsynth_group_fetch(g, s, bp.ctypes.data_as(llsm.P_fp), bap.ctypes.data_as(llsm.P_fp), 256)
                outp[s].append(bp[:n].copy()); outap[s].append(bap[:n].copy())
    L.llsm_delete_rtsynth_group(g)
    worst = 0.0
    for s in range(S):
        yp, yap = np.concatenate(outp[s]), np.concatenate(outap[s])
        assert len(yp) == len(singles[s][0])
        assert np.array_equal(yp, singles[s][0].astype(np.float32)), s
        e = rel_rms(yap, singles[s][1]); worst = max(worst, e)
        assert e < 2e-6, (s, e)
    report("config4_group64", dict(streams=S, frames=nfrm, noise_rel_rms_worst=worst))
    for ch in chunks:
        L.llsm_delete_chunk(ch)


def test_synthesis_at_another_sampling_rate(ctx, o64):
    """Parameters analysed at 44.1 kHz (FNYQ 22050) synthesised at 48 kHz and at 32 kHz: the stored PSD is
    interpolated from linspace(0, FNYQ, npsd) onto the synthesis bins (clamped above FNYQ); every other
    size follows options->fs.  Through the batch API (set_fnyq) and through llsm_synthesize on a chunk."""
    L = llsm.load()
    x, f0 = make_speechlike(13, nx=16000)
    ao = llsm.make_aoptions(f0_refine=0)
    pr, _ = oracle_analyze(o64, ao, FS, x, f0)
    p32 = pr.astype(np.float32).astype(np.float64)
    rep = {}
    for fs2 in (48000.0, 32000.0):
        yo, yso, yno = o64.synthesize(o64.soptions(fs2), p32, seed=5)
        b = llsm.Batch(ctx, ao, fs2, [0], [len(f0)])
        assert L.llsm_gpu_batch_set_fnyq(b.h, FS / 2) == 0
        b.upload_params(params_to_gpu_rows(pr))
        b.synthesize(llsm.make_soptions(fs2), seed=5)
        ctx.sync()
        y, ys, yn = b.download(llsm.A_Y), b.download(llsm.A_YSIN), b.download(llsm.A_YNOISE)
        b.close()
        assert len(y) == len(yo)
        m = dict(ysin=rel_rms(ys, yso), ynoise=rel_rms(yn, yno), y=rel_rms(y, yo))
        rep[str(int(fs2))] = m
        for k, v in m.items():
            assert v <= SYN_TOL, (fs2, k, v)
        # drop-in: a 44.1 kHz chunk through llsm_synthesize with options->fs = fs2 (the reference accepts it)
        ch = chunk_from_oracle(L, ao, pr, FS)
        so = llsm.make_soptions(fs2)
        L.llsm_gpu_set_default_seed(77)
        out = L.llsm_synthesize(C.byref(so), ch)
        assert bool(out), L.llsm_gpu_last_error()
        assert out.contents.ny == len(yo) and abs(out.contents.fs - fs2) < 1e-3
        yd = np.ctypeslib.as_array(out.contents.y_sin, (out.contents.ny,)).copy()
        assert rel_rms(yd, yso) <= SYN_TOL
        L.llsm_delete_output(out); L.llsm_delete_chunk(ch)
    report("synthesis_other_rate", rep)


def test_degenerate_batches(ctx):
    """Frameless batch: x_res = x.  Frames over empty signals: the constant rows the path gives on silence."""
    ao = llsm.make_aoptions(f0_refine=0)
    x = make_utterance(3, 150.0, nx=4000)
    b = llsm.Batch(ctx, ao, FS, [len(x)], [0])
    b.upload(llsm.A_X, x); b.analyze(); ctx.sync()
    assert np.array_equal(b.download(llsm.A_XRES), x)
    b.close()
    f0 = np.array([0, 150, 150, 0], np.float32)
    b = llsm.Batch(ctx, ao, FS, [0], [4])
    b.upload(llsm.A_F0, f0); b.analyze(); ctx.sync()
    g = b.download_params()
    floor_db = 10 * np.log10(np.exp(np.log(1e-10) + 0.57721566) * 44100.0 / FS + 1e-12)
    assert np.allclose(g[llsm.A_PSD], floor_db, atol=1e-3) and np.all(g[llsm.A_HAS_PSDRES] == 1)
    assert np.array_equal(g[llsm.A_NHAR], [0, 100, 100, 0]) and np.array_equal(g[llsm.A_NHAR_E], [0, 4, 4, 0])
    assert not np.any(g[llsm.A_AMPL]) and not np.any(g[llsm.A_EDC]) and not np.any(g[llsm.A_PSDRES])
    # ... and a zero signal of real length gives the same PSD rows away from the Kalman edges
    b2, g2, _ = gpu_analyze(ctx, ao, FS, [np.zeros(8000, np.float32)], [np.full(30, 150.0, np.float32)])
    assert np.allclose(g2[llsm.A_PSD][10:20], floor_db, atol=1e-2)
    b.close(); b2.close()


def test_batch_rejects_mixed_confs(ctx, o64):
    """llsm_synthesize_batch must compare FNYQ and every CHANFREQ entry across the chunks."""
    L = llsm.load()
    L.llsm_synthesize_batch.argtypes = [C.POINTER(llsm.SOptions), C.POINTER(C.POINTER(llsm.Chunk)), C.c_int,
                                        C.POINTER(C.POINTER(llsm.Output))]
    x, f0 = make_speechlike(2, nx=6000)
    ao = llsm.make_aoptions(f0_refine=0)
    pr, _ = oracle_analyze(o64, ao, FS, x, f0)
    a = chunk_from_oracle(L, ao, pr, FS)
    so = llsm.make_soptions(FS)
    outs = (C.POINTER(llsm.Output) * 2)()
    for what in ("fnyq", "chanfreq"):
        bch = L.llsm_copy_chunk(a)
        if what == "fnyq":
            C.cast(L.llsm_container_get(bch.contents.conf, llsm.CONF_FNYQ), llsm.P_fp)[0] = 24000.0
        else:
            C.cast(L.llsm_container_get(bch.contents.conf, llsm.CONF_CHANFREQ), llsm.P_fp)[1] = 4500.0
        arr = (C.POINTER(llsm.Chunk) * 2)(a, bch)
        assert L.llsm_synthesize_batch(C.byref(so), arr, 2, outs) != 0
        assert b"share" in L.llsm_gpu_last_error()
        L.llsm_delete_chunk(bch)
    arr = (C.POINTER(llsm.Chunk) * 2)(a, a)
    assert L.llsm_synthesize_batch(C.byref(so), arr, 2, outs) == 0
    L.llsm_delete_output(outs[0]); L.llsm_delete_output(outs[1]); L.llsm_delete_chunk(a)


CONVENTIONS = dict(hann_periodic=(0, 1), moving_avg_half=(3, 1), filtfilt_pad=(15, 12), interp1u_exclusive=(0, 1), kalman_init=(0, 1))


def test_convention_switches_move_product_and_oracle_together(ctx, o64):
    """The ciglet conventions the reference cannot confirm are switches shared by the
    kernels and the oracle: with every switch at its non-default value parity holds exactly as at the
    defaults, and each switch really changes the result."""
    L = llsm.load()
    x, f0 = make_speechlike(17, nx=14000)
    ao = llsm.make_aoptions(f0_refine=0)

    def run_gpu():
        c2 = llsm.Context(0)
        b, g, xres = gpu_analyze(c2, ao, FS, [x], [f0])
        b.synthesize(llsm.make_soptions(FS), seed=9); c2.sync()
        out = (g, xres, b.download(llsm.A_YSIN), b.download(llsm.A_YNOISE))
        b.close(); c2.close()
        return out

    def run_oracle():
        pr, xr = oracle_analyze(o64, ao, FS, x, f0)
        p32 = pr.astype(np.float32).astype(np.float64)
        return pr, xr, p32

    base_g, _, base_ys, base_yn = run_gpu()
    try:
        for name, (dflt, alt) in CONVENTIONS.items():
            assert L.llsm_gpu_get_convention(name.encode()) == dflt
            assert L.llsm_gpu_set_convention(name.encode(), alt) == 0
            o64.set_convention(name, alt)
        assert L.llsm_gpu_set_convention(b"hann_periodic", 7) != 0            # out-of-range values are refused
        g, xres, ys, yn = run_gpu()
        pr, xr, p32 = run_oracle()
        m = analysis_metrics(g, slice(0, len(f0)), pr, xres, xr)
        # synthesis of the oracle's parameters on both sides
        b = llsm.Batch(ctx, ao, FS, [0], [len(f0)])
        b.upload_params(params_to_gpu_rows(pr)); b.synthesize(llsm.make_soptions(FS), seed=5); ctx.sync()
        ys2, yn2 = b.download(llsm.A_YSIN), b.download(llsm.A_YNOISE); b.close()
        yo, yso, yno = o64.synthesize(o64.soptions(FS), p32, seed=5)
        m.update(ysin_rel_rms=rel_rms(ys2, yso), ynoise_rel_rms=rel_rms(yn2, yno))
        report("conventions_alt", m)
        assert_contract(m, None, "conventions_alt")
        assert m["ysin_rel_rms"] <= SYN_TOL and m["ynoise_rel_rms"] <= SYN_TOL, m
        # and the switches are not no-ops: PSD rows (interp1u / filtfilt), waveforms (Hann, moving average) move
        assert np.abs(g[llsm.A_PSD] - base_g[llsm.A_PSD]).max() > 1e-3
        assert np.abs(g[llsm.A_EDC] - base_g[llsm.A_EDC]).max() > 0
        assert rel_rms(ys, base_ys) > 1e-4 and rel_rms(yn, base_yn) > 1e-3
    finally:
        for name, (dflt, alt) in CONVENTIONS.items():
            L.llsm_gpu_set_convention(name.encode(), dflt); o64.set_convention(name, dflt)


@pytest.mark.parametrize("virtual_devices", [0, 2, 8])
def test_fanout_blocks_and_workers_do_not_change_results(ctx, monkeypatch, virtual_devices):
    """llsm_analyze_batch / llsm_synthesize_batch through the worker pool (2 workers on this device, blocks of 3
    utterances, page-locked staging) give exactly what one worker with one block gives: analysis rows bit-identical,
    synthesis identical for the same call seed (utterance u always draws from seed + u).
    virtual_devices = 2: LLSM_GPU_VIRTUAL_DEVICES makes the one GPU of the box two LOGICAL devices (own contexts,
    streams, worker pools), so the multi-device branch of the fan-out -- LLSM_GPU_DEVICES=all, workers spread over
    devices, the queue balancing between them -- executes for real."""
    L = llsm.load()
    if virtual_devices:
        monkeypatch.setenv("LLSM_GPU_VIRTUAL_DEVICES", str(virtual_devices))
        assert L.llsm_gpu_device_count() == virtual_devices
    AB = L.llsm_analyze_batch
    AB.argtypes = [C.POINTER(llsm.AOptions), C.POINTER(llsm.P_fp), llsm.P_int, C.c_float, C.POINTER(llsm.P_fp), llsm.P_int,
                   C.c_int, C.POINTER(C.POINTER(llsm.Chunk)), C.POINTER(llsm.P_fp)]
    SB = L.llsm_synthesize_batch
    SB.argtypes = [C.POINTER(llsm.SOptions), C.POINTER(C.POINTER(llsm.Chunk)), C.c_int, C.POINTER(C.POINTER(llsm.Output))]
    U = 8
    xs, f0s = [], []
    for u in range(U):
        x, f0 = make_speechlike(60 + u, nx=7000 + 900 * u)
        xs.append(np.ascontiguousarray(x, np.float32)); f0s.append(np.ascontiguousarray(f0, np.float32))
    ao = llsm.make_aoptions(f0_refine=0)
    so = llsm.make_soptions(FS)
    nx = np.array([len(x) for x in xs], np.int32); nf = np.array([len(f) for f in f0s], np.int32)
    xp = (llsm.P_fp * U)(*[x.ctypes.data_as(llsm.P_fp) for x in xs]); fp_ = (llsm.P_fp * U)(*[f.ctypes.data_as(llsm.P_fp) for f in f0s])

    def run(devices, workers, block):
        L.llsm_gpu_set_fanout(devices, workers, block)
        chunks = (C.POINTER(llsm.Chunk) * U)(); xap = (llsm.P_fp * U)()
        assert AB(C.byref(ao), xp, nx.ctypes.data_as(llsm.P_int), FS, fp_, nf.ctypes.data_as(llsm.P_int), U, chunks, xap) == 0, L.llsm_gpu_last_error()
        L.llsm_gpu_set_default_seed(555)
        outs = (C.POINTER(llsm.Output) * U)()
        assert SB(C.byref(so), chunks, U, outs) == 0, L.llsm_gpu_last_error()
        res = []
        for u in range(U):
            fr = chunks[u].contents.frames[int(nf[u]) // 2]
            hm = C.cast(L.llsm_container_get(fr, llsm.FRAME_HM), C.POINTER(llsm.HMFrame))
            nm = C.cast(L.llsm_container_get(fr, llsm.FRAME_NM), C.POINTER(llsm.NMFrame)).contents
            a = np.ctypeslib.as_array(hm.contents.ampl, (hm.contents.nhar,)).copy() if hm else np.zeros(0)
            res.append((a, np.ctypeslib.as_array(nm.psd, (nm.npsd,)).copy(), np.ctypeslib.as_array(xap[u], (int(nx[u]),)).copy(),
                        np.ctypeslib.as_array(outs[u].contents.y, (outs[u].contents.ny,)).copy()))
            L.llsm_delete_output(outs[u]); L.llsm_delete_chunk(chunks[u])
        return res

    try:
        ref = run(1, 1, 1000)
        # (8 logical devices: the shape of one node -- all of them with one worker each and blocks of one utterance, so
        # every device gets work; then four of them with two workers)
        plans = {0: ((1, 2, 3), (0, 3, 1)), 2: ((0, 2, 1), (2, 1, 3)), 8: ((0, 1, 1), (4, 2, 1))}[virtual_devices]
        for devices, workers, block in plans:
            got = run(devices, workers, block)
            for u in range(U):
                for k in range(4):
                    assert np.array_equal(got[u][k], ref[u][k]), (devices, workers, block, u, k)
    finally:
        L.llsm_gpu_set_fanout(-1, -1, -1)


def test_dropin_calls_from_concurrent_host_threads():
    """The reference is re-entrant (SURVEY 8b "Threading": no internal state besides libc rand()); hosts call
    llsm_analyze / llsm_synthesize from several threads.  Here the calls share the default context and its device
    memory cache: results of 6 threads x 3 rounds must equal the single-threaded ones bit for bit (analysis) and,
    with the seed pinned per call, sample for sample (synthesis)."""
    import threading
    L = llsm.load()
    L.llsm_analyze.restype = C.POINTER(llsm.Chunk)
    L.llsm_synthesize.restype = C.POINTER(llsm.Output)
    ao = llsm.make_aoptions(f0_refine=0)
    so = llsm.make_soptions(FS)
    cases = []
    for k in range(6):
        x, f0 = make_speechlike(400 + k, nx=9000 + 1100 * k)
        cases.append((x, f0.astype(np.float32)))

    def run(k):
        x, f0 = cases[k]
        f = f0.copy()
        ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f.ctypes.data_as(llsm.P_fp), len(f), None)
        assert ch, L.llsm_gpu_last_error()
        nm = C.cast(L.llsm_container_get(ch.contents.frames[len(f) // 2], llsm.FRAME_NM), C.POINTER(llsm.NMFrame)).contents
        psd = np.ctypeslib.as_array(nm.psd, (nm.npsd,)).copy()
        out = L.llsm_synthesize(C.byref(so), ch)
        assert out, L.llsm_gpu_last_error()
        ys = np.ctypeslib.as_array(out.contents.y_sin, (out.contents.ny,)).copy()
        L.llsm_delete_output(out); L.llsm_delete_chunk(ch)
        return psd, ys

    ref = [run(k) for k in range(6)]
    got = [[None] * 3 for _ in range(6)]
    errs = []

    def worker(k):
        try:
            for r in range(3):
                got[k][r] = run(k)
        except Exception as e:                                   # noqa: BLE001
            errs.append((k, repr(e)))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    for k in range(6):
        for r in range(3):
            assert np.array_equal(got[k][r][0], ref[k][0]), (k, r, "psd")
            assert np.array_equal(got[k][r][1], ref[k][1]), (k, r, "y_sin")


def test_rt_buffers_from_concurrent_host_threads(o64):
    """Several llsmrt buffers fed from several host threads at once (each thread owns its buffer; they share the
    default context's stream): the deterministic part of every stream equals the one produced alone."""
    import threading
    L = llsm.load()
    ao = llsm.make_aoptions(f0_refine=0)
    so = llsm.make_soptions(FS)
    chunks = []
    for k in range(4):
        x, f0 = make_speechlike(500 + k, nx=12000 + 1500 * k)
        pr, _ = oracle_analyze(o64, ao, FS, x, f0.astype(np.float32))
        chunks.append((chunk_from_oracle(L, ao, pr, FS), pr.nfrm))
    ref = [rt_run(L, so, ch, n)[0] for ch, n in chunks]
    got = [None] * 4; errs = []

    def worker(k):
        try:
            for _ in range(2):
                got[k] = rt_run(L, so, chunks[k][0], chunks[k][1])[0]
        except Exception as e:                                   # noqa: BLE001
            errs.append((k, repr(e)))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    for ch, _ in chunks:
        L.llsm_delete_chunk(ch)
    assert not errs, errs
    for k in range(4):
        assert np.array_equal(got[k], ref[k]), k


def test_analysis_overlap_does_not_change_results(ctx):
    """llsm_gpu_analysis_overlap: the Kalman smoother on a second stream beside the band filter -- every analysed row must
    be bit-identical with the single-stream order, also when the next call (synthesis, another analysis) follows at once."""
    from conftest import make_speechlike
    from gpu_common import gpu_analyze
    L = llsm.load()
    xs, f0s = [], []
    for k in range(6):
        x, f0 = make_speechlike(60 + k, nx=20000 + 1500 * k); xs.append(x); f0s.append(f0.astype(np.float32))
    ao = llsm.make_aoptions(f0_refine=0)
    prev = L.llsm_gpu_analysis_overlap(-1)
    try:
        rows = {}
        for on in (0, 1):
            L.llsm_gpu_analysis_overlap(on)
            b, g, xres = gpu_analyze(ctx, ao, FS, xs, f0s)
            b.analyze(); b.analyze(); ctx.sync()                       # back-to-back analyses reuse the planes the smoother reads
            g2 = {k: b.download(k) for k in (llsm.A_PSD, llsm.A_PSDRES, llsm.A_EDC, llsm.A_EENV_AMPL, llsm.A_AMPL)}
            rows[on] = (g, g2); b.close()
        for k in (llsm.A_PSD, llsm.A_PSDRES, llsm.A_EDC, llsm.A_EENV_AMPL, llsm.A_EENV_PHSE, llsm.A_AMPL, llsm.A_PHSE, llsm.A_HAS_PSDRES):
            assert np.array_equal(rows[0][0][k], rows[1][0][k]), k
        for k, v in rows[1][1].items():
            assert np.array_equal(v, rows[0][1][k]) and np.array_equal(v, rows[1][0][k]), k
    finally:
        L.llsm_gpu_analysis_overlap(prev)


def test_kept_batches_do_not_change_results():
    """A worker keeps the device batch of its last block and reuses it for an equally shaped one (capi.cpp worker_batch):
    the same utterance through a fresh and through a reused batch, another utterance of the same shape after it, a
    differently shaped one in between and llsm_gpu_release_cached_batches() must all give the rows and samples of a first
    call -- every PSD row, every harmonic amplitude, y_sin and (seed pinned) y_noise bit for bit."""
    L = llsm.load()
    L.llsm_analyze.restype = C.POINTER(llsm.Chunk)
    L.llsm_synthesize.restype = C.POINTER(llsm.Output)
    ao = llsm.make_aoptions(f0_refine=0)
    so = llsm.make_soptions(FS)
    xa, fa = make_speechlike(700, nx=15000); xb, fb = make_speechlike(701, nx=15000)      # two utterances of one shape
    xc, fc = make_speechlike(702, nx=11000)                                               # and one of another
    assert len(fa) == len(fb) != len(fc)

    def run(x, f0):
        f = f0.astype(np.float32).copy()
        ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f.ctypes.data_as(llsm.P_fp), len(f), None)
        assert ch, L.llsm_gpu_last_error()
        rows = []
        for i in range(len(f)):
            fr = ch.contents.frames[i]
            nm = C.cast(L.llsm_container_get(fr, llsm.FRAME_NM), C.POINTER(llsm.NMFrame)).contents
            rows.append(np.ctypeslib.as_array(nm.psd, (nm.npsd,)).copy())
            hm = L.llsm_container_get(fr, llsm.FRAME_HM)
            if hm:
                h = C.cast(hm, C.POINTER(llsm.HMFrame)).contents
                rows.append(np.ctypeslib.as_array(h.ampl, (h.nhar,)).copy())
        L.llsm_gpu_set_default_seed(77)
        out = L.llsm_synthesize(C.byref(so), ch)
        assert out, L.llsm_gpu_last_error()
        ny = out.contents.ny
        ys = np.ctypeslib.as_array(out.contents.y_sin, (ny,)).copy(); yn = np.ctypeslib.as_array(out.contents.y_noise, (ny,)).copy()
        L.llsm_delete_output(out); L.llsm_delete_chunk(ch)
        return np.concatenate(rows), ys, yn

    def same(p, q, what):
        for k in range(3):
            assert np.array_equal(p[k], q[k]), (what, k)

    L.llsm_gpu_release_cached_batches()
    a1 = run(xa, fa)                        # fresh batches
    a2 = run(xa, fa)                        # reused
    b1 = run(xb, fb)                        # reused, other data
    L.llsm_gpu_release_cached_batches()
    b2 = run(xb, fb)                        # fresh again
    c1 = run(xc, fc)                        # another shape replaces the kept batches
    a3 = run(xa, fa)                        # and back
    same(a1, a2, "reused batch"); same(b1, b2, "reused batch, other data"); same(a1, a3, "after another shape")
    assert np.abs(a1[1]).max() > 0.01 and np.abs(c1[2]).max() > 0 and not np.array_equal(a1[1], b1[1])
    # a convention that a batch bakes into its tables when it is created (the synthesis window): a kept batch of the old
    # convention must not serve a call under the new one
    prev = L.llsm_gpu_get_convention(b"hann_periodic")
    try:
        assert L.llsm_gpu_set_convention(b"hann_periodic", 1 - prev) == 0
        h1 = run(xa, fa)                    # same shape as the kept batches, other convention
        L.llsm_gpu_release_cached_batches()
        h2 = run(xa, fa)                    # fresh batches under that convention
        same(h1, h2, "convention changed between equally shaped calls")
        assert not np.array_equal(h1[1], a1[1])
    finally:
        L.llsm_gpu_set_convention(b"hann_periodic", prev)
    same(run(xa, fa), a1, "convention restored")


def test_transfer_many_matches_the_single_array_calls(ctx):
    """llsm_gpu_batch_transfer_many: several arrays enqueued and waited for once -- the same bytes as one call per array in
    both directions, and the single-array calls' refusal of a wrong id or byte count (nothing is copied then)."""
    L = llsm.load()
    L.llsm_gpu_batch_transfer_many.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    x, f0 = make_speechlike(810, nx=9000); f0 = f0.astype(np.float32)
    b = llsm.Batch(ctx, llsm.make_aoptions(f0_refine=0), FS, [len(x)], [len(f0)])
    try:
        ids = (C.c_int * 2)(llsm.A_X, llsm.A_F0)
        ptrs = (C.c_void_p * 2)(x.ctypes.data, f0.ctypes.data)
        sizes = (C.c_size_t * 2)(x.nbytes, f0.nbytes)
        assert L.llsm_gpu_batch_transfer_many(b.h, 1, 2, ids, ptrs, sizes) == 0, L.llsm_gpu_last_error()
        assert np.array_equal(b.download(llsm.A_X), x) and np.array_equal(b.download(llsm.A_F0), f0)
        b.analyze(); ctx.sync()
        one = {a: b.download(a) for a in (llsm.A_PSD, llsm.A_AMPL, llsm.A_NHAR)}
        outs = [np.empty_like(one[a]) for a in (llsm.A_PSD, llsm.A_AMPL, llsm.A_NHAR)]
        ids3 = (C.c_int * 3)(llsm.A_PSD, llsm.A_AMPL, llsm.A_NHAR)
        ptrs3 = (C.c_void_p * 3)(*[o.ctypes.data for o in outs]); sizes3 = (C.c_size_t * 3)(*[o.nbytes for o in outs])
        assert L.llsm_gpu_batch_transfer_many(b.h, 0, 3, ids3, ptrs3, sizes3) == 0, L.llsm_gpu_last_error()
        for o, a in zip(outs, (llsm.A_PSD, llsm.A_AMPL, llsm.A_NHAR)):
            assert np.array_equal(o, one[a]), a
        outs[0][:] = -7.0
        bad = (C.c_size_t * 3)(outs[0].nbytes, outs[1].nbytes - 4, outs[2].nbytes)      # one wrong size: the whole call is refused
        assert L.llsm_gpu_batch_transfer_many(b.h, 0, 3, ids3, ptrs3, bad) != 0
        assert (outs[0] == -7.0).all()
        badid = (C.c_int * 1)(9999)
        assert L.llsm_gpu_batch_transfer_many(b.h, 0, 1, badid, ptrs3, sizes3) != 0
    finally:
        b.close()


@pytest.mark.parametrize("refine", [0, 1])
def test_batch_frames_over_packed_records_equal_the_drop_in_frames(refine):
    """llsm_analyze_batch (round 5): the device packs each frame's rows into one record and copies an utterance's records
    straight into the chunk's registered slab; the host lays the frame structs over them (model.cpp
    llsm_frames_packed_finish).  Against llsm_analyze (heap frames, staged rows) on the same utterances: every frame member
    -- F0, harmonic counts and rows, PSD, band energies, envelope harmonics, the PSDRES fparray and its length -- bit for
    bit, the refined F0 written back to the caller; then the reference's object operations on such frames (deep copy, in-place
    growth beyond the record, attach, single-frame deletion, chunk deletion) and the synthesis of both chunk sets; no slab
    left at the end."""
    L = llsm.load()
    AB = L.llsm_analyze_batch
    AB.argtypes = [C.POINTER(llsm.AOptions), C.POINTER(llsm.P_fp), llsm.P_int, C.c_float, C.POINTER(llsm.P_fp), llsm.P_int,
                   C.c_int, C.POINTER(C.POINTER(llsm.Chunk)), C.POINTER(llsm.P_fp)]
    L.llsm_slab_stats.argtypes = [C.POINTER(C.c_longlong)] * 3
    L.llsm_copy_hmframe_inplace.argtypes = [C.POINTER(llsm.HMFrame), C.POINTER(llsm.HMFrame)]
    U = 5
    xs, f0s = [], []
    for u in range(U):
        x, f0 = make_speechlike(80 + u, nx=5000 + 1700 * u)
        xs.append(np.ascontiguousarray(x, np.float32)); f0s.append(np.ascontiguousarray(f0, np.float32))
    ao = llsm.make_aoptions(f0_refine=refine, maxnhar_e=3)
    nx = np.array([len(x) for x in xs], np.int32); nf = np.array([len(f) for f in f0s], np.int32)
    f0_b = [f.copy() for f in f0s]; f0_s = [f.copy() for f in f0s]
    xp = (llsm.P_fp * U)(*[x.ctypes.data_as(llsm.P_fp) for x in xs]); fp_ = (llsm.P_fp * U)(*[f.ctypes.data_as(llsm.P_fp) for f in f0_b])
    live0 = C.c_longlong(0); L.llsm_slab_stats(C.byref(live0), None, None)
    chunks = (C.POINTER(llsm.Chunk) * U)(); xap = (llsm.P_fp * U)()
    assert AB(C.byref(ao), xp, nx.ctypes.data_as(llsm.P_int), FS, fp_, nf.ctypes.data_as(llsm.P_int), U, chunks, xap) == 0, L.llsm_gpu_last_error()
    live = C.c_longlong(0); L.llsm_slab_stats(C.byref(live), None, None)
    assert live.value - live0.value == U                                        # one slab per chunk
    singles = []
    for u in range(U):
        ch = L.llsm_analyze(C.byref(ao), xs[u].ctypes.data_as(llsm.P_fp), len(xs[u]), FS, f0_s[u].ctypes.data_as(llsm.P_fp), len(f0_s[u]), None)
        assert ch, L.llsm_gpu_last_error()
        singles.append(ch)
        assert np.array_equal(f0_b[u], f0_s[u])                                 # (refined) F0 written back the same
        if refine and np.any(f0s[u] > 0):
            assert not np.array_equal(f0_b[u], f0s[u])

    def members(fr):
        f0 = C.cast(L.llsm_container_get(fr, llsm.FRAME_F0), llsm.P_fp)[0]
        hm = C.cast(L.llsm_container_get(fr, llsm.FRAME_HM), C.POINTER(llsm.HMFrame)).contents
        nm = C.cast(L.llsm_container_get(fr, llsm.FRAME_NM), C.POINTER(llsm.NMFrame)).contents
        rp = L.llsm_container_get(fr, llsm.FRAME_PSDRES)
        out = [np.float32(f0), hm.nhar, np.ctypeslib.as_array(hm.ampl, (max(hm.nhar, 1),))[:hm.nhar].copy(),
               np.ctypeslib.as_array(hm.phse, (max(hm.nhar, 1),))[:hm.nhar].copy(), nm.npsd, nm.nchannel,
               np.ctypeslib.as_array(nm.psd, (nm.npsd,)).copy(), np.ctypeslib.as_array(nm.edc, (nm.nchannel,)).copy()]
        for c in range(nm.nchannel):
            e = nm.eenv[c].contents
            out += [e.nhar, np.ctypeslib.as_array(e.ampl, (max(e.nhar, 1),))[:e.nhar].copy(), np.ctypeslib.as_array(e.phse, (max(e.nhar, 1),))[:e.nhar].copy()]
        out.append(bool(rp))
        if rp:
            n = L.llsm_fparray_length(C.cast(rp, llsm.P_fp))
            out += [n, np.ctypeslib.as_array(C.cast(rp, llsm.P_fp), (n,)).copy()]
        return out

    for u in range(U):
        for i in range(int(nf[u])):
            a, b_ = members(chunks[u].contents.frames[i]), members(singles[u].contents.frames[i])
            assert len(a) == len(b_)
            for va, vb in zip(a, b_):
                assert np.array_equal(va, vb), (u, i)
    # the reference's object operations on frames that lie over the records
    cp = L.llsm_copy_chunk(chunks[0])
    fr = chunks[1].contents.frames[int(nf[1]) // 2]
    hm = C.cast(L.llsm_container_get(fr, llsm.FRAME_HM), C.POINTER(llsm.HMFrame))
    big = L.llsm_create_hmframe(400)                                           # beyond the record's maxnhar = 100 values
    L.llsm_copy_hmframe_inplace(hm, big); L.llsm_delete_hmframe(big)
    assert hm.contents.nhar == 400
    L.llsm_container_attach_(chunks[2].contents.frames[1], llsm.FRAME_PBPSYN, C.cast(L.llsm_create_int(1), C.c_void_p),
                             C.cast(L.llsm_delete_int, C.c_void_p), C.cast(L.llsm_copy_int, C.c_void_p))
    L.llsm_delete_container(chunks[3].contents.frames[0]); chunks[3].contents.frames[0] = L.llsm_copy_container(chunks[3].contents.frames[1])
    # synthesis of both sets (chunks 1 .. 3 were edited: compare 0 and 4)
    so = llsm.make_soptions(FS)
    for u in (0, 4):
        L.llsm_gpu_set_default_seed(300 + u); oa = L.llsm_synthesize(C.byref(so), chunks[u])
        L.llsm_gpu_set_default_seed(300 + u); ob = L.llsm_synthesize(C.byref(so), singles[u])
        ya = np.ctypeslib.as_array(oa.contents.y, (oa.contents.ny,)).copy(); yb = np.ctypeslib.as_array(ob.contents.y, (ob.contents.ny,)).copy()
        L.llsm_delete_output(oa); L.llsm_delete_output(ob)
        assert np.array_equal(ya, yb), u
    for u in range(U):
        L.llsm_delete_chunk(chunks[u]); L.llsm_delete_chunk(singles[u])
        if xap[u]:
            libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
            libc.free(C.cast(xap[u], C.c_void_p))
    L.llsm_delete_chunk(cp)
    L.llsm_slab_stats(C.byref(live), None, None)
    assert live.value == live0.value


def test_batch_synthesis_reads_untouched_chunks_where_they_lie():
    """llsm_synthesize_batch (round 5): chunks whose frames still lie over the records llsm_analyze_batch landed in their
    page-locked slabs are not flattened -- the device reads the records in place (k_unpack_frames) and writes the waveforms into
    page-locked pooled outputs (k_scatter_outputs).  Against deep copies of the same chunks (ordinary heap frames: the
    staged path) with the same call seed: bit-identical y / y_sin / y_noise -- also after VALUES were edited through the
    structs (amplitudes scaled, a phase row shifted, F0 changed, a harmonic count reduced: all read where they lie), and
    after one chunk's frame was replaced (that block falls back to the staged path).  Outputs are deleted in any order."""
    L = llsm.load()
    AB = L.llsm_analyze_batch
    AB.argtypes = [C.POINTER(llsm.AOptions), C.POINTER(llsm.P_fp), llsm.P_int, C.c_float, C.POINTER(llsm.P_fp), llsm.P_int,
                   C.c_int, C.POINTER(C.POINTER(llsm.Chunk)), C.POINTER(llsm.P_fp)]
    SB = L.llsm_synthesize_batch
    SB.argtypes = [C.POINTER(llsm.SOptions), C.POINTER(C.POINTER(llsm.Chunk)), C.c_int, C.POINTER(C.POINTER(llsm.Output))]
    U = 6
    xs, f0s = [], []
    for u in range(U):
        x, f0 = make_speechlike(90 + u, nx=6000 + 1300 * u)
        xs.append(np.ascontiguousarray(x, np.float32)); f0s.append(np.ascontiguousarray(f0, np.float32))
    ao = llsm.make_aoptions(f0_refine=0)
    so = llsm.make_soptions(FS)
    nx = np.array([len(x) for x in xs], np.int32); nf = np.array([len(f) for f in f0s], np.int32)
    xp = (llsm.P_fp * U)(*[x.ctypes.data_as(llsm.P_fp) for x in xs]); fp_ = (llsm.P_fp * U)(*[f.ctypes.data_as(llsm.P_fp) for f in f0s])
    L.llsm_gpu_set_fanout(1, 2, 3)                              # two blocks of three
    try:
        chunks = (C.POINTER(llsm.Chunk) * U)()
        assert AB(C.byref(ao), xp, nx.ctypes.data_as(llsm.P_int), FS, fp_, nf.ctypes.data_as(llsm.P_int), U, chunks, None) == 0, L.llsm_gpu_last_error()

        def synth(cs, seed):
            L.llsm_gpu_set_default_seed(seed)
            outs = (C.POINTER(llsm.Output) * U)()
            assert SB(C.byref(so), cs, U, outs) == 0, L.llsm_gpu_last_error()
            res = [tuple(np.ctypeslib.as_array(getattr(outs[u].contents, k), (outs[u].contents.ny,)).copy() for k in ("y", "y_sin", "y_noise"))
                   for u in range(U)]
            for u in reversed(range(U)):
                L.llsm_delete_output(outs[u])
            return res

        def copies():
            cs = (C.POINTER(llsm.Chunk) * U)()
            for u in range(U):
                cs[u] = L.llsm_copy_chunk(chunks[u])
            return cs

        def same(a, b, what):
            for u in range(U):
                for k in range(3):
                    assert len(a[u][k]) == len(b[u][k]) and np.array_equal(a[u][k], b[u][k]), (what, u, k)

        cp = copies()
        a, b = synth(chunks, 700), synth(cp, 700)
        same(a, b, "untouched")
        assert float(np.sqrt(np.mean(a[2][0] ** 2))) > 0.01
        # values edited through the structs, on the slab frames AND on the copies
        for cs in (chunks, cp):
            fr = cs[1].contents.frames[7]                                    # (a voiced frame: 0 .. 5 and the middle of these utterances are not)
            hm = C.cast(L.llsm_container_get(fr, llsm.FRAME_HM), C.POINTER(llsm.HMFrame)).contents
            for k in range(hm.nhar):
                hm.ampl[k] *= 2.0
            if hm.nhar > 5:
                hm.nhar = hm.nhar - 3
            L.llsm_frame_phaseshift.argtypes = [C.POINTER(llsm.Container), C.c_float]
            L.llsm_frame_phaseshift(cs[1].contents.frames[6], 0.4)
            C.cast(L.llsm_container_get(cs[4].contents.frames[int(nf[4]) // 4], llsm.FRAME_F0), llsm.P_fp)[0] *= 1.01
        a2, b2 = synth(chunks, 701), synth(cp, 701)
        same(a2, b2, "edited values")
        assert not np.array_equal(a2[1][1], a[1][1])                  # ... and the edits were heard
        # a frame replaced in chunk 3: its block is flattened the ordinary way, the other block is still read in place
        for cs in (chunks, cp):
            old = C.cast(cs[3].contents.frames[2], C.c_void_p).value      # (the address: a ctypes pointer taken from the slot is a view of the slot)
            cs[3].contents.frames[2] = L.llsm_copy_container(cs[3].contents.frames[3])
            L.llsm_delete_container(C.cast(C.c_void_p(old), C.POINTER(llsm.Container)))
        same(synth(chunks, 702), synth(cp, 702), "frame replaced")
        for u in range(U):
            L.llsm_delete_chunk(chunks[u]); L.llsm_delete_chunk(cp[u])
    finally:
        L.llsm_gpu_set_fanout(-1, -1, -1)
