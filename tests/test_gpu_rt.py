"""-m gpu: llsmrt pull loop on the GPU vs the oracle's restatement of llsmrt.c,
and vs offline synthesis (test/test-llsmrt.c:161-164)."""
import ctypes as C

import numpy as np
import pytest

import libllsm2_amd as llsm
from conftest import FS, make_speechlike
from gpu_common import oracle_analyze, rel_rms, report
from verify_utils import assert_reference_acceptance

pytestmark = pytest.mark.gpu


def chunk_from_oracle(L, ao, pr, fs):
    """oracle Params (float64) -> product llsm_chunk via llsm_flat_to_chunk."""
    conf = L.llsm_aoptions_toconf(C.byref(ao), fs / 2.0)
    C.cast(L.llsm_container_get(conf, llsm.CONF_NFRM), llsm.P_int)[0] = pr.nfrm
    ch = L.llsm_create_chunk(conf, 1)
    L.llsm_delete_container(conf)
    keep = dict(f0=pr.f0.astype(np.float32), nhar=pr.nhar.astype(np.int32), ampl=pr.ampl.astype(np.float32),
                phse=pr.phse.astype(np.float32), psd=pr.psd.astype(np.float32), psdres=pr.psdres.astype(np.float32),
                has=np.ones(pr.nfrm, np.int32), edc=pr.edc.astype(np.float32), nhe=pr.nhar_e.astype(np.int32),
                ea=np.ascontiguousarray(pr.eenv_ampl.astype(np.float32)), ep=np.ascontiguousarray(pr.eenv_phse.astype(np.float32)))
    v = llsm.FlatParams()
    v.maxnhar, v.maxnhar_e, v.npsd, v.nchannel = pr.maxnhar, pr.maxnhar_e, pr.npsd, pr.nchannel
    v.f0 = keep["f0"].ctypes.data_as(llsm.P_fp); v.nhar = keep["nhar"].ctypes.data_as(llsm.P_int)
    v.ampl = keep["ampl"].ctypes.data_as(llsm.P_fp); v.phse = keep["phse"].ctypes.data_as(llsm.P_fp)
    v.psd = keep["psd"].ctypes.data_as(llsm.P_fp); v.psdres = keep["psdres"].ctypes.data_as(llsm.P_fp)
    v.has_psdres = keep["has"].ctypes.data_as(llsm.P_int); v.edc = keep["edc"].ctypes.data_as(llsm.P_fp)
    v.nhar_e = keep["nhe"].ctypes.data_as(llsm.P_int)
    v.eenv_ampl = keep["ea"].ctypes.data_as(llsm.P_fp); v.eenv_phse = keep["ep"].ctypes.data_as(llsm.P_fp)
    assert L.llsm_flat_to_chunk(C.byref(v), 0, ch) == 0
    return ch


def rt_run(L, so, ch, nfrm, capacity=4096):
    rt = L.llsm_create_rtsynth_buffer(C.byref(so), ch.contents.conf, capacity)
    assert rt, L.llsm_gpu_last_error()
    lat = L.llsm_rtsynth_buffer_getlatency(rt)
    yp, yap = [], []
    p, ap = C.c_float(0), C.c_float(0)
    for i in range(nfrm):                                  # single-thread pattern, test-llsmrt.c:129-145
        L.llsm_rtsynth_buffer_feed(rt, ch.contents.frames[i])
        assert L.llsm_rtsynth_buffer_numoutput(rt) > 0
        while L.llsm_rtsynth_buffer_fetch_decomposed(rt, C.byref(p), C.byref(ap)):
            yp.append(p.value); yap.append(ap.value)
    L.llsm_delete_rtsynth_buffer(rt)
    return np.array(yp), np.array(yap), lat


@pytest.mark.parametrize("thop", [0.005, 128 / 44100.0])
def test_rt_matches_oracle_rt(o64, thop):
    L = llsm.load()
    x, _ = make_speechlike(3, nx=26000)
    nfrm = int(len(x) / FS / thop)
    t = np.arange(nfrm) * thop
    f0 = (150 + 40 * np.sin(2 * np.pi * 1.1 * t)).astype(np.float32)
    f0[:4] = 0; f0[nfrm // 2: nfrm // 2 + 7] = 0
    ao = llsm.make_aoptions(f0_refine=0, thop=thop)
    pr, _ = oracle_analyze(o64, ao, FS, x, f0)
    p32 = pr.astype(np.float32).astype(np.float64)
    so_o = o64.soptions(FS)
    seed = 4242
    yp_o, yap_o, lat_o = o64.rt_run(so_o, p32, capacity=4096, seed=seed)
    ch = chunk_from_oracle(L, ao, pr, FS)
    so = llsm.make_soptions(FS)
    L.llsm_gpu_set_default_seed(seed)
    yp, yap, lat = rt_run(L, so, ch, nfrm)
    L.llsm_delete_chunk(ch)
    assert lat == lat_o and len(yp) == len(yp_o)
    m = dict(latency=lat, n=len(yp), p_rel_rms=rel_rms(yp, yp_o), ap_rel_rms=rel_rms(yap, yap_o),
             p_abs_max=float(np.abs(yp - yp_o).max()), ap_abs_max=float(np.abs(yap - yap_o).max()))
    report(f"rt_thop{int(round(thop * FS))}", m)
    assert m["p_rel_rms"] <= 1e-4 and m["ap_rel_rms"] <= 1e-4, m


def test_rt_vs_offline_and_threaded(o64):
    """RT output == offline llsm_synthesize after latency alignment (deterministic part
    sample-level; total through the reference's KLD / correlation thresholds), with
    producer and consumer on separate threads (test-llsmrt.c:29-64)."""
    import threading
    L = llsm.load()
    # hop of exactly 128 samples as in test-llsmrt.c:71-86: with a fractional hop (220.5) the
    # reference's RT path windows with 2*curr_nhop = 440/442 samples against 442 offline, so
    # RT and offline are then only statistically equal, by construction of llsmrt.c.
    # The signal is the reference's own fixture (speech): the spectral-correlation threshold is
    # calibrated for speech, where two independent noise realisations barely move the STFT.
    import os
    from verify_utils import GOLDEN, read_wav
    thop = 128 / 44100.0
    x, _ = read_wav(os.path.join(GOLDEN, "arctic_a0001.wav"))
    x = np.ascontiguousarray(x[:60000]); nfrm = 60000 // 128
    f0 = np.ascontiguousarray(np.load(os.path.join(GOLDEN, "arctic_a0001_f0_hop128.npy"))[:nfrm])
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, npsd=128, maxnhar=400, maxnhar_e=5)
    f0c = f0.copy()
    ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f0c.ctypes.data_as(llsm.P_fp), nfrm, None)
    assert bool(ch), L.llsm_gpu_last_error()
    so = llsm.make_soptions(FS)
    out = L.llsm_synthesize(C.byref(so), ch)
    ny = out.contents.ny
    y_off = np.ctypeslib.as_array(out.contents.y, (ny,)).copy()
    ys_off = np.ctypeslib.as_array(out.contents.y_sin, (ny,)).copy()
    L.llsm_delete_output(out)
    rt = L.llsm_create_rtsynth_buffer(C.byref(so), ch.contents.conf, 2048)
    lat = L.llsm_rtsynth_buffer_getlatency(rt)
    assert lat == 128 + 256                              # curr_nhop + nfft/2 (llsmrt.c:568-571)
    got_p, got_ap = [], []
    done = threading.Event()

    def producer():
        for i in range(nfrm):
            L.llsm_rtsynth_buffer_feed(rt, ch.contents.frames[i])    # blocks while the ring is full
        done.set()

    def consumer():
        p, ap = C.c_float(0), C.c_float(0)
        while True:
            if L.llsm_rtsynth_buffer_fetch_decomposed(rt, C.byref(p), C.byref(ap)):
                got_p.append(p.value); got_ap.append(ap.value)
            elif done.is_set() and L.llsm_rtsynth_buffer_numoutput(rt) == 0:
                break

    tp, tc = threading.Thread(target=producer), threading.Thread(target=consumer)
    tp.start(); tc.start(); tp.join(120); tc.join(120)
    assert not tp.is_alive() and not tc.is_alive()
    L.llsm_delete_rtsynth_buffer(rt); L.llsm_delete_chunk(ch)
    yp, yap = np.array(got_p), np.array(got_ap)
    n = min(len(yp) - lat, ny) - 600
    d = yp[lat:lat + n] - ys_off[:n]
    assert np.sqrt(np.mean(d[600:] ** 2)) <= 1e-4 * np.sqrt(np.mean(ys_off[600:n] ** 2))
    assert_reference_acceptance(y_off[:n], (yp + yap)[lat:lat + n], "rt vs offline (GPU)")


def test_rt_create_rejects_bad_conf():
    L = llsm.load()
    so = llsm.make_soptions(FS)
    bare = L.llsm_create_container(12)
    assert not L.llsm_create_rtsynth_buffer(C.byref(so), bare, 4096)      # llsmrt.c:163
    L.llsm_delete_container(bare)
    ao = llsm.make_aoptions()
    conf = L.llsm_aoptions_toconf(C.byref(ao), FS / 2)
    so1 = llsm.make_soptions(FS, use_l1=1)
    assert not L.llsm_create_rtsynth_buffer(C.byref(so1), conf, 4096)
    L.llsm_delete_container(conf)


def test_rt_group_equals_single_streams(o64):
    """BASELINE.json config 4 shape: a group of lock-stepped streams fed one hop per call and
    drained with 256-sample pulls gives, per stream, exactly what a single-stream buffer with
    the same seed gives (stream s uses seed + s)."""
    L = llsm.load()
    S = 5
    thop = 0.005
    chunks, nfrm = [], None
    ao = llsm.make_aoptions(f0_refine=0, thop=thop)
    for s in range(S):
        x, _ = make_speechlike(20 + s, nx=15000)
        n = int(len(x) / FS / thop)
        t = np.arange(n) * thop
        f0 = (130 + 20 * s + 30 * np.sin(2 * np.pi * 1.3 * t + s)).astype(np.float32)
        f0[:3] = 0
        if s == 2:
            f0[:] = 0                                            # an all-unvoiced stream
        pr, _ = oracle_analyze(o64, ao, FS, x, f0)
        chunks.append(chunk_from_oracle(L, ao, pr, FS)); nfrm = n
    so = llsm.make_soptions(FS)
    seed = 900
    # reference: each stream alone
    singles = []
    for s in range(S):
        L.llsm_gpu_set_default_seed(seed + s)
        yp, yap, lat = rt_run(L, so, chunks[s], nfrm)
        singles.append((yp, yap))
    # the group, 256-sample pulls
    L.llsm_gpu_set_default_seed(seed)
    g = L.llsm_create_rtsynth_group(C.byref(so), chunks[0].contents.conf, 4096, S)
    assert g, L.llsm_gpu_last_error()
    assert L.llsm_rtsynth_group_getlatency(g) == lat
    outp = [[] for _ in range(S)]; outap = [[] for _ in range(S)]
    bp = np.zeros(256, np.float32); bap = np.zeros(256, np.float32)
    FrameArr = C.POINTER(llsm.Container) * S
    for i in range(nfrm):
        fr = FrameArr(*[chunks[s].contents.frames[i] for s in range(S)])
        L.llsm_rtsynth_group_feed(g, fr)
        for s in range(S):
            while L.llsm_rtsynth_group_numoutput(g, s) >= 256 or (i == nfrm - 1 and L.llsm_rtsynth_group_numoutput(g, s) > 0):
                n = L.llsm_rtsynth_group_fetch(g, s, bp.ctypes.data_as(llsm.P_fp), bap.ctypes.data_as(llsm.P_fp), 256)
                outp[s].append(bp[:n].copy()); outap[s].append(bap[:n].copy())
    L.llsm_delete_rtsynth_group(g)
    # the same group again, every stream pulled by one llsm_rtsynth_group_fetch_all per 256 samples: identical samples
    L.llsm_gpu_set_default_seed(seed)
    g = L.llsm_create_rtsynth_group(C.byref(so), chunks[0].contents.conf, 4096, S)
    allp = [[] for _ in range(S)]; allap = [[] for _ in range(S)]
    bp2 = np.zeros((S, 256), np.float32); bap2 = np.zeros((S, 256), np.float32); cnt = (C.c_int * S)()
    for i in range(nfrm):
        fr = FrameArr(*[chunks[s].contents.frames[i] for s in range(S)])
        L.llsm_rtsynth_group_feed(g, fr)
        while L.llsm_rtsynth_group_numoutput(g, 0) >= 256 or (i == nfrm - 1 and L.llsm_rtsynth_group_numoutput(g, 0) > 0):
            least = L.llsm_rtsynth_group_fetch_all(g, bp2.ctypes.data_as(llsm.P_fp), bap2.ctypes.data_as(llsm.P_fp), 256, cnt)
            assert least == min(cnt) and least > 0
            for s in range(S):
                allp[s].append(bp2[s, :cnt[s]].copy()); allap[s].append(bap2[s, :cnt[s]].copy())
    L.llsm_delete_rtsynth_group(g)
    for s in range(S):
        assert np.array_equal(np.concatenate(allp[s]), np.concatenate(outp[s])), s
        assert np.array_equal(np.concatenate(allap[s]), np.concatenate(outap[s])), s
    for s in range(S):
        yp, yap = np.concatenate(outp[s]), np.concatenate(outap[s])
        assert len(yp) == len(singles[s][0])
        assert np.array_equal(yp, singles[s][0].astype(np.float32)), s
        # the noise filter packs two streams into one complex FFT, so a stream's rounding
        # depends on its pair partner: equal to float32 rounding, not bit-for-bit
        assert rel_rms(yap, singles[s][1]) < 2e-6, s
    for ch in chunks:
        L.llsm_delete_chunk(ch)


def test_rt_large_group_packed_by_helper_threads(o64):
    """Groups of >= 128 streams pack their frames on the feeding thread AND helper threads (rt.cpp RtPackPool; round 6): stream s
    of a 136-stream group whose streams cycle through 5 chunks carries exactly the harmonic part, and within float32 rounding
    the noise part (pair partner in the complex transform), of that chunk fed alone with seed + s."""
    L = llsm.load()
    S, K, thop = 136, 5, 0.005
    ao = llsm.make_aoptions(f0_refine=0, thop=thop)
    chunks, nfrm = [], None
    for s in range(K):
        x, _ = make_speechlike(40 + s, nx=6000)
        n = int(len(x) / FS / thop)
        f0 = (140 + 15 * s + 20 * np.sin(2 * np.pi * 1.1 * np.arange(n) * thop + s)).astype(np.float32)
        f0[:2] = 0
        pr, _ = oracle_analyze(o64, ao, FS, x, f0)
        chunks.append(chunk_from_oracle(L, ao, pr, FS)); nfrm = n
    so = llsm.make_soptions(FS)
    seed = 4100
    L.llsm_gpu_set_default_seed(seed)
    g = L.llsm_create_rtsynth_group(C.byref(so), chunks[0].contents.conf, 4096, S)
    assert g, L.llsm_gpu_last_error()
    FrameArr = C.POINTER(llsm.Container) * S
    outp = np.zeros((S, 4096), np.float32); outap = np.zeros((S, 4096), np.float32); cnt = (C.c_int * S)()
    got_p = [[] for _ in range(S)]; got_ap = [[] for _ in range(S)]
    for i in range(nfrm):
        L.llsm_rtsynth_group_feed(g, FrameArr(*[chunks[s % K].contents.frames[i] for s in range(S)]))
        if L.llsm_rtsynth_group_fetch_all(g, outp.ctypes.data_as(llsm.P_fp), outap.ctypes.data_as(llsm.P_fp), 4096, cnt) > 0:
            for s in range(S):
                got_p[s].append(outp[s, :cnt[s]].copy()); got_ap[s].append(outap[s, :cnt[s]].copy())
    L.llsm_delete_rtsynth_group(g)
    for s in (0, 1, 63, 64, 127, 128, 131, 135):
        L.llsm_gpu_set_default_seed(seed + s)
        yp, yap, _ = rt_run(L, so, chunks[s % K], nfrm)
        gp, gap = np.concatenate(got_p[s]), np.concatenate(got_ap[s])
        assert len(gp) == len(yp) and np.array_equal(gp, yp.astype(np.float32)), s
        assert rel_rms(gap, yap) < 2e-6, s
    for ch in chunks:
        L.llsm_delete_chunk(ch)


@pytest.mark.parametrize("seed", range(8))
def test_rt_random_configurations(o64, seed):
    """Seeded fuzz of llsmrt over sampling rate, hop (integer and fractional), band plan and harmonic limits: the
    streaming output must match the oracle's restatement of llsmrt.c sample for sample (same latency, same length)."""
    from test_gpu_configs import _fuzz_case
    from gpu_common import aopt_kwargs
    L = llsm.load()
    fs, thop, kw, nx = _fuzz_case(200 + seed)
    nx = min(nx, int(0.3 * fs))
    x, f0 = make_speechlike(700 + seed, nx=nx, fs=fs, thop=thop)
    f0 = f0.astype(np.float32)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)
    okw = aopt_kwargs(ao); okw["chanfreq"] = kw["chanfreq"]
    pr, _ = o64.analyze(o64.aoptions(**okw), x, fs, f0, want_res=True)
    p32 = pr.astype(np.float32).astype(np.float64)
    seed_rng = 555 + seed
    ypo, yapo, lato = o64.rt_run(o64.soptions(fs), p32, seed=seed_rng)
    ch = chunk_from_oracle(L, ao, pr, fs)
    so = llsm.make_soptions(fs)
    L.llsm_gpu_set_default_seed(seed_rng)
    yp, yap, lat = rt_run(L, so, ch, pr.nfrm)
    L.llsm_delete_chunk(ch)
    m = dict(fs=fs, thop=thop, latency=lat, n=len(yp), p_rel_rms=rel_rms(yp, ypo) if len(yp) == len(ypo) else 1.0,
             ap_rel_rms=rel_rms(yap, yapo) if len(yap) == len(yapo) else 1.0)
    report("rt_fuzz_%02d" % seed, m)
    assert lat == lato and len(yp) == len(ypo), (lat, lato, len(yp), len(ypo))
    assert m["p_rel_rms"] <= 1e-5 and m["ap_rel_rms"] <= 1e-5, m


def _group_run(L, so, chunks, nfrm, seed):
    """S lock-stepped streams, one hop per feed, 256-sample pulls of every stream at once."""
    S = len(chunks)
    L.llsm_gpu_set_default_seed(seed)
    g = L.llsm_create_rtsynth_group(C.byref(so), chunks[0].contents.conf, 4096, S)
    assert g, L.llsm_gpu_last_error()
    outp = [[] for _ in range(S)]; outap = [[] for _ in range(S)]
    bp = np.zeros((S, 256), np.float32); bap = np.zeros((S, 256), np.float32); cnt = (C.c_int * S)()
    FrameArr = C.POINTER(llsm.Container) * S
    for i in range(nfrm):
        fr = FrameArr(*[chunks[s].contents.frames[i] for s in range(S)])
        L.llsm_rtsynth_group_feed(g, fr)
        while L.llsm_rtsynth_group_numoutput(g, 0) >= 256 or (i == nfrm - 1 and L.llsm_rtsynth_group_numoutput(g, 0) > 0):
            L.llsm_rtsynth_group_fetch_all(g, bp.ctypes.data_as(llsm.P_fp), bap.ctypes.data_as(llsm.P_fp), 256, cnt)
            for s in range(S):
                outp[s].append(bp[s, :cnt[s]].copy()); outap[s].append(bap[s, :cnt[s]].copy())
    L.llsm_delete_rtsynth_group(g)
    return [np.concatenate(v) for v in outp], [np.concatenate(v) for v in outap]


@pytest.mark.parametrize("thop", [0.005, 200.5 / 44100.0])
def test_rt_launch_modes_agree(o64, thop):
    """The ways a hop reaches the device -- five single-purpose launches (llsm_gpu_rt_fused(0)), two launches (1) or one
    (2, the default), each between a copy in and a copy out (llsm_gpu_rt_direct(0)) or reading and writing the pinned
    blocks themselves (the default) -- give the same streams: the one- and two-launch hops run the same device functions
    on the same numbers and must agree bit for bit (odd stream count: the last pair of the noise filter is half empty;
    a fractional hop: the output rows change length from hop to hop); five launches differ by the float32 rounding
    of the noise part only."""
    L = llsm.load()
    S = 3
    chunks, nfrm = [], None
    ao = llsm.make_aoptions(f0_refine=0, thop=thop)
    for s in range(S):
        x, _ = make_speechlike(40 + s, nx=12000)
        n = int(len(x) / FS / thop)
        t = np.arange(n) * thop
        f0 = (110 + 45 * s + 25 * np.sin(2 * np.pi * 1.7 * t + s)).astype(np.float32)
        f0[:3] = 0; f0[n // 2: n // 2 + 5] = 0
        pr, _ = oracle_analyze(o64, ao, FS, x, f0)
        chunks.append(chunk_from_oracle(L, ao, pr, FS)); nfrm = n
    so = llsm.make_soptions(FS)
    prev_f, prev_d = L.llsm_gpu_rt_fused(-1), L.llsm_gpu_rt_direct(-1)
    try:
        runs = {}
        for name, fused, direct in (("five", 0, 0), ("copies", 1, 0), ("direct", 1, 1), ("one_copies", 2, 0), ("one", 2, 1),
                                    ("chip_copies", 3, 0), ("chip", 3, 1)):
            L.llsm_gpu_rt_fused(fused); L.llsm_gpu_rt_direct(direct)
            runs[name] = _group_run(L, so, chunks, nfrm, 4242)
    finally:
        L.llsm_gpu_rt_fused(prev_f); L.llsm_gpu_rt_direct(prev_d)
    for ch in chunks:
        L.llsm_delete_chunk(ch)
    for s in range(S):
        assert len(runs["direct"][0][s]) == len(runs["copies"][0][s]) == len(runs["five"][0][s]) > 10000
        assert np.sqrt(np.mean(runs["direct"][0][s] ** 2)) > 0.01
        for name in ("direct", "one_copies", "one", "chip_copies", "chip"):
            assert np.array_equal(runs[name][0][s], runs["copies"][0][s]), (name, s)
            assert np.array_equal(runs[name][1][s], runs["copies"][1][s]), (name, s)
        assert np.array_equal(runs["five"][0][s], runs["copies"][0][s]), s      # the sinusoid path is the same arithmetic
        assert rel_rms(runs["five"][1][s], runs["copies"][1][s]) < 2e-6, s


def test_pipelined_consumer_never_blocks_on_another_streams_ring(o64):
    """ADVICE r4 (rt.cpp complete_pending): pipelined feeds, a group of two streams with a ring barely larger than two
    hops, an idle producer, and ONE thread draining the streams one after another.  After three feeds two hops lie in
    each ring and the third is in flight with no room for it; the consumer drains stream 0 dry -- the dry fetch must
    return 0 (the hop stays in flight) instead of waiting, with the feed lock held, for room in stream 1's ring that
    only this very thread can make.  Once stream 1 is drained the hop arrives for both.  Run in a helper thread with a
    time-out so that a regression fails instead of hanging the suite."""
    import threading
    L = llsm.load()
    thop = 0.005
    x, f0 = make_speechlike(3, nx=8000)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop)
    pr, _ = oracle_analyze(o64, ao, FS, x, f0)
    ch = chunk_from_oracle(L, ao, pr, FS)
    so = llsm.make_soptions(FS)
    prev = L.llsm_gpu_rt_pipeline(1)
    result = {}

    def body():
        S = 2
        g = C.c_void_p(L.llsm_create_rtsynth_group(C.byref(so), ch.contents.conf, 512, S))
        assert g.value, L.llsm_gpu_last_error()
        Frames = C.POINTER(llsm.Container) * S
        buf = np.zeros(4096, np.float32)
        for i in range(3):
            L.llsm_rtsynth_group_feed(g, Frames(*[ch.contents.frames[4 + i]] * S))
        n0 = L.llsm_rtsynth_group_numoutput(g, 0); n1 = L.llsm_rtsynth_group_numoutput(g, 1)
        got0 = L.llsm_rtsynth_group_fetch(g, 0, buf.ctypes.data_as(llsm.P_fp), None, 4096)
        dry = L.llsm_rtsynth_group_fetch(g, 0, buf.ctypes.data_as(llsm.P_fp), None, 4096)       # used to deadlock here
        got1 = L.llsm_rtsynth_group_fetch(g, 1, buf.ctypes.data_as(llsm.P_fp), None, 4096)
        late1 = L.llsm_rtsynth_group_fetch(g, 1, buf.ctypes.data_as(llsm.P_fp), None, 4096)     # dry -> the hop in flight arrives
        late0 = L.llsm_rtsynth_group_fetch(g, 0, buf.ctypes.data_as(llsm.P_fp), None, 4096)
        L.llsm_delete_rtsynth_group(g)
        result.update(n0=n0, n1=n1, got0=got0, dry=dry, got1=got1, late1=late1, late0=late0)

    t = threading.Thread(target=body, daemon=True)
    t.start(); t.join(60.0)
    try:
        assert not t.is_alive(), "consumer blocked inside complete_pending (deadlock)"
        assert result["n0"] == result["n1"] == result["got0"] == result["got1"] and 400 <= result["n0"] <= 512, result
        assert result["dry"] == 0 and result["late1"] == result["late0"] and 200 <= result["late0"] <= 230, result
    finally:
        L.llsm_gpu_rt_pipeline(prev)
        if not t.is_alive():
            L.llsm_delete_chunk(ch)


@pytest.mark.parametrize("K", [2, 4, 7])
def test_feed_many_equals_single_feeds(o64, K):
    """llsm_rtsynth_group_feed_many (K hops per call, hop k + 1 packed and enqueued while hop k is on the device): the same
    kernels on the same numbers as K single feeds -- bit-identical samples of every stream, and on return the samples of
    ALL K hops are in the rings (what a synchronous feed promises).  Two streams of different material, a frame count that
    is not a multiple of K (the tail goes as a shorter call)."""
    L = llsm.load()
    L.llsm_rtsynth_group_feed_many.argtypes = [C.c_void_p, C.POINTER(C.POINTER(llsm.Container)), C.c_int]
    thop = 0.005
    ao = llsm.make_aoptions(f0_refine=0, thop=thop)
    chunks = []
    for k in range(2):
        x, f0 = make_speechlike(30 + k, nx=9000)
        pr, _ = oracle_analyze(o64, ao, FS, x, f0)
        chunks.append(chunk_from_oracle(L, ao, pr, FS))
        nfrm = pr.nfrm
    so = llsm.make_soptions(FS)
    ref_p, ref_ap = _group_run(L, so, chunks, nfrm, 77)
    S = 2
    L.llsm_gpu_set_default_seed(77)
    g = C.c_void_p(L.llsm_create_rtsynth_group(C.byref(so), chunks[0].contents.conf, 4096, S))
    assert g.value, L.llsm_gpu_last_error()
    outp = [[] for _ in range(S)]; outap = [[] for _ in range(S)]
    bp = np.zeros((S, 4096), np.float32); bap = np.zeros((S, 4096), np.float32); cnt = (C.c_int * S)()
    lat = L.llsm_rtsynth_group_getlatency(g)
    fed = 0
    for i0 in range(0, nfrm, K):
        k = min(K, nfrm - i0)
        arr = (C.POINTER(llsm.Container) * (S * k))(*[chunks[s].contents.frames[i0 + h] for h in range(k) for s in range(S)])
        before = L.llsm_rtsynth_group_numoutput(g, 0)
        L.llsm_rtsynth_group_feed_many(g, arr, k)
        after = L.llsm_rtsynth_group_numoutput(g, 0)
        assert 218 * k <= after - before <= 223 * k, (i0, k, before, after)      # every hop of the call has arrived
        L.llsm_rtsynth_group_fetch_all(g, bp.ctypes.data_as(llsm.P_fp), bap.ctypes.data_as(llsm.P_fp), 4096, cnt)
        for s in range(S):
            outp[s].append(bp[s, :cnt[s]].copy()); outap[s].append(bap[s, :cnt[s]].copy())
        fed += k
    L.llsm_delete_rtsynth_group(g)
    for c in chunks:
        L.llsm_delete_chunk(c)
    for s in range(S):
        a, r = np.concatenate(outp[s]), ref_p[s]
        assert len(a) == len(r) and np.array_equal(a, r), (s, len(a), len(r))
        assert np.array_equal(np.concatenate(outap[s]), ref_ap[s]), s
