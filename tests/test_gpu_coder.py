"""-m gpu: the frame coder on the HIP path (csrc/coder.cpp, k_coder_encode / k_coder_decode) vs the float64
oracle (oracle/coder_oracle.c) on identical frames, and the reference's own acceptance (test/test-coder.c:31-51)
through the chunk API."""
import ctypes as C
import os

import numpy as np
import pytest

import libllsm2_amd as llsm
from conftest import FS, make_speechlike, wrap
from gpu_common import oracle_analyze, report
from test_gpu_l1 import l1_chunk_from_oracle, q32
from verify_utils import GOLDEN, data_distribution_klds, read_wav

pytestmark = pytest.mark.gpu


def coder_lib():
    L = llsm.load()
    L.llsm_create_coder.restype = C.c_void_p; L.llsm_create_coder.argtypes = [C.POINTER(llsm.Container), C.c_int, C.c_int]
    L.llsm_delete_coder.argtypes = [C.c_void_p]
    L.llsm_coder_dimension.argtypes = [C.c_void_p]
    L.llsm_coder_encode.restype = llsm.P_fp; L.llsm_coder_encode.argtypes = [C.c_void_p, C.POINTER(llsm.Container)]
    L.llsm_coder_encode_frames.argtypes = [C.c_void_p, C.POINTER(C.POINTER(llsm.Container)), C.c_int, llsm.P_fp]
    L.llsm_coder_decode_frames.argtypes = [C.c_void_p, llsm.P_fp, C.c_int, C.c_int, C.POINTER(C.POINTER(llsm.Container))]
    for n in ("llsm_coder_decode_layer0", "llsm_coder_decode_layer1"):
        getattr(L, n).restype = C.POINTER(llsm.Container); getattr(L, n).argtypes = [C.c_void_p, llsm.P_fp]
    return L


@pytest.fixture(scope="module")
def L():
    return coder_lib()


CODER_CASES = {
    # id: (fs, thop, vocal-tract transform, order_spec, order_bap, analysis options, utterance seed)
    "default": (FS, 0.005, 2048, 64, 5, dict(), 1),
    "16k_low_order": (16000.0, 0.005, 1024, 24, 3, dict(nchannel=3, chanfreq=[1000.0, 3000.0], npsd=128), 2),
    "48k_high_order": (48000.0, 0.004, 4096, 120, 8, dict(maxnhar=160), 3),
    "22k_hop128": (22050.0, 128.0 / 22050.0, 2048, 40, 2, dict(npsd=64, maxnhar=60), 4),
    # band aperiodicities of 0.9999998 / 1 / 1 in the encoded vectors: the ill-conditioned decode described below
    "bap_near_one": (44100.0, 0.005, 4096, 75, 9, dict(nchannel=1, chanfreq=[], npsd=64, maxnhar=100, maxnhar_e=3), 3128),
}


@pytest.mark.parametrize("cid", sorted(CODER_CASES))
def test_coder_parity(L, o64, cid):
    coder_parity(L, o64, cid, CODER_CASES[cid])


def coder_parity(L, o64, cid, case):
    """One configuration (also driven with random ones by tools/fuzz_soak.py).  Decoded phases are compared on the
    harmonics above 1e-4 of the frame's largest amplitude (as the layer-1 -> layer-0 test does: below that the harmonic
    is the 1e-10 floor of dsputils.c:491 and its phase carries no signal); the worst one at any amplitude is reported."""
    FS, thop, nfft, osp, obap, kw, useed = case
    dim, ns = 3 + osp + obap, nfft // 2 + 1
    x, f0 = make_speechlike(useed, nx=int(0.45 * FS), fs=FS, thop=thop)
    f0 = f0.astype(np.float32)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)
    pr, _ = oracle_analyze(o64, ao, FS, x, f0)
    pr = pr.astype(np.float32).astype(np.float64)
    q = q32(o64.chunk_tolayer1(pr, nfft))
    ch = l1_chunk_from_oracle(L, ao, pr, q, FS, nfft=nfft)
    nfrm = pr.nfrm
    coder = L.llsm_create_coder(ch.contents.conf, osp, obap)
    assert coder and L.llsm_coder_dimension(coder) == dim
    enc = np.zeros((nfrm, dim), np.float32)
    assert L.llsm_coder_encode_frames(coder, ch.contents.frames, nfrm, enc.ctypes.data_as(llsm.P_fp)) == 0
    enco = o64.coder_encode_chunk(pr, q, osp, obap)
    m = dict(enc_spec_abs_max=float(np.abs(enc[:, 3:3 + osp] - enco[:, 3:3 + osp]).max()), enc_bap_abs_max=float(np.abs(enc[:, 3 + osp:] - enco[:, 3 + osp:]).max()),
             enc_head_abs_max=float(np.abs(enc[:, :3] - enco[:, :3]).max()))
    # the single-frame entry point gives the same vector
    one = L.llsm_coder_encode(coder, ch.contents.frames[nfrm // 2])
    assert np.array_equal(np.ctypeslib.as_array(one, (dim,)), enc[nfrm // 2])
    # decode the ORACLE's vectors on both sides.  The decoder is ill-conditioned where the interpolated aperiodicity comes
    # within 1e-4 of 1: the harmonic part is sqrt(psd (1 - ap)) (coder.c:212), so a bin next to a band value of 1 -- above
    # all a bin that coincides with such a knot of apaxis, where float rounding of the two axes decides between
    # 1 - ap = 0 and 6e-8 -- has its amplitude decided by the last bits of ap; log(ampl + 1e-10) (dsputils.c:491) then
    # moves by O(1) and, being a Hilbert transform of that log-amplitude, the minimum phase of EVERY harmonic of the frame
    # follows.  (Found by tools/fuzz_soak.py: 1.02 rad on one frame, where a float32 build of the oracle moves by
    # 1.02 rad too; 3.01 dB = sqrt(2) on one vocal-tract bin, 1 - ap = 2^-23 against 2^-24.)  The float64 oracle gives
    # the aperiodicity per bin (o_coder_aperiodicity); bins within 1e-4 of 1 are left out of the vocal-tract comparison
    # and frames with such a bin (3e-4) below their top harmonic out of the phase comparison -- both counted in the
    # report -- except inside a run of exact ones, where the harmonic part is exactly 0 on both sides.
    e32 = np.ascontiguousarray(enco.astype(np.float32))
    mh = int(FS / 2 / 20.0)
    margin = 1.0 - o64.coder_aperiodicity_chunk(e32.astype(np.float64), pr, ns, 1.5, osp, obap)
    zero = margin == 0
    inner = zero & np.hstack([zero[:, :1], zero[:, :-1]]) & np.hstack([zero[:, 1:], zero[:, -1:]])   # inside a run of exact ones
    for use_l1 in (0, 1):
        out = (C.POINTER(llsm.Container) * nfrm)()
        assert L.llsm_coder_decode_frames(coder, e32.ctypes.data_as(llsm.P_fp), nfrm, use_l1, out) == 0
        po, qo = o64.coder_decode_chunk(e32.astype(np.float64), bool(use_l1), pr, ns, 1.5, osp, obap, mh)
        da = dp = dv = ds = dn = dp_ill = 0.0; worst = (0.0, -1, -1, 0.0, 0.0); n_ill = n_voiced = 0; bins_ill = bins_all = 0
        for i in range(nfrm):
            fr = out[i]
            nm = C.cast(L.llsm_container_get(fr, llsm.FRAME_NM), C.POINTER(llsm.NMFrame)).contents
            dn = max(dn, np.abs(np.ctypeslib.as_array(nm.psd, (nm.npsd,)) - po.psd[i]).max())
            assert C.cast(L.llsm_container_get(fr, llsm.FRAME_F0), llsm.P_fp)[0] == np.float32(po.f0[i])
            hm = C.cast(L.llsm_container_get(fr, llsm.FRAME_HM), C.POINTER(llsm.HMFrame))
            if use_l1:
                n = int(qo.nvsphse[i])
                if n:
                    assert not bool(hm)
                    vt = C.cast(L.llsm_container_get(fr, llsm.FRAME_VTMAGN), llsm.P_fp); vs = C.cast(L.llsm_container_get(fr, llsm.FRAME_VSPHSE), llsm.P_fp)
                    assert L.llsm_fparray_length(vs) == n and L.llsm_fparray_length(vt) == ns
                    g = np.ctypeslib.as_array(vt, (ns,)).astype(np.float64); o = qo.vtmagn[i]
                    assert not np.isnan(g).any() and not np.isposinf(g).any(), i
                    assert np.all(np.isneginf(g[inner[i]])) and np.all(np.isneginf(o[inner[i]])), i     # exactly zero harmonic part
                    mg = margin[i].copy(); mg[0] = mg[1]                                                 # (bin 0 is a copy of bin 1, coder.c:239)
                    well = (mg >= 1e-4)
                    bins_ill += int(np.count_nonzero(~well & ~inner[i])); bins_all += ns
                    assert np.all(np.isfinite(g[well])) and np.all(np.isfinite(o[well])), i
                    dv = max(dv, np.abs(g[well] - o[well]).max())
                    ds = max(ds, np.abs(wrap(np.ctypeslib.as_array(vs, (n,)) - qo.vsphse[i, :n])).max())
            else:
                n = int(po.nhar[i]); assert hm.contents.nhar == n
                if n:
                    a = np.ctypeslib.as_array(hm.contents.ampl, (n,)); p = np.ctypeslib.as_array(hm.contents.phse, (n,))
                    da = max(da, (np.abs(a - po.ampl[i, :n]) / po.ampl[i, :n].max()).max())
                    e = np.abs(wrap(p - po.phse[i, :n])); k = int(np.argmax(e))
                    if e[k] > worst[0]:
                        worst = (float(e[k]), i, k, float(po.ampl[i, k] / po.ampl[i, :n].max()), float(po.f0[i]))
                    big = po.ampl[i, :n] > 1e-4 * po.ampl[i, :n].max()
                    jtop = min(int(np.ceil(n * po.f0[i] / (FS / 2) * (ns - 1))) + 1, ns)
                    ill = bool(np.any((margin[i, :jtop] < 3e-4) & ~inner[i, :jtop]))
                    n_voiced += 1; n_ill += ill
                    if ill:
                        dp_ill = max(dp_ill, e[big].max())
                    else:
                        dp = max(dp, e[big].max())
            L.llsm_delete_container(fr)
        m.update({f"dec{use_l1}_psd_db_max": float(dn), f"dec{use_l1}_ampl_over_max": float(da), f"dec{use_l1}_phse_rad": float(dp),
                  f"dec{use_l1}_vtmagn_db": float(dv), f"dec{use_l1}_vsphse_rad": float(ds)})
        if use_l1:
            m["dec1_vtmagn_bins_left_out"] = "%d of %d" % (bins_ill, bins_all)
        else:
            m["dec0_phse_worst_any_ampl"] = dict(zip(("rad", "frame", "harmonic", "ampl_over_max", "f0"), worst))
            m["dec0_phse_rad_ill_conditioned_frames"] = float(dp_ill)
            m["dec0_frames_ill_conditioned"] = "%d of %d voiced" % (n_ill, n_voiced)
    report("coder_parity_" + cid, m)
    L.llsm_delete_coder(coder); L.llsm_delete_chunk(ch)
    assert m["enc_head_abs_max"] == 0 and m["enc_spec_abs_max"] <= 2e-4 and m["enc_bap_abs_max"] <= 1e-4, m
    assert m["dec0_psd_db_max"] <= 0.02 and m["dec1_psd_db_max"] <= 0.02, m
    assert m["dec0_ampl_over_max"] <= 1e-4 and m["dec0_phse_rad"] <= 1e-3, m
    assert m["dec1_vtmagn_db"] <= 0.02 and m["dec1_vsphse_rad"] <= 1e-3, m


def test_coder_acceptance_through_the_chunk_api(L):
    """test/test-coder.c:31-51 on the product: analyze -> llsm_chunk_tolayer1(2048) -> encode(64, 5) -> decode layer 0
    and layer 1 (-> llsm_chunk_tolayer0) -> phasepropagate -> llsm_synthesize; KLD < 0.05 against the input."""
    x, fs = read_wav(os.path.join(GOLDEN, "arctic_a0001.wav"))
    f0 = np.load(os.path.join(GOLDEN, "arctic_a0001_f0_hop128.npy")).copy()
    nfrm = len(f0)
    ao = llsm.make_aoptions(thop=128.0 / fs, f0_refine=0)
    so = llsm.make_soptions(fs)
    ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), fs, f0.ctypes.data_as(llsm.P_fp), nfrm, None)
    assert bool(ch), L.llsm_gpu_last_error()
    L.llsm_chunk_tolayer1(ch, 2048)
    coder = L.llsm_create_coder(ch.contents.conf, 64, 5)
    enc = np.zeros((nfrm, 72), np.float32)
    assert L.llsm_coder_encode_frames(coder, ch.contents.frames, nfrm, enc.ctypes.data_as(llsm.P_fp)) == 0
    rep = {}
    for use_l1 in (0, 1):
        rec = L.llsm_create_chunk(ch.contents.conf, 0)
        out = (C.POINTER(llsm.Container) * nfrm)()
        assert L.llsm_coder_decode_frames(coder, enc.ctypes.data_as(llsm.P_fp), nfrm, use_l1, out) == 0
        for i in range(nfrm):
            rec.contents.frames[i] = out[i]
        if use_l1:
            L.llsm_chunk_tolayer0(rec)
        L.llsm_chunk_phasepropagate(rec, 1)
        o = L.llsm_synthesize(C.byref(so), rec)
        assert bool(o), L.llsm_gpu_last_error()
        y = np.ctypeslib.as_array(o.contents.y, (o.contents.ny,)).copy()
        L.llsm_delete_output(o); L.llsm_delete_chunk(rec)
        klds = data_distribution_klds(x, y)
        rep[f"layer{use_l1}"] = klds
        assert all(k < 0.05 for k in klds), (use_l1, klds)
    report("coder_acceptance", rep)
    L.llsm_delete_coder(coder); L.llsm_delete_chunk(ch)
