"""Flat wire format of a layer-0 chunk (csrc/wire.cpp, llsm_gpu.h): chunk -> blob -> chunk
round trip, zero-copy view, and rejection of malformed input.  Host-only: runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

import libllsm2_amd as llsm


def _bind(L):
    L.llsm_chunk_blob_size.restype = C.c_size_t
    L.llsm_chunk_blob_size.argtypes = [C.POINTER(llsm.Chunk)]
    L.llsm_chunk_to_blob.restype = C.c_longlong
    L.llsm_chunk_to_blob.argtypes = [C.POINTER(llsm.Chunk), C.c_void_p, C.c_size_t]
    L.llsm_blob_view.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(llsm.FlatParams), llsm.P_int,
                                 llsm.P_fp, llsm.P_fp]
    L.llsm_blob_to_chunk.restype = C.POINTER(llsm.Chunk)
    L.llsm_blob_to_chunk.argtypes = [C.c_void_p, C.c_size_t]
    return L


def _make_chunk(L, F=9, mh=12, me=3, npsd=16, nch=4, seed=0):
    """analysis-shaped chunk: voiced frames carry HM + eenv, every frame PSD / PSDRES / edc"""
    rng = np.random.default_rng(seed)
    f0 = np.where(rng.random(F) < 0.7, rng.uniform(80, 300, F), 0).astype(np.float32)
    f0[0] = 0; f0[1] = 123.0
    nhar = np.where(f0 > 0, rng.integers(2, mh + 1, F), 0).astype(np.int32)
    nhar[1] = mh
    nhe = np.where(f0 > 0, me, 0).astype(np.int32)
    rows = dict(ampl=rng.random((F, mh), np.float32), phse=rng.random((F, mh), np.float32),
                psd=rng.random((F, npsd), np.float32), psdres=rng.random((F, npsd), np.float32),
                edc=rng.random((F, nch), np.float32), ea=rng.random((F, nch, me), np.float32),
                ep=rng.random((F, nch, me), np.float32))
    for i in range(F):
        rows["ampl"][i, nhar[i]:] = 0; rows["phse"][i, nhar[i]:] = 0
        rows["ea"][i, :, nhe[i]:] = 0; rows["ep"][i, :, nhe[i]:] = 0
    has = np.ones(F, np.int32)
    v = llsm.FlatParams()
    v.maxnhar, v.maxnhar_e, v.npsd, v.nchannel = mh, me, npsd, nch
    v.f0 = f0.ctypes.data_as(llsm.P_fp); v.nhar = nhar.ctypes.data_as(llsm.P_int)
    v.ampl = rows["ampl"].ctypes.data_as(llsm.P_fp); v.phse = rows["phse"].ctypes.data_as(llsm.P_fp)
    v.psd = rows["psd"].ctypes.data_as(llsm.P_fp); v.psdres = rows["psdres"].ctypes.data_as(llsm.P_fp)
    v.has_psdres = has.ctypes.data_as(llsm.P_int); v.edc = rows["edc"].ctypes.data_as(llsm.P_fp)
    v.nhar_e = nhe.ctypes.data_as(llsm.P_int)
    v.eenv_ampl = rows["ea"].ctypes.data_as(llsm.P_fp); v.eenv_phse = rows["ep"].ctypes.data_as(llsm.P_fp)
    ao = llsm.make_aoptions(maxnhar=mh, maxnhar_e=me, npsd=npsd, thop=0.004)
    conf = L.llsm_aoptions_toconf(C.byref(ao), 24000.0)
    C.cast(L.llsm_container_get(conf, llsm.CONF_NFRM), llsm.P_int)[0] = F
    ch = L.llsm_create_chunk(conf, 1)
    L.llsm_delete_container(conf)
    assert L.llsm_flat_to_chunk(C.byref(v), 0, ch) == 0
    return ch, dict(f0=f0, nhar=nhar, nhe=nhe, has=has, **rows)


def _flat_of(L, ch, F, mh, me, npsd, nch):
    out = dict(f0=np.zeros(F, np.float32), nhar=np.zeros(F, np.int32), nhe=np.zeros(F, np.int32),
               has=np.zeros(F, np.int32), ampl=np.zeros((F, mh), np.float32), phse=np.zeros((F, mh), np.float32),
               psd=np.zeros((F, npsd), np.float32), psdres=np.zeros((F, npsd), np.float32),
               edc=np.zeros((F, nch), np.float32), ea=np.zeros((F, nch, me), np.float32),
               ep=np.zeros((F, nch, me), np.float32))
    v = llsm.FlatParams()
    v.maxnhar, v.maxnhar_e, v.npsd, v.nchannel = mh, me, npsd, nch
    v.f0 = out["f0"].ctypes.data_as(llsm.P_fp); v.nhar = out["nhar"].ctypes.data_as(llsm.P_int)
    v.ampl = out["ampl"].ctypes.data_as(llsm.P_fp); v.phse = out["phse"].ctypes.data_as(llsm.P_fp)
    v.psd = out["psd"].ctypes.data_as(llsm.P_fp); v.psdres = out["psdres"].ctypes.data_as(llsm.P_fp)
    v.has_psdres = out["has"].ctypes.data_as(llsm.P_int); v.edc = out["edc"].ctypes.data_as(llsm.P_fp)
    v.nhar_e = out["nhe"].ctypes.data_as(llsm.P_int)
    v.eenv_ampl = out["ea"].ctypes.data_as(llsm.P_fp); v.eenv_phse = out["ep"].ctypes.data_as(llsm.P_fp)
    assert L.llsm_chunk_to_flat(ch, C.byref(v), 0) == 0
    return out


def test_blob_roundtrip_and_view():
    L = _bind(llsm.load())
    F, mh, me, npsd, nch = 9, 12, 3, 16, 4
    ch, ref = _make_chunk(L, F, mh, me, npsd, nch)
    n = L.llsm_chunk_blob_size(ch)
    assert n > 0 and n % 8 == 0
    buf = (C.c_ubyte * n)()
    assert L.llsm_chunk_to_blob(ch, buf, n) == n
    assert L.llsm_chunk_to_blob(ch, buf, n - 1) == -1          # destination too small
    assert bytes(buf[:8]) == b"LLSM2L0\x00"

    # zero-copy view: pointers into the blob, widths = largest counts present
    v = llsm.FlatParams(); nfrm = C.c_int(); thop = llsm.fp(); fnyq = llsm.fp()
    assert L.llsm_blob_view(buf, n, C.byref(v), C.byref(nfrm), C.byref(thop), C.byref(fnyq)) == 0
    assert nfrm.value == F and abs(thop.value - 0.004) < 1e-9 and fnyq.value == 24000.0
    assert (v.maxnhar, v.maxnhar_e, v.npsd, v.nchannel) == (mh, me, npsd, nch)
    base = C.addressof(buf)
    assert base <= C.addressof(v.ampl.contents) < base + n
    assert np.array_equal(np.ctypeslib.as_array(v.f0, (F,)), ref["f0"])
    assert np.array_equal(np.ctypeslib.as_array(v.ampl, (F, mh)), ref["ampl"])

    # blob -> chunk -> flat equals the original rows; conf survives
    ch2 = L.llsm_blob_to_chunk(buf, n)
    assert bool(ch2)
    got = _flat_of(L, ch2, F, mh, me, npsd, nch)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    conf = ch2.contents.conf
    assert C.cast(L.llsm_container_get(conf, llsm.CONF_NPSD), llsm.P_int)[0] == npsd
    assert C.cast(L.llsm_container_get(conf, llsm.CONF_NCHANNEL), llsm.P_int)[0] == nch
    cf = C.cast(L.llsm_container_get(conf, llsm.CONF_CHANFREQ), llsm.P_fp)
    assert [cf[i] for i in range(3)] == [2000.0, 4000.0, 8000.0]
    assert L.llsm_conf_checklayer0(conf) == 1
    # serialising the rebuilt chunk gives the same bytes
    buf2 = (C.c_ubyte * n)()
    assert L.llsm_chunk_blob_size(ch2) == n and L.llsm_chunk_to_blob(ch2, buf2, n) == n
    assert bytes(buf2) == bytes(buf)
    L.llsm_delete_chunk(ch); L.llsm_delete_chunk(ch2)


def test_blob_of_empty_and_unvoiced_chunks():
    L = _bind(llsm.load())
    ao = llsm.make_aoptions()
    for F in (0, 5):
        conf = L.llsm_aoptions_toconf(C.byref(ao), 22050.0)
        C.cast(L.llsm_container_get(conf, llsm.CONF_NFRM), llsm.P_int)[0] = F
        ch = L.llsm_create_chunk(conf, 1)
        L.llsm_delete_container(conf)
        n = L.llsm_chunk_blob_size(ch)
        buf = (C.c_ubyte * n)()
        assert L.llsm_chunk_to_blob(ch, buf, n) == n
        v = llsm.FlatParams(); nfrm = C.c_int(-1)
        assert L.llsm_blob_view(buf, n, C.byref(v), C.byref(nfrm), None, None) == 0
        assert nfrm.value == F and v.maxnhar == 0 and v.maxnhar_e == 0
        ch2 = L.llsm_blob_to_chunk(buf, n)
        assert bool(ch2)
        k = C.c_int(0)
        f0 = L.llsm_chunk_getf0(ch2, C.byref(k))
        assert k.value == F
        C.CDLL(None).free(f0)
        L.llsm_delete_chunk(ch); L.llsm_delete_chunk(ch2)


@pytest.mark.parametrize("damage", ["truncate", "magic", "version", "size_field", "offset", "nfrm", "nhar_row"])
def test_malformed_blobs_are_rejected(damage):
    L = _bind(llsm.load())
    ch, _ = _make_chunk(L)
    n = L.llsm_chunk_blob_size(ch)
    buf = (C.c_ubyte * n)()
    assert L.llsm_chunk_to_blob(ch, buf, n) == n
    L.llsm_delete_chunk(ch)
    raw = bytearray(bytes(buf))
    size = n
    if damage == "truncate":
        size = n - 8
    elif damage == "magic":
        raw[0] = ord("X")
    elif damage == "version":
        raw[8] = 2
    elif damage == "size_field":
        raw[56] ^= 0x10                                       # total_bytes
    elif damage == "offset":
        raw[64 + 8 * 3] ^= 0x08                               # offset of the amplitude rows
    elif damage == "nfrm":
        raw[16:20] = (10 ** 6).to_bytes(4, "little")          # rows no longer fit the blob
    elif damage == "nhar_row":
        v = llsm.FlatParams()
        b0 = (C.c_ubyte * n).from_buffer(raw)
        assert L.llsm_blob_view(b0, n, C.byref(v), None, None, None) == 0
        v.nhar[1] = 10 ** 6                                   # a harmonic count beyond its row
        del b0
    b = (C.c_ubyte * len(raw)).from_buffer(raw)
    v = llsm.FlatParams()
    assert L.llsm_blob_view(b, size, C.byref(v), None, None, None) == -1
    assert not bool(L.llsm_blob_to_chunk(b, size))
    assert L.llsm_gpu_last_error()
