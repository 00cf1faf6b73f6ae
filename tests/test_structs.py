"""Re-statement of test/test-structs.c (container / fparray / hmframe / nmframe /
chunk semantics) against the library's host-side data model, through the C ABI."""
import ctypes as C

import numpy as np

import libllsm2_amd as llsm


def fpval(ptr):
    return C.cast(ptr, llsm.P_fp)[0]


def test_container_attach_copy_remove():
    L = llsm.load()
    libc = C.CDLL(None)
    free = C.cast(libc.free, C.c_void_p)
    copy_fp = C.cast(L.llsm_copy_fp, C.c_void_p)
    c1 = L.llsm_create_container(10)
    L.llsm_container_attach_(c1, 0, C.cast(L.llsm_create_fp(5.0), C.c_void_p), free, copy_fp)
    L.llsm_container_attach_(c1, 1, C.cast(L.llsm_create_fp(10.0), C.c_void_p), None, copy_fp)
    assert fpval(c1.contents.members[0]) == 5.0 and fpval(c1.contents.members[1]) == 10.0
    L.llsm_container_attach_(c1, 15, C.cast(L.llsm_create_fp(50.0), C.c_void_p), free, None)
    assert c1.contents.nmember >= 16 and fpval(c1.contents.members[15]) == 50.0
    # copy: deep where a copy-ctor exists, shared otherwise (test-structs.c:29-35)
    c2 = L.llsm_copy_container(c1)
    assert fpval(c2.contents.members[0]) == 5.0 and fpval(c2.contents.members[15]) == 50.0
    C.cast(c2.contents.members[15], llsm.P_fp)[0] = 45.0
    assert fpval(c1.contents.members[15]) == 45.0
    C.cast(c2.contents.members[0], llsm.P_fp)[0] = 6.0
    assert fpval(c1.contents.members[0]) == 5.0
    C.cast(c2.contents.members[0], llsm.P_fp)[0] = 5.0
    c3 = L.llsm_create_container(5)
    L.llsm_container_attach_(c3, 0, C.cast(L.llsm_create_fp(-5.0), C.c_void_p), free, copy_fp)
    L.llsm_copy_container_inplace(c3, c2)
    assert fpval(c3.contents.members[0]) == 5.0 and fpval(c3.contents.members[1]) == 10.0
    assert fpval(c3.contents.members[15]) == 45.0
    C.cast(c3.contents.members[15], llsm.P_fp)[0] = 50.0
    assert fpval(c1.contents.members[15]) == 50.0
    L.llsm_container_remove(c1, 0)
    assert not c1.contents.members[0] and fpval(c2.contents.members[0]) == 5.0
    assert L.llsm_container_get(c1, 99) is None
    for c in (c1, c2, c3):
        libc.free(C.c_void_p(c.contents.members[1]))
    L.llsm_delete_container(c1); L.llsm_delete_container(c2); L.llsm_delete_container(c3)


def test_fparray_length_prefix():
    L = llsm.load()
    a = L.llsm_create_fparray(7)
    assert L.llsm_fparray_length(a) == 7
    assert C.cast(a, C.POINTER(C.c_int))[-1] == 7          # length lives right before the data
    for i in range(7):
        a[i] = i * 0.5
    b = L.llsm_copy_fparray(a)
    assert L.llsm_fparray_length(b) == 7 and [b[i] for i in range(7)] == [i * 0.5 for i in range(7)]
    z = L.llsm_create_fparray(0)
    assert L.llsm_fparray_length(z) == 0
    for p in (a, b, z):
        L.llsm_delete_fparray(p)


def test_hmframe_copy_and_phaseshift_roundtrip():
    L = llsm.load()
    h1 = L.llsm_create_hmframe(3)
    for i, (a, p) in enumerate(((1.0, 1.0), (0.5, -0.5), (0.2, 2.5))):
        h1.contents.ampl[i] = a; h1.contents.phse[i] = p
    h2 = L.llsm_copy_hmframe(h1)
    assert h2.contents.nhar == 3 and abs(h2.contents.ampl[2] - 0.2) < 1e-7
    L.llsm_hmframe_phaseshift(h2, 3.14); L.llsm_hmframe_phaseshift(h2, 3.14); L.llsm_hmframe_phaseshift(h2, -6.28)
    for i, p in enumerate((1.0, -0.5, 2.5)):               # test-structs.c:103-108 (1e-6 relative)
        assert abs(h2.contents.phse[i] - p) <= 1e-6 * abs(p) + 2e-6
    L.llsm_delete_hmframe(h1); L.llsm_delete_hmframe(h2)


def test_nmframe_defaults_and_copy():
    L = llsm.load()
    n1 = L.llsm_create_nmframe(3, 2, 20)
    assert all(n1.contents.psd[i] == -120.0 for i in range(20))          # frame.c:79-80
    assert all(abs(n1.contents.edc[c] - 1e-5) < 1e-12 for c in range(3))  # frame.c:86
    for i in range(20):
        n1.contents.psd[i] = i - 10.0
    for c in range(3):
        n1.contents.edc[c] = c * 0.1
        n1.contents.eenv[c].contents.ampl[0] = 1.0; n1.contents.eenv[c].contents.ampl[1] = 0.5
    n2 = L.llsm_copy_nmframe(n1)
    assert n2.contents.npsd == 20 and n2.contents.nchannel == 3
    assert all(n2.contents.psd[i] == i - 10.0 for i in range(20))
    for c in range(3):
        assert abs(n2.contents.edc[c] - c * 0.1) < 1e-7
        assert n2.contents.eenv[c].contents.nhar == 2 and n2.contents.eenv[c].contents.ampl[1] == 0.5
    L.llsm_delete_nmframe(n1); L.llsm_delete_nmframe(n2)


def test_chunk_create_copy_and_conf():
    L = llsm.load()
    opt = L.llsm_create_aoptions()
    o = opt.contents
    assert (abs(o.thop - 0.005) < 1e-9 and o.maxnhar == 100 and o.maxnhar_e == 4 and o.npsd == 256 and
            o.nchannel == 4 and o.f0_refine == 1 and o.hm_method == 1 and o.rel_winsize == 4.0)
    assert [o.chanfreq[i] for i in range(3)] == [2000.0, 4000.0, 8000.0]
    conf = L.llsm_aoptions_toconf(opt, 22050.0)
    assert L.llsm_conf_checklayer0(conf) == 1
    npsd = C.cast(L.llsm_container_get(conf, llsm.CONF_NPSD), llsm.P_int)[0]
    C.cast(L.llsm_container_get(conf, llsm.CONF_NFRM), llsm.P_int)[0] = 100
    assert L.llsm_fparray_length(C.cast(L.llsm_container_get(conf, llsm.CONF_CHANFREQ), llsm.P_fp)) == 3
    c1 = L.llsm_create_chunk(conf, 1)
    for i in range(100):
        nm = C.cast(L.llsm_container_get(c1.contents.frames[i], llsm.FRAME_NM), C.POINTER(llsm.NMFrame))
        for j in range(npsd):
            nm.contents.psd[j] = np.sin(j * 0.1)
    c2 = L.llsm_copy_chunk(c1)
    for i in (0, 50, 99):
        nm = C.cast(L.llsm_container_get(c2.contents.frames[i], llsm.FRAME_NM), C.POINTER(llsm.NMFrame))
        assert all(abs(nm.contents.psd[j] - np.float32(np.sin(j * 0.1))) < 1e-6 for j in range(npsd))
        assert L.llsm_frame_checklayer0(c2.contents.frames[i]) == 1
    n = C.c_int(0)
    f0 = L.llsm_chunk_getf0(c2, C.byref(n))
    assert n.value == 100 and all(f0[i] == 0 for i in range(100))
    C.CDLL(None).free(f0)
    so = L.llsm_create_soptions(44100.0)
    s = so.contents
    assert s.fs == 44100.0 and s.use_iczt == 1 and s.use_l1 == 0
    assert abs(s.iczt_param_a - 0.275) < 1e-7 and abs(s.iczt_param_b - 2.26) < 1e-6
    L.llsm_delete_soptions(so)
    L.llsm_delete_chunk(c1); L.llsm_delete_chunk(c2)
    L.llsm_delete_container(conf); L.llsm_delete_aoptions(opt)
    # a conf without NCHANNEL / NPSD yields no chunk (container.c:163)
    bare = L.llsm_create_container(2)
    assert not bool(L.llsm_create_chunk(bare, 1))
    L.llsm_delete_container(bare)


def test_flat_roundtrip_through_chunk():
    """llsm_flat_to_chunk / llsm_chunk_to_flat are inverse on analysis-shaped rows."""
    L = llsm.load()
    rng = np.random.default_rng(0)
    F, mh, me, npsd, nch = 6, 10, 4, 16, 4
    f0 = np.array([0, 100, 120, 0, 130, 140], np.float32)
    nhar = np.where(f0 > 0, rng.integers(3, mh + 1, F), 0).astype(np.int32)
    nhe = np.where(f0 > 0, 3, 0).astype(np.int32)
    rows = dict(ampl=rng.random((F, mh), np.float32), phse=rng.random((F, mh), np.float32),
                psd=rng.random((F, npsd), np.float32), psdres=rng.random((F, npsd), np.float32),
                edc=rng.random((F, nch), np.float32), ea=rng.random((F, nch, me), np.float32),
                ep=rng.random((F, nch, me), np.float32))
    for i in range(F):
        rows["ampl"][i, nhar[i]:] = 0; rows["phse"][i, nhar[i]:] = 0
        rows["ea"][i, :, nhe[i]:] = 0; rows["ep"][i, :, nhe[i]:] = 0
    has = np.ones(F, np.int32)

    def view(d, f0_, nhar_, nhe_, has_):
        v = llsm.FlatParams()
        v.maxnhar, v.maxnhar_e, v.npsd, v.nchannel = mh, me, npsd, nch
        v.f0 = f0_.ctypes.data_as(llsm.P_fp); v.nhar = nhar_.ctypes.data_as(llsm.P_int)
        v.ampl = d["ampl"].ctypes.data_as(llsm.P_fp); v.phse = d["phse"].ctypes.data_as(llsm.P_fp)
        v.psd = d["psd"].ctypes.data_as(llsm.P_fp); v.psdres = d["psdres"].ctypes.data_as(llsm.P_fp)
        v.has_psdres = has_.ctypes.data_as(llsm.P_int); v.edc = d["edc"].ctypes.data_as(llsm.P_fp)
        v.nhar_e = nhe_.ctypes.data_as(llsm.P_int)
        v.eenv_ampl = d["ea"].ctypes.data_as(llsm.P_fp); v.eenv_phse = d["ep"].ctypes.data_as(llsm.P_fp)
        return v

    ao = llsm.make_aoptions(maxnhar=mh, maxnhar_e=me, npsd=npsd)
    conf = L.llsm_aoptions_toconf(C.byref(ao), 22050.0)
    C.cast(L.llsm_container_get(conf, llsm.CONF_NFRM), llsm.P_int)[0] = F
    ch = L.llsm_create_chunk(conf, 1)
    src = view(rows, f0, nhar, nhe, has)
    assert L.llsm_flat_to_chunk(C.byref(src), 0, ch) == 0
    back = {k: np.full_like(v, -7) for k, v in rows.items()}
    f0b, nharb, nheb, hasb = np.zeros_like(f0), np.zeros_like(nhar), np.zeros_like(nhe), np.zeros_like(has)
    dst = view(back, f0b, nharb, nheb, hasb)
    assert L.llsm_chunk_to_flat(ch, C.byref(dst), 0) == 0
    assert np.array_equal(f0b, f0) and np.array_equal(nharb, nhar) and np.array_equal(nheb, nhe)
    assert np.array_equal(hasb, has)
    for k in rows:
        assert np.array_equal(back[k], rows[k]), k
    L.llsm_delete_chunk(ch); L.llsm_delete_container(conf)


def test_chunk_to_flat_pads_and_truncates():
    """llsm_chunk_to_flat into rows of ANOTHER shape than the frames': more harmonics than the rows hold are cut
    (nhar reports the cut count), missing PSD bins read -120 dB, missing channels 1e-5 with empty envelopes, envelope
    harmonics beyond the rows' limit are cut, frames without PSDRES give zeros and has_psdres = 0, a shorter PSDRES is
    zero-padded (the block-copy form of the loop must keep every one of these)."""
    L = llsm.load()
    rng = np.random.default_rng(1)
    F, mh, me, npsd, nch = 5, 10, 4, 16, 4
    ao = llsm.make_aoptions(maxnhar=mh, maxnhar_e=me, npsd=npsd)
    conf = L.llsm_aoptions_toconf(C.byref(ao), 22050.0)
    C.cast(L.llsm_container_get(conf, llsm.CONF_NFRM), llsm.P_int)[0] = F
    ch = L.llsm_create_chunk(conf, 1)
    f0 = np.array([0, 100, 120, 130, 140], np.float32)
    nhar = np.array([0, 9, 4, 10, 7], np.int32); nhe = np.array([0, 4, 3, 1, 4], np.int32)
    rows = dict(ampl=rng.random((F, mh), np.float32) + 1, phse=rng.random((F, mh), np.float32) + 1,
                psd=rng.random((F, npsd), np.float32) + 1, psdres=rng.random((F, npsd), np.float32) + 1,
                edc=rng.random((F, nch), np.float32) + 1, ea=rng.random((F, nch, me), np.float32) + 1,
                ep=rng.random((F, nch, me), np.float32) + 1)
    has = np.array([1, 1, 0, 1, 1], np.int32)

    def view(d, shape, f0_, nhar_, nhe_, has_):
        v = llsm.FlatParams()
        v.maxnhar, v.maxnhar_e, v.npsd, v.nchannel = shape
        v.f0 = f0_.ctypes.data_as(llsm.P_fp); v.nhar = nhar_.ctypes.data_as(llsm.P_int)
        v.ampl = d["ampl"].ctypes.data_as(llsm.P_fp); v.phse = d["phse"].ctypes.data_as(llsm.P_fp)
        v.psd = d["psd"].ctypes.data_as(llsm.P_fp); v.psdres = d["psdres"].ctypes.data_as(llsm.P_fp)
        v.has_psdres = has_.ctypes.data_as(llsm.P_int); v.edc = d["edc"].ctypes.data_as(llsm.P_fp)
        v.nhar_e = nhe_.ctypes.data_as(llsm.P_int)
        v.eenv_ampl = d["ea"].ctypes.data_as(llsm.P_fp); v.eenv_phse = d["ep"].ctypes.data_as(llsm.P_fp)
        return v

    src = view(rows, (mh, me, npsd, nch), f0, nhar, nhe, has)
    assert L.llsm_flat_to_chunk(C.byref(src), 0, ch) == 0
    # frame 4 gets a PSDRES of 5 values only
    short = L.llsm_create_fparray(5)
    for j in range(5):
        short[j] = 40.0 + j
    L.llsm_container_attach_(ch.contents.frames[4], llsm.FRAME_PSDRES, C.cast(short, C.c_void_p),
                             C.cast(L.llsm_delete_fparray, C.c_void_p), C.cast(L.llsm_copy_fparray, C.c_void_p))
    mh2, me2, npsd2, nch2 = 6, 2, 20, 5
    back = dict(ampl=np.full((F, mh2), -7, np.float32), phse=np.full((F, mh2), -7, np.float32),
                psd=np.full((F, npsd2), -7, np.float32), psdres=np.full((F, npsd2), -7, np.float32),
                edc=np.full((F, nch2), -7, np.float32), ea=np.full((F, nch2, me2), -7, np.float32),
                ep=np.full((F, nch2, me2), -7, np.float32))
    f0b, nharb, nheb, hasb = np.zeros_like(f0), np.zeros_like(nhar), np.zeros_like(nhe), np.zeros_like(has)
    dst = view(back, (mh2, me2, npsd2, nch2), f0b, nharb, nheb, hasb)
    assert L.llsm_chunk_to_flat(ch, C.byref(dst), 0) == 0
    assert np.array_equal(f0b, f0) and np.array_equal(hasb, has)
    assert np.array_equal(nharb, np.minimum(nhar, mh2)) and np.array_equal(nheb, np.minimum(nhe, me2))
    for i in range(F):
        n = min(int(nhar[i]), mh2)
        assert np.array_equal(back["ampl"][i, :n], rows["ampl"][i, :n]) and np.all(back["ampl"][i, n:] == 0)
        assert np.array_equal(back["phse"][i, :n], rows["phse"][i, :n]) and np.all(back["phse"][i, n:] == 0)
        assert np.array_equal(back["psd"][i, :npsd], rows["psd"][i]) and np.all(back["psd"][i, npsd:] == -120.0)
        assert np.array_equal(back["edc"][i, :nch], rows["edc"][i]) and back["edc"][i, nch] == np.float32(1e-5)
        k = min(int(nhe[i]), me2) if f0[i] > 0 else 0
        assert np.array_equal(back["ea"][i, :nch, :k], rows["ea"][i, :, :k]) and np.all(back["ea"][i, :nch, k:] == 0)
        assert np.array_equal(back["ep"][i, :nch, :k], rows["ep"][i, :, :k]) and np.all(back["ep"][i, nch:] == 0)
    assert np.array_equal(back["psdres"][1, :npsd], rows["psdres"][1]) and np.all(back["psdres"][1, npsd:] == 0)
    assert np.all(back["psdres"][2] == 0)                                     # no PSDRES on that frame
    assert np.array_equal(back["psdres"][4, :5], 40.0 + np.arange(5, dtype=np.float32)) and np.all(back["psdres"][4, 5:] == 0)
    L.llsm_delete_chunk(ch); L.llsm_delete_container(conf)
