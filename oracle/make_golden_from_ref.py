"""Regenerates tests/golden/ref_*.npz from the REAL reference built by oracle/build_ref.sh
(oracle/_ref/libllsm2_ref.so).  Not runnable in this image (ciglet absent); committed so that parity can be
pinned the moment a ciglet checkout is available.  The vectors are data: inputs and the reference's outputs.

    python oracle/make_golden_from_ref.py

Written: tests/golden/ref_arctic_layer0.npz   (test/test-layer0-anasynth.c options, 'czt' method, committed F0
track): per-frame nhar / ampl / phse / psd / psdres / edc / eenv, the residual, and y_sin of llsm_synthesize
(the noise part depends on libc rand() and is not stored).  tests/test_ref_golden.py then checks the oracle
(float32 build) and, on the GPU box, the HIP path against them with the tolerances of SURVEY 8(d)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import libllsm2_amd as llsm                      # only for the ctypes struct layouts (ABI of llsm.h)
from verify_utils import GOLDEN, read_wav

REF = os.path.join(ROOT, "oracle", "_ref", "libllsm2_ref.so")


def main():
    if not os.path.exists(REF):
        raise SystemExit(f"{REF} missing: run oracle/build_ref.sh /path/to/ciglet first")
    R = C.CDLL(REF)
    fp, P, PI = C.c_float, llsm.P_fp, llsm.P_int
    R.llsm_create_aoptions.restype = C.POINTER(llsm.AOptions)
    R.llsm_create_soptions.restype = C.POINTER(llsm.SOptions); R.llsm_create_soptions.argtypes = [fp]
    R.llsm_analyze.restype = C.POINTER(llsm.Chunk)
    R.llsm_analyze.argtypes = [C.POINTER(llsm.AOptions), P, C.c_int, fp, P, C.c_int, C.POINTER(P)]
    R.llsm_synthesize.restype = C.POINTER(llsm.Output); R.llsm_synthesize.argtypes = [C.POINTER(llsm.SOptions), C.POINTER(llsm.Chunk)]
    R.llsm_container_get.restype = C.c_void_p; R.llsm_container_get.argtypes = [C.POINTER(llsm.Container), C.c_int]
    x, fs = read_wav(os.path.join(GOLDEN, "arctic_a0001.wav"))
    f0 = np.load(os.path.join(GOLDEN, "arctic_a0001_f0_hop128.npy")).copy()
    nfrm = len(f0)
    ao = R.llsm_create_aoptions().contents
    ao.thop, ao.npsd, ao.maxnhar, ao.maxnhar_e, ao.f0_refine, ao.hm_method = 128.0 / fs, 128, 400, 5, 0, llsm.HMCZT
    xap = P()
    ch = R.llsm_analyze(C.byref(ao), x.ctypes.data_as(P), len(x), fs, f0.ctypes.data_as(P), nfrm, C.byref(xap))
    mh, me, nch, npsd = 400, 5, 4, 128
    out = dict(f0=f0, nhar=np.zeros(nfrm, np.int32), ampl=np.zeros((nfrm, mh), np.float32), phse=np.zeros((nfrm, mh), np.float32),
               psd=np.zeros((nfrm, npsd), np.float32), psdres=np.zeros((nfrm, npsd), np.float32), edc=np.zeros((nfrm, nch), np.float32),
               nhar_e=np.zeros(nfrm, np.int32), eenv_ampl=np.zeros((nfrm, nch, me), np.float32), eenv_phse=np.zeros((nfrm, nch, me), np.float32),
               xres=np.ctypeslib.as_array(xap, (len(x),)).copy())
    for i in range(nfrm):
        fr = ch.contents.frames[i]
        hm = C.cast(R.llsm_container_get(fr, llsm.FRAME_HM), C.POINTER(llsm.HMFrame))
        nm = C.cast(R.llsm_container_get(fr, llsm.FRAME_NM), C.POINTER(llsm.NMFrame)).contents
        res = C.cast(R.llsm_container_get(fr, llsm.FRAME_PSDRES), P)
        if hm and f0[i] > 0:
            n = hm.contents.nhar; out["nhar"][i] = n
            out["ampl"][i, :n] = np.ctypeslib.as_array(hm.contents.ampl, (n,)); out["phse"][i, :n] = np.ctypeslib.as_array(hm.contents.phse, (n,))
        out["psd"][i] = np.ctypeslib.as_array(nm.psd, (npsd,)); out["edc"][i] = np.ctypeslib.as_array(nm.edc, (nch,))
        if res:
            out["psdres"][i] = np.ctypeslib.as_array(res, (npsd,))
        for c in range(nch):
            e = nm.eenv[c].contents
            out["nhar_e"][i] = max(out["nhar_e"][i], e.nhar)
            out["eenv_ampl"][i, c, :e.nhar] = np.ctypeslib.as_array(e.ampl, (e.nhar,)) if e.nhar else 0
            out["eenv_phse"][i, c, :e.nhar] = np.ctypeslib.as_array(e.phse, (e.nhar,)) if e.nhar else 0
    so = R.llsm_create_soptions(fs)
    y = R.llsm_synthesize(so, ch).contents
    out["y_sin"] = np.ctypeslib.as_array(y.y_sin, (y.ny,)).copy()
    np.savez_compressed(os.path.join(GOLDEN, "ref_arctic_layer0.npz"), **out)
    print("wrote tests/golden/ref_arctic_layer0.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
