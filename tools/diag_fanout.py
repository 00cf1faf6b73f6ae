import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import libllsm2_amd as llsm
from conftest import FS, make_speechlike
L = llsm.load()
AB = L.llsm_analyze_batch
AB.argtypes = [C.POINTER(llsm.AOptions), C.POINTER(llsm.P_fp), llsm.P_int, C.c_float, C.POINTER(llsm.P_fp), llsm.P_int, C.c_int, C.POINTER(C.POINTER(llsm.Chunk)), C.POINTER(llsm.P_fp)]
U = 8
xs, f0s = [], []
for u in range(U):
    x, f0 = make_speechlike(60 + u, nx=7000 + 900 * u)
    xs.append(np.ascontiguousarray(x, np.float32)); f0s.append(np.ascontiguousarray(f0, np.float32))
ao = llsm.make_aoptions(f0_refine=0)
nx = np.array([len(x) for x in xs], np.int32); nf = np.array([len(f) for f in f0s], np.int32)
xp = (llsm.P_fp * U)(*[x.ctypes.data_as(llsm.P_fp) for x in xs]); fp_ = (llsm.P_fp * U)(*[f.ctypes.data_as(llsm.P_fp) for f in f0s])
def run(d, w, b):
    L.llsm_gpu_set_fanout(d, w, b)
    chunks = (C.POINTER(llsm.Chunk) * U)()
    assert AB(C.byref(ao), xp, nx.ctypes.data_as(llsm.P_int), FS, fp_, nf.ctypes.data_as(llsm.P_int), U, chunks, None) == 0
    res = []
    for u in range(U):
        rows = []
        for i in range(int(nf[u])):
            nm = C.cast(L.llsm_container_get(chunks[u].contents.frames[i], llsm.FRAME_NM), C.POINTER(llsm.NMFrame)).contents
            rows.append(np.r_[np.ctypeslib.as_array(nm.psd, (nm.npsd,)), np.ctypeslib.as_array(nm.edc, (4,))])
        res.append(np.array(rows)); L.llsm_delete_chunk(chunks[u])
    return res
ref = run(1, 1, 1000)
for cfg in ((1, 1, 1000), (1, 1, 3), (1, 1, 1), (1, 2, 3), (1, 2, 3), (1, 3, 1)):
    got = run(*cfg)
    out = []
    for u in range(U):
        d = np.abs(got[u] - ref[u])
        if d.max() > 0:
            fr = np.flatnonzero(d.max(axis=1) > 0)
            out.append((u, float(d.max()), int(len(fr)), fr[:6].tolist(), int(nf[u])))
    print(cfg, out)
