import sys, os, ctypes as C, threading
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, psutil, torch
import libllsm2_amd as llsm
from conftest import make_speechlike, FS
from test_gpu_rt import rt_run
L = llsm.load()
L.llsm_analyze.restype = C.POINTER(llsm.Chunk); L.llsm_synthesize.restype = C.POINTER(llsm.Output)
ao = llsm.make_aoptions(f0_refine=1); so = llsm.make_soptions(FS)
x, f0 = make_speechlike(7, nx=20000); f0 = f0.astype(np.float32)
proc = psutil.Process()
def mem():
    free, tot = torch.cuda.mem_get_info()
    return proc.memory_info().rss / 1e6, (tot - free) / 1e6
def once(l1):
    f = f0.copy()
    ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f.ctypes.data_as(llsm.P_fp), len(f), None)
    assert ch
    if l1:
        L.llsm_chunk_tolayer1(ch, 2048); L.llsm_chunk_tolayer0(ch)
    s2 = llsm.make_soptions(FS, use_l1=1 if l1 else 0)
    out = L.llsm_synthesize(C.byref(s2), ch); assert out, L.llsm_gpu_last_error()
    if not l1:
        yp, yap, lat = rt_run(L, so, ch, len(f))
    L.llsm_delete_output(out); L.llsm_delete_chunk(ch)
for i in range(5): once(i % 2)
m0 = mem(); print("start rss %.1f MB, device %.1f MB" % m0, flush=True)
for i in range(300):
    once(i % 2)
    if i % 100 == 99: print(i, "rss %.1f MB, device %.1f MB" % mem(), flush=True)
m1 = mem()
print("growth rss %.1f MB device %.1f MB over 300 rounds" % (m1[0] - m0[0], m1[1] - m0[1]))
