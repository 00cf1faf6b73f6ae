"""In-process A/B of llsm_gpu_analysis_overlap: analysis step times alternating off / on (the same batch, the same clocks)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import libllsm2_amd as llsm
from conftest import make_utterance, FS
L = llsm.load()
L.llsm_gpu_analysis_overlap.argtypes = [C.c_int]
ctx = llsm.Context(0)
U = 1024
xs = [make_utterance(u, 120.0) for u in range(4)]
x = np.concatenate([xs[u % 4] for u in range(U)])
f0 = np.full(200 * U, 120.0, np.float32)
b = llsm.Batch(ctx, llsm.make_aoptions(f0_refine=0), FS, [44100] * U, [200] * U)
b.upload(llsm.A_X, x); b.upload(llsm.A_F0, f0)
for _ in range(3):
    b.analyze()
ctx.sync()
res = {0: [], 1: []}
for rnd in range(12):
    for on in (0, 1):
        L.llsm_gpu_analysis_overlap(on)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(5):
            b.analyze()
        ctx.sync(); res[on].append((time.perf_counter() - t0) / 5 * 1e3)
for on in (0, 1):
    v = np.array(res[on]); print("overlap", on, "analysis ms/step: median %.4f  min %.4f  mean %.4f" % (np.median(v), v.min(), v.mean()))
