"""ms per analyse + synthesise step of a resident batch outside bench.py (AB_PROF=1: with the per-kernel event timing
the bench keeps on in its timed region; AB_DISTINCT=1: 1024 distinct utterances instead of 8 repeated ones).
    python tools/ab_step.py [utts] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import libllsm2_amd as llsm
from conftest import FS, make_utterance
U = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = llsm.Context(0)
ao = llsm.make_aoptions(f0_refine=0); so = llsm.make_soptions(FS)
nb = U if os.environ.get("AB_DISTINCT") else 8
base = [make_utterance(u, 120.0) for u in range(nb)]
x = np.concatenate([base[u % nb] for u in range(U)]); f0 = np.full(U * 200, 120.0, np.float32)
b = llsm.Batch(ctx, ao, FS, [44100] * U, [200] * U)
b.upload(llsm.A_X, x); b.upload(llsm.A_F0, f0)
for i in range(3):
    b.analyze(); b.synthesize(so, seed=i)
ctx.sync()
if os.environ.get("AB_PROF"):
    ctx.set_profiling(True)
ta = []
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(K):
        b.analyze(); b.synthesize(so, seed=i)
    ctx.sync()
    ta.append((time.perf_counter() - t0) / K * 1e3)
t0 = time.perf_counter()
for i in range(K):
    b.analyze()
ctx.sync()
tan = (time.perf_counter() - t0) / K * 1e3
print("prof=%s distinct=%s: %.3f ms per step (best of 3 x %d), analysis alone %.3f ms" % (os.environ.get("AB_PROF", "0"), os.environ.get("AB_DISTINCT", "0"), min(ta), K, tan))
