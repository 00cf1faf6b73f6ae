"""Depth-first sub-batches against one resident batch ("spend the 256 MB Infinity Cache").
The 1024-utterance batch of the bench as P batches of 1024 / P utterances on one context; a step = for every
sub-batch: analyse + synthesise (so a sub-batch's intermediates -- two 513-bin planes, four sub-band planes -- are
re-read by the next kernel while they still sit in the memory-side cache), against the breadth-first step of the one
big batch.  Same kernels, same results; what changes is the working set per launch and the grid sizes.
    python tools/ab_subbatch.py [utts] [steps]   ->  one line per P in 1, 2, 4, 8, 16"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import libllsm2_amd as llsm
from conftest import FS, make_utterance
U = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = llsm.Context(0)
ao = llsm.make_aoptions(f0_refine=0); so = llsm.make_soptions(FS)
base = [make_utterance(u, 120.0) for u in range(8)]
out = []
for P in (1, 2, 4, 8, 16):
    n = U // P
    bs = []
    for p in range(P):
        b = llsm.Batch(ctx, ao, FS, [44100] * n, [200] * n)
        b.upload(llsm.A_X, np.concatenate([base[(p * n + u) % 8] for u in range(n)])); b.upload(llsm.A_F0, np.full(n * 200, 120.0, np.float32))
        bs.append(b)
    def step(i, analysis_only=False):
        for b in bs:
            b.analyze()
            if not analysis_only:
                b.synthesize(so, seed=i)
    for i in range(2):
        step(i)
    ctx.sync()
    best, best_a = 1e9, 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(K):
            step(i)
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
        t0 = time.perf_counter()
        for i in range(K):
            step(i, True)
        ctx.sync()
        best_a = min(best_a, (time.perf_counter() - t0) / K * 1e3)
    ctx.set_profiling(True); ctx.reset_profile()
    for i in range(3):
        step(i)
    ctx.sync()
    prof = ctx.profile(); ctx.set_profiling(False)
    r = {"sub_batches": P, "utts_each": n, "ms_per_step": round(best, 3), "analysis_ms": round(best_a, 3),
         "kernels_ms_per_step": {k: round(v[0] / 3, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]}}
    print(json.dumps(r), flush=True)
    out.append(r)
    for b in bs:
        b.close()
ctx.close()
