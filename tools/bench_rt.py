"""BASELINE.json config 4: llsmrt pull loop, 64 concurrent streams per GPU, 256-sample pulls
(harmonic-model path; the PbP variant needs layer 1, out of scope).

    python tools/bench_rt.py [--streams 64] [--seconds 2]

One JSON line: synthesised frames/s over all streams, real-time factor per stream, and the
distribution of the time one feed (one hop for all streams) + the pulls take.
"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import libllsm2_amd as llsm
from conftest import FS, make_utterance

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=64)
ap.add_argument("--seconds", type=float, default=2.0)
a = ap.parse_args()
L = llsm.load()
S, nfrm = a.streams, int(a.seconds * 200)
ao = llsm.make_aoptions(f0_refine=0)
so = llsm.make_soptions(FS)
chunks = []
for u in range(4):                                   # four distinct parameter sets, reused round-robin
    x = np.tile(make_utterance(u, 120.0 + 15 * u), int(np.ceil(a.seconds)))[: int(a.seconds * FS)]
    f0 = np.full(nfrm, 120.0 + 15 * u, np.float32)
    ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f0.ctypes.data_as(llsm.P_fp), nfrm, None)
    assert bool(ch), L.llsm_gpu_last_error()
    chunks.append(ch)
g = L.llsm_create_rtsynth_group(C.byref(so), chunks[0].contents.conf, 8192, S)
assert g, L.llsm_gpu_last_error()
FrameArr = C.POINTER(llsm.Container) * S
bp = np.zeros(256, np.float32); bap = np.zeros(256, np.float32)
pp, pap = bp.ctypes.data_as(llsm.P_fp), bap.ctypes.data_as(llsm.P_fp)
frames = [FrameArr(*[chunks[s % 4].contents.frames[i] for s in range(S)]) for i in range(nfrm)]
for i in range(5):                                   # warm-up hops
    L.llsm_rtsynth_group_feed(g, frames[i])
feed_ms, pulled = [], 0
t0 = time.perf_counter()
for i in range(5, nfrm):
    t1 = time.perf_counter()
    L.llsm_rtsynth_group_feed(g, frames[i])
    feed_ms.append((time.perf_counter() - t1) * 1e3)
    for s in range(S):
        while L.llsm_rtsynth_group_numoutput(g, s) >= 256:
            pulled += L.llsm_rtsynth_group_fetch(g, s, pp, pap, 256)
dt = time.perf_counter() - t0
L.llsm_delete_rtsynth_group(g)
n = nfrm - 5
feed_ms = np.array(feed_ms)
print(json.dumps({"metric": "llsmrt synthesised frames/sec (harmonic path)", "value": S * n / dt, "unit": "frames/s",
                  "streams": S, "hops": n, "pull": 256, "samples_pulled": pulled,
                  "realtime_factor_per_stream": (n * 0.005) / dt,
                  "feed_ms": {"median": float(np.median(feed_ms)), "p99": float(np.percentile(feed_ms, 99)), "max": float(feed_ms.max())},
                  "hop_ms": 5.0, "latency_samples": 732}))
