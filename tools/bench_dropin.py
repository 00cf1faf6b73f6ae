"""Latency of the drop-in entry points (llsm_analyze / llsm_synthesize, one utterance per call),
BASELINE.json configs[0]: test/arctic_a0001.wav with the test-layer0-anasynth options."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libllsm2_amd as llsm  # noqa: E402


def main():
    from scipy.io import wavfile
    fs, x = wavfile.read(os.path.join(ROOT, "tests", "golden", "arctic_a0001.wav"))
    x = (x.astype(np.float32) / 32768.0) if x.dtype != np.float32 else x
    f0 = np.load(os.path.join(ROOT, "tests", "golden", "arctic_a0001_f0_hop128.npy")).astype(np.float32)
    L = llsm.load()
    ao = llsm.make_aoptions(thop=128.0 / fs, npsd=128, maxnhar=400, maxnhar_e=5, f0_refine=0)
    so = llsm.make_soptions(float(fs))
    ta, ts = [], []
    for it in range(12):
        f = f0.copy()
        t0 = time.perf_counter()
        ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), float(fs),
                            f.ctypes.data_as(llsm.P_fp), len(f), None)
        t1 = time.perf_counter()
        out = L.llsm_synthesize(C.byref(so), ch)
        t2 = time.perf_counter()
        assert bool(ch) and bool(out)
        L.llsm_delete_output(out); L.llsm_delete_chunk(ch)
        if it >= 2:
            ta.append(t1 - t0); ts.append(t2 - t1)
    print(json.dumps({"metric": "drop-in call latency, arctic_a0001 (%.2f s, %d frames)" % (len(x) / fs, len(f0)),
                      "llsm_analyze_ms": {"median": float(np.median(ta)) * 1e3, "max": float(np.max(ta)) * 1e3},
                      "llsm_synthesize_ms": {"median": float(np.median(ts)) * 1e3, "max": float(np.max(ts)) * 1e3},
                      "frames_per_s_analyze_plus_synth": len(f0) / (float(np.median(ta)) + float(np.median(ts)))}))


if __name__ == "__main__":
    main()
