"""Frame coder throughput: llsm_coder_encode_frames / llsm_coder_decode_frames over the frames of
analysed + layer-1 converted config-2 utterances (containers in, containers out: the object model's host cost
included; the kernel times are in the rocprofv3 trace of this command, profiles/r02_*_coder_kernel_stats.txt).

    python tools/bench_coder.py [--utts 64] [--reps 5]
"""
import argparse
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
from conftest import FS, make_utterance  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    L = llsm.load()
    nfrm = 200
    ao = llsm.make_aoptions(f0_refine=0)
    L.llsm_analyze.restype = C.POINTER(llsm.Chunk)
    chunks = []
    for u in range(a.utts):
        x = make_utterance(u % 8, 120.0); f0 = np.full(nfrm, 120.0, np.float32)
        ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f0.ctypes.data_as(llsm.P_fp), nfrm, None)
        assert ch, L.llsm_gpu_last_error()
        L.llsm_chunk_tolayer1(ch, 2048)
        chunks.append(ch)
    n = a.utts * nfrm
    frames = (C.POINTER(llsm.Container) * n)()
    for u, ch in enumerate(chunks):
        for i in range(nfrm):
            frames[u * nfrm + i] = ch.contents.frames[i]
    L.llsm_create_coder.restype = C.c_void_p
    coder = C.c_void_p(L.llsm_create_coder(chunks[0].contents.conf, 64, 5))
    L.llsm_coder_dimension.argtypes = [C.c_void_p]
    dim = L.llsm_coder_dimension(coder)
    enc = np.zeros((n, dim), np.float32)
    L.llsm_coder_encode_frames.argtypes = [C.c_void_p, C.POINTER(C.POINTER(llsm.Container)), C.c_int, llsm.P_fp]
    L.llsm_coder_decode_frames.argtypes = [C.c_void_p, llsm.P_fp, C.c_int, C.c_int, C.POINTER(C.POINTER(llsm.Container))]
    outs = (C.POINTER(llsm.Container) * n)()
    te, t0l, t1l = [], [], []
    for it in range(a.reps + 1):
        t = time.perf_counter()
        assert L.llsm_coder_encode_frames(coder, frames, n, enc.ctypes.data_as(llsm.P_fp)) == 0, L.llsm_gpu_last_error()
        t1 = time.perf_counter()
        assert L.llsm_coder_decode_frames(coder, enc.ctypes.data_as(llsm.P_fp), n, 0, outs) == 0, L.llsm_gpu_last_error()
        t2 = time.perf_counter()
        for i in range(n):
            L.llsm_delete_container(outs[i])
        t3 = time.perf_counter()
        assert L.llsm_coder_decode_frames(coder, enc.ctypes.data_as(llsm.P_fp), n, 1, outs) == 0, L.llsm_gpu_last_error()
        t4 = time.perf_counter()
        for i in range(n):
            L.llsm_delete_container(outs[i])
        if it:
            te.append(t1 - t); t0l.append(t2 - t1); t1l.append(t4 - t3)
    med = lambda v: float(np.median(v))
    print(json.dumps({"metric": "frames/s, frame coder (order_spec 64, order_bap 5)", "frames": n, "dimension": dim,
                      "encode_frames_per_s": n / med(te), "decode_layer0_frames_per_s": n / med(t0l),
                      "decode_layer1_frames_per_s": n / med(t1l),
                      "encode_ms": med(te) * 1e3, "decode_layer0_ms": med(t0l) * 1e3, "decode_layer1_ms": med(t1l) * 1e3,
                      "finite": bool(np.all(np.isfinite(enc)))}))


if __name__ == "__main__":
    main()
