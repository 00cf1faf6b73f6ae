"""Flat chunk wire format throughput: llsm_chunk_to_blob, llsm_blob_view (validation only),
llsm_blob_to_chunk and llsm_gpu_batch_upload_blob + use of the rows by a synthesis, for config-2 chunks.

    python tools/bench_wire.py [--utts 256] [--reps 5]
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
    ap.add_argument("--utts", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    L = llsm.load()
    nfrm, U = 200, a.utts
    ao = llsm.make_aoptions(f0_refine=0)
    L.llsm_analyze.restype = C.POINTER(llsm.Chunk)
    L.llsm_chunk_blob_size.restype = C.c_size_t
    L.llsm_chunk_blob_size.argtypes = [C.POINTER(llsm.Chunk)]
    L.llsm_chunk_to_blob.restype = C.c_longlong
    L.llsm_chunk_to_blob.argtypes = [C.POINTER(llsm.Chunk), C.c_void_p, C.c_size_t]
    L.llsm_blob_to_chunk.restype = C.POINTER(llsm.Chunk)
    L.llsm_blob_to_chunk.argtypes = [C.c_void_p, C.c_size_t]
    L.llsm_blob_view.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(llsm.FlatParams), llsm.P_int, llsm.P_fp, llsm.P_fp]
    L.llsm_gpu_batch_upload_blob.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    chunks = []
    for u in range(U):
        x = make_utterance(u % 8, 120.0); f0 = np.full(nfrm, 120.0, np.float32)
        ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), len(x), FS, f0.ctypes.data_as(llsm.P_fp), nfrm, None)
        assert ch, L.llsm_gpu_last_error()
        chunks.append(ch)
    sizes = [L.llsm_chunk_blob_size(ch) for ch in chunks]
    bufs = [np.zeros((s + 7) // 8, np.uint64) for s in sizes]           # 8-byte aligned
    tot = sum(sizes)
    t_enc, t_view, t_dec, t_up, t_upb = [], [], [], [], []
    L.llsm_gpu_batch_upload_blobs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    ptrs = (C.c_void_p * U)(*[bf.ctypes.data for bf in bufs]); szs = (C.c_size_t * U)(*sizes)
    ctx = llsm.Context(0)
    b = llsm.Batch(ctx, ao, FS, [44100] * U, [nfrm] * U)
    v = llsm.FlatParams(); nf = C.c_int(0); th = C.c_float(0); fn = C.c_float(0)
    for it in range(a.reps + 1):
        t0 = time.perf_counter()
        for ch, bf, s in zip(chunks, bufs, sizes):
            assert L.llsm_chunk_to_blob(ch, bf.ctypes.data, s) == s
        t1 = time.perf_counter()
        for bf, s in zip(bufs, sizes):
            assert L.llsm_blob_view(bf.ctypes.data, s, C.byref(v), C.byref(nf), C.byref(th), C.byref(fn)) == 0
        t2 = time.perf_counter()
        back = [L.llsm_blob_to_chunk(bf.ctypes.data, s) for bf, s in zip(bufs, sizes)]
        t3 = time.perf_counter()
        for c2 in back:
            assert c2
            L.llsm_delete_chunk(c2)
        t4 = time.perf_counter()
        for u, (bf, s) in enumerate(zip(bufs, sizes)):
            assert L.llsm_gpu_batch_upload_blob(b.h, u, bf.ctypes.data, s) == 0, L.llsm_gpu_last_error()
        ctx.sync()
        t5 = time.perf_counter()
        assert L.llsm_gpu_batch_upload_blobs(b.h, 0, U, ptrs, szs) == 0, L.llsm_gpu_last_error()
        ctx.sync()
        t6 = time.perf_counter()
        if it:
            t_enc.append(t1 - t0); t_view.append(t2 - t1); t_dec.append(t3 - t2); t_up.append(t5 - t4); t_upb.append(t6 - t5)
    b.synthesize(llsm.make_soptions(FS), seed=3); ctx.sync()
    y = b.download(llsm.A_Y)
    med = lambda q: float(np.median(q))
    fr = U * nfrm
    print(json.dumps({"metric": "wire format v2, config-2 chunks", "chunks": U, "frames": fr, "blob_bytes_per_frame": tot / fr,
                      "chunk_to_blob": {"MB_per_s": tot / med(t_enc) / 1e6, "frames_per_s": fr / med(t_enc)},
                      "blob_view_validate": {"MB_per_s": tot / med(t_view) / 1e6, "frames_per_s": fr / med(t_view)},
                      "blob_to_chunk": {"MB_per_s": tot / med(t_dec) / 1e6, "frames_per_s": fr / med(t_dec)},
                      "batch_upload_blob": {"MB_per_s": tot / med(t_up) / 1e6, "frames_per_s": fr / med(t_up)},
                      "batch_upload_blobs": {"MB_per_s": tot / med(t_upb) / 1e6, "frames_per_s": fr / med(t_upb)},
                      "synthesis_from_blob_rows_finite": bool(np.all(np.isfinite(y)) and float(np.abs(y).max()) > 0.05)}))
    b.close(); ctx.close()


if __name__ == "__main__":
    main()
