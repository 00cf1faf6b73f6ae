"""Throughput of the reference-object-model batch calls (llsm_analyze_batch / llsm_synthesize_batch of
include/llsm_gpu.h: arrays of waveforms in, llsm_chunk / llsm_output objects out) -- what a libllsm2 host that
keeps its containers pays, host-side object construction and PCIe included.  Config 2 utterances.

    python tools/bench_chunk_api.py [--utts 1024] [--workers 2] [--block 256] [--reps 3]
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
    ap.add_argument("--utts", type=int, default=1024)
    ap.add_argument("--workers", type=int, default=2)
    ap.add_argument("--block", type=int, default=256)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--batch-delete", type=int, default=0, help="1: llsm_delete_chunks(chunks, n) instead of n llsm_delete_chunk calls")
    a = ap.parse_args()
    L = llsm.load()
    U, nfrm = a.utts, 200
    base = [make_utterance(u, 120.0) for u in range(8)]
    xs = [base[u % 8] for u in range(U)]
    f0s = [np.full(nfrm, 120.0, np.float32) for _ in range(U)]
    xp = (llsm.P_fp * U)(*[x.ctypes.data_as(llsm.P_fp) for x in xs])
    fp = (llsm.P_fp * U)(*[f.ctypes.data_as(llsm.P_fp) for f in f0s])
    nx = (C.c_int * U)(*[len(x) for x in xs]); nf = (C.c_int * U)(*([nfrm] * U))
    chunks = (C.POINTER(llsm.Chunk) * U)(); outs = (C.POINTER(llsm.Output) * U)()
    ao = llsm.make_aoptions(f0_refine=0); so = llsm.make_soptions(FS)
    L.llsm_analyze_batch.argtypes = [C.POINTER(llsm.AOptions), C.POINTER(llsm.P_fp), llsm.P_int, C.c_float, C.POINTER(llsm.P_fp),
                                     llsm.P_int, C.c_int, C.POINTER(C.POINTER(llsm.Chunk)), C.POINTER(llsm.P_fp)]
    L.llsm_synthesize_batch.argtypes = [C.POINTER(llsm.SOptions), C.POINTER(C.POINTER(llsm.Chunk)), C.c_int,
                                        C.POINTER(C.POINTER(llsm.Output))]
    L.llsm_gpu_set_fanout(1, a.workers, a.block)
    ta, ts, td, tdo = [], [], [], []
    for it in range(a.reps + 1):
        t0 = time.perf_counter()
        rc = L.llsm_analyze_batch(C.byref(ao), xp, nx, FS, fp, nf, U, chunks, None)
        t1 = time.perf_counter()
        assert rc == 0, L.llsm_gpu_last_error()
        rc = L.llsm_synthesize_batch(C.byref(so), chunks, U, outs)
        t2 = time.perf_counter()
        assert rc == 0, L.llsm_gpu_last_error()
        for u in range(U):
            L.llsm_delete_output(outs[u])
        t3o = time.perf_counter()
        if a.batch_delete:
            L.llsm_delete_chunks(chunks, U)
        else:
            for u in range(U):
                L.llsm_delete_chunk(chunks[u])
        t3 = time.perf_counter()
        if it:
            ta.append(t1 - t0); ts.append(t2 - t1); td.append(t3 - t2); tdo.append(t3o - t2)
    fr = U * nfrm
    print(json.dumps({"metric": "frames/s through llsm_analyze_batch + llsm_synthesize_batch (llsm_chunk objects)",
                      "utterances": U, "workers": a.workers, "block": a.block,
                      "analyze_ms": float(np.median(ta)) * 1e3, "synthesize_ms": float(np.median(ts)) * 1e3,
                      "delete_objects_ms": float(np.median(td)) * 1e3, "delete_outputs_ms": float(np.median(tdo)) * 1e3,
                      "delete_chunks_ms": float(np.median(td) - np.median(tdo)) * 1e3,
                      "value_excluding_delete": fr / (float(np.median(ta)) + float(np.median(ts))),
                      "value": fr / (float(np.median(ta)) + float(np.median(ts)) + float(np.median(td))), "unit": "frames/s",
                      "note": "value = analyse + synthesise + delete every object (VERDICT r4 item 3)"}))


if __name__ == "__main__":
    main()
