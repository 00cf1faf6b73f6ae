"""PCIe-inclusive rate of the bench workload (DESIGN.md "Roofline accounting"): upload of the
waveforms and F0 tracks, analysis + synthesis, download of every parameter row and waveform,
through llsm_gpu_batch_upload / _download from ordinary (pageable) host memory."""
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


def main(utts=1024):
    ctx = llsm.Context(0)
    xs = [make_utterance(u, 120.0) for u in range(4)]
    x = np.concatenate([xs[u % 4] for u in range(utts)])
    f0 = np.full(200 * utts, 120.0, np.float32)
    b = llsm.Batch(ctx, llsm.make_aoptions(f0_refine=0), FS, [44100] * utts, [200] * utts)
    so = llsm.make_soptions(FS)
    ids = [llsm.A_F0, llsm.A_NHAR, llsm.A_AMPL, llsm.A_PHSE, llsm.A_PSD, llsm.A_PSDRES, llsm.A_EDC,
           llsm.A_NHAR_E, llsm.A_EENV_AMPL, llsm.A_EENV_PHSE, llsm.A_Y, llsm.A_YSIN, llsm.A_YNOISE]
    frames = 200 * utts
    out = {"utterances": utts}
    for mode in ("pageable", "pinned"):
        if mode == "pinned":                                    # llsm_gpu_alloc_host buffers
            xin = b.pinned_array(llsm.A_X); xin[:] = x
            fin = b.pinned_array(llsm.A_F0); fin[:] = f0
            dst = {i: b.pinned_array(i) for i in ids}
        else:
            xin, fin, dst = x, f0, {i: None for i in ids}
        res = []
        for it in range(4):
            t0 = time.perf_counter()
            b.upload(llsm.A_X, xin); b.upload(llsm.A_F0, fin)
            t1 = time.perf_counter()
            b.analyze(); b.synthesize(so, seed=it); ctx.sync()
            t2 = time.perf_counter()
            nbytes = 0
            for i in ids:
                nbytes += b.download(i, out=dst[i]).nbytes
            t3 = time.perf_counter()
            res.append((t1 - t0, t2 - t1, t3 - t2, nbytes))
        up, comp, down, nbytes = (np.median([r[k] for r in res[1:]]) for k in range(4))
        out[mode] = {"upload_MB": (x.nbytes + f0.nbytes) / 1e6, "upload_ms": up * 1e3, "compute_ms": comp * 1e3,
                     "download_MB": nbytes / 1e6, "download_ms": down * 1e3,
                     "frames_per_s_resident": frames / comp,
                     "frames_per_s_pcie_inclusive": frames / (up + comp + down)}
        if mode == "pinned":
            for a in [xin, fin] + list(dst.values()):
                b.free_pinned(a)
    print(json.dumps(out))
    b.close(); ctx.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
