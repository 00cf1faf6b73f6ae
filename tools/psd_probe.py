"""Where the largest smoothed-PSD error of a fuzz seed sits: frame, PSD point, the F0 track around it, the error profile of
that PSD point along time and of that frame along frequency.   python tools/psd_probe.py seed [seed ...]"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import libllsm2_amd as llsm
from conftest import make_speechlike
from oracle.oracle import Oracle
from test_gpu_configs import _fuzz_case
from gpu_common import aopt_kwargs, gpu_analyze

o64 = Oracle(np.float64)
ctx = llsm.Context(0)
for seed in [int(a) for a in sys.argv[1:]]:
    fs, thop, kw, nx = _fuzz_case(seed)
    x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop); f0 = f0.astype(np.float32)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)
    okw = aopt_kwargs(ao); okw["chanfreq"] = kw["chanfreq"]
    pr, xr = o64.analyze(o64.aoptions(**okw), x, fs, f0, want_res=True)
    b, g, xres = gpu_analyze(ctx, ao, fs, [x], [f0]); b.close()
    d = g[llsm.A_PSD].astype(np.float64).reshape(pr.psd.shape) - pr.psd
    i, j = np.unravel_index(np.argmax(np.abs(d)), d.shape)
    n = len(f0)
    print("seed", seed, "fs", fs, "thop", round(thop, 6), "nfrm", n, "npsd", pr.psd.shape[1], "worst", round(float(d[i, j]), 4), "at frame", int(i), "point", int(j))
    print("  f0 around:", [round(float(v), 1) for v in f0[max(0, i - 6):i + 7]])
    print("  error of that point along time:", [round(float(v), 3) for v in d[max(0, i - 8):i + 9, j]])
    print("  error of that frame along frequency:", [round(float(v), 3) for v in d[i, max(0, j - 6):j + 7]])
    big = np.argwhere(np.abs(d) > 0.05)
    print("  values over 0.05 dB:", len(big), "frames", sorted(set(int(a) for a in big[:, 0]))[:20], "points", sorted(set(int(a) for a in big[:, 1]))[:20])
    print("  oracle psd there:", round(float(pr.psd[i, j]), 2), "frame max", round(float(pr.psd[i].max()), 2), "psdres there", round(float(pr.psdres[i, j]), 2))
