"""How far the reference's own FP_TYPE = float arithmetic sits from exact arithmetic on a fuzz seed: the float32 build of
the oracle against the float64 build (CPU only), in the metrics of the GPU parity tests.  Beside tools/fuzz_one.py's numbers of
the product for the same seed this says whether a PSD tail is the product's or the algorithm's conditioning.
    python tools/oracle_f32_spread.py seed [seed ...]"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import libllsm2_amd as llsm
from conftest import make_speechlike
from oracle.oracle import Oracle
from test_gpu_configs import _fuzz_case
from gpu_common import aopt_kwargs

for seed in [int(a) for a in sys.argv[1:]]:
    fs, thop, kw, nx = _fuzz_case(seed)
    x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop); f0 = f0.astype(np.float32)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)
    okw = aopt_kwargs(ao); okw["chanfreq"] = kw["chanfreq"]
    p = {}
    for dt in (np.float64, np.float32):
        o = Oracle(dt)
        p[dt] = o.analyze(o.aoptions(**okw), x, fs, f0, want_res=True)[0]
    d = np.abs(np.asarray(p[np.float64].psd, np.float64) - np.asarray(p[np.float32].psd, np.float64))
    d = d[np.isfinite(d)]
    print(seed, fs, round(thop, 6), {"psd_db_max": float("%.4g" % d.max()), "psd_over_0p05_db": int((d > 0.05).sum()),
                                     "psd_db_p99": float("%.4g" % np.quantile(d, 0.99)), "values": int(d.size)}, flush=True)
