"""How far the reference's own FP_TYPE = float arithmetic sits from exact arithmetic on a fuzz seed: the float32 build of
the oracle stands in for the product and is measured against the float64 build with the metrics of the GPU parity tests
(tests/gpu_common.py analysis_metrics), CPU only.  Beside tools/fuzz_one.py --json (the product, same seeds) this says
whether a value over a bound of SURVEY 8(d) is the product's arithmetic or the conditioning of the algorithm in float32.
    python tools/oracle_f32_spread.py seed [seed ...]          one JSON line per seed"""
import json
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import libllsm2_amd as llsm
from conftest import make_speechlike
from oracle.oracle import Oracle
from test_gpu_configs import _fuzz_case
from gpu_common import analysis_metrics, aopt_kwargs

for seed in [int(a) for a in sys.argv[1:]]:
    fs, thop, kw, nx = _fuzz_case(seed)
    x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop); f0 = f0.astype(np.float32)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)
    okw = aopt_kwargs(ao); okw["chanfreq"] = kw["chanfreq"]
    p, r = {}, {}
    for dt in (np.float64, np.float32):
        o = Oracle(dt)
        p[dt], r[dt] = o.analyze(o.aoptions(**okw), x, fs, f0, want_res=True)
    q = p[np.float32]
    g = {llsm.A_NHAR: q.nhar, llsm.A_NHAR_E: q.nhar_e, llsm.A_AMPL: q.ampl, llsm.A_PHSE: q.phse, llsm.A_PSD: q.psd,
         llsm.A_PSDRES: q.psdres, llsm.A_EDC: q.edc, llsm.A_EENV_AMPL: q.eenv_ampl, llsm.A_EENV_PHSE: q.eenv_phse}
    m = analysis_metrics(g, slice(0, len(f0)), p[np.float64], np.asarray(r[np.float32], np.float64), r[np.float64])
    print(json.dumps(dict(seed=seed, fs=fs, thop=thop, who="oracle_f32", **m)), flush=True)
