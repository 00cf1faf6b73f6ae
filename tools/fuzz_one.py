"""Metrics of single seeds of the configuration fuzz (tests/test_gpu_configs.py::_fuzz_case), for looking at a soak failure
under different switches:   [LLSM_GPU_FILT_FUSE=0] python tools/fuzz_one.py [--hmpp] [--json] seed [seed ...]
(--json: every metric of analysis_metrics as one JSON line per seed, the format of tools/oracle_f32_spread.py)"""
import json
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import libllsm2_amd as llsm
from conftest import make_speechlike
from oracle.oracle import Oracle
from test_gpu_configs import _fuzz_case
from gpu_common import analysis_metrics, aopt_kwargs, gpu_analyze

args = sys.argv[1:]
hmpp = "--hmpp" in args
as_json = "--json" in args
seeds = [int(a) for a in args if not a.startswith("--")]
o64 = Oracle(np.float64)
ctx = llsm.Context(0)
keys = ("ampl_rel_max", "ampl_abs_over_max", "phse_max_rad", "psd_db_max", "psd_over_0p05_db_excess", "edc_rel_max",
        "eenv_ampl_abs_over_max")
for seed in seeds:
    fs, thop, kw, nx = _fuzz_case(seed)
    x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop); f0 = f0.astype(np.float32)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, **(dict(kw, hm_method=llsm.HMPP) if hmpp else kw))
    okw = aopt_kwargs(ao); okw["chanfreq"] = kw["chanfreq"]
    pr, xr = o64.analyze(o64.aoptions(**okw), x, fs, f0, want_res=True)
    b, g, xres = gpu_analyze(ctx, ao, fs, [x], [f0]); b.close()
    m = analysis_metrics(g, slice(0, len(f0)), pr, xres, xr)
    if as_json: print(json.dumps(dict(seed=seed, fs=fs, thop=thop, who="product", **m)), flush=True)
    else: print(seed, fs, round(thop, 6), {k: float("%.4g" % m[k]) for k in keys}, flush=True)
