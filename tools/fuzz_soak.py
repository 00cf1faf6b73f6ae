"""One-off soak: the seeded configuration fuzz of tests/test_gpu_configs.py over many more seeds than the suite runs.
    python tools/fuzz_soak.py [first_seed] [count]        (SOAK_ONLY=layer0,l1rt,hmpp,alt,coder picks sweeps;
                                                           SOAK_PROCS=n oracle worker processes, default = CPUs)
Layer 0: the float64 oracle runs in worker processes (forked BEFORE the device is opened; they never touch it), the
product in this one.  One line per seed outside the contract of tests/gpu_common.py (FAIL), one per seed that the
superseded round-2 ... round-4 tolerances would have flagged (MARGINAL: these go into tests/test_gpu_regressions.py),
and the worst value of every metric over the sweep with the seed it came from (WORST)."""
import json
import multiprocessing as mp
import os
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import libllsm2_amd as llsm
from conftest import make_speechlike
from oracle.oracle import Oracle
from test_gpu_configs import _fuzz_case, _run_parity, other_rate_case
from gpu_common import CONDITIONED, CONTRACT, aopt_kwargs

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
only = [t for t in os.environ.get("SOAK_ONLY", "").split(",") if t]

# the bounds the suite used until round 4 (tests/test_gpu_parity.py TOL of that time): kept HERE only, to find the
# inputs that were marginal under them
OLD_TOL = dict(ampl_abs_over_max=1e-5, ampl_rel_max=1e-3, ampl_rel_max_above_m40db=1e-4, phse_max_rad=1e-3, xres_rel_rms=1e-4,
               psd_db_p99=0.01, psd_db_max=0.2, psd_over_0p05_db_excess=1.0,
               psdres_db_p99=0.01, psdres_db_max=1.0, psdres_over_0p05_db_excess=1.0,
               edc_rel_max=1e-4, eenv_ampl_abs_over_max=1e-4, eenv_phse_max_rad=1e-3)
_o = None


def _inputs(seed):
    fs, thop, kw, nx = _fuzz_case(seed)
    x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop)
    return fs, thop, kw, nx, x, f0.astype(np.float32)


def _oracle_job(seed):
    """worker process: analysis and synthesis of one configuration by the float64 oracle (CPU only)"""
    global _o
    if _o is None:
        _o = Oracle(np.float64)
    fs, thop, kw, nx, x, f0 = _inputs(seed)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)
    okw = aopt_kwargs(ao); okw["chanfreq"] = kw["chanfreq"]
    pr, xr = _o.analyze(_o.aoptions(**okw), x, fs, f0, want_res=True)
    ys = _o.synthesize(_o.soptions(fs), pr.astype(np.float32).astype(np.float64), seed=5)
    return seed, pr, xr, ys


def want(name, n):
    return n if (not only or name in only) else 0


n0 = want("layer0", count)
pool = None
if n0:
    procs = int(os.environ.get("SOAK_PROCS", "0")) or max(1, len(os.sched_getaffinity(0)) - 1)
    pool = mp.get_context("fork").Pool(procs)                # before the HIP runtime exists in this process
    from collections import deque
    pending, nxt = deque(), first                            # a sliding window of jobs: results are ~1 MB each

    def next_job():
        global nxt
        while nxt < first + n0 and len(pending) < 4 * procs:
            pending.append(pool.apply_async(_oracle_job, (nxt,))); nxt += 1
        return pending.popleft().get()
o64 = Oracle(np.float64)
ctx = llsm.Context(0)

bad = 0
marginal = []
worst = {}
for k in range(n0):
    seed, pr, xr, ys = next_job()
    fs, thop, kw, nx, x, f0 = _inputs(seed)
    m = None
    try:
        m = _run_parity(ctx, o64, "soak", fs, thop, kw, x, f0, oracle_out=(pr, xr, ys), quiet=True)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, fs, thop, kw, nx, repr(e)[:600], flush=True)
        for tup in (e.args[0][1] if e.args and isinstance(e.args[0], tuple) and len(e.args[0]) > 1 and isinstance(e.args[0][1], list) else []):
            if tup[0] in CONDITIONED and tup[2] > CONDITIONED[tup[0]][0]:      # how far over its (largest) yardstick a FAILED seed was
                r = tup[1] / tup[2]
                if tup[0] + "/bound" not in worst or r > worst[tup[0] + "/bound"][0]:
                    worst[tup[0] + "/bound"] = (r, seed)
    except Exception as e:                                    # noqa: BLE001
        bad += 1
        print("FAIL seed", seed, fs, thop, kw, nx, "(not an assertion)", repr(e)[:300], flush=True)
    if m is None:
        continue
    old = [(t, float("%.4g" % m[t])) for t, tol in OLD_TOL.items() if not m[t] <= tol]
    if old:
        marginal.append(seed)
        f32 = {t: m.get(t + "_f32_oracle") for t in CONDITIONED if m.get(t + "_f32_oracle") is not None}
        print("MARGINAL seed", seed, fs, round(thop, 7), old, "float32 oracle:", f32, flush=True)
    for t in list(CONTRACT) + list(CONDITIONED) + ["ampl_abs_over_max", "ysin_rel_rms", "ynoise_rel_rms", "y_rel_rms"] + \
            [k_ for k_ in m if k_.startswith(("psd_db_max_", "psdraw_db_max_", "psd_pow", "psdraw_pow", "psd_over_0p05_db_", "psdres_db_max"))]:
        if t not in worst or m[t] > worst[t][0]:
            worst[t] = (m[t], seed)
    for t, (tol, kappa, yard, kulp, *_add) in CONDITIONED.items():   # how much of the float32 oracle's distance the product used
        v32 = m.get(t + "_f32_oracle")
        if v32:
            r = m[t] / v32
            if t + "/f32" not in worst or r > worst[t + "/f32"][0]:
                worst[t + "/f32"] = (r, seed)
        vu = m.get(t + "_ulp_response")
        if vu:                                                # (consulted only where the float32 oracle did not cover the value)
            r = m[t] / vu
            if t + "/ulp" not in worst or r > worst[t + "/ulp"][0]:
                worst[t + "/ulp"] = (r, seed)
if pool is not None:
    pool.close(); pool.join()
print("soak: %d configurations, %d failures; %d marginal under the superseded tolerances: %s" % (n0, bad, len(marginal), marginal))
print("WORST " + json.dumps({t: [float("%.4g" % v), s] for t, (v, s) in worst.items()}))

# layer-1 and llsmrt sweeps (the parametrised test functions called directly with further seeds)
import test_gpu_l1, test_gpu_rt
counts = {"layer-1": [0, 0], "llsmrt": [0, 0], "llsmrt PbP": [0, 0]}


def run(kind, fn, *args):
    try:
        fn(*args)
        counts[kind][0] += 1
    except BaseException as e:                                # noqa: BLE001 (pytest.skip raises a BaseException)
        if type(e).__name__ == "Skipped":
            return
        if isinstance(e, KeyboardInterrupt):
            raise
        counts[kind][0] += 1; counts[kind][1] += 1
        print("FAIL", kind, "seed", args[-1], repr(e)[:300], flush=True)


n1 = want("l1rt", max(count // 5, 1))
for seed in range(first, first + n1):
    run("layer-1", test_gpu_l1.test_random_layer1_configurations, ctx, o64, seed)
    run("llsmrt", test_gpu_rt.test_rt_random_configurations, o64, seed)
    run("llsmrt PbP", test_gpu_l1.test_random_rt_pbp_configurations, o64, seed)
print("soak: " + "; ".join("%d %s cases, %d failures" % (v[0], k, v[1]) for k, v in counts.items()))

# HMPP analysis and F0 refinement over the same random configurations (bounds of tests/test_gpu_parity.py's HMPP test;
# refined F0 against the oracle's estimator)
from gpu_common import HMPP_CONDITIONED, analysis_metrics, aopt_kwargs, assert_hmpp_contract, gpu_analyze, Yard, oracle32_metrics
worst_h = {}
nh = want("hmpp", max(count // 5, 1)); badh = badf = fliph = 0
for seed in range(first, first + nh):
    fs, thop, kw, nx = _fuzz_case(seed)
    x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop); f0 = f0.astype(np.float32)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, hm_method=llsm.HMPP, **kw)
    okw = aopt_kwargs(ao); okw["chanfreq"] = kw["chanfreq"]
    try:
        pr, xr = o64.analyze(o64.aoptions(**okw), x, fs, f0, want_res=True)
        b, g, xres = gpu_analyze(ctx, ao, fs, [x], [f0]); b.close()
        m = analysis_metrics(g, slice(0, len(f0)), pr, xres, xr)
        # peak picking is an arg-max over neighbouring bins: a near-tie resolves differently in float32 and float64 and
        # moves ONE harmonic to another local maximum (a float32 build of the oracle does the same on the same frames),
        # and the residual and its PSD follow: gpu_common.HMPP_CONTRACT bounds every such metric by the float32 oracle's
        # own distance from the float64 oracle.  Cases with moved harmonics are counted.
        z_g = (np.asarray(g[llsm.A_AMPL], np.float64) * np.exp(1j * np.asarray(g[llsm.A_PHSE], np.float64))).reshape(len(f0), -1)
        z_o = (pr.ampl * np.exp(1j * pr.phse)).reshape(len(f0), -1)
        moved = int(np.count_nonzero(np.abs(z_g - z_o) > 1e-5 * np.abs(z_o).max()))   # (float32 noise: 3.5e-7 of the maximum)
        assert_hmpp_contract(m, Yard(okw, x, fs, f0), "hmpp")
        fliph += 1 if m.get("hmpp_branch") == "B" else 0
        if m.get("hmpp_b_residual_ratio", 0) > worst_h.get("branch_B_residual_rows/ceiling", (0, 0))[0]:
            worst_h["branch_B_residual_rows/ceiling"] = (m["hmpp_b_residual_ratio"], seed)
        for t, (tol, kappa, yard, kulp, *_add) in HMPP_CONDITIONED.items():
            v32 = m.get(t + "_f32_oracle")
            if v32 and m[t] / v32 > worst_h.get(t, (0, 0))[0]:
                worst_h[t] = (m[t] / v32, seed)
    except Exception as e:                                    # noqa: BLE001
        badh += 1; print("FAIL hmpp seed", seed, fs, thop, kw, repr(e)[:1500], flush=True)
    # F0 refinement: perturb the track by +-1.5 %, both estimators must land on the same values
    try:
        f1 = (f0 * (1.0 + 0.015 * np.sign(np.sin(np.arange(len(f0)))))).astype(np.float32)
        ref = o64.refine_f0(x, fs, f1, thop)
        ao2 = llsm.make_aoptions(f0_refine=1, thop=thop, **kw)
        b = llsm.Batch(ctx, ao2, fs, [len(x)], [len(f1)])
        b.upload(llsm.A_X, x); b.upload(llsm.A_F0, f1); b.analyze(); ctx.sync()
        got = b.download(llsm.A_F0); b.close()
        v = ref > 0
        assert np.array_equal(got > 0, ref > 0) and (not v.any() or np.abs(got[v] - ref[v]).max() < 5e-2), float(np.abs(got[v] - ref[v]).max())
    except Exception as e:                                    # noqa: BLE001
        badf += 1; print("FAIL refine seed", seed, fs, thop, repr(e)[:300], flush=True)
print("soak: %d HMPP cases, %d failures, %d accepted under branch B (a handful of harmonics / envelope values on another local maximum or bin); %d F0-refinement cases, %d failures" % (nh, badh, fliph, nh, badf))
print("WORST-HMPP share of the float32 oracle's distance where the plain bound was exceeded: " + json.dumps({t: [float("%.4g" % v), s_] for t, (v, s_) in worst_h.items()}))

# the alternative conventions on both sides, over random configurations
from test_gpu_round2 import CONVENTIONS
L = llsm.load()
nc = want("alt", max(count // 5, 1)); badc = 0
try:
    for name, (dflt, alt) in CONVENTIONS.items():
        assert L.llsm_gpu_set_convention(name.encode(), alt) == 0
        o64.set_convention(name, alt)
    ctx2 = llsm.Context(0)
    for seed in range(first, first + nc):
        fs, thop, kw, nx = _fuzz_case(seed)
        x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop)
        try:
            _run_parity(ctx2, o64, "soak_alt", fs, thop, kw, x, f0.astype(np.float32))
        except Exception as e:                                # noqa: BLE001
            badc += 1; print("FAIL alt-conventions seed", seed, fs, thop, kw, repr(e)[:300], flush=True)
    ctx2.close()
finally:
    for name, (dflt, alt) in CONVENTIONS.items():
        L.llsm_gpu_set_convention(name.encode(), dflt); o64.set_convention(name, dflt)
print("soak: %d configurations under the alternative conventions, %d failures" % (nc, badc))

# the frame coder over random configurations (vocal-tract size, orders) and synthesis at a rate other than the analysis rate
import test_gpu_coder
Lc = test_gpu_coder.coder_lib()
nk = want("coder", max(count // 10, 1)); badk = bado = refused = 0
for seed in range(first, first + nk):
    fs, thop, kw, nx = _fuzz_case(seed)
    r = np.random.default_rng(4000 + seed)
    nfft = int(r.choice([512, 1024, 2048, 4096]))
    osp = int(r.integers(8, min(nfft // 4, 128))); obap = int(r.integers(1, 10))
    try:
        test_gpu_coder.coder_parity(Lc, o64, "soak", (fs, thop, nfft, osp, obap, kw, 100 + seed))
    except Exception as e:                                    # noqa: BLE001
        badk += 1; print("FAIL coder seed", seed, fs, thop, nfft, osp, obap, kw, repr(e)[:1800], flush=True)
    try:
        other_rate_case(ctx, o64, seed)
    except Exception as e:                                    # noqa: BLE001
        if "outside the supported range" in repr(e):          # e.g. a 25 ms hop synthesised at 96 kHz: refused, loudly
            refused += 1
        else:
            bado += 1; print("FAIL other-rate seed", seed, fs, thop, kw, repr(e)[:300], flush=True)
print("soak: %d coder cases, %d failures; %d other-rate synthesis cases, %d failures, %d refused (transform size)" % (nk, badk, nk, bado, refused))
