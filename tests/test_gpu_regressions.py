"""-m gpu regressions: EVERY input that was ever marginal in a soak (tools/fuzz_soak.py, rounds 2 ... 5) under the
tolerances of its time, asserted under the contract of gpu_common.py -- so the suite is not a selection of survivors.

For each such input three things run: the product (HIP, float32), the float64 oracle and the FLOAT32 build of the oracle
(the reference's own FP_TYPE = float arithmetic, makefile:20).  Asserted:
  * gpu_common.CONTRACT as it stands (every harmonic as a complex number within 1e-5 of the largest amplitude; SURVEY
    8(d)'s relative 1e-4 / 1e-3 rad above -40 dB; residual, envelope harmonics),
  * for the smoothed PSD and the band energies  err(HIP, f64) <= max(8(d) value, 1 x err(f32 oracle, f64), 4 x the float64
    oracle's own response to a one-ulp perturbation of the float32 input)  (gpu_common.CONDITIONED: never further from
    exact arithmetic than the reference's own float build on that input, or than a four-ulp change of the input),
and the table product / float32 oracle / share is written to gpurun_out/parity_regression_*.json.

The seed lists are what tools/fuzz_soak.py prints as MARGINAL (superseded tolerances exceeded) or FAIL."""
import numpy as np
import pytest

import libllsm2_amd as llsm
from conftest import make_speechlike
from gpu_common import (CONDITIONED, CONTRACT, HMPP_CONDITIONED, analysis_metrics, aopt_kwargs, assert_contract,
                        assert_hmpp_contract, gpu_analyze, oracle32_metrics, oracle_ulp_response, report)
from test_gpu_configs import _fuzz_case, _run_parity

pytestmark = pytest.mark.gpu

# layer-0 configuration-fuzz seeds (tests/test_gpu_configs.py::_fuzz_case), by the round that found them and what was over:
LAYER0_SEEDS = [
    5242,   # r4: 5 Kalman-smoothed PSD values over 0.05 dB (48 kHz, 20 harmonics removed: the residual keeps strong partials)
    7244,   # r4: 2 PSD values over 0.05 dB, worst 0.069 dB (float32 oracle: 9.8 dB)
    7286,   # r4: weak-harmonic phase 1.78e-3 rad
    8224,   # r4: weak-harmonic phase 1.36e-3 rad
    8619,   # r4: 6 PSD values over 0.05 dB at 8 kHz
    9146,   # r4: weak-harmonic amplitude ratio 1.57e-3 (1.6 ms hop at 48 kHz)
    9198,   # r4: weak-harmonic phase 1.03e-3 rad
]
# round 5, seeds 1000 ... 3999 (the ranges of the round-2 / round-3 soaks, whose marginal seeds were recorded by value only:
# re-found by number under the tolerances of that time) and 10000 ... 12999 (fresh); profiles/r05_a_*, r05_b_*:
LAYER0_SEEDS += [
    1444, 1988, 2046,          # weak-harmonic phase 1.05 / 1.64 / 1.09e-3 rad
    1621, 2248, 2563, 3053, 3553,   # weak-harmonic amplitude ratio 1.01 ... 1.23e-3
    2291,                      # 2 PSD values over 0.05 dB
    2790,                      # PSD 0.36 dB at a bin 10 dB below the frame's maximum (its raw periodogram: 1e-3 dB), 8 values over 0.05 dB
    1833, 2769,                # PSD 0.31 / 0.095 dB: further than HALF the float32 oracle's distance (the first restatement's KAPPA)
    2240, 2675, 10586,         # PSDRES 0.051 / 0.051 / 0.053 dB at values 60 dB below the frame's maximum
    10466, 10709, 11724, 12517, 12817,   # 2 - 3 PSD values over 0.05 dB (worst 0.24 dB)
    10681, 11294, 12879, 12898,          # weak-harmonic amplitude ratio 1.09 ... 1.2e-3
    11339, 11460, 11597, 12053, 12995,   # weak-harmonic phase 1.0 ... 1.6e-3 rad
]
# round 5, seeds 20000 ... 39999 (profiles/r05_c_soak_layer0.txt): the six over the second restatement (joint float32
# yardstick at KAPPA = 1, raw periodogram asserted down to -40 dB) -- 25591: PSD 0.405 dB at the DC point against 0.155 dB for
# the float32 oracle (2.6 x; the float64 oracle itself moves by 0.13 dB under a one-ulp change of the input); 25349, 30930:
# PSDRES 0.085 / 0.151 dB at Rayleigh nulls; 20586, 26207, 29526: raw periodogram 0.058 ... 0.102 dB between -40 and -20 dB --
LAYER0_SEEDS += [20586, 25349, 25591, 26207, 29526, 30930]
# ... and the 65 the superseded tolerances would have flagged (weak-harmonic ratios 1.0 ... 1.3e-3, PSD values over 0.05 dB,
# worst 2.1 dB at the DC point of seed 25083)
LAYER0_SEEDS += [
    20052, 20307, 20687, 20767, 20796, 21650, 21894, 22048, 22077, 22116, 22186, 22425, 22629, 22688, 23145, 23976, 23997, 24414, 24605, 25083,
    25472, 25615, 25933, 26030, 26149, 26363, 27434, 27927, 28136, 28337, 28371, 29153, 29260, 29427, 29819, 30252, 30457, 30483, 30646, 31796,
    32135, 32308, 32578, 32838, 33223, 33316, 33682, 33782, 33947, 34161, 34194, 34261, 34469, 34567, 35177, 35331, 35485, 35700, 35982, 36062,
    37245, 37464, 38463, 39118, 39659,
]
HMPP_SEEDS = [7037,            # r4: band energy 2.2e-4 (band 5.4 - 8 kHz at 16 kHz)
              # r5 (profiles/r05_e_soak_others.txt): harmonics on another local maximum (40044, 40115, 40157, 40183, 40290 ...),
              # every-harmonic values of 1.1 ... 1.9e-5 (40015, 40047, 40099, 40198), envelope phases of 1.1 ... 1.4e-3 rad
              40015, 40044, 40047, 40052, 40078, 40079, 40099, 40104, 40115, 40157, 40171, 40182, 40183, 40198, 40240, 40246, 40250,
              40259, 40290, 40367, 40375, 40419, 40587,
              # r5, soak at the HEAD of the round (profiles/r05_zz2_soak_all.txt): ONE harmonic 35 dB down at 4.8e-4 relative
              # (8.6e-6 of the maximum: inside the complex bound, outside the relative one; no harmonic "moved" as branch (B)
              # counted them then)
              80189,
              # r5, 3 000 more (profiles/r05_zz3_soak_hmpp.txt): 8 of 72 envelope values (the float32 oracle: 18) behind ONE moved harmonic
              92774]
L1_SEEDS = [80586]              # r5, same soak: Rd of one frame 2.86e-4 off -- in the float32 oracle exactly as in the product
# r5, the last soak of the round (30 000 fresh configurations, seeds 100000 ..., profiles/r05_zz4_soak_layer0.txt): the ONE that was
# outside the conditioned PSD bound (1.95 dB at PSD point 0, six frames after a voicing onset).  Round 6 found the cause -- a DC bin
# of the spectrogram 147 dB under the frame's harmonics -- and removed it (test below); the seed stays as a regression input.
KNOWN_OUTSIDE = [123208]
ALT_CONVENTION_SEEDS = [5078]   # r4: band energy 1.23e-4 (band edge 256 Hz at 8 kHz) under the alternative conventions


@pytest.fixture(scope="module")
def ctx():
    c = llsm.Context(0)
    yield c
    c.close()


def _case(seed):
    fs, thop, kw, nx = _fuzz_case(seed)
    x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop)
    return fs, thop, kw, x, f0.astype(np.float32)


def _conditioning_table(m, m32, mu, conditioned, check=True):
    """product / float32 oracle / exact algorithm's one-ulp response for every conditioned metric, and the assertion with
    BOTH yardsticks always evaluated (the lazy form of assert_contract only looks when the plain value is exceeded)."""
    tab = {}
    for k, (tol, kappa, yard, kappa_ulp, *add) in conditioned.items():
        y32 = max(m32[t] for t in yard)
        bound = max(tol, kappa * y32 + (add[0] if add else 0.0) * tol, kappa_ulp * mu[k])
        tab[k] = dict(product=m[k], oracle_f32=y32, ulp_response_f64=mu[k], contract=tol, kappa_f32=kappa, kappa_ulp=kappa_ulp,
                      f32_plus_contract=bool(add and add[0]), bound=bound)
        assert not check or m[k] <= bound, (k, tab[k])
    return tab


@pytest.mark.parametrize("seed", LAYER0_SEEDS)
def test_marginal_layer0_seeds(ctx, o64, seed):
    fs, thop, kw, x, f0 = _case(seed)
    m = _run_parity(ctx, o64, "regression_%d" % seed, fs, thop, kw, x, f0)            # asserts the contract (+ synthesis)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)
    okw = aopt_kwargs(ao); okw["chanfreq"] = kw["chanfreq"]
    m32 = oracle32_metrics(okw, x, fs, f0)
    tab = _conditioning_table(m, m32, oracle_ulp_response(okw, x, fs, f0), CONDITIONED)
    report("regression_%d_conditioning" % seed, dict(fs=fs, thop=thop, options=kw, conditioned=tab,
                                                     product={k: m[k] for k in CONTRACT}, oracle_f32={k: m32[k] for k in CONTRACT}))


@pytest.mark.parametrize("seed", HMPP_SEEDS)
def test_marginal_hmpp_seeds(ctx, o64, seed):
    fs, thop, kw, x, f0 = _case(seed)
    ao = llsm.make_aoptions(f0_refine=0, thop=thop, hm_method=llsm.HMPP, **kw)
    okw = aopt_kwargs(ao); okw["chanfreq"] = kw["chanfreq"]
    pr, xr = o64.analyze(o64.aoptions(**okw), x, fs, f0, want_res=True)
    b, g, xres = gpu_analyze(ctx, ao, fs, [x], [f0]); b.close()
    m = analysis_metrics(g, slice(0, len(f0)), pr, xres, xr)
    m32 = oracle32_metrics(okw, x, fs, f0)
    mu = oracle_ulp_response(okw, x, fs, f0)
    assert_hmpp_contract(m, lambda: m32, "hmpp_%d" % seed, ulp_response=lambda: mu)
    report("regression_hmpp_%d" % seed, dict(fs=fs, thop=thop, options=kw, branch=m["hmpp_branch"], moved=m["harm_over_count"],
                                             conditioned=_conditioning_table(m, m32, mu, HMPP_CONDITIONED, check=m["hmpp_branch"] == "A")))


@pytest.mark.parametrize("seed", KNOWN_OUTSIDE)
def test_the_configuration_that_was_outside_until_round_5(ctx, o64, seed):
    """Seed 123208 (32 kHz, 4 ms hop): rounds 3 - 5 had its smoothed PSD 1.95 dB off at the DC point, six frames after a voicing
    onset -- the one configuration in 60 000 outside the conditioned bound.  Cause (tools/psd_bisect.py --product, round 6):
    frame 43's spectrogram has its DC bin 147 dB under the frame's harmonics (a real sum changing sign), the float32
    transform and window returned their own rounding there, the cepstral envelope came out 1.3 nepers low, the Kalman process
    variance with it.  k_spgm_env_wf now lists such frames and a second launch recomputes the two real bins exactly; the
    configuration passes the contract as written, WITHOUT either yardstick, and its value at that point is 1e-3 dB."""
    fs, thop, kw, x, f0 = _case(seed)
    m = _run_parity(ctx, o64, "regression_formerly_outside_%d" % seed, fs, thop, kw, x, f0, quiet=True)
    assert m["psd_db_max"] <= 0.05, m["psd_db_max"]
    assert "psd_db_max_f32_oracle" not in m          # (neither yardstick was consulted)


@pytest.mark.parametrize("seed", L1_SEEDS)
def test_marginal_layer1_seeds(ctx, o64, seed):
    import test_gpu_l1
    test_gpu_l1.test_random_layer1_configurations(ctx, o64, seed)


@pytest.mark.parametrize("seed", ALT_CONVENTION_SEEDS)
def test_marginal_seeds_under_the_alternative_conventions(ctx, o64, seed):
    from test_gpu_round2 import CONVENTIONS
    L = llsm.load()
    fs, thop, kw, x, f0 = _case(seed)
    try:
        for name, (dflt, alt) in CONVENTIONS.items():
            assert L.llsm_gpu_set_convention(name.encode(), alt) == 0
            o64.set_convention(name, alt)
        c2 = llsm.Context(0)
        try:
            m = _run_parity(c2, o64, "regression_alt_%d" % seed, fs, thop, kw, x, f0)
            ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)
            okw = aopt_kwargs(ao); okw["chanfreq"] = kw["chanfreq"]
            report("regression_alt_%d_conditioning" % seed,
                   _conditioning_table(m, oracle32_metrics(okw, x, fs, f0), oracle_ulp_response(okw, x, fs, f0), CONDITIONED))
        finally:
            c2.close()
    finally:
        for name, (dflt, alt) in CONVENTIONS.items():
            L.llsm_gpu_set_convention(name.encode(), dflt); o64.set_convention(name, dflt)


def test_edge_probe_96k_2p5ms_300_harmonics(ctx, o64):
    """tools/edge_probe.py's one value outside the round-4 bounds: a harmonic 70 dB down at 1.05e-3 relative (96 kHz,
    2.5 ms hop, 300 harmonics, 512 PSD points)."""
    fs, thop, kw = 96000.0, 0.0025, dict(maxnhar=300, npsd=512)
    x, f0 = make_speechlike(9, nx=int(0.4 * fs), fs=fs, thop=thop)
    _run_parity(ctx, o64, "regression_edge_96k_2p5ms", fs, thop, kw, x, f0.astype(np.float32))
