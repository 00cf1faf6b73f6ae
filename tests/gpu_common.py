"""Shared helpers of the -m gpu parity tests: run a batch through the C-ABI,
run the float64 oracle on the same inputs, compute error metrics."""
import json
import os

import numpy as np

import libllsm2_amd as llsm
from conftest import wrap

REPORT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def report(name, obj):
    os.makedirs(REPORT_DIR, exist_ok=True)
    with open(os.path.join(REPORT_DIR, f"parity_{name}.json"), "w") as f:
        json.dump(obj, f, indent=1, default=float)
    print(f"[parity:{name}] " + json.dumps(obj, default=float))


def aopt_kwargs(ao):
    kw = dict(thop=ao.thop, maxnhar=ao.maxnhar, maxnhar_e=ao.maxnhar_e, npsd=ao.npsd,
              nchannel=ao.nchannel, f0_refine=ao.f0_refine, hm_method=ao.hm_method,
              rel_winsize=ao.rel_winsize)
    if ao.nchannel > 1 and bool(ao.chanfreq):       # the band plan travels with the options (it used to be dropped here)
        kw["chanfreq"] = [float(ao.chanfreq[i]) for i in range(ao.nchannel - 1)]
    return kw


def gpu_analyze(ctx, ao, fs, xs, f0s):
    """xs, f0s: lists of float32 arrays. Returns (batch, params dict, xres)."""
    b = llsm.Batch(ctx, ao, fs, [len(x) for x in xs], [len(f) for f in f0s])
    b.upload(llsm.A_X, np.concatenate(xs) if xs else np.zeros(0, np.float32))
    b.upload(llsm.A_F0, np.concatenate(f0s) if f0s else np.zeros(0, np.float32))
    b.analyze()
    ctx.sync()
    return b, b.download_params(), b.download(llsm.A_XRES)


def oracle_analyze(o, ao, fs, x, f0):
    oo = o.aoptions(**aopt_kwargs(ao))
    return o.analyze(oo, x, fs, f0, want_res=True)


def _eenv_rows(pr, a):
    """[nfrm][nchannel][max(maxnhar_e, 1)] float32 (one unused zero column when maxnhar_e = 0)"""
    if pr.maxnhar_e == 0:
        return np.zeros((pr.nfrm, pr.nchannel, 1), np.float32)
    return a.astype(np.float32).reshape(pr.nfrm, pr.nchannel, pr.maxnhar_e)


def params_to_gpu_rows(pr):
    """oracle Params -> dict of flat float32/int32 rows in batch layout."""
    return {llsm.A_F0: pr.f0.astype(np.float32), llsm.A_NHAR: pr.nhar.astype(np.int32),
            llsm.A_AMPL: pr.ampl.astype(np.float32), llsm.A_PHSE: pr.phse.astype(np.float32),
            llsm.A_PSD: pr.psd.astype(np.float32), llsm.A_PSDRES: pr.psdres.astype(np.float32),
            llsm.A_HAS_PSDRES: np.ones(pr.nfrm, np.int32), llsm.A_EDC: pr.edc.astype(np.float32),
            llsm.A_NHAR_E: pr.nhar_e.astype(np.int32),
            llsm.A_EENV_AMPL: _eenv_rows(pr, pr.eenv_ampl), llsm.A_EENV_PHSE: _eenv_rows(pr, pr.eenv_phse)}


def rel_rms(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    den = np.sqrt(np.mean(b ** 2))
    return float(np.sqrt(np.mean((a - b) ** 2)) / den) if den > 0 else float(np.sqrt(np.mean((a - b) ** 2)))


def analysis_metrics(g, sl, pr, xres_g, xres_o):
    """g: GPU param dict; sl: frame slice of this utterance; pr: oracle Params (f64)."""
    m = {}
    nh_g, nh_o = g[llsm.A_NHAR][sl], pr.nhar
    m["nhar_mismatch"] = int(np.sum(nh_g != nh_o))
    m["nhar_e_mismatch"] = int(np.sum(g[llsm.A_NHAR_E][sl] != pr.nhar_e))
    a_g, a_o = g[llsm.A_AMPL][sl].astype(np.float64), pr.ampl
    p_g, p_o = g[llsm.A_PHSE][sl].astype(np.float64), pr.phse
    amax = max(a_o.max(), 1e-30)
    big = a_o > 1e-4 * amax
    m["ampl_rel_max"] = float(np.max(np.abs(a_g - a_o)[big] / a_o[big])) if big.any() else 0.0
    m["ampl_abs_over_max"] = float(np.max(np.abs(a_g - a_o)) / amax)
    # the same relative error by level: harmonics above -40 dB re the largest one (SURVEY 8d's 1e-4 is asserted
    # there) and the band between -80 and -40 dB, where the ABSOLUTE float32 error (a few 1e-7 of the maximum:
    # twiddle recurrences, not accumulation) is no longer small against the harmonic itself
    hi = a_o > 1e-2 * amax
    lo = big & ~hi
    m["ampl_rel_max_above_m40db"] = float(np.max(np.abs(a_g - a_o)[hi] / a_o[hi])) if hi.any() else 0.0
    m["ampl_rel_max_m80_to_m40db"] = float(np.max(np.abs(a_g - a_o)[lo] / a_o[lo])) if lo.any() else 0.0
    dph = np.abs(wrap(p_g - p_o))
    # the harmonic as ONE complex number: |a_g e^{j phi_g} - a_o e^{j phi_o}| over the largest amplitude, EVERY harmonic
    # (the form of the bound that is independent of a harmonic's own level: float32 leaves an absolute error)
    zerr = np.abs(a_g * np.exp(1j * p_g) - a_o * np.exp(1j * p_o)) / amax
    m["harm_cplx_abs_over_max"] = float(np.max(zerr))
    m["harm_cplx_over_1e5_count"] = int(np.count_nonzero(zerr > 1e-5))      # (peak picking: harmonics on another local maximum)
    m["harm_count"] = int(np.count_nonzero(a_o > 0))
    # ... and outside ANY harmonic bound of the contract: the complex one, or above -40 dB the relative amplitude / phase ones
    with np.errstate(divide="ignore", invalid="ignore"):
        relbad = hi & ((np.abs(a_g - a_o) > 1e-4 * a_o) | (dph > 1e-3))
    m["harm_over_count"] = int(np.count_nonzero((zerr > 1e-5) | relbad))
    m["phse_max_rad"] = float(np.max(dph[big])) if big.any() else 0.0
    # by level and as a distribution (the peak-picking method interpolates WRAPPED bin phases, dsputils.c:140-141: its
    # error is bimodal -- SURVEY 8d's 1e-3 rad where the two bins sit on one branch, ~1e-2 where a float32 difference
    # in the refined peak position meets a phase slope of pi per bin)
    m["phse_max_rad_above_m40db"] = float(np.max(dph[hi])) if hi.any() else 0.0
    m["phse_p50_rad"] = float(np.percentile(dph[big], 50)) if big.any() else 0.0
    m["phse_p99_rad"] = float(np.percentile(dph[big], 99)) if big.any() else 0.0
    m["phse_frac_within_1e3_rad"] = float(np.mean(dph[big] <= 1e-3)) if big.any() else 1.0
    m["xres_rel_rms"] = rel_rms(xres_g, xres_o)
    m["xres_abs_max"] = float(np.max(np.abs(xres_g - xres_o))) if len(xres_o) else 0.0
    d = np.abs(g[llsm.A_PSD][sl].astype(np.float64) - pr.psd)
    m["psd_db_max"] = float(d.max()); m["psd_db_p99"] = float(np.percentile(d, 99)); m["psd_db_mean"] = float(d.mean())
    m["psd_over_0p05_db_excess"] = float(np.count_nonzero(d > 0.05) / max(1.0, 1e-5 * d.size))
    d = np.abs(g[llsm.A_PSDRES][sl].astype(np.float64) - pr.psdres)
    m["psdres_db_max"] = float(d.max()); m["psdres_db_p99"] = float(np.percentile(d, 99)); m["psdres_db_mean"] = float(d.mean())
    # values over the 0.05 dB of the contract, relative to the allowance max(2, 1e-4 of the values)
    m["psdres_over_0p05_db_excess"] = float(np.count_nonzero(d > 0.05) / max(2.0, 1e-4 * d.size))
    # the same errors BY LEVEL: a PSD value is the logarithm of (smoothed) periodogram bins, and float32 leaves an error
    # that is absolute against the frame's strong bins -- so the dB error of a value grows as the value sinks below the
    # frame's maximum (Rayleigh nulls of a noise periodogram sit 60 .. 100 dB down).  Level of a PSD value: below the
    # largest smoothed PSD value of its frame; of a PSDRES value: the raw log-periodogram it stands for (psd + psdres).
    pg, po = g[llsm.A_PSD][sl].astype(np.float64).reshape(pr.psd.shape), pr.psd
    rg, ro = pg + g[llsm.A_PSDRES][sl].astype(np.float64).reshape(pr.psd.shape), po + pr.psdres
    lmax = po.max(axis=-1, keepdims=True) if po.ndim > 1 else po.max()
    for name, vg, vo in (("psd", pg, po), ("psdraw", rg, ro)):
        lev = vo - lmax
        err = np.abs(vg - vo)
        for lo_, hi_, tag in ((-20.0, np.inf, "above_m20db"), (-40.0, -20.0, "m40_to_m20db"), (-60.0, -40.0, "m60_to_m40db"),
                              (-np.inf, -60.0, "below_m60db")):
            sel = (lev >= lo_) & (lev < hi_)
            m[f"{name}_db_max_{tag}"] = float(err[sel].max()) if sel.any() else 0.0
        m[f"{name}_db_max_above_m40db"] = max(m[f"{name}_db_max_above_m20db"], m[f"{name}_db_max_m40_to_m20db"])
        # float32 leaves an ABSOLUTE error on a spectrum value (a few 1e-7 of the frame's strong bins), so the error of its
        # logarithm grows as the amplitude sinks: err_dB x amplitude ratio (10^(level / 20), level <= 0) is the level-free form
        m[f"{name}_db_scaled_max"] = float(np.max(err * 10.0 ** (np.minimum(lev, 0.0) / 20.0)))
        # linear (power) form: |10^(g/10) - 10^(o/10)| over the frame's largest smoothed PSD value
        m[f"{name}_pow_abs_over_max"] = float(np.max(np.abs(10.0 ** ((vg - lmax) / 10.0) - 10.0 ** ((vo - lmax) / 10.0))))
    # WHERE the smoothed-PSD values over 0.05 dB sit (the conditioning argument of CONDITIONED says: only next to DC and to
    # Nyquist) and how many they are; PSDRES where the raw periodogram it completes is above -20 dB re the frame's maximum
    # (what layer0.c:606 adds back where the signal is)
    npsd_ = po.shape[-1]
    errp = np.abs(pg - po).reshape(-1, npsd_)
    over = errp > 0.05
    inner = np.zeros(npsd_, bool); inner[4:max(4, npsd_ - 2)] = True          # PSD points 4 .. npsd - 3
    m["psd_values"] = int(errp.size)
    m["psd_over_0p05_db_count"] = int(np.count_nonzero(over))
    m["psd_over_0p05_db_interior_count"] = int(np.count_nonzero(over[:, inner]))
    m["psd_over_0p05_db_frac"] = float(np.count_nonzero(over) / max(1, errp.size))
    m["psd_db_max_interior"] = float(errp[:, inner].max()) if inner.any() else 0.0
    errr = np.abs((rg - pg) - (ro - po)).reshape(-1, npsd_)
    strong = ((ro - lmax) >= -20.0).reshape(-1, npsd_)
    m["psdres_db_max_above_m20db"] = float(errr[strong].max()) if strong.any() else 0.0
    e_g, e_o = g[llsm.A_EDC][sl].astype(np.float64), pr.edc
    m["edc_rel_max"] = float(np.max(np.abs(e_g - e_o) / np.maximum(np.abs(e_o), 1e-30)))
    if pr.eenv_ampl.size == 0:                       # maxnhar_e = 0: the rows are one (unused) column wide
        m["eenv_ampl_abs_over_max"] = 0.0; m["eenv_phse_max_rad"] = 0.0; m["eenv_over_count"] = 0; m["eenv_count"] = 0
        return m
    ea_g = g[llsm.A_EENV_AMPL][sl].astype(np.float64).reshape(pr.eenv_ampl.shape)
    ep_g = g[llsm.A_EENV_PHSE][sl].astype(np.float64).reshape(pr.eenv_phse.shape)
    emax = max(pr.eenv_ampl.max(), 1e-30)
    m["eenv_ampl_abs_over_max"] = float(np.max(np.abs(ea_g - pr.eenv_ampl)) / emax)
    bige = pr.eenv_ampl > 1e-2 * emax
    dpe = np.abs(wrap(ep_g - pr.eenv_phse))
    m["eenv_phse_max_rad"] = float(np.max(dpe[bige])) if bige.any() else 0.0
    m["eenv_over_count"] = int(np.count_nonzero((np.abs(ea_g - pr.eenv_ampl) > 1e-4 * emax) | (bige & (dpe > 1e-3))))
    m["eenv_count"] = int(np.count_nonzero(pr.eenv_ampl > 0))
    return m


# ---- the parity contract: SURVEY 8(d), in the form that holds WITHOUT exceptions ----
# Harmonics: every harmonic, as one complex number, within 1e-5 of the largest amplitude of the utterance (float32
# leaves an ABSOLUTE error of a few 1e-7 of the maximum -- twiddle recurrences and accumulation --, so a relative bound
# on a harmonic 60 ... 80 dB down is a bound on that floor, not on the arithmetic), PLUS SURVEY 8(d)'s relative 1e-4 /
# 1e-3 rad for every harmonic above -40 dB re the largest.  Residual waveform, band energies, envelope harmonics: 8(d).
CONTRACT = dict(harm_cplx_abs_over_max=1e-5, ampl_rel_max_above_m40db=1e-4, phse_max_rad_above_m40db=1e-3,
                xres_rel_rms=1e-4, eenv_ampl_abs_over_max=1e-4, eenv_phse_max_rad=1e-3,
                # the RAW log-periodogram the analysis stores (psd + PSDRES, layer0.c:398-403) within SURVEY 8(d)'s 0.05 dB
                # for every value above -20 dB re the frame's largest PSD value: where the signal is, the transforms,
                # the residual and the PSD frames of the product are exact to float32 rounding (measured 0.016 dB over
                # 26 000 configurations; -40 ... -20 dB: 0.10 dB, below: the Rayleigh nulls).  psd + PSDRES is what the
                # synthesis filters towards (layer0.c:606): the analysis -> synthesis chain never sees the split.
                psdraw_db_max_above_m20db=0.05,
                # ... and at EVERY level once the error is weighed by the value's amplitude re the frame's largest PSD value
                # (err_dB x 10^(level / 20)): float32 leaves an absolute error on a spectrum value, so its logarithm's error
                # grows as the value sinks into the Rayleigh nulls (measured 0.35 dB at -50 dB, 0.83 dB below -60 dB)
                psdraw_db_scaled_max=0.05)
# Metrics whose float32 error is the ALGORITHM's conditioning, not an implementation's:
#  * PSD (the Kalman / RTS-smoothed level): the smoother's process variance Q_i is the variance of THREE neighbouring
#    values of a cepstrally smoothed log spectrogram (layer0.c:365-385).  Where a spectrogram bin lies 80 dB below the
#    frame's harmonics (the bins next to DC and to Nyquist of voiced speech) the float32 transform error is a visible
#    fraction of the bin, the envelope moves in its third digit, Q -- a variance of nearly equal numbers -- moves by tens
#    of per cent, the smoother's gain with it, and the smoothed value slides along the +-5.6 dB scatter of the
#    log-periodogram: every value over 0.05 dB in 29 000 random configurations sits at PSD points 0, 1 or npsd - 1
#    (tools/psd_probe.py), bit-identical with a correctly rounded log and with the recursions in float64 (profiles/r05_d)
#    -- it is the transform's rounding, which no float32 implementation avoids -- while the RAW periodogram of the same
#    point (psd + PSDRES) is exact to 1e-3 dB.  The float64 oracle ITSELF moves by 0.06 ... 0.47 dB on those inputs when
#    the float32 input samples are perturbed by one unit in the last place.
#  * band energies come through a float Chebyshev recursion whose poles sit near the unit circle for low band edges.
# For these the bound is
#     err(HIP, float64 oracle) <= max(contract,
#                                     KAPPA_F32 * err(float32 oracle, float64 oracle),        KAPPA_F32 = 1
#                                     KAPPA_ULP * max_k |float64 oracle(x (1 +- 2^-24)_k) - float64 oracle(x)|)   KAPPA_ULP = 4, k < 4
# i.e. never further from exact arithmetic than the reference's own FP_TYPE = float arithmetic (makefile:20) on the
# SAME input (its PSD and PSDRES errors are two views of one smoother: the larger one counts), or -- the float32 oracle
# being ONE draw of a chaotic error: the product exceeded it on 1 input in 7 000, by 2.6 x -- than four times the EXACT
# algorithm's own response to a one-ulp perturbation of the float32 input (a backward-error bound: the product's
# result is the exact result of an input four ulps away).  Both yardsticks are only consulted when the plain contract
# value is exceeded.  PSDRES = raw - PSD carries no bound of its own: |d PSDRES| <= |d raw| + |d PSD|.
#   metric: (contract, KAPPA_F32, float32-oracle metrics (the largest counts), KAPPA_ULP)
CONDITIONED = dict(psd_db_max=(0.05, 1.0, ("psd_db_max", "psdres_db_max"), 4.0),
                   edc_rel_max=(1e-4, 1.0, ("edc_rel_max",), 4.0))


# The CEILING: absolute statements that hold WHATEVER the two yardsticks say -- a noisy
# float32 oracle (its own smoothed PSD is up to 18 dB off on fresh random inputs) or a large one-ulp response cannot
# whitewash a systematic error of the product.  Calibrated in round 6, AFTER the spectrogram's DC / Nyquist nulls are
# recomputed exactly (kernels.hip k_spgm_env_wf FIX: the cause of every large value of rounds 3 - 5), on 6 000 fresh
# random configurations + the 131 regression inputs + the configuration matrix (profiles/r06_c_*; worst value in brackets):
#  * smoothed PSD never more than 1 dB off [0.22], PSDRES 2 dB [0.68], band energies 1e-3 [1.2e-4];
#  * away from DC / Nyquist -- PSD points 4 ... npsd - 3 -- never more than 0.2 dB [0.064: one value in one input is
#    over 0.05 dB at all];
#  * the NUMBER of smoothed-PSD values over 0.05 dB (the count clause of the earlier tiers, kept beside the yardsticks):
#    at most max(8, 2e-4 x the utterance's PSD values) [6 of 27 648];
#  * PSDRES has a bound of its own where it matters -- where the raw periodogram it completes lies above -20 dB re the
#    frame's largest PSD value (what layer0.c:606 adds back where the signal is): 0.5 dB [0.12].
# A build with the Kalman gain x 1.5 (-DKAL_BREAK, tools/kbench.py) fails 190 of 240 tests of the parity modules.
CEILING = dict(psd_db_max=1.0, psdres_db_max=2.0, edc_rel_max=1e-3, psd_db_max_interior=0.2, psdres_db_max_above_m20db=0.5)
CEIL_COUNT_MIN, CEIL_FRAC = 8, 2e-4


def ceiling_violations(m):
    bad = [(k + " (ceiling)", m[k], tol) for k, tol in CEILING.items() if k in m and not m[k] <= tol]
    if "psd_values" in m and not m["psd_over_0p05_db_count"] <= max(CEIL_COUNT_MIN, CEIL_FRAC * m["psd_values"]):
        bad.append(("psd_over_0p05_db_count (ceiling)", m["psd_over_0p05_db_count"], max(CEIL_COUNT_MIN, CEIL_FRAC * m["psd_values"])))
    return bad


CONVENTION_NAMES = ("hann_periodic", "moving_avg_half", "filtfilt_pad", "interp1u_exclusive", "kalman_init", "spec2env_lobe_1e6",
                    "lf_rd_clamp")


def oracle32_metrics(okw, x, fs, f0):
    """analysis_metrics of the FLOAT32 build of the oracle against the float64 build on the same input: how far the
    reference's own FP_TYPE = float arithmetic sits from exact arithmetic (tools/oracle_f32_spread.py as a function)."""
    from oracle.oracle import Oracle
    o64, o32 = Oracle(np.float64), Oracle(np.float32)
    L = llsm.load()
    for name in CONVENTION_NAMES:             # the float32 build is its own library: same conventions as product / float64 build
        o32.set_convention(name, L.llsm_gpu_get_convention(name.encode()))
    p64, r64 = o64.analyze(o64.aoptions(**okw), x, fs, f0, want_res=True)
    q, r32 = o32.analyze(o32.aoptions(**okw), x, fs, f0, want_res=True)
    g = {llsm.A_NHAR: q.nhar, llsm.A_NHAR_E: q.nhar_e, llsm.A_AMPL: q.ampl, llsm.A_PHSE: q.phse, llsm.A_PSD: q.psd,
         llsm.A_PSDRES: q.psdres, llsm.A_EDC: q.edc, llsm.A_EENV_AMPL: q.eenv_ampl, llsm.A_EENV_PHSE: q.eenv_phse}
    return analysis_metrics(g, slice(0, len(f0)), p64, np.asarray(r32, np.float64), r64)


def oracle_ulp_response(okw, x, fs, f0, n=4, seed=7):
    """{metric: max over n draws} of the float64 oracle's response to a one-ulp perturbation of its float32 input,
    x -> x (1 +- 2^-24) with random signs: how far the EXACT algorithm moves when the input moves by its own resolution."""
    from oracle.oracle import Oracle
    o64 = Oracle(np.float64)
    p0, r0 = o64.analyze(o64.aoptions(**okw), x, fs, f0, want_res=True)
    rng = np.random.default_rng(seed)
    out = {}
    for _ in range(n):
        xp = (np.asarray(x, np.float32) * (1 + rng.choice([-1, 1], size=len(x)) * 2.0 ** -24)).astype(np.float32)
        q, rq = o64.analyze(o64.aoptions(**okw), xp, fs, f0, want_res=True)
        g = {llsm.A_NHAR: q.nhar, llsm.A_NHAR_E: q.nhar_e, llsm.A_AMPL: q.ampl, llsm.A_PHSE: q.phse, llsm.A_PSD: q.psd,
             llsm.A_PSDRES: q.psdres, llsm.A_EDC: q.edc, llsm.A_EENV_AMPL: q.eenv_ampl, llsm.A_EENV_PHSE: q.eenv_phse}
        mm = analysis_metrics(g, slice(0, len(f0)), p0, rq, r0)
        for k, v in mm.items():
            out[k] = max(out.get(k, 0.0), v)
    return out


def contract_violations(m, f32_metrics=None, contract=None, conditioned=None, ulp_response=None):
    """[(metric, value, bound)] of everything in m outside the contract.  f32_metrics / ulp_response: callables returning
    oracle32_metrics(...) / oracle_ulp_response(...) of the same input (each evaluated at most once, and only if a
    conditioned metric is over its plain value -- the second only if the first does not cover it) or None."""
    bad = []
    for k, tol in (CONTRACT if contract is None else contract).items():
        if not m[k] <= tol:
            bad.append((k, m[k], tol))
    m32 = mu = None
    for k, (tol, kappa, yard, kappa_ulp, *add) in (CONDITIONED if conditioned is None else conditioned).items():
        if m[k] <= tol:
            continue
        if m32 is None and f32_metrics is not None:
            m32 = f32_metrics()
        y32 = None if m32 is None else max(m32[t] for t in yard)
        bound = tol if y32 is None else max(tol, kappa * y32 + (add[0] if add else 0.0) * tol)
        m[k + "_f32_oracle"] = y32
        if not m[k] <= bound and ulp_response is not None:
            if mu is None:
                mu = ulp_response()
            m[k + "_ulp_response"] = mu[k]
            bound = max(bound, kappa_ulp * mu[k])
        if not m[k] <= bound:
            bad.append((k, m[k], bound))
    if m.get("nhar_mismatch", 0) or m.get("nhar_e_mismatch", 0):
        bad.append(("nhar", m["nhar_mismatch"], m["nhar_e_mismatch"]))
    if conditioned is None:                                   # the layer-0 contract proper (HMPP: assert_hmpp_contract)
        bad += ceiling_violations(m)
    return bad


class Yard:
    """The two yardsticks of one input for the conditioned metrics, evaluated lazily and once: calling it gives
    oracle32_metrics(...), .ulp() gives oracle_ulp_response(...)."""
    def __init__(self, okw, x, fs, f0):
        self.args = (okw, x, fs, f0)
        self._m32 = self._mu = None

    def __call__(self):
        if self._m32 is None:
            self._m32 = oracle32_metrics(*self.args)
        return self._m32

    def ulp(self):
        if self._mu is None:
            self._mu = oracle_ulp_response(*self.args)
        return self._mu


def assert_contract(m, f32_metrics=None, where="", **kw):
    if isinstance(f32_metrics, Yard) and "ulp_response" not in kw:
        kw["ulp_response"] = f32_metrics.ulp
    bad = contract_violations(m, f32_metrics, **kw)
    assert not bad, (where, bad)


# Peak picking (LLSM_AOPTION_HMPP, dsputils.c:126-143): a harmonic is the ARG-MAX of the spectrum over the bins around
# k f0, refined by a parabola, its phase a linear interpolation of WRAPPED bin phases.  Above -40 dB SURVEY 8(d)'s bounds
# hold as they stand (measured 2.3e-4 rad).  The method itself is discontinuous: a near-tie of two local maxima resolves
# differently in float32 and float64 (in the float32 build of the oracle as in the product, though not always on the same
# harmonic), the harmonic lands on the other maximum, and the residual, its PSD and the band envelopes follow.  So:
#   (A) the contract above with EVERY metric conditioned on the two yardsticks: err(HIP, f64) <= 8(d)'s value, or
#       <= err(float32 oracle, f64) + 8(d)'s value (the product on the SAME other maximum as the reference's float build
#       is as far from float64 as that build to five digits -- seeds 40419, 40587: 0.0268968 against 0.0268952 -- so the
#       plain 1 x of the layer-0 contract is a knife edge here; the sum is what "within 8(d) of the float32 oracle" implies
#       by the triangle inequality), or <= 4 x the float64 oracle's one-ulp response; or
#   (B) at most max(3, 0.5 %) of the utterance's harmonics outside the harmonic bounds (1e-5 of the largest amplitude as a
#       complex number; above -40 dB the relative 1e-4 / 1e-3 rad -- seed 80189: ONE harmonic 35 dB down whose parabola was
#       fitted around the neighbouring bin, 4.8e-4 relative and 8.6e-6 of the maximum) and at most max(3, 5 %) of its
#       envelope-harmonic values outside 8(d) (the band envelopes are peak-picked too -- a handful of values per band;
#       the arg-max took another maximum for them) -- or, either count, at most what the FLOAT32 ORACLE moves on the same input
#       (the yardstick of (A) applied to the counts) --, the harmonic counts equal, and the residual-derived rows not asserted.
#       Soak of 1 000 random configurations (profiles/r05_e_soak_others.txt): 35 under (B); the two beyond its fractions
#       (40587: 8 harmonics of 840; 40419: 17 envelope values of 276, where the float32 oracle moves 49) pass under (A).
HMPP_CONTRACT = {}
HMPP_CONDITIONED = {_k: _v + (1.0,) for _k, _v in CONDITIONED.items()}
for _k, _tol in CONTRACT.items():
    HMPP_CONDITIONED[_k] = (_tol, 1.0, (_k,), 4.0, 1.0)
HMPP_MAX_MOVED = 3
HMPP_B_CEILING = dict(xres_rel_rms=2e-2, psdraw_db_max_above_m20db=6.0, psd_db_max=3.0, edc_rel_max=2e-2)


def assert_hmpp_contract(m, f32_metrics=None, where="", **kw):
    if isinstance(f32_metrics, Yard) and "ulp_response" not in kw:
        kw["ulp_response"] = f32_metrics.ulp
    bad = contract_violations(m, f32_metrics, contract=HMPP_CONTRACT, conditioned=HMPP_CONDITIONED, **kw)
    m["hmpp_branch"] = "A" if not bad else "B"
    if bad:
        moved, emoved = m["harm_over_count"], m["eenv_over_count"]
        # how many the reference's own float build moves on this input (seed 92774, 8 kHz, 72 envelope values in all: 8 here,
        # 18 in the float32 oracle -- one harmonic on another maximum changes the residual under several envelope frames)
        m32 = f32_metrics() if f32_metrics is not None else {}
        m["harm_over_count_f32_oracle"], m["eenv_over_count_f32_oracle"] = m32.get("harm_over_count", 0), m32.get("eenv_over_count", 0)
        # (B) no longer skips the residual-derived rows (VERDICT r5 item 2 iv).  A harmonic on another local maximum leaves
        # its difference in the residual, so these rows cannot be held to the float32 oracle's distance (which may have kept
        # that harmonic: seed 92774, residual 2.7e-3 here against 3.1e-5 there); they get absolute ceilings instead -- a
        # handful of weak harmonics cannot move the residual by more than a few per cent or its spectrum by more than a few dB
        bad_b = [(k, m[k], c) for k, c in HMPP_B_CEILING.items() if not m[k] <= c]
        m["hmpp_b_residual_ratio"] = max(m[k] / c for k, c in HMPP_B_CEILING.items())
        assert not bad_b, (where, "branch B residual rows", bad_b)
        assert 0 < moved + emoved and moved <= max(HMPP_MAX_MOVED, 0.005 * m["harm_count"], m32.get("harm_over_count", 0)) and \
            emoved <= max(HMPP_MAX_MOVED, 0.05 * m["eenv_count"], m32.get("eenv_over_count", 0)) and \
            not (m["nhar_mismatch"] or m["nhar_e_mismatch"]), (where, bad, moved, emoved, m["harm_count"], m["eenv_count"],
                                                               m32.get("harm_over_count"), m32.get("eenv_over_count"))
