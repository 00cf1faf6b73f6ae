"""Which float32 stage moves a smoothed-PSD value?  CPU only (seed 123208).

    python tools/psd_bisect.py [seed] [frame] [point] [--product]

--product (GPU box): additionally the PRODUCT's own envelope / log-periodogram planes (llsm_gpu_batch_debug_plane) are
substituted into the float64 chain one at a time -- which of the two carries the product's error.

The float64 oracle's noise-PSD chain (layer0.c:325-408) is run with ONE intermediate array at a time taken from the float32
build of the same oracle (oracle/llsm_oracle.c: o_set_stage_hook): residual, spectrogram magnitudes, resampled envelope,
log periodogram, process variance Q, filtered means / variances, smoothed means.  For every substitution: the smoothed PSD's
error against the all-float64 result at (frame, point) and its maximum over the utterance.  Then, for the stage that
carries it, finer cuts (only the DC bin of the spectrogram, only bins 0 ... 3, everything but those; values merely ROUNDED
to float32 instead of computed in float32)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as omod
from oracle.oracle import Oracle

omod.build()
import libllsm2_amd as llsm                       # make_aoptions only (no device call)
from conftest import make_speechlike
from test_gpu_configs import _fuzz_case
from gpu_common import aopt_kwargs

PRODUCT = "--product" in sys.argv
if PRODUCT:
    sys.argv.remove("--product")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 123208
fs, thop, kw, nx = _fuzz_case(seed)
x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop)
f0 = f0.astype(np.float32)
ao = llsm.make_aoptions(f0_refine=0, thop=thop, **kw)
okw = aopt_kwargs(ao); okw["chanfreq"] = kw["chanfreq"]
o64, o32 = Oracle(np.float64), Oracle(np.float32)
NAMES = ["x_res", "spectrogram |X|", "envelope (spec2env, resampled)", "log periodogram of x_res", "Q", "filtered mean", "filtered variance", "smoothed mean"]


def run(o, hook=None):
    fpt = C.c_double if o.dtype == np.float64 else C.c_float
    HT = C.CFUNCTYPE(None, C.c_int, C.c_int, C.POINTER(fpt), C.c_long)
    cb = HT(hook) if hook else C.cast(None, HT)
    o.lib.o_set_stage_hook(cb)
    try:
        pr = o.analyze(o.aoptions(**okw), x, fs, f0)
    finally:
        o.lib.o_set_stage_hook(C.cast(None, HT))
    return pr.psd.astype(np.float64).reshape(len(f0), -1)


cap32, cap64 = {}, {}


def capture(store, dtype):
    def h(stage, index, data, n):
        store[(stage, index)] = np.ctypeslib.as_array(data, shape=(n,)).astype(dtype).copy()
    return h


psd64 = run(o64, capture(cap64, np.float64))
psd32 = run(o32, capture(cap32, np.float32))
e32 = psd32 - psd64
if len(sys.argv) > 3:
    fr, pt = int(sys.argv[2]), int(sys.argv[3])
elif seed == 123208:
    fr, pt = 42, 0                                 # where the product is 1.95 dB off (profiles/r05_zz4_psd_probe_123208.txt)
else:
    fr, pt = np.unravel_index(np.argmax(np.abs(e32)), e32.shape)
print(f"seed {seed} fs {fs} thop {thop} nfrm {len(f0)} npsd {psd64.shape[1]}; looking at frame {fr} point {pt}")
print(f"float32 oracle as a whole: {e32[fr, pt]:+.4f} dB there, max |.| {np.abs(e32).max():.4f} dB at {np.unravel_index(np.argmax(np.abs(e32)), e32.shape)}")
ns_spgm = len(cap64[(1, 0)]) // len(f0)


def substitute(stages, mask=None, round_only=False):
    """float64 run with the arrays of `stages` replaced by the float32 build's (or rounded to float32); mask(stage, a64, a32)
    -> array restricts the replacement"""
    def h(stage, index, data, n):
        if stage not in stages:
            return
        a = np.ctypeslib.as_array(data, shape=(n,))
        new = a.astype(np.float32).astype(np.float64) if round_only else cap32[(stage, index)].astype(np.float64)
        if mask is not None:
            new = mask(stage, a.copy(), new)
        a[:] = new
    p = run(o64, h)
    return p - psd64


print("\none stage at a time from the float32 build (everything else float64):")
print(f"  {'stage':42s} {'at (fr, pt)':>12s} {'max |.|':>9s}  where")
for s in range(8):
    e = substitute({s})
    w = np.unravel_index(np.argmax(np.abs(e)), e.shape)
    print(f"  {s} {NAMES[s]:40s} {e[fr, pt]:+12.4f} {np.abs(e).max():9.4f}  {w}")
print("\none stage at a time merely ROUNDED to float32 (computed in float64):")
for s in range(8):
    e = substitute({s}, round_only=True)
    print(f"  {s} {NAMES[s]:40s} {e[fr, pt]:+12.4f} {np.abs(e).max():9.4f}")


def bins(sel):
    def m(stage, a64, a32):
        a64 = a64.reshape(len(f0), ns_spgm); a32 = a32.reshape(len(f0), ns_spgm)
        out = a64.copy(); out[:, sel] = a32[:, sel]
        return out.reshape(-1)
    return m


print("\nthe spectrogram, by bins (float32 build's values in the bins named, float64 elsewhere):")
allb = np.arange(ns_spgm)
for name, sel in (("bin 0 (DC)", allb == 0), ("bins 0 .. 3", allb < 4), ("bins 0 .. 7", allb < 8), ("all but bins 0 .. 7", allb >= 8),
                  ("bin %d (Nyquist)" % (ns_spgm - 1), allb == ns_spgm - 1)):
    e = substitute({1}, bins(sel))
    print(f"  {name:24s} {e[fr, pt]:+12.4f} {np.abs(e).max():9.4f}")
s64 = cap64[(1, 0)].reshape(len(f0), ns_spgm); s32 = cap32[(1, 0)].reshape(len(f0), ns_spgm).astype(np.float64)
rel = np.abs(s32 - s64) / s64.max(axis=1, keepdims=True)
print(f"\nspectrogram of frames {fr - 2} .. {fr + 2}: level of bins 0 .. 3 re the frame's largest bin (dB), and the float32 build's relative error there")
for i in range(max(0, fr - 2), min(len(f0), fr + 3)):
    lv = 20 * np.log10(np.maximum(s64[i, :4], 1e-300) / s64[i].max())
    er = np.abs(s32[i, :4] - s64[i, :4]) / np.maximum(s64[i, :4], 1e-300)
    print(f"  frame {i}: level {np.round(lv, 1)}  rel err {np.array2string(er, precision=2)}  abs err / max {np.array2string(rel[i, :4], precision=2)}")
q64 = np.array([cap64[(4, j)] for j in range(4)]); q32 = np.array([cap32[(4, j)] for j in range(4)]).astype(np.float64)
print(f"\nprocess variance Q of PSD bins 0 .. 3 around frame {fr} (float64 | float32 build):")
for j in range(4):
    print(f"  bin {j}: {np.array2string(q64[j, max(0, fr - 3):fr + 3], precision=3)} | {np.array2string(q32[j, max(0, fr - 3):fr + 3], precision=3)}")

if PRODUCT:
    ctx = llsm.Context(0)
    b = llsm.Batch(ctx, ao, fs, [len(x)], [len(f0)])
    b.upload(llsm.A_X, x); b.upload(llsm.A_F0, f0)
    b.analyze(); ctx.sync()
    g = b.download_params()
    pg = g[llsm.A_PSD].astype(np.float64).reshape(psd64.shape)
    eg = pg - psd64
    print(f"\nPRODUCT: {eg[fr, pt]:+.4f} dB at (frame {fr}, point {pt}); max |.| {np.abs(eg).max():.4f} at {np.unravel_index(np.argmax(np.abs(eg)), eg.shape)}")
    planes = {}
    for which, stage in ((0, 2), (1, 3)):
        n = ctx.L.llsm_gpu_batch_debug_plane(b.h, which, None, 0)
        a = np.zeros(n, np.float32)
        assert ctx.L.llsm_gpu_batch_debug_plane(b.h, which, a.ctypes.data_as(C.c_void_p), n) == n
        planes[stage] = a.astype(np.float64).reshape(len(f0), -1)
    nspec = planes[2].shape[1]
    # the oracle keeps the log periodogram transposed ([nspec][nfrm])
    cap_prod = {2: planes[2].reshape(-1), 3: planes[3].T.copy().reshape(-1)}
    for stage in (2, 3):
        d = cap_prod[stage] - cap64[(stage, 0)]
        dd = d.reshape(len(f0), nspec) if stage == 2 else d.reshape(nspec, len(f0)).T
        w = np.unravel_index(np.argmax(np.abs(dd)), dd.shape)
        print(f"  product's {NAMES[stage]}: largest difference to the float64 chain {np.abs(dd).max():.4g} (natural-log units) at frame {w[0]} bin {w[1]}; "
              f"at frames {fr - 1} .. {fr + 1}, bins 0 .. 3: {np.array2string(dd[max(0, fr - 1):fr + 2, :4], precision=3)}")

    def sub_prod(stages):
        def h(stage, index, data, n):
            if stage in stages:
                np.ctypeslib.as_array(data, shape=(n,))[:] = cap_prod[stage]
        return run(o64, h) - psd64
    for stages, name in (({2}, "envelope plane from the product"), ({3}, "log periodogram plane from the product"), ({2, 3}, "both")):
        e = sub_prod(stages)
        print(f"  float64 chain with the {name:42s} {e[fr, pt]:+10.4f} at (fr, pt), max |.| {np.abs(e).max():.4f}; product minus this: {np.abs(eg - e).max():.4f}")
    b.close(); ctx.close()
