"""k_filtfilt cuts FEW LONG signals along time into segments with halos (engine.cpp build_jobs, round 6): the drop-in
llsm_analyze of one utterance runs its four band signals as ~40 short jobs each instead of four wavefronts on the whole chip.
The cut must not show: band energies (edc) and envelope harmonics of an analysis with the cut equal those without it
($LLSM_GPU_FILT_SEGMENTS=0, read once per process: two child processes) to float32 rounding; what does not pass through the
band filters (the PSD rows) is bit-identical."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r"""
import sys, json, numpy as np, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, %r)
import libllsm2_amd as llsm
from conftest import make_speechlike
L = llsm.load()
fs = 44100.0
x, f0 = make_speechlike(7, nx=150000, fs=fs, thop=0.005)
f0 = f0.astype(np.float32)
ctx = llsm.Context(0)
ao = llsm.make_aoptions(f0_refine=0, thop=0.005)
b = llsm.Batch(ctx, ao, fs, [len(x)], [len(f0)])
b.upload(llsm.A_X, x); b.upload(llsm.A_F0, f0)
b.analyze(); ctx.sync()
g = b.download_params()
np.savez(sys.argv[1], edc=g[llsm.A_EDC], ea=g[llsm.A_EENV_AMPL], ep=g[llsm.A_EENV_PHSE], psd=g[llsm.A_PSD])
"""


def _run(tmp, tag, env):
    path = os.path.join(tmp, tag + ".npz")
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, HERE), path], env=dict(os.environ, **env), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)


@pytest.mark.gpu
def test_time_segments_of_long_signals_do_not_show(tmp_path):
    a = _run(str(tmp_path), "cut", {})
    w = _run(str(tmp_path), "whole", {"LLSM_GPU_FILT_SEGMENTS": "0"})
    m = {}
    m["edc_rel"] = float(np.max(np.abs(a["edc"] - w["edc"]) / np.maximum(np.abs(w["edc"]), 1e-12)))
    m["eenv_ampl_over_max"] = float(np.max(np.abs(a["ea"] - w["ea"])) / max(float(np.max(w["ea"])), 1e-30))
    m["psd_db"] = float(np.max(np.abs(a["psd"] - w["psd"])))
    print(json.dumps(m))
    assert m["edc_rel"] <= 1e-5 and m["eenv_ampl_over_max"] <= 1e-5 and m["psd_db"] == 0.0, m
