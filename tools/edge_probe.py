import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import libllsm2_amd as llsm
from conftest import make_speechlike
from oracle.oracle import Oracle
from test_gpu_configs import _run_parity
o64 = Oracle(np.float64)
ctx = llsm.Context(0)
for fs, thop, kw in [(44100.0, 0.001, dict()), (44100.0, 0.002, dict()), (44100.0, 0.02, dict()), (16000.0, 0.025, dict(nchannel=2, chanfreq=[3000.0])),
                     (8000.0, 0.002, dict(nchannel=2, chanfreq=[1500.0], maxnhar=40)), (96000.0, 0.0025, dict(maxnhar=300, npsd=512))]:
    x, f0 = make_speechlike(9, nx=int(0.4 * fs), fs=fs, thop=thop)
    try:
        _run_parity(ctx, o64, "probe_%d_%g" % (fs, thop), fs, thop, kw, x, f0.astype(np.float32))
        print("OK  ", fs, thop, kw, flush=True)
    except Exception as e:
        print("FAIL", fs, thop, kw, repr(e)[:400], flush=True)
