"""One-off soak: the seeded configuration fuzz of tests/test_gpu_configs.py over many more seeds than the suite runs.
    python tools/fuzz_soak.py [first_seed] [count]
Prints one line per failing seed (configuration + the assertion) and a summary."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import libllsm2_amd as llsm
from conftest import make_speechlike
from oracle.oracle import Oracle
from test_gpu_configs import _fuzz_case, _run_parity

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
o64 = Oracle(np.float64)
ctx = llsm.Context(0)
bad = 0
for seed in range(first, first + count):
    fs, thop, kw, nx = _fuzz_case(seed)
    x, f0 = make_speechlike(100 + seed, nx=nx, fs=fs, thop=thop)
    try:
        _run_parity(ctx, o64, "soak", fs, thop, kw, x, f0.astype(np.float32))
    except Exception as e:                                    # noqa: BLE001
        bad += 1
        print("FAIL seed", seed, fs, thop, kw, nx, repr(e)[:300], flush=True)
print("soak: %d configurations, %d failures" % (count, bad))

# layer-1 and llsmrt sweeps (the parametrised test functions called directly with further seeds)
import test_gpu_l1, test_gpu_rt
bad1 = badr = 0
n1 = max(count // 5, 1)
orig = test_gpu_l1._l1_fuzz_case
for seed in range(first, first + n1):
    try:
        test_gpu_l1.test_random_layer1_configurations(ctx, o64, seed)
    except Exception as e:                                    # noqa: BLE001
        bad1 += 1; print("FAIL l1 seed", seed, orig(seed), repr(e)[:300], flush=True)
    try:
        test_gpu_rt.test_rt_random_configurations(o64, seed)
    except Exception as e:                                    # noqa: BLE001
        badr += 1; print("FAIL rt seed", seed, repr(e)[:300], flush=True)
print("soak: %d layer-1 cases, %d failures; %d llsmrt cases, %d failures" % (n1, bad1, n1, badr))
