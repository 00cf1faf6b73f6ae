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


n1 = max(count // 5, 1)
for seed in range(first, first + n1):
    run("layer-1", test_gpu_l1.test_random_layer1_configurations, ctx, o64, seed)
    run("llsmrt", test_gpu_rt.test_rt_random_configurations, o64, seed)
    run("llsmrt PbP", test_gpu_l1.test_random_rt_pbp_configurations, o64, seed)
print("soak: " + "; ".join("%d %s cases, %d failures" % (v[0], k, v[1]) for k, v in counts.items()))
