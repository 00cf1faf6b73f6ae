#!/bin/bash
# Round 5, visit F: the padded in-place LDS transform in k_l1_frame / k_l1_to_l0 / k_pbp_pulse (layer-1 tests, l1 and rt64pbp
# bench legs), the HMPP contract with its arg-max branch (regression seeds + the HMPP sweep of the soak).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest l1 / rt / regressions(hmpp) / c_host =="
timeout 1200 python -m pytest tests/test_gpu_l1.py tests/test_gpu_rt.py tests/test_gpu_frameapi.py tests/test_c_host.py "tests/test_gpu_regressions.py::test_marginal_hmpp_seeds" tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-600 | head -20
echo "== bench l1 / rt64pbp =="
for w in l1 rt64pbp; do timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:40], round(d['value']), d['ms_per_step'], d.get('kernels_ms_per_step'))"; done
echo "== soak HMPP 40000 .. =="
( time SOAK_ONLY=hmpp timeout 900 python tools/fuzz_soak.py 40000 2500 ) 2>&1 | grep -E "^soak: 500|^FAIL hmpp|^WORST-HMPP|^real" | cut -c1-700
echo "== object path (delete with prefetch) =="
timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 4 --batch-delete 1 2>/dev/null | cut -c100-640
timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 4 --batch-delete 0 2>/dev/null | cut -c100-640
