#!/bin/bash
# Round 5, visit G: analysed frames laid over packed records in registered slabs (tests, object-path bench against the staged
# path), the HMPP contract with its envelope branch.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest round2 / c_host / regressions(hmpp) / l1 =="
timeout 1200 python -m pytest tests/test_gpu_round2.py tests/test_c_host.py tests/test_gpu_full.py tests/test_gpu_l1.py "tests/test_gpu_regressions.py::test_marginal_hmpp_seeds" -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-600 | head -20
echo "== object path: packed (default) =="
for cfg in "8 32" "12 16" "8 64"; do set -- $cfg
  timeout 300 python tools/bench_chunk_api.py --workers $1 --block $2 --reps 4 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_g_chunk_api_packed_w$1_b$2.json | cut -c100-640; done
echo "== object path: staged (LLSM_PACKED_FRAMES=0) =="
LLSM_PACKED_FRAMES=0 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 4 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_g_chunk_api_staged_w8_b32.json | cut -c100-640
LLSM_TIMING=1 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 2 --batch-delete 1 2>&1 | grep -E "^\[analyze_block" | tail -8 | cut -c1-300 | tee gpurun_out/r05_g_chunk_api_analysis_phases.txt
echo "== soak HMPP 40000 .. =="
( time SOAK_ONLY=hmpp timeout 900 python tools/fuzz_soak.py 40000 5000 ) 2>&1 | grep -E "^soak: 1000|^FAIL hmpp|^real" | cut -c1-700
