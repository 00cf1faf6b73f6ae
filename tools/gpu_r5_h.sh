#!/bin/bash
# Round 5, visit H: frames over packed records in PAGE-LOCKED slabs (hipHostMalloc; registered memory measured slow in visit G).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest round2 / c_host =="
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_c_host.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-600 | head -12
echo "== object path: packed (default) =="
for cfg in "8 32" "8 64" "12 32"; do set -- $cfg
  timeout 300 python tools/bench_chunk_api.py --workers $1 --block $2 --reps 4 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_h_chunk_api_packed_w$1_b$2.json | cut -c100-640; done
echo "== object path: staged (LLSM_PACKED_FRAMES=0) =="
LLSM_PACKED_FRAMES=0 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 4 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_h_chunk_api_staged_w8_b32.json | cut -c100-640
LLSM_TIMING=1 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 3 --batch-delete 1 2>&1 | grep -E "^\[analyze_block" | tail -8 | cut -c1-300 | tee gpurun_out/r05_h_chunk_api_analysis_phases.txt
