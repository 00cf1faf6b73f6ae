#!/bin/bash
# Round 5, visit J: the object path with both directions on packed records (analysis lands records in page-locked slabs by a
# kernel; synthesis reads them where they lie and writes the waveforms into page-locked pooled outputs).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest round2 / c_host / full / l1 / rt =="
timeout 1200 python -m pytest tests/test_gpu_round2.py tests/test_c_host.py tests/test_gpu_full.py tests/test_gpu_l1.py tests/test_gpu_rt.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-600 | head -12
echo "== object path: packed (default) =="
for cfg in "8 32" "12 32" "16 32" "16 16" "12 64"; do set -- $cfg
  timeout 300 python tools/bench_chunk_api.py --workers $1 --block $2 --reps 4 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_j_chunk_api_packed_w$1_b$2.json | cut -c100-640; done
echo "== object path: staged (LLSM_PACKED_FRAMES=0) =="
LLSM_PACKED_FRAMES=0 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 4 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_j_chunk_api_staged_w8_b32.json | cut -c100-640
LLSM_TIMING=1 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 3 --batch-delete 1 2>&1 | grep -E "^\[(analyze|synthesize)_block" | tail -10 | cut -c1-300 | tee gpurun_out/r05_j_chunk_api_phases.txt
