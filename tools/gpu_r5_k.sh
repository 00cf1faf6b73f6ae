#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest round2 =="
timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25 | cut -c1-400
echo "== phases =="
LLSM_TIMING=1 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 3 --batch-delete 1 2>&1 | grep -E "^\[(synthesize)_block" | tail -6 | cut -c1-400
LLSM_TIMING=1 timeout 300 python tools/bench_chunk_api.py --workers 1 --block 32 --reps 2 --batch-delete 1 2>&1 | grep -E "^\[(analyze|synthesize)_block" | tail -4 | cut -c1-400
