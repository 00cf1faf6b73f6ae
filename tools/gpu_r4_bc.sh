#!/bin/bash
# Round 4, visit BC: per-worker batch objects kept between blocks of equal shape (LLSM_GPU_BATCH_CACHE, default on) against
# a fresh batch per block (=0): llsm_analyze_batch + llsm_synthesize_batch with 8 workers, drop-in latency, the host tests.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
{
for r in 1 2 3; do for c in 0 1; do
  LLSM_GPU_BATCH_CACHE=$c timeout 300 python tools/bench_chunk_api.py --utts 1024 --workers 8 --block 32 --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('batch cache $c', 'analyze', round(d['analyze_ms'],1), 'synthesize', round(d['synthesize_ms'],1), 'M frames/s', round(d['value']/1e6,2))"
done; done
for c in 0 1; do LLSM_GPU_BATCH_CACHE=$c timeout 300 python tools/bench_dropin.py 2>/dev/null | tail -1 | cut -c1-400; done
} | tee gpurun_out/r04_bc_batch_cache.txt
timeout 900 python -m pytest tests/test_c_host.py tests/test_gpu_round2.py tests/test_gpu_full.py tests/test_gpu_l1.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 600 python tools/leak_probe.py 2>&1 | tail -1
