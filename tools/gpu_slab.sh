#!/bin/bash
# Frame slabs: whole gpu suite, then the object-path batch API with 1 and 8 workers, slabs on and off.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
for s in 1 0; do for w in 1 8; do
  echo "slabs=$s workers=$w"
  LLSM_FRAME_SLABS=$s LLSM_TIMING=1 timeout 200 python tools/bench_chunk_api.py --workers $w --block 128 --reps 3 2> gpurun_out/chunk_api_${s}_$w.err | tee gpurun_out/chunk_api_${s}_$w.json | cut -c90-400
  grep -i "analyze_block" gpurun_out/chunk_api_${s}_$w.err | tail -1 | cut -c1-300
done; done
timeout 200 python tools/bench_dropin.py 2>/dev/null | tee gpurun_out/dropin.json | cut -c1-400
