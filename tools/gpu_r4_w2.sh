#!/bin/bash
# Round 4, visit W2: a longer soak at the final HEAD (seeds 8000 ..., 1000 layer-0 configurations and the other sweeps in proportion).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 2400 python tools/fuzz_soak.py 8000 1000 2>&1 | grep -E "^soak|^FAIL" | cut -c1-420 | tee gpurun_out/r04_w2_fuzz_soak.txt
