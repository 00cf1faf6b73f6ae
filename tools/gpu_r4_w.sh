#!/bin/bash
# Round 4, visit W: soak at HEAD after the Kalman variance form, the closed-form lobes and the streaming stores (fresh seeds).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 1500 python tools/fuzz_soak.py 7000 300 2>&1 | grep -E "^soak|^FAIL" | tee gpurun_out/r04_w_fuzz_soak.txt
