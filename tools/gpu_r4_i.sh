#!/bin/bash
# Round 4, visit I: tabulated LF phase in the llsmrt pulse tracker.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_rt.py tests/test_gpu_l1.py tests/test_c_host.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5
for i in 1 2; do
LLSM_TIMING=1 timeout 300 python bench.py --workload rt64pbp --steps 5 --warmup 1 2>gpurun_out/rt_timing.err | cut -c1-160
grep "llsmrt" gpurun_out/rt_timing.err | tail -2
done | tee gpurun_out/r04_i_rt.txt
