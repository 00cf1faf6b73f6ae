#!/bin/bash
# One GPU-box visit for the layer-1 path: its tests, the l1 bench with the scheduler's phase times, kernel stats.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_l1.py tests/test_gpu_coder.py tests/test_gpu_frameapi.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
LLSM_L1_TIMING=1 timeout 300 python bench.py --workload l1 --steps 4 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench_l1.err | tee gpurun_out/bench_l1.json | cut -c1-600
grep "l1 synth" gpurun_out/bench_l1.err | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_l1 -o l1 -- python $REPO/bench.py --workload l1 --steps 4 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof_l1.log 2>&1
python $REPO/tools/rocpd_summary.py $(find $REPO/gpurun_out/prof_l1 -name "*.db" | head -1) > $REPO/gpurun_out/l1_kernel_stats.txt
find $REPO/gpurun_out/prof_l1 -name "*.db" -delete
grep -v "at::native\|^void  " $REPO/gpurun_out/l1_kernel_stats.txt | head -16 | cut -c1-110
