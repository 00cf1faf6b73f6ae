#!/bin/bash
# llsmrt pulse-by-pulse hop anatomy: the layer-1 and llsmrt tests, host phase times (LLSM_TIMING=1), per-kernel times.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_rt.py tests/test_gpu_l1.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
LLSM_TIMING=1 timeout 200 python bench.py --workload rt64pbp --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/rtpbp_timing.err | tee gpurun_out/bench_rt64pbp.json | cut -c1-200
grep "llsmrt feed" gpurun_out/rtpbp_timing.err | tail -2
timeout 200 python bench.py --workload l1 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tee gpurun_out/bench_l1.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_rt -o rt -- python $REPO/bench.py --workload rt64pbp --steps 2 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof_rt.log 2>&1
python $REPO/tools/rocpd_summary.py $(find $REPO/gpurun_out/prof_rt -name "*.db" | head -1) | grep -E "^kernel|k_|copy" | head -8 > $REPO/gpurun_out/rt64pbp_kernel_stats.txt
cat $REPO/gpurun_out/rt64pbp_kernel_stats.txt
find $REPO/gpurun_out/prof_rt -name "*.db" -delete
