#!/bin/bash
# llsmrt hop anatomy: its tests, host phase times (LLSM_TIMING=1) and the device timeline of a hop (rocprofv3 kernel trace).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_rt.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
for d in 0 1; do
  LLSM_RT_DIRECT=$d LLSM_TIMING=1 timeout 200 python bench.py --workload rt64 --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/rt_timing_$d.err | tee gpurun_out/bench_rt64_direct$d.json | cut -c1-200
  grep "llsmrt feed" gpurun_out/rt_timing_$d.err | tail -2
done
timeout 200 python bench.py --workload rt64pbp --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_rt64pbp.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_rt -o rt -- python $REPO/bench.py --workload rt64 --steps 2 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof_rt.log 2>&1
python $REPO/tools/rt_timeline.py $(find $REPO/gpurun_out/prof_rt -name "*.db" | head -1) | tee $REPO/gpurun_out/rt_timeline.txt
python $REPO/tools/rocpd_summary.py $(find $REPO/gpurun_out/prof_rt -name "*.db" | head -1) | grep -E "^kernel|k_rt|copy" > $REPO/gpurun_out/rt64_kernel_stats.txt
cat $REPO/gpurun_out/rt64_kernel_stats.txt
find $REPO/gpurun_out/prof_rt -name "*.db" -delete
