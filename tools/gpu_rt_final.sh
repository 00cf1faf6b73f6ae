#!/bin/bash
# llsmrt closing visit: RT + layer-1 tests, rt64 / rt64pbp benches (default mode and with copies), kernel stats, hop timeline.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_rt.py tests/test_gpu_l1.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for w in rt64 rt64pbp; do
  LLSM_TIMING=1 timeout 200 python bench.py --workload $w --steps 5 --warmup 2 2>gpurun_out/timing_$w.err | tee gpurun_out/bench_$w.json | cut -c1-200
  grep "llsmrt feed" gpurun_out/timing_$w.err | tail -1 | tee gpurun_out/feed_phases_$w.txt
  LLSM_RT_DIRECT=0 LLSM_RT_FUSED=1 timeout 200 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_${w}_two_launches_copies.json | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
for w in rt64 rt64pbp; do
  timeout 300 rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_$w -o rt -- python $REPO/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof_$w.log 2>&1
  python $REPO/tools/rocpd_summary.py $(find $REPO/gpurun_out/prof_$w -name "*.db" | head -1) | grep -E "^kernel|k_|copy" | head -8 > $REPO/gpurun_out/${w}_kernel_stats.txt
  cat $REPO/gpurun_out/${w}_kernel_stats.txt
  [ $w = rt64 ] && python $REPO/tools/rt_timeline.py $(find $REPO/gpurun_out/prof_$w -name "*.db" | head -1) | tee $REPO/gpurun_out/rt64_timeline.txt
  find $REPO/gpurun_out/prof_$w -name "*.db" -delete
done
