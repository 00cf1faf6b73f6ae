#!/bin/bash
# One GPU-box visit for llsmrt: its tests, the rt64 / rt64pbp benches (plain and one-hipGraph-per-hop), kernel stats.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
timeout 300 python -m pytest tests/test_gpu_rt.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for w in rt64 rt64pbp; do
  timeout 200 python bench.py --workload $w --steps 3 --warmup 1 2>gpurun_out/bench_$w.err | tee gpurun_out/bench_$w.json | cut -c1-330
done
timeout 200 python bench.py --workload rt64 --steps 3 --warmup 1 --rt-graph 1 2>/dev/null | tee gpurun_out/bench_rt64_graph.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_rt -o rt -- python $REPO/bench.py --workload rt64 --steps 2 --warmup 1 > $REPO/gpurun_out/prof_rt.log 2>&1
python $REPO/tools/rocpd_summary.py $(find $REPO/gpurun_out/prof_rt -name "*.db" | head -1) > $REPO/gpurun_out/rt64_kernel_stats.txt
find $REPO/gpurun_out/prof_rt -name "*.db" -delete
head -10 $REPO/gpurun_out/rt64_kernel_stats.txt | cut -c1-110
