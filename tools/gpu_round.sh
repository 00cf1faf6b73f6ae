#!/bin/bash
# One GPU-box visit: new tests first, whole gpu suite, default bench, rocprof stats + PMC traffic.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
echo "== pytest -m gpu (${TESTS:-tests}) =="
timeout 1500 python -m pytest ${TESTS:-tests} -m gpu -q -x -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | tail -40
echo "== bench default =="
timeout 600 python bench.py 2>gpurun_out/bench_default.err | tee gpurun_out/bench_default.json | cut -c1-1500
if [ -n "$PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_stats -o bench -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/gpurun_out/prof_bench.log 2>&1
  python $REPO/tools/rocpd_summary.py $(find $REPO/gpurun_out/prof_stats -name "*.db" | head -1) > $REPO/gpurun_out/kernel_stats.txt
  head -20 $REPO/gpurun_out/kernel_stats.txt
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $REPO/gpurun_out/prof_pmc_fetch -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/gpurun_out/prof_pmc_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $REPO/gpurun_out/prof_pmc_write -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/gpurun_out/prof_pmc_write.log 2>&1
  python $REPO/tools/rocpd_traffic.py $(find $REPO/gpurun_out/prof_pmc_fetch -name "*.db" | head -1) $(find $REPO/gpurun_out/prof_pmc_write -name "*.db" | head -1) > $REPO/gpurun_out/prof_traffic.json
  head -c 400 $REPO/gpurun_out/prof_traffic.json
  rm -rf $REPO/gpurun_out/prof_pmc_fetch $REPO/gpurun_out/prof_pmc_write
  find $REPO/gpurun_out/prof_stats -name "*.db" -delete
  cd $REPO
fi
if [ -n "$EXTRA" ]; then eval "$EXTRA"; fi
