#!/bin/bash
# Round 4, visit G: llsmrt pack (row copies on the helper threads), object path (chunk API), drop-in latency.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 600 python -m pytest tests/test_gpu_rt.py tests/test_c_host.py tests/test_gpu_round2.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
for v in "" "LLSM_RT_PACK_THREADS=0" "LLSM_RT_PACK_THREADS=7"; do
  echo "-- rt64pbp ${v:-default}"
  env $v LLSM_TIMING=1 timeout 300 python bench.py --workload rt64pbp --steps 3 --warmup 1 2>gpurun_out/rt_timing.err | cut -c1-160
  grep "llsmrt feed" gpurun_out/rt_timing.err | tail -1
done | tee gpurun_out/r04_g_rt.txt
echo "-- rt64"
LLSM_TIMING=1 timeout 300 python bench.py --workload rt64 --steps 3 --warmup 1 2>gpurun_out/rt_timing.err | cut -c1-160
grep "llsmrt feed" gpurun_out/rt_timing.err | tail -1
for blk in 128 64 32; do
  echo "-- chunk api workers 8 block $blk"
  LLSM_TIMING=1 timeout 300 python tools/bench_chunk_api.py --workers 8 --block $blk --reps 3 2> gpurun_out/chunk_api.err | tee gpurun_out/r04_g_chunk_api_$blk.json | cut -c90-400
  grep -i "synthesize_block" gpurun_out/chunk_api.err | tail -2 | cut -c1-300
  grep -i "analyze_block" gpurun_out/chunk_api.err | tail -1 | cut -c1-300
done
timeout 200 python tools/bench_dropin.py 2>/dev/null | tee gpurun_out/r04_g_dropin.json | cut -c1-400
