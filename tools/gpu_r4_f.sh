#!/bin/bash
# Round 4, visit F: float32 LF spectrum in the layer-1 / pulse kernels, helper threads for the llsmrt LF solves.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== l1 + rt + frameapi + coder tests =="
timeout 900 python -m pytest tests/test_gpu_l1.py tests/test_gpu_rt.py tests/test_gpu_frameapi.py tests/test_gpu_coder.py -m gpu -q -p no:cacheprovider 2>&1 | tail -12
for lib in "" exp_build/lib_LF_FAST_0.so; do
  echo "-- l1 bench ${lib:-product}"
  if [ -n "$lib" ]; then export LLSM_AMD_LIB=$PWD/$lib; else unset LLSM_AMD_LIB; fi
  timeout 300 python bench.py --workload l1 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), round(d['value']/1e6,2), {k: round(v,3) for k,v in list(d['kernels_ms_per_step'].items())}, d['host_ms_per_step'])"
done | tee gpurun_out/r04_f_l1.txt
unset LLSM_AMD_LIB
for v in "" "LLSM_RT_PACK_THREADS=0" "LLSM_RT_PACK_THREADS=7"; do
  echo "-- rt64pbp ${v:-default}"
  env $v LLSM_TIMING=1 timeout 300 python bench.py --workload rt64pbp --steps 3 --warmup 1 2>gpurun_out/rt_timing.err | cut -c1-220
  grep "llsmrt feed" gpurun_out/rt_timing.err | tail -1
done | tee -a gpurun_out/r04_f_l1.txt
nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"
