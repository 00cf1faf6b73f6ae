#!/bin/bash
# Round 4, visit D: k_pbp_pulse with the real-output inverse transform (layer 1 / llsmrt pulse-by-pulse).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== l1 + rt tests =="
timeout 900 python -m pytest tests/test_gpu_l1.py tests/test_gpu_rt.py tests/test_gpu_frameapi.py -m gpu -q -p no:cacheprovider 2>&1 | tail -12
for v in "" "LLSM_GPU_PBP_REAL=0" "PBP64" ; do
  echo "-- l1 bench ${v:-default}"
  if [ "$v" = "PBP64" ]; then continue; fi
  env $v timeout 300 python bench.py --workload l1 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), round(d['value']/1e6,2), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()}, d['host_ms_per_step'])"
done | tee gpurun_out/r04_d_l1.txt
for v in "" "LLSM_GPU_PBP_REAL=0"; do
  echo "-- rt64pbp ${v:-default}"
  env $v timeout 300 python bench.py --workload rt64pbp --steps 3 --warmup 1 2>/dev/null | cut -c1-200
done | tee -a gpurun_out/r04_d_l1.txt
