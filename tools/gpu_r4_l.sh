#!/bin/bash
# Round 4, visit L: table-based Rd fit, whole LF solution cached per frame.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_l1.py tests/test_gpu_rt.py tests/test_gpu_frameapi.py tests/test_gpu_coder.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6
for v in "" "LLSM_GPU_RD_FIT_TABLES=0"; do
  echo "-- l1 bench ${v:-default}"
  env $v timeout 300 python bench.py --workload l1 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), round(d['value']/1e6,2), {k: round(v,3) for k,v in list(d['kernels_ms_per_step'].items())}, d['host_ms_per_step'])"
done | tee gpurun_out/r04_l_l1.txt
