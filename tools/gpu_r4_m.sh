#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_l1.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
for i in 1 2; do
LLSM_L1_TIMING=1 timeout 300 python bench.py --workload l1 --steps 5 --warmup 2 2>gpurun_out/l1_timing.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), round(d['value']/1e6,2), round(d['gpu_ms_per_step'],2), d['host_ms_per_step'])"
grep "l1 synth" gpurun_out/l1_timing.err | tail -2
done | tee gpurun_out/r04_m_l1.txt
