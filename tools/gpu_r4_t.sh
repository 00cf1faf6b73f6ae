#!/bin/bash
# Round 4, visit T: closed-form Hann lobes in k_l1_env_wf (hann_lobe_fast): layer-1 parity, then the l1 bench with kernel times.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_l1.py tests/test_gpu_coder.py tests/test_gpu_frameapi.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
for r in 1 2; do
timeout 300 python bench.py --workload l1 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value']/1e6,2),'M frames/s', round(d['ms_per_step'],2),'ms', 'gpu', round(d.get('gpu_ms_per_step',0),2), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if k.startswith('k_l1') or k.startswith('k_pbp')})"
done | tee gpurun_out/r04_t_l1_lobes.txt
