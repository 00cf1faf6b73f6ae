#!/bin/bash
# Round 4, visit SPIN: a synchronous llsmrt feed waiting on the hop kernel's completion word (page-locked, LLSM_RT_SPIN=1, default)
# against the stream wait (=0): tests, then rt64 / rt64pbp with synchronous feeds.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_rt.py tests/test_gpu_l1.py tests/test_gpu_round2.py tests/test_c_host.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
{
for r in 1 2 3; do for sp in 0 1; do for wl in rt64 rt64pbp; do
  LLSM_RT_SPIN=$sp timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --rt-pipeline 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$wl spin $sp', round(d['value']/1e6,3), 'M frames/s', round(d['ms_per_hop']*1e3,1), 'us per hop, max pull', round(d['max_pull_ms']*1e3,1), 'us')"
done; done; done
} | tee gpurun_out/r04_spin_rt.txt
