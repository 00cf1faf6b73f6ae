#!/bin/bash
# Round 4, visit O: pipelined llsmrt feeds.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_l1.py tests/test_gpu_rt.py tests/test_gpu_parity.py tests/test_c_host.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6
for wl in rt64 rt64pbp; do for p in 0 1; do
  echo "-- $wl pipeline $p"
  timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --rt-pipeline $p 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value']/1e6,3), 'M frames/s', round(d['ms_per_hop']*1e3,1), 'us per hop, max pull', round(d['max_pull_ms']*1e3,1), 'us', d['pipelined_feeds'])"
done; done | tee gpurun_out/r04_o_rt_pipeline.txt
