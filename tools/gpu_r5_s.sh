#!/bin/bash
# Round 5, visit S: the object path against the number of hardware queues the HIP runtime gives the process
# (GPU_MAX_HW_QUEUES, default 4: eight workers' streams share four queues), and the layer-1 line three times.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
for q in 4 8 16; do
  for cfg in "2 8 32" "2 12 32" "1 6 64" "2 8 64"; do set -- $cfg
    echo -n "queues $q mode $1 w $2 b $3: "
    GPU_MAX_HW_QUEUES=$q LLSM_PACKED_FRAMES=$1 timeout 300 python tools/bench_chunk_api.py --workers $2 --block $3 --reps 5 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_s_chunk_api_q${q}_mode$1_w$2_b$3.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ana %.1f syn %.1f del %.1f -> %.2f M' % (d['analyze_ms'], d['synthesize_ms'], d['delete_objects_ms'], d['value'] / 1e6))"
  done
done
for i in 1 2 3; do
  timeout 300 python bench.py --workload l1 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tee gpurun_out/r05_s_bench_l1_$i.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('l1', round(d['value'] / 1e6, 2), 'M', round(d['ms_per_step'], 3), 'ms', d.get('host_ms_of_each_step'))"
done
