#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest round2 / c_host / full =="
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_c_host.py tests/test_gpu_full.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-500 | head
echo "== object path: workers x block =="
for cfg in "8 32" "4 128" "6 64"; do set -- $cfg
  timeout 300 python tools/bench_chunk_api.py --workers $1 --block $2 --reps 5 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_m_chunk_api_packed_w$1_b$2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('w',d['workers'],'b',d['block'],'ana %.1f syn %.1f del %.1f  -> %.2f M (%.2f M excl. delete)'%(d['analyze_ms'],d['synthesize_ms'],d['delete_objects_ms'],d['value']/1e6,d['value_excluding_delete']/1e6))"; done
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 5 --batch-delete 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=8: w',d['workers'],'b',d['block'],'ana %.1f syn %.1f del %.1f  -> %.2f M'%(d['analyze_ms'],d['synthesize_ms'],d['delete_objects_ms'],d['value']/1e6))"
