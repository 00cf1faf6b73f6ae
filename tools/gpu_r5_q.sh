#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
for m in 2 1; do
echo "== pytest round2 / c_host / full  (LLSM_PACKED_FRAMES=$m) =="
LLSM_PACKED_FRAMES=$m timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_c_host.py tests/test_gpu_full.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|FAILED|Error|Fatal" | cut -c1-400 | head -6
done
echo "== object path: mode x workers x block =="
for m in 2 1 0; do for cfg in "8 32" "6 64" "4 128"; do set -- $cfg
  LLSM_PACKED_FRAMES=$m timeout 300 python tools/bench_chunk_api.py --workers $1 --block $2 --reps 5 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_q_chunk_api_mode${m}_w$1_b$2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mode $m w',d['workers'],'b',d['block'],'ana %.1f syn %.1f del %.1f  -> %.2f M (%.2f M excl. delete)'%(d['analyze_ms'],d['synthesize_ms'],d['delete_objects_ms'],d['value']/1e6,d['value_excluding_delete']/1e6))"; done; done
LLSM_TIMING=1 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 3 --batch-delete 1 2>&1 | grep -E "^\[(analyze|synthesize)_block" | tail -6 | cut -c1-330
