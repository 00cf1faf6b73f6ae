#!/bin/bash
# Round 5, visit X: workers x block of the object path's default mode once more at the HEAD (the link is busy 80 % of a call pair).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
for cfg in "8 32" "6 32" "8 16" "8 24" "10 32" "10 24" "12 16" "16 16" "8 32"; do set -- $cfg
  echo -n "w $1 b $2: "
  timeout 300 python tools/bench_chunk_api.py --workers $1 --block $2 --reps 5 --batch-delete 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ana %.1f syn %.1f del %.1f -> %.2f M' % (d['analyze_ms'], d['synthesize_ms'], d['delete_objects_ms'], d['value'] / 1e6))"
done
