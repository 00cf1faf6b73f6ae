#!/bin/bash
# Round 4, visit V: streaming stores in llsm_chunk_to_flat (LLSM_FLAT_NT=1, the default) against ordinary stores (=0), through
# llsm_analyze_batch + llsm_synthesize_batch with 1 and 8 workers; then the host-side tests that go through the flatten.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
for w in 8 1; do for nt in 0 1 0 1; do
  LLSM_FLAT_NT=$nt timeout 300 python tools/bench_chunk_api.py --utts 1024 --workers $w --block 32 --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('workers $w nt $nt', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k not in ('metric','config')})"
done; done | tee gpurun_out/r04_v_flat_nt.txt
timeout 600 python -m pytest tests/test_c_host.py tests/test_gpu_round2.py tests/test_gpu_full.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
