#!/bin/bash
# Round 5, visit A: the gpu suite under the restated contract, a first soak with the oracle worker pool (seeds 1000 ...:
# the ranges of the round-3 soaks, to re-find their marginal seeds by number), the default bench with the new e2e block.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
nproc; lscpu | grep -E "^CPU\(s\)|NUMA|Socket|Model name" | head -8
cat /sys/fs/cgroup/cpu.max 2>/dev/null
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tee gpurun_out/r05_a_pytest_gpu.log | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-600 | head -30
echo "== soak layer0 1000..3999 =="
( time SOAK_ONLY=layer0 timeout 1500 python tools/fuzz_soak.py 1000 3000 ) 2>&1 | grep -E "^soak|^FAIL|^MARGINAL|^WORST|^real" | cut -c1-700 | tee gpurun_out/r05_a_soak_layer0.txt
echo "== bench default =="
timeout 900 python bench.py 2>gpurun_out/r05_a_bench_default.err | tee gpurun_out/r05_a_bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
print(json.dumps(d['value_e2e'], indent=1)[:3500])
print({k:(v.get('value'), v.get('ms_per_step')) for k,v in d.get('other_workloads',{}).items()})
"
tail -5 gpurun_out/r05_a_bench_default.err
