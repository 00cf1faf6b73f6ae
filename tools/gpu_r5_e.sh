#!/bin/bash
# Round 5, visit E: the whole gpu suite under the two-yardstick contract (107 regression seeds), 10 000 fresh layer-0 seeds,
# the other sweeps of the soak, the object path over workers x block, the default bench (new e2e block, llsmrt capacity sweep,
# feed_many legs).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest -m gpu =="
( time timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tee gpurun_out/r05_e_pytest_gpu.log | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-700 | head -30 ) 2>&1 | grep -v "^$\|user\|sys"
echo "== soak layer0 40000..49999 =="
( time SOAK_ONLY=layer0 timeout 1200 python tools/fuzz_soak.py 40000 10000 ) 2>&1 | grep -E "^soak: 10000|^FAIL|^WORST \{|^real" | cut -c1-2500 | tee gpurun_out/r05_e_soak_layer0.txt
echo "== soak: the other sweeps, seeds 40000 ... =="
( time SOAK_ONLY=l1rt,hmpp,alt,coder timeout 1500 python tools/fuzz_soak.py 40000 1500 ) 2>&1 | grep -E "^soak|^FAIL|^WORST|^real" | cut -c1-900 | tee gpurun_out/r05_e_soak_others.txt
echo "== object path =="
for cfg in "8 32" "12 32" "12 16" "8 16"; do
  set -- $cfg
  timeout 300 python tools/bench_chunk_api.py --workers $1 --block $2 --reps 4 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_e_chunk_api_w$1_b$2.json | cut -c100-640
done
timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 4 --batch-delete 0 2>/dev/null | tee gpurun_out/r05_e_chunk_api_w8_b32_perchunk.json | cut -c100-640
echo "== bench default =="
timeout 900 python bench.py 2>gpurun_out/r05_e_bench_default.err | tee gpurun_out/r05_e_bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', {k: d['roofline'][k] for k in ('kernel','bound','frac','achieved')})
e=d['value_e2e']; print('e2e', e['value'], e['ms_per_step_min_median_max'], e['pcie_frac'], e['parts'], e['directions_alone'])
o=d.get('other_workloads',{})
print({k:(round(v.get('value',0)), v.get('ms_per_step')) for k,v in o.items() if 'value' in v})
print(json.dumps(o.get('rt_capacity',{}).get('streams',{}), indent=0)[:2500])
print('cpu', d.get('cpu_baseline',{}).get('value'), 'legs wall', d.get('other_workloads_wall_s'))
"
tail -3 gpurun_out/r05_e_bench_default.err
