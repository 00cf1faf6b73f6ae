#!/bin/bash
# Round 5, visit R (after the final profile visit): the HMPP regression seeds under the contract with the float32 yardstick
# taken as "float32 oracle + 8(d)" (tests/gpu_common.py), fresh HMPP and layer-0 soaks at this HEAD, and the layer-1 /
# headline workloads with the process inside / outside the device's NUMA node.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest regressions (hmpp) =="
timeout 900 python -m pytest "tests/test_gpu_regressions.py::test_marginal_hmpp_seeds" -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-600 | head -10
echo "== soak HMPP 60000 .. (1000 cases + 1000 F0-refinement cases) =="
( time SOAK_ONLY=hmpp timeout 900 python tools/fuzz_soak.py 60000 5000 ) > gpurun_out/r05_zz_soak_hmpp.txt 2>&1
grep -E "^soak:|^FAIL|^real|^WORST" gpurun_out/r05_zz_soak_hmpp.txt | cut -c1-500 | head -12
echo "== soak layer 0, 60000 .. 79999 =="
( time SOAK_ONLY=layer0 timeout 1200 python tools/fuzz_soak.py 60000 20000 ) > gpurun_out/r05_zz_soak_layer0.txt 2>&1
grep -E "^soak:|^FAIL|^real|^WORST" gpurun_out/r05_zz_soak_layer0.txt | cut -c1-600 | head -12
echo "== NUMA: where the process runs =="
NODE=$(python -c "import libllsm2_amd as l; print(l.load().llsm_gpu_device_numa_node(0))" 2>/dev/null | tail -1)
echo "device node $NODE; nodes: $(ls -d /sys/devices/system/node/node* | wc -l); allowed: $(grep Cpus_allowed_list /proc/self/status)"
for n in $(ls -d /sys/devices/system/node/node* | sed 's/.*node//'); do
  CPUS=$(cat /sys/devices/system/node/node$n/cpulist)
  for wl in l1 fixed120; do
    extra=""; [ $wl = fixed120 ] && extra="--no-other"
    echo "-- node $n ($CPUS) workload $wl"
    timeout 300 taskset -c $CPUS python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --no-e2e $extra 2>/dev/null | tee gpurun_out/r05_zz_numa_node${n}_$wl.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'] / 1e6, 2), 'M frames/s', round(d['ms_per_step'], 3), 'ms', d.get('host_ms_per_step'))"
  done
done
echo "-- unbound, workload l1"
timeout 300 python bench.py --workload l1 --steps 8 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tee gpurun_out/r05_zz_numa_unbound_l1.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'] / 1e6, 2), 'M frames/s', round(d['ms_per_step'], 3), 'ms', d.get('host_ms_per_step'))"
