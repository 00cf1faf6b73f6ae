#!/bin/bash
# diagnose: standalone `--workload sweep` ran k_harm_speech_rest at 11.6 ms (0.02 ms as a leg of the default run)
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
show() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels_ms_per_step']; print(round(d['ms_per_step'],3), {n: round(k[n],3) for n in ('k_harm_speech_tile','k_harm_speech_rest') if n in k})"; }
echo "-- sweep 5/2"; timeout 300 python bench.py --workload sweep --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | show
echo "-- sweep 3/1"; timeout 300 python bench.py --workload sweep --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | show
echo "-- sweep 5/2 utts 512"; timeout 300 python bench.py --workload sweep --utts 512 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | show
echo "-- fixed120 5/2"; timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-other 2>/dev/null | show
echo "-- sweep with e2e"; timeout 300 python bench.py --workload sweep --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | show
