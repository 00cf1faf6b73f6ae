#!/bin/bash
# Round 4, visit R: process variance of the Kalman smoother as a mean squared deviation (new) against m2/3 - m1^2/9 (old,
# exp_build/lib_kalnaive.so) on the soak seeds with PSD tails, then the parity suites on the new build.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
SEEDS="5242 5000 5001 5002 5003 5004 5005 5006 5007 5008 5009 5010 5011"
{
echo "== old (m2/3 - m1^2/9 in float32)"
LLSM_AMD_LIB=$PWD/exp_build/lib_kalnaive.so timeout 600 python tools/fuzz_one.py $SEEDS 2>&1 | grep -v Warning
echo "== new (mean squared deviation)"
timeout 600 python tools/fuzz_one.py $SEEDS 2>&1 | grep -v Warning
} | tee gpurun_out/r04_r_kalman_variance.txt
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6
timeout 300 python bench.py --steps 5 --warmup 2 --no-other 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v for k,v in d.get('kernel_ms',{}).items() if 'kalman' in k})"
