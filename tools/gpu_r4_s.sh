#!/bin/bash
# Round 4, visit S: every metric of the product on fuzz seeds 5000 ... 5039 and 5242 as JSON lines, to set beside the float32
# build of the oracle on the same seeds (tools/oracle_f32_spread.py, CPU).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 900 python tools/fuzz_one.py --json $(seq 5000 5039) 5242 2>/dev/null | grep '^{' > gpurun_out/r04_s_product_metrics.jsonl
wc -l gpurun_out/r04_s_product_metrics.jsonl
