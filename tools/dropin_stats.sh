#!/bin/bash
# Kernel trace of single drop-in calls (tools/bench_dropin.py): which launches make up one llsm_analyze / llsm_synthesize
TAG=${1:-dropin}; R=$PWD; export PYTHONPATH=$PWD
python tools/bench_dropin.py 2>/dev/null | cut -c1-400 | tee gpurun_out/${TAG}_dropin_latency.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_dd -o dp -- python $R/tools/bench_dropin.py > /dev/null 2>&1)
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_dd -name "*.db" | head -1) | grep -E "^kernel|k_|copy|Copy" > gpurun_out/${TAG}_dropin_kernel_stats.txt
rm -rf gpurun_out/${TAG}_dd; head -40 gpurun_out/${TAG}_dropin_kernel_stats.txt
