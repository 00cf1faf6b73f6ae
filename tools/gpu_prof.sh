#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench (-> gpurun_out/prof_*), PMC pass separately.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_stats -o bench -- python $REPO/bench.py --utts ${UTTS:-1024} --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof_bench.log 2>&1
tail -2 $REPO/gpurun_out/prof_bench.log
find $REPO/gpurun_out/prof_stats -name "*kernel_stats*" | head -3
f=$(find $REPO/gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f"
if [ -n "$PMC" ]; then
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $REPO/gpurun_out/prof_pmc_fetch -o bench -- python $REPO/bench.py --utts ${UTTS:-1024} --steps 2 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof_pmc_fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $REPO/gpurun_out/prof_pmc_write -o bench -- python $REPO/bench.py --utts ${UTTS:-1024} --steps 2 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof_pmc_write.log 2>&1
  python $REPO/tools/rocpd_traffic.py $REPO/gpurun_out/prof_pmc_fetch/bench_results.db $REPO/gpurun_out/prof_pmc_write/bench_results.db > $REPO/gpurun_out/prof_traffic.json
  head -c 600 $REPO/gpurun_out/prof_traffic.json
fi
