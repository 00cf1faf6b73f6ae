#!/bin/bash
# Round-3 closing GPU visit: whole gpu suite, default bench, other workloads, rocprof kernel stats, HBM traffic, SQ counters.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
echo "== pytest -m gpu =="
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | grep -E "^E  |passed|failed|FAILED" | cut -c1-300 | head -20
echo "== bench default =="
timeout 600 python bench.py 2>gpurun_out/bench_default.err | tee gpurun_out/bench_default.json | cut -c1-400
for w in sweep rt64 rt64pbp l1; do
  echo "== bench $w =="
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>gpurun_out/bench_$w.err | tee gpurun_out/bench_$w.json | cut -c1-300
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_stats -o bench -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/gpurun_out/prof_bench.log 2>&1
python $REPO/tools/rocpd_summary.py $(find $REPO/gpurun_out/prof_stats -name "*.db" | head -1) | grep -E "^kernel|k_" > $REPO/gpurun_out/kernel_stats.txt
head -16 $REPO/gpurun_out/kernel_stats.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $REPO/gpurun_out/prof_pmc_fetch -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/gpurun_out/prof_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $REPO/gpurun_out/prof_pmc_write -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/gpurun_out/prof_pmc_write.log 2>&1
python $REPO/tools/rocpd_traffic.py $(find $REPO/gpurun_out/prof_pmc_fetch -name "*.db" | head -1) $(find $REPO/gpurun_out/prof_pmc_write -name "*.db" | head -1) > $REPO/gpurun_out/prof_traffic.json
rm -rf $REPO/gpurun_out/prof_pmc_fetch $REPO/gpurun_out/prof_pmc_write
find $REPO/gpurun_out/prof_stats -name "*.db" -delete
timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_rt -o rt -- python $REPO/bench.py --workload rt64 --steps 2 --warmup 1 > $REPO/gpurun_out/prof_rt.log 2>&1
python $REPO/tools/rocpd_summary.py $(find $REPO/gpurun_out/prof_rt -name "*.db" | head -1) | grep -E "^kernel|k_|copy" > $REPO/gpurun_out/rt64_kernel_stats.txt
find $REPO/gpurun_out/prof_rt -name "*.db" -delete
cd $REPO
bash tools/gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1
grep -E "k_harm_speech_tile|k_filtfilt" gpurun_out/pmc_2.txt | head -12
