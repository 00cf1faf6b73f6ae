#!/bin/bash
# Round-5 profile visit: whole gpu suite, default bench (with its other_workloads legs), the other workloads on their own,
# rocprofv3 kernel stats + HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) + three SQ counter sets for the headline
# workload, kernel stats + traffic for the layer-1 workload, kernel stats for llsmrt.  Everything lands in gpurun_out/;
# the summaries are copied to profiles/r05_<tag>_* afterwards.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
echo "== pytest -m gpu =="
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | grep -E "^E  |passed|failed|FAILED" | cut -c1-300 | head -20
echo "== bench default =="
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench_default.err | tee gpurun_out/bench_default.json | cut -c1-300
echo "== object path =="
timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 5 --batch-delete 1 2>/dev/null | tee gpurun_out/chunk_api_mode2_w8_b32.json | cut -c100-640
timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 5 --batch-delete 0 2>/dev/null | tee gpurun_out/chunk_api_mode2_w8_b32_perchunk.json | cut -c100-640
timeout 300 python tools/bench_dropin.py 2>/dev/null | tee gpurun_out/dropin_latency.json | cut -c1-400
for w in sweep rt64 rt64pbp l1; do
  echo "== bench $w =="
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>gpurun_out/bench_$w.err | tee gpurun_out/bench_$w.json | cut -c1-220
done
cd /tmp && export TMPDIR=/tmp
stats() {   # stats <outdir> <summary file> <bench args...>
  local od=$1 sm=$2; shift 2
  timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/$od -o bench -- python $REPO/bench.py "$@" > $REPO/gpurun_out/$od.log 2>&1
  python $REPO/tools/rocpd_summary.py $(find $REPO/gpurun_out/$od -name "*.db" | head -1) | grep -E "^kernel|k_|copy|Copy" > $REPO/gpurun_out/$sm
  rm -rf $REPO/gpurun_out/$od
}
traffic() { # traffic <summary file> <bench args...>
  local sm=$1; shift
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $REPO/gpurun_out/pf -o bench -- python $REPO/bench.py "$@" > $REPO/gpurun_out/pf.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $REPO/gpurun_out/pw -o bench -- python $REPO/bench.py "$@" > $REPO/gpurun_out/pw.log 2>&1
  python $REPO/tools/rocpd_traffic.py $(find $REPO/gpurun_out/pf -name "*.db" | head -1) $(find $REPO/gpurun_out/pw -name "*.db" | head -1) > $REPO/gpurun_out/$sm
  rm -rf $REPO/gpurun_out/pf $REPO/gpurun_out/pw
}
stats prof_stats kernel_stats.txt --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-other
head -16 $REPO/gpurun_out/kernel_stats.txt
traffic prof_traffic.json --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other
stats prof_l1 l1_kernel_stats.txt --workload l1 --steps 3 --warmup 1
head -12 $REPO/gpurun_out/l1_kernel_stats.txt
traffic l1_traffic.json --workload l1 --steps 2 --warmup 1
stats prof_rt rt64_kernel_stats.txt --workload rt64 --steps 2 --warmup 1
stats prof_rtp rt64pbp_kernel_stats.txt --workload rt64pbp --steps 2 --warmup 1
cd $REPO
bash tools/gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1
grep -E "k_synth_ola|k_harm_speech_tile" gpurun_out/pmc_2.txt | head -8
for i in 1 2 3; do mv gpurun_out/pmc_$i.txt gpurun_out/pmc_l0_$i.txt; done
WL=l1 bash tools/gpu_pmc.sh > gpurun_out/pmc_run_l1.log 2>&1
for i in 1 2 3; do mv gpurun_out/pmc_$i.txt gpurun_out/pmc_l1_$i.txt; done
grep -E "k_pbp_pulse" gpurun_out/pmc_l1_3.txt | head -4
