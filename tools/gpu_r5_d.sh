#!/bin/bash
# Round 5, visit D: where the smoothed-PSD tails come from (correctly rounded log / sqrt in the spectrogram envelope; the
# Kalman recursions in float64) on the worst seeds of the soaks; the object path with pooled outputs.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
SEEDS="25591 1833 2790 25083 30457 12817 5242 39118 29819 36062 2769 7244"
for lib in base SPGM_PRECISE_LOG_1 KAL_F64_1+KAL_WPE_2; do
  echo "== psd tails: $lib =="
  if [ $lib = base ]; then unset LLSM_AMD_LIB; else export LLSM_AMD_LIB=$PWD/exp_build/lib_$lib.so; fi
  timeout 600 python tools/fuzz_one.py --json $SEEDS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    print(d['seed'], {k: float('%.3g' % d[k]) for k in ('psd_db_max', 'psdres_db_max', 'psdraw_db_max_above_m20db', 'edc_rel_max')})
"
done 2>&1 | tee gpurun_out/r05_d_psd_tails_experiments.txt
echo "== kbench base / KAL_F64 =="
unset LLSM_AMD_LIB
timeout 300 python tools/kbench.py --utts 1024 --steps 5 --ablate KAL_F64=1,KAL_WPE=2 SPGM_PRECISE_LOG=1 2>&1 | tail -3 | cut -c1-330
echo "== object path =="
for cfg in "8 32" "8 64" "16 32" "16 64"; do
  set -- $cfg
  timeout 300 python tools/bench_chunk_api.py --workers $1 --block $2 --reps 4 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_d_chunk_api_w$1_b$2.json | cut -c100-700
done
timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 4 --batch-delete 0 2>/dev/null | tee gpurun_out/r05_d_chunk_api_w8_b32_perchunk.json | cut -c100-700
LLSM_TIMING=1 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 2 --batch-delete 1 2>&1 | grep -E "^\[analyze_block" | tail -12 | cut -c1-300 | tee gpurun_out/r05_d_chunk_api_analysis_phases.txt
LLSM_TIMING=1 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 2 --batch-delete 1 2>&1 | grep -E "^\[synthesize_block" | tail -8 | cut -c1-300 | tee gpurun_out/r05_d_chunk_api_synthesis_phases.txt
echo "== pytest c_host + round2 (pooled outputs, fan-out) =="
timeout 900 python -m pytest tests/test_c_host.py tests/test_gpu_round2.py tests/test_gpu_full.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-500 | head
