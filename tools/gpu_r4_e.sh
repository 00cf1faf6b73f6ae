#!/bin/bash
# Round 4, visit E: register budget of k_pbp_pulse (experiment builds under exp_build/).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
for lib in "" exp_build/lib_PBP_WPE_4.so exp_build/lib_PBP_WPE_5.so "exp_build/lib_PBP_WPE_4+PBP_NT_64.so"; do
  echo "-- l1 bench ${lib:-product}"
  if [ -n "$lib" ]; then export LLSM_AMD_LIB=$PWD/$lib; else unset LLSM_AMD_LIB; fi
  timeout 300 python bench.py --workload l1 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), round(d['value']/1e6,2), {k: round(v,3) for k,v in list(d['kernels_ms_per_step'].items())[:4]})"
done | tee gpurun_out/r04_e_pbp.txt
