#!/bin/bash
# Round 4, visit U: where k_l1_env_wf's time goes -- timing builds (results are garbage) without the lobe evaluation, the lifter,
# the two transforms (exp_build/lib_l1env_*.so, made from temporary #ifdefs that are not in the tree).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
for v in product nolobes nolobes_nolifter nolobes_nolifter_nofft nofft; do
  lib=$PWD/exp_build/lib_l1env_$v.so; [ $v = product ] && lib=$PWD/libllsm2_amd/libllsm2_amd.so
  LLSM_AMD_LIB=$lib timeout 300 python bench.py --workload l1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],2),'ms', {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if k.startswith('k_l1_env')})"
done | tee gpurun_out/r04_u_l1_env_ablation.txt
