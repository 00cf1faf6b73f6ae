#!/bin/bash
# Round 4, visit H: float32 in-lane recursion of k_filtfilt (experiment build): time and parity; llsmrt pack anatomy.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== kbench: IIR variants =="
timeout 600 python tools/kbench.py --utts 1024 --steps 5 --ablate IIR_F32_LOCAL=1 IIR_F32_LOCAL=1,IIR_SEG=24 2>&1 | cut -c1-330 | tee gpurun_out/r04_h_kbench.txt
echo "== parity with the float32 in-lane recursion =="
LLSM_AMD_LIB=$PWD/exp_build/lib_IIR_F32_LOCAL_1.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_full.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15
python - <<'PY'
import json, glob
worst = {}
for f in glob.glob("gpurun_out/parity_config_*.json"):
    d = json.load(open(f))
    for k in ("edc_rel_max", "eenv_ampl_rel_max", "eenv_ampl_abs_over_max"):
        if k in d: worst[k] = max(worst.get(k, 0), d[k])
print("worst over the configuration reports (float32 in-lane recursion):", worst)
PY
echo "== llsmrt pack anatomy =="
LLSM_TIMING=1 timeout 300 python bench.py --workload rt64pbp --steps 3 --warmup 1 2>gpurun_out/rt_timing.err | cut -c1-160
grep "llsmrt" gpurun_out/rt_timing.err | tail -2
