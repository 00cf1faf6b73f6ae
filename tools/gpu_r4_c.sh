#!/bin/bash
# Round 4, visit C: transforms beyond 8192 points (global-scratch kernels), unit sizes of the table kernel.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== new tests =="
timeout 900 python -m pytest "tests/test_gpu_parity.py::test_hmpp_below_the_lds_transform" "tests/test_gpu_parity.py::test_unsupported_configurations_fail_loudly" "tests/test_gpu_parity.py::test_hmpp_low_f0_uses_the_8192_point_transform" "tests/test_gpu_configs.py::test_config_matrix_parity" -m gpu -q -p no:cacheprovider 2>&1 | tail -25
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | tail -8
echo "== kbench =="
for v in "" "LLSM_GPU_SIN_UNIT=25"; do
  echo "-- ${v:-default}"
  env $v timeout 300 python tools/kbench.py --utts 1024 --steps 5 2>&1 | tail -1 | cut -c1-420
done | tee gpurun_out/r04_c_kbench.txt
