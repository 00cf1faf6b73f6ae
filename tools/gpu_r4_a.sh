#!/bin/bash
# Round 4, visit A: the table kernel of the harmonic resynthesis (k_synth_ola4) -- parity first, then A/B timing,
# then the default bench with the other_workloads legs.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== new tests =="
timeout 600 python -m pytest tests/test_gpu_synth_tables.py tests/test_c_host.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | tail -8
echo "== kbench: tables on / off / unit sizes =="
for v in "" "LLSM_GPU_SYNTH_TABLES=0" "LLSM_GPU_SIN_UNIT=50" "LLSM_GPU_SIN_UNIT=34" "LLSM_GPU_SIN_UNIT=100"; do
  echo "-- ${v:-default}"
  env $v timeout 300 python tools/kbench.py --utts 1024 --steps 5 2>&1 | tail -1 | cut -c1-420
done | tee gpurun_out/r04_a_kbench.txt
echo "== bench default =="
timeout 900 python bench.py 2>gpurun_out/bench_default.err | tee gpurun_out/bench_default.json | cut -c1-600
tail -3 gpurun_out/bench_default.err
