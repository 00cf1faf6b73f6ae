#!/bin/bash
# Round 4, visit B: k_synth_ola4 with pinned products (bit identity) + row / x prefetch; sub-batch experiment.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== new tests =="
timeout 600 python -m pytest tests/test_gpu_synth_tables.py tests/test_c_host.py "tests/test_gpu_parity.py::test_overlap_add_units_do_not_change_results" -m gpu -q -p no:cacheprovider 2>&1 | tail -15
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | tail -8
echo "== kbench: tables on / off / unit sizes =="
for v in "" "LLSM_GPU_SYNTH_TABLES=0" "LLSM_GPU_SIN_UNIT=50" "LLSM_GPU_SIN_UNIT=40"; do
  echo "-- ${v:-default}"
  env $v timeout 300 python tools/kbench.py --utts 1024 --steps 5 2>&1 | tail -1 | cut -c1-420
done | tee gpurun_out/r04_b_kbench.txt
echo "== sub-batches =="
timeout 600 python tools/ab_subbatch.py 1024 10 2>&1 | tee gpurun_out/r04_b_subbatch.txt | cut -c1-500
echo "== bench default =="
timeout 900 python bench.py 2>gpurun_out/bench_default.err | tee gpurun_out/bench_default.json | cut -c1-300
tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print({k: (v.get("value"), v.get("ms_per_step")) for k, v in d.get("other_workloads", {}).items()}, d.get("other_workloads_wall_s"))
print(d["kernels_ms_per_step"])
PY
