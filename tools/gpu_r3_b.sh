#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== tile tests =="
timeout 600 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tee gpurun_out/r3b_pytest.log | tail -30
echo "== bench (tiles on) =="
timeout 300 python bench.py --no-cpu-baseline --no-e2e 2>gpurun_out/r3b_bench.err | tee gpurun_out/r3b_bench_tiles.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['kernels_ms_per_step'])"
echo "== bench (tiles off) =="
LLSM_GPU_F0_TILES=0 timeout 300 python bench.py --no-cpu-baseline --no-e2e 2>>gpurun_out/r3b_bench.err | tee gpurun_out/r3b_bench_notiles.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['kernels_ms_per_step'])"
echo "== mfma + valu =="
timeout 120 tools/ubench/mfma_valu | tee gpurun_out/r3b_mfma_valu.txt | tail -45
