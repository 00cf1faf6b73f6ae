#!/bin/bash
# Round 5, visit C: k_excite_env4 with explicit FMAs (tests + per-kernel times), the object path with its phase times and the
# new deletion (per chunk and batched), 20 000 fresh layer-0 seeds under the joint-yardstick contract.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest subset =="
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_regressions.py tests/test_c_host.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-700 | head -20
echo "== kbench =="
LLSM_GPU_EXCITE4=0 timeout 300 python tools/kbench.py --utts 1024 --steps 5 2>&1 | tail -1 | cut -c1-400
LLSM_GPU_EXCITE4=1 timeout 300 python tools/kbench.py --utts 1024 --steps 5 2>&1 | tail -1 | cut -c1-400
echo "== object path =="
for bd in 0 1; do
  timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 4 --batch-delete $bd 2>gpurun_out/r05_c_chunk_api_$bd.err | tee gpurun_out/r05_c_chunk_api_w8_b32_bd$bd.json
done
LLSM_SLAB_POOL_MB=64 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 4 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_c_chunk_api_w8_b32_pool64.json
timeout 300 python tools/bench_chunk_api.py --workers 16 --block 32 --reps 4 --batch-delete 1 2>/dev/null | tee gpurun_out/r05_c_chunk_api_w16_b32.json
LLSM_TIMING=1 timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 1 --batch-delete 1 2>&1 | grep -E "^\[analyze_block|^\[synthesize_block" | tail -24 | cut -c1-300 | tee gpurun_out/r05_c_chunk_api_phases.txt
echo "== soak layer0 20000..39999 =="
( time SOAK_ONLY=layer0 timeout 1500 python tools/fuzz_soak.py 20000 20000 ) 2>&1 | grep -E "^soak: 20000|^FAIL|^MARGINAL|^WORST \{|^real" | cut -c1-1800 | tee gpurun_out/r05_c_soak_layer0.txt
