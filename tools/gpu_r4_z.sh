#!/bin/bash
# Round 4, visit Z: k_harm_speech_tile with running operand cursors (product) against the library before (index arithmetic per
# load), with two operand sets taking turns (HT_PINGPONG), and both at 2 wavefronts / SIMD; then the tile parity tests.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
for r in 1 2; do for v in before product HT_PINGPONG_1 HT_PINGPONG_1+HT_WPE_2 HT_WPE_2; do
  lib=$PWD/exp_build/lib_$v.so; [ $v = product ] && lib=$PWD/libllsm2_amd/libllsm2_amd.so
  LLSM_AMD_LIB=$lib timeout 200 python tools/kbench.py --child --utts 1024 --steps 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$v', d.get('k_harm_speech_tile'), d.get('k_spgm_env_wf'))"
done; done | tee gpurun_out/r04_z_tile_cursors.txt
timeout 600 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
