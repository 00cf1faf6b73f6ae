#!/bin/bash
# Round 4, visit Y: what a per-F0 twiddle table would buy k_harm_speech_tile (timing builds, garbage results:
# tools/kbench_experiments.h HT_TABLE_EXPERIMENT) at 4 and 2 wavefronts per SIMD.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 900 python tools/kbench.py --utts 1024 --steps 4 --ablate HT_TABLE_EXPERIMENT=1 HT_TABLE_EXPERIMENT=1,HT_WPE=2 HT_TABLE_EXPERIMENT=1,HT_WPE=2,HT_CHUNK=4 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    n,_,j=l.partition(' ')
    try: d=json.loads(j)
    except Exception: print(l.strip()[:300]); continue
    print(n, {k:v for k,v in d.items() if k in ('k_harm_speech_tile','k_spgm_env_wf','k_filtfilt')})" | tee gpurun_out/r04_y_tile_table_experiment.txt
