#!/bin/bash
# Round 5, visit B: k_excite_env4 against the per-sample kernel (tests + per-kernel times), level-resolved PSD errors on the
# seeds that exceeded the first restated contract and over 3000 fresh seeds.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest parity subset =="
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rt.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|FAILED|Error" | cut -c1-600 | head -20
echo "== kbench: per-sample excitation kernel, then by template position =="
LLSM_GPU_EXCITE4=0 timeout 300 python tools/kbench.py --utts 1024 --steps 5 2>&1 | tail -1 | cut -c1-900
LLSM_GPU_EXCITE4=1 timeout 300 python tools/kbench.py --utts 1024 --steps 5 2>&1 | tail -1 | cut -c1-900
echo "== level-resolved PSD errors of the seeds over the first restated contract =="
timeout 600 python tools/fuzz_one.py --json 1833 2240 2675 2769 2790 5242 7244 8619 2291 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    print(d['seed'], d['fs'], {k: float('%.3g' % v) for k, v in d.items() if k.startswith(('psd_db_max', 'psdres_db_max', 'psdraw', 'psd_pow'))})
" | tee gpurun_out/r05_b_psd_levels_seeds.txt
echo "== soak layer0 10000..12999 =="
( time SOAK_ONLY=layer0 timeout 900 python tools/fuzz_soak.py 10000 3000 ) 2>&1 | grep -E "^soak: 3000|^FAIL|^MARGINAL|^WORST \{|^real" | cut -c1-1800 | tee gpurun_out/r05_b_soak_layer0.txt
