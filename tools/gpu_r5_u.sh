#!/bin/bash
# Round 5, visit U: larger fresh soaks of the two sweeps whose last run found a case each (HMPP, layer 1 / llsmrt), under the
# contract as it stands after them.  The per-case report files are removed before gpurun merges gpurun_out/ (2 000-file limit).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== soak HMPP + F0 refinement, 90000 .. (3000 cases each) =="
( time SOAK_ONLY=hmpp timeout 900 python tools/fuzz_soak.py 90000 15000 ) > gpurun_out/r05_zz3_soak_hmpp.txt 2>&1
grep -E "^soak: [0-9]+ HMPP|^FAIL|^real" gpurun_out/r05_zz3_soak_hmpp.txt | cut -c1-600 | head -12
echo "== soak layer 1 / llsmrt / PbP, 90000 .. (1000 cases each) =="
( time SOAK_ONLY=l1rt timeout 1200 python tools/fuzz_soak.py 90000 5000 ) > gpurun_out/r05_zz3_soak_l1rt.txt 2>&1
grep -E "^soak: [0-9]+ layer-1|^FAIL|^real" gpurun_out/r05_zz3_soak_l1rt.txt | cut -c1-600 | head -12
find gpurun_out -name "parity_*.json" -delete
