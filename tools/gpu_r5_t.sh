#!/bin/bash
# Round 5, visit T: the whole gpu suite and the driver's bench command at the HEAD of the round (after the HMPP yardstick
# change), and every sweep of tools/fuzz_soak.py on fresh seeds (80000 ...).
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== pytest -m gpu =="
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tee gpurun_out/r05_zz2_pytest_gpu.log | grep -E "^E  |passed|failed|FAILED" | cut -c1-300 | head -20
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench default =="
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r05_zz2_bench_default.err | tee gpurun_out/r05_zz2_bench_default.json | cut -c1-300
echo "== soak, every sweep, 80000 .. =="
( time timeout 1500 python tools/fuzz_soak.py 80000 5000 ) > gpurun_out/r05_zz2_soak_all.txt 2>&1
grep -E "^soak:|^FAIL|^real|^WORST" gpurun_out/r05_zz2_soak_all.txt | cut -c1-400 | head -20
