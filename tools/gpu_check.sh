#!/bin/bash
# One GPU-box visit: parity tests, smoke, short bench.  Logs -> gpurun_out/.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
rocminfo 2>/dev/null | grep -m1 gfx || true
echo "== pytest -m gpu ==" 
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | tail -60
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log | tail -5
echo "== bench small =="
timeout 600 python bench.py --utts ${UTTS:-128} --steps 3 --warmup 1 2>&1 | tee gpurun_out/bench_small.log | tail -3
