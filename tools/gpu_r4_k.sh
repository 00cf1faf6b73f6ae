#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_l1.py tests/test_gpu_parity.py tests/test_gpu_frameapi.py tests/test_gpu_synth_tables.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8
python - <<'PY'
import json
for f in ("analysis_hmpp", "analysis_hmpp_f0_30", "analysis_hmpp_f0_15"):
    try:
        d = json.load(open(f"gpurun_out/parity_{f}.json"))
    except Exception as e:
        print(f, e); continue
    d = d if "phse_max_rad" in d else d
    for k, v in (d.items() if "phse_max_rad" not in d else [("", d)]):
        print(f, k, {q: v[q] for q in v if q.startswith("phse")})
PY
