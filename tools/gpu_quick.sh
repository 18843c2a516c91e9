#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -k "(golden and (w1 or w2)) or weight_draws" > gpurun_out/parity_draws.log 2>&1; echo "rc=$?"; tail -n 15 gpurun_out/parity_draws.log | cut -c1-300
python - <<'PY'
import json
r=json.load(open("gpurun_out/parity_report.json"))
for k,v in sorted(r.items()):
    if "w1" in k or "w2" in k or "draw" in k: print(k, {a:(round(b,6) if isinstance(b,float) else b) for a,b in v.items()})
PY
