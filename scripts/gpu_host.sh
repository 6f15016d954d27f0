#!/bin/bash
# host-side cost of issuing a step vs the GPU step time
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out/host
for i in 1 2; do
python bench.py --no-cpu-baseline --profile-steps 0 --steps 24 --warmup 8 > gpurun_out/host/b$i.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/host/b$i.json")); print("steps 24: ms/step", d["ms_per_step"], "host enqueue ms/step", d["host_enqueue_ms_per_step"], "value", d["value"])
PY
done
