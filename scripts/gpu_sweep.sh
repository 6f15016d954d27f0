#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/sweep"; mkdir -p "$OUT"
for V in "$@"; do
  env $SWEEP_VAR=$V timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > "$OUT/b_$V.json" 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/b_$V.json"))
print("$SWEEP_VAR=$V", d["value"], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if "dw" in k})
PY
done
