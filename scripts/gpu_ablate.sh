#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/abl"; mkdir -p "$OUT"
for D in "$@"; do
  CLSTM_DBG=$D timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > "$OUT/b$D.json" 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/b$D.json"))
print("dbg=$D", {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("lstm")})
PY
done
