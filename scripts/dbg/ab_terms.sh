#!/bin/bash
# A/B of the headline bench: operand-exact 3-term split (default) vs the 2-term split, interleaved
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/ab_terms"; mkdir -p "$OUT"
for r in 1 2 3; do
  for v in 3 2; do
    CLSTM_DEBUG="split_terms=$v" python bench.py --no-cpu-baseline --no-secondary > "$OUT/b_${v}_$r.json" 2> "$OUT/b_${v}_$r.err"
    python - <<PY
import json
d=json.load(open("$OUT/b_${v}_$r.json")); print("terms $v run $r:", d["value"], d["ms_per_step"], {k:round(x["ms_per_step"],4) for k,x in d["kernels"].items()})
PY
  done
done
