#!/bin/bash
# A/B of the headline bench: experiment option ctc_float (float-only log_add in the CTC recursion) off / on, interleaved
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/ab_ctcfloat"; mkdir -p "$OUT"
for r in 1 2 3; do for v in 0 1; do
  CLSTM_DEBUG="ctc_float=$v" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > "$OUT/b_${v}_$r.json" 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/b_${v}_$r.json")); print("ctc_float $v run $r:", d["value"], d["ms_per_step"], {k:round(x["ms_per_step"],4) for k,x in d["kernels"].items()})
PY
done; done
