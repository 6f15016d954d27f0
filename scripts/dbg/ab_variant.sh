#!/bin/bash
# A/B of two BUILDS of the library (make variant VARIANT=name EXTRA=...): ab_variant.sh <variant> ; BENCH_ARGS as ab_opt.sh
VAR=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/abv_$VAR"; mkdir -p "$OUT"
for r in 1 2 3; do for v in default $VAR; do
  if [ $v = default ]; then unset CLSTM_HIP_VARIANT; else export CLSTM_HIP_VARIANT=$v; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} > "$OUT/b_${v}_$r.json" 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/b_${v}_$r.json")); print("build $v run $r:", d["value"], d["ms_per_step"], {k:round(x["ms_per_step"],4) for k,x in d["kernels"].items()})
PY
done; done
