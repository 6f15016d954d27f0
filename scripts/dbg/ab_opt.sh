#!/bin/bash
# A/B of the headline bench for one CLSTM_DEBUG experiment option: ab_opt.sh <option> [values...] (default 0 1), interleaved, three rounds
OPT=$1; shift; VALS=${@:-0 1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/ab_$OPT"; mkdir -p "$OUT"
for r in 1 2 3; do for v in $VALS; do
  CLSTM_DEBUG="$OPT=$v" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} > "$OUT/b_${v}_$r.json" 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/b_${v}_$r.json")); print("$OPT $v run $r:", d["value"], d["ms_per_step"], {k:round(x["ms_per_step"],4) for k,x in d["kernels"].items()})
PY
done; done
