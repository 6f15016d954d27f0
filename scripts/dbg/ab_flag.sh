#!/bin/bash
# A/B of the headline bench with / without one bench.py flag, interleaved, three rounds: ab_flag.sh --no-declare-next
FLAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/abf"; mkdir -p "$OUT"
for r in 1 2 3; do for v in with without; do
  if [ $v = with ]; then F="$FLAG"; else F=""; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} $F > "$OUT/b_${v}_$r.json" 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/b_${v}_$r.json")); print("$v $FLAG run $r:", d["value"], d["ms_per_step"], {k:round(x["ms_per_step"],4) for k,x in d["kernels"].items()})
PY
done; done
