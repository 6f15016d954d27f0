#!/bin/bash
# same-box A/B of the headline bench between library builds, interleaved: VARS="head v4 -" names clstm_amd/lib/libclstm_hip_<name>.so
# (CLSTM_HIP_VARIANT; "-" = the working tree's libclstm_hip.so; "head" = built from HEAD's sources into libclstm_hip_head.so)
TAG=${1:-abhead}; ROUNDS=${2:-2}; ARGS=${3:-}
VARS=${VARS:-"head -"}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
for r in $(seq 1 $ROUNDS); do
for v in $VARS; do
vv=$v; [ "$v" = "-" ] && vv=""
CLSTM_HIP_VARIANT=$vv timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $ARGS > "$OUT/bench_${v}_$r.json" 2> "$OUT/bench.err"
python - "$OUT/bench_${v}_$r.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("[%5s]" % sys.argv[2], d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
done
