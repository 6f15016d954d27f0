#!/bin/bash
# north_star's literal design for narrow layers -- R.h batched across the minibatch on the MFMA (lock-step path,
# CLSTM_FORCE_WIDE=1) -- against the per-line register-resident recurrence, at saturating minibatch sizes.
# Usage: gpurun --timeout 900 -- 'bash scripts/gpu_narrow_lockstep.sh r04'
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
B() { timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --profile-steps 5 "$@"; }
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], d["value"], "lines/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels"].items() if "lstm" in k})
except Exception as e:
    print(sys.argv[1].split("/")[-1], "FAILED", e)
PY
}
for MB in 64 256 1024; do
  B --minibatch $MB > "$OUT/narrow_perline_mb$MB.json" 2> "$OUT/narrow_perline_mb$MB.err"; show "$OUT/narrow_perline_mb$MB.json"
  CLSTM_FORCE_WIDE=1 B --minibatch $MB > "$OUT/narrow_lockstep_f32_mb$MB.json" 2> "$OUT/narrow_lockstep_f32_mb$MB.err"; show "$OUT/narrow_lockstep_f32_mb$MB.json"
  CLSTM_FORCE_WIDE=1 B --minibatch $MB --bf16 > "$OUT/narrow_lockstep_bf16_mb$MB.json" 2> "$OUT/narrow_lockstep_bf16_mb$MB.err"; show "$OUT/narrow_lockstep_bf16_mb$MB.json"
done
