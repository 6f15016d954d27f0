#!/bin/bash
# lstm_xcd_bwd_bf16_c32 (32 cells per workgroup, two groups per XCD) against the 16-cell persistent backward kernel:
# GPU parity selection, bench configs[4] both ways, the 128- / 256-line shapes, per-phase stamps of both kernels.
# Usage: gpurun --timeout 900 -- 'bash scripts/gpu_c32.sh TAG'
TAG=${1:-c32}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 900 python -m pytest tests -m gpu -q -x -k "configs4 or bidi2 or wide or lockstep or 32_cells" > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"; grep -E "^E  " "$OUT/pytest.log" | head -10
B() { timeout 300 python bench.py --config b2 --bf16 --steps 10 --warmup 3 --profile-steps 3 "$@"; }
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("[%s]" % sys.argv[2], d["value"], "lines/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels"].items() if "lstm" in k})
except Exception as e:
    print("[%s]" % sys.argv[2], "FAILED", e)
PY
}
for MB in 64 256; do
  for C in 0 1; do
    CLSTM_DEBUG=bwd_c32=$C B --minibatch $MB > "$OUT/b2_mb${MB}_c32_$C.json" 2> "$OUT/b2_mb${MB}_c32_$C.err"; show "$OUT/b2_mb${MB}_c32_$C.json" "mb=$MB CLSTM_DEBUG=bwd_c32=$C"
  done
done
for C in 0 1; do
  echo "=== stamps, CLSTM_DEBUG=bwd_c32=$C"
  CLSTM_DEBUG=bwd_c32=$C CLSTM_HIP_VARIANT=prof timeout 300 python scripts/gpu_xcdprof.py 2>&1 | grep -A 12 -E "^backward|^forward" | tee "$OUT/xcd_phase_cycles_c32_$C.txt"
done
