#!/bin/bash
# configs[4] A/B on the GPU: a pytest selection first (parity before speed), then bench.py --config b2 --bf16 under each of the
# environment settings given as arguments ("-" = none).
# Usage: gpurun --timeout 900 -- 'bash scripts/gpu_b2ab.sh TAG "pytest -k expr" - CLSTM_DEBUG=fuse_wx=0 ...'
TAG=$1; KEXPR=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
if [ -n "$KEXPR" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x -k "$KEXPR" > "$OUT/pytest.log" 2>&1; tail -5 "$OUT/pytest.log"; grep -E "^E  " "$OUT/pytest.log" | head -10
fi
for ENVS in "$@"; do
  [ "$ENVS" = "-" ] && ENVS=""
  NAME=$(echo "b2_${ENVS:-default}" | tr ' =' '__')
  env $ENVS timeout 300 python bench.py --config b2 --bf16 --steps 10 --warmup 3 --profile-steps 3 > "$OUT/$NAME.json" 2> "$OUT/$NAME.err"
  python - "$OUT/$NAME.json" "$ENVS" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("[%s]" % sys.argv[2], d["value"], "lines/s", d["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in d["kernels"].items()})
except Exception as e:
    print("[%s]" % sys.argv[2], "FAILED", e)
PY
done
