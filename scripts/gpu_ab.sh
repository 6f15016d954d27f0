#!/bin/bash
# A/B on the GPU: parity tests, then the default bench under several environment settings
# usage: bash scripts/gpu_ab.sh TAG "ENV1=.. ENV2=.." "ENV=.." ...   (each quoted argument is one configuration)
TAG=${1:-ab}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2; grep -E "^E  .*mismatch|Error|FAILED|assert" "$OUT/pytest_gpu.log" | head -10
fi
i=0
for CFG in "" "$@"; do
  i=$((i+1))
  env $CFG timeout 300 python bench.py --no-cpu-baseline $BENCH_ARGS > "$OUT/bench_$i.json" 2> "$OUT/bench_$i.err"; tail -2 "$OUT/bench_$i.err" | grep -v amdgpu.ids
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$i.json"))
    print("[$CFG]", "value", d["value"], "ms/step", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], {k:round(v["ms_per_step"]*1e3,1) for k,v in d["kernels"].items()})
except Exception as e:
    print("[$CFG]", "FAILED", e)
PY
done
