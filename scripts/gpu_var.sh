#!/bin/bash
# quick GPU iteration: parity tests on the base library, then the default bench for the base and for each
# experiment variant (CLSTM_HIP_VARIANT=<name> -> clstm_amd/lib/libclstm_hip_<name>.so)
# usage: bash scripts/gpu_var.sh TAG [variant ...]
TAG=${1:-v}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2; grep -E "^E  .*mismatch|Error|FAILED" "$OUT/pytest_gpu.log" | head -10
fi
for V in "" "$@"; do
  CLSTM_HIP_VARIANT=$V timeout 300 python bench.py --no-cpu-baseline $BENCH_ARGS > "$OUT/bench_${V:-base}.json" 2> "$OUT/bench_${V:-base}.err"; tail -2 "$OUT/bench_${V:-base}.err" | grep -v amdgpu.ids
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${V:-base}.json"))
    print("${V:-base}", "value", d["value"], "ms/step", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], {k:round(v["ms_per_step"]*1e3,1) for k,v in d["kernels"].items()})
except Exception as e:
    print("${V:-base}", "FAILED", e)
PY
done
