#!/bin/bash
# full GPU suite + default / ragged / mb256 bench lines (no CPU baseline, no secondary)
TAG=${1:-qf}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 1200 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2; grep -E "^(FAILED|ERROR)|^E  " "$OUT/pytest_gpu.log" | head -20
for extra in "" "--ragged" "--minibatch 256 --steps 50 --warmup 10" "--config b2 --bf16 --steps 10 --warmup 3"; do
timeout 600 python bench.py --no-cpu-baseline --no-secondary $extra > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -2 "$OUT/bench.err" | grep -v amdgpu
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("$extra | value", d["value"], "ms/step", d["ms_per_step"], "repeats", d["repeats"])
print("  ", {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
