#!/bin/bash
# quick iteration on the recurrence kernels: a parity selection, the pure-kernel width sweep, the default bench line
TAG=${1:-stag}; SEL=${2:-"forward_as_one or full_bench or full_shape or bidi_small or uw3_shape or overlapped"}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 900 python -m pytest tests -m gpu -q -x -k "$SEL" > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"; grep -E "^E  " "$OUT/pytest.log" | head -10
CLSTM_OVERLAP=0 timeout 300 python scripts/gpu_width_sweep.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/width_sweep.txt"
for extra in "" "--ragged"; do
timeout 600 python bench.py --no-cpu-baseline --no-secondary $extra > "$OUT/bench$extra.json" 2> "$OUT/bench$extra.err"; tail -2 "$OUT/bench$extra.err" | grep -v amdgpu
python - <<PY
import json
d = json.load(open("$OUT/bench$extra.json"))
print("$extra value", d["value"], "ms/step", d["ms_per_step"], "repeats", d["repeats"])
print({k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
