#!/bin/bash
# quick iteration: a pytest selection + the default bench line (no CPU baseline, no secondary)
TAG=${1:-r3b}; SEL=${2:-"forward_as_one or full_bench or full_shape or overlapped"}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 900 python -m pytest tests -m gpu -q -x -k "$SEL" > "$OUT/pytest.log" 2>&1; tail -4 "$OUT/pytest.log"; grep -E "^E  " "$OUT/pytest.log" | head -10
for extra in "" "--ragged"; do
timeout 600 python bench.py --no-cpu-baseline --no-secondary $extra > "$OUT/bench$extra.json" 2> "$OUT/bench$extra.err"; tail -2 "$OUT/bench$extra.err" | grep -v amdgpu
python - <<PY
import json
d = json.load(open("$OUT/bench$extra.json"))
print("$extra value", d["value"], "ms/step", d["ms_per_step"], "repeats", d["repeats"])
print({k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
CLSTM_OVERLAP=0 timeout 600 python bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_ov0.json" 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_ov0.json')); print('overlap0', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"
