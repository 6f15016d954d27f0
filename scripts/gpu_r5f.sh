#!/bin/bash
# round 5: whole GPU suite + configs[4] bench + default bench
TAG=${1:-r5f}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 200 python bench.py --config b2 --bf16 --steps 10 --warmup 3 --profile-steps 3 > "$OUT/bench_b2_bf16.json" 2> "$OUT/bench_b2_bf16.err"
python - "$OUT/bench_b2_bf16.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("b2 bf16", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
timeout 1500 python -m pytest tests -m gpu -q -s > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"; grep -E "^E  |FAILED|Error" "$OUT/pytest_gpu.log" | head -20
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("default", d["value"], "lines/s", d["ms_per_step"], "ms; strict_f32", (d.get("strict_f32") or {}).get("value"))
print({k: v["ms_per_step"] for k, v in d["kernels"].items()})
for k in ("saturated", "secondary", "secondary_f32"):
    s = d.get(k)
    if s: print(k, s["value"], s["ms_per_step"], {a: b["ms_per_step"] for a, b in (s.get("kernels") or {}).items()})
PY
