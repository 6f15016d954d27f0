#!/bin/bash
# round-4 iteration: the whole GPU parity suite, the default bench line (with its strict_f32 / secondary / secondary_f32 legs),
# the configs[4] lines in both precisions.  Usage: gpurun --timeout 1800 -- 'bash scripts/gpu_r4.sh TAG ["pytest -k expr"]'
TAG=${1:-r04}; KEXPR=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
if [ -n "$KEXPR" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -s -k "$KEXPR" > "$OUT/pytest_gpu.log" 2>&1
else timeout 1500 python -m pytest tests -m gpu -q -s > "$OUT/pytest_gpu.log" 2>&1; fi
tail -4 "$OUT/pytest_gpu.log"; grep -E "^E  |FAILED|drop-in|ragged configs|configs\[4\]" "$OUT/pytest_gpu.log" | head -20
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -2 "$OUT/bench_default.err" | grep -v amdgpu
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("default", d["value"], "lines/s", d["ms_per_step"], "ms; strict_f32", d.get("strict_f32"))
print({k: v["ms_per_step"] for k, v in d["kernels"].items()})
for k in ("secondary", "secondary_f32"):
    s = d.get(k)
    if s: print(k, s["value"], s["ms_per_step"], {a: b["ms_per_step"] for a, b in (s.get("kernels") or {}).items()})
print("cpu_baseline", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None, "dtype:", d["dtype"][:60])
PY
timeout 300 python bench.py --config b2 --steps 5 --warmup 2 --profile-steps 2 > "$OUT/bench_b2_f32.json" 2>/dev/null
python - "$OUT/bench_b2_f32.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("b2 f32", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
