#!/bin/bash
# GPU iteration: parity tests + default bench + the 2xBiLSTM(512) shape
TAG=${1:-q}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"; grep -E "^E  .*mismatch|Error" "$OUT/pytest_gpu.log" | head -10
timeout 600 python bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -2 "$OUT/bench.err"
timeout 900 python bench.py --config b2 --steps 5 --warmup 2 --profile-steps 2 "$@" > "$OUT/bench_b2.json" 2> "$OUT/bench_b2.err"; tail -3 "$OUT/bench_b2.err"
python - <<PY
import json
for f in ("bench.json", "bench_b2.json"):
    try:
        d=json.load(open("$OUT/"+f))
    except Exception as e:
        print(f, "failed", e); continue
    print(f, "value", d["value"], "ms/step", d["ms_per_step"])
    print({k:v["ms_per_step"] for k,v in d["kernels"].items()})
    print(d["roofline"])
PY
