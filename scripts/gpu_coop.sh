#!/bin/bash
# cooperative lock-step recurrence: parity first (bounded), then the 2xBiLSTM(512) bench with and without it
TAG=${1:-coop}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 300 python -m pytest tests/test_net_parity.py -m gpu -q -k lockstep > "$OUT/pytest_lockstep.log" 2>&1; echo "rc=$?"; tail -5 "$OUT/pytest_lockstep.log"
grep -E "^E  " "$OUT/pytest_lockstep.log" | head -10
CLSTM_COOP=1 timeout 300 python bench.py --config b2 --steps 5 --warmup 2 --profile-steps 2 > "$OUT/bench_b2.json" 2> "$OUT/bench_b2.err"; echo "rc=$?"; tail -3 "$OUT/bench_b2.err"
CLSTM_COOP=0 timeout 300 python bench.py --config b2 --steps 5 --warmup 2 --profile-steps 2 > "$OUT/bench_b2_nocoop.json" 2> "$OUT/bench_b2_nocoop.err"
python - <<PY
import json
for f in ("bench_b2.json", "bench_b2_nocoop.json"):
    try:
        d=json.load(open("$OUT/"+f))
    except Exception as e:
        print(f, "failed", e); continue
    print(f, "value", d["value"], "ms/step", d["ms_per_step"])
    print({k:v["ms_per_step"] for k,v in d["kernels"].items()})
PY
