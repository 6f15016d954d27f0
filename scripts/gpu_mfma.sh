#!/bin/bash
# MFMA narrow recurrence (lstm_mfma.h): parity tests, then the 256- and 1024-line steps with the path on and off
TAG=${1:-mfma}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 900 python -m pytest tests/test_mfma_recurrence.py -m gpu -q -x > "$OUT/pytest_mfma.log" 2>&1; tail -15 "$OUT/pytest_mfma.log"
for mb in 256 1024; do
  for mode in 1 0; do
    CLSTM_DEBUG="fwd_mfma=$mode,bwd_mfma=$mode" timeout 600 python bench.py --no-cpu-baseline --no-secondary --minibatch $mb --steps 20 --warmup 5 \
        > "$OUT/bench_mb${mb}_mfma${mode}.json" 2> "$OUT/bench_mb${mb}_mfma${mode}.err"
    python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_mb${mb}_mfma${mode}.json"))
    print("mb $mb mfma $mode: value", d["value"], "ms/step", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})
except Exception as e:
    print("mb $mb mfma $mode: FAILED", e); print(open("$OUT/bench_mb${mb}_mfma${mode}.err").read()[-1500:])
PY
  done
done
