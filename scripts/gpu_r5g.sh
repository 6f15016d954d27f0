#!/bin/bash
TAG=${1:-r5g}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
for st in 2; do
CLSTM_DEBUG=gemm_stag=$st timeout 200 python bench.py --config b2 --bf16 --steps 10 --warmup 3 --profile-steps 3 > "$OUT/bench_b2_bf16_stag$st.json" 2> "$OUT/bench_b2_bf16.err"
python - "$OUT/bench_b2_bf16_stag$st.json" $st <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("b2 bf16 GEMM_STAG", sys.argv[2], d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
timeout 900 python -m pytest tests -m gpu -q -x -s -k "bf16 or bias or c32 or configs4 or gemm or lazy or one_launch" > "$OUT/pytest_gpu_sel.log" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu_sel.log" | tail -2; grep -E "^E  |FAILED|Error" "$OUT/pytest_gpu_sel.log" | head -20
