#!/bin/bash
# Round 5: where the small launches of the headline step go (scripts/gpu_b1micro.py, CTC phase stamps, bench kernel times
# under the ingest experiment switches)
TAG=${1:-b1micro}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 300 python scripts/gpu_b1micro.py > "$OUT/xd.txt" 2>&1; cat "$OUT/xd.txt" | grep -v amdgpu.ids
timeout 300 python scripts/gpu_ctcprof.py > "$OUT/ctc_phase_cycles.txt" 2>&1; head -3 "$OUT/ctc_phase_cycles.txt" | grep -v amdgpu.ids
for opt in "" $EXTRA_OPTS; do
CLSTM_DEBUG="$opt" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > "$OUT/bench_$opt.json" 2> "$OUT/bench.err"
python - "$OUT/bench_$opt.json" "$opt" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("b1 [%s]" % sys.argv[2], d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
timeout 900 python -m pytest tests -m gpu -q -x -k "ctc or train_step or e2e or parity" > "$OUT/pytest_gpu_sel.log" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu_sel.log" | tail -2; grep -E "^E  |FAILED|Error" "$OUT/pytest_gpu_sel.log" | head -20
