#!/bin/bash
# round 5, second GPU call: GEMM phase stamps / leave-outs (diagnostics build), GEMM timings with the 192-row tile, configs[4] bench
# with the bias row outside the weight-gradient product, the bf16 / configs[4] GPU tests.
TAG=${1:-r5b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
CLSTM_HIP_VARIANT=gprof timeout 300 python scripts/gpu_gemmprof_r5.py > "$OUT/gemmprof.txt" 2>&1; tail -32 "$OUT/gemmprof.txt"
timeout 300 python scripts/gpu_gemm_r5.py 5 > "$OUT/gemm_r5.txt" 2>&1; tail -10 "$OUT/gemm_r5.txt"
timeout 200 python bench.py --config b2 --bf16 --steps 10 --warmup 3 --profile-steps 3 > "$OUT/bench_b2_bf16.json" 2>/dev/null
python - "$OUT/bench_b2_bf16.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("b2 bf16", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
timeout 900 python -m pytest tests -m gpu -q -x -s -k "bf16 or bias or c32 or configs4 or contraction_major or lazy" > "$OUT/pytest_gpu_sel.log" 2>&1
tail -3 "$OUT/pytest_gpu_sel.log"; grep -E "^E  |FAILED|Error" "$OUT/pytest_gpu_sel.log" | head -20
