#!/bin/bash
# round 5, third GPU call: the LDS-DMA kk GEMM against the register-staged loops, configs[4] bench A/B (GEMM tiles by DMA or not,
# forward per-frame stores behind the arrival or not), GEMM + bf16 GPU tests.
TAG=${1:-r5c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 300 python scripts/gpu_gemm_r5.py 5 > "$OUT/gemm_r5.txt" 2>&1; tail -12 "$OUT/gemm_r5.txt"
for cfg in "2 1" "1 1"; do
  set -- $cfg
  CLSTM_DEBUG=gemm_stag=$1 timeout 200 python bench.py --config b2 --bf16 --steps 10 --warmup 3 --profile-steps 3 > "$OUT/bench_b2_bf16_stag$1_late$2.json" 2>/dev/null
  python - "$OUT/bench_b2_bf16_stag$1_late$2.json" "$cfg" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("b2 bf16 [GEMM_STAG FWD_LATE] =", sys.argv[2], d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
timeout 900 python -m pytest tests -m gpu -q -x -s -k "bf16 or bias or c32 or configs4 or gemm or lazy" > "$OUT/pytest_gpu_sel.log" 2>&1
tail -3 "$OUT/pytest_gpu_sel.log"; grep -E "^E  |FAILED|Error" "$OUT/pytest_gpu_sel.log" | head -20
