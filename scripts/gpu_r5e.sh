#!/bin/bash
# round 5 quick A/B: GEMM variants + configs[4] bench under CLSTM_DEBUG=gemm_stag= 2 / 1
TAG=${1:-r5e}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
timeout 300 python scripts/gpu_gemm_r5.py 5 > "$OUT/gemm_r5.txt" 2>&1; grep -v amdgpu "$OUT/gemm_r5.txt" | tail -12
for st in 2 1; do
  CLSTM_DEBUG=gemm_stag=$st timeout 200 python bench.py --config b2 --bf16 --steps 10 --warmup 3 --profile-steps 3 > "$OUT/bench_b2_bf16_stag$st.json" 2>/dev/null
  python - "$OUT/bench_b2_bf16_stag$st.json" $st <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("b2 bf16 GEMM_STAG =", sys.argv[2], d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
