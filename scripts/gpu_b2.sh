#!/bin/bash
# BASELINE configs[4] shape (2 x BiLSTM(512), H=64, T=400, 64 lines): parity tests of the bf16 paths, then f32 / bf16 GEMMs / bf16 MFMA everywhere
TAG=${1:-b2}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 600 python -m pytest tests/test_net_parity.py -m gpu -q -x -k "bf16 or lockstep" > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"; grep -E "^E  |Error" "$OUT/pytest.log" | head
for M in "" "--bf16-gemm" "--bf16"; do
  timeout 600 python bench.py --config b2 --steps 10 --warmup 3 --profile-steps 2 $M > "$OUT/bench_b2$M.json" 2> "$OUT/bench_b2$M.err"; tail -2 "$OUT/bench_b2$M.err" | grep -v amdgpu
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_b2$M.json"))
    print("[$M]", d["value"], "lines/s", d["ms_per_step"], "ms/step", {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items()}); print("   ", d["roofline"])
except Exception as e: print("[$M] FAILED", e)
PY
done
