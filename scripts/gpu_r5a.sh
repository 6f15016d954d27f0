#!/bin/bash
# round 5, first GPU call: the new GPU tests (multi-rank on one GPU at 2/4/8 ranks, trajectories, 256 lines, placement fallback
# with the all-or-nothing step, GEMM bit-identity), the staggered bf16-source GEMMs against the one-barrier loop at the configs[4]
# shapes, configs[4] bench both ways, the default bench line.   gpurun --timeout 1500 -- 'bash scripts/gpu_r5a.sh r5a'
TAG=${1:-r5a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 300 python scripts/gpu_gemm_r5.py 5 > "$OUT/gemm_r5.txt" 2>&1; cat "$OUT/gemm_r5.txt" | tail -12
for st in 0 1; do
  CLSTM_DEBUG=gemm_stag=$st timeout 200 python bench.py --config b2 --bf16 --steps 10 --warmup 3 --profile-steps 3 > "$OUT/bench_b2_bf16_stag$st.json" 2>/dev/null
  python - "$OUT/bench_b2_bf16_stag$st.json" $st <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("b2 bf16 stag", sys.argv[2], d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
timeout 900 python -m pytest tests -m gpu -q -x -s -k "distributed or intrinsics or three_minibatch or 256_lines or two_minibatch or placement_fallback or test_states_and_step" > "$OUT/pytest_gpu_sel.log" 2>&1
tail -4 "$OUT/pytest_gpu_sel.log"; grep -E "^E  |FAILED|Error" "$OUT/pytest_gpu_sel.log" | head -20
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -2 "$OUT/bench_default.err" | grep -v amdgpu
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("default", d["value"], "lines/s", d["ms_per_step"], "ms; strict_f32", d.get("strict_f32"))
print({k: v["ms_per_step"] for k, v in d["kernels"].items()})
for k in ("saturated", "secondary", "secondary_f32"):
    s = d.get(k)
    if s: print(k, s["value"], s["ms_per_step"], {a: b["ms_per_step"] for a, b in (s.get("kernels") or {}).items()})
print("saturated roofline", json.dumps((d.get("saturated") or {}).get("roofline"))[:900])
PY
