#!/bin/bash
# bf16-input hoisted GEMMs: kernel tests, net test, then the 2xBiLSTM(512) bench with and without
TAG=${1:-bf16}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 300 python -m pytest tests/test_intrinsics.py tests/test_net_parity.py -m gpu -q -k "bf16 or gemm" > "$OUT/pytest.log" 2>&1; echo "rc=$?"; tail -5 "$OUT/pytest.log"
grep -E "^E  " "$OUT/pytest.log" | head -10
timeout 300 python bench.py --config b2 --bf16-gemm --steps 5 --warmup 2 --profile-steps 2 > "$OUT/bench_b2_bf16.json" 2> "$OUT/bench_b2_bf16.err"; echo "rc=$?"; tail -3 "$OUT/bench_b2_bf16.err"
timeout 300 python bench.py --config b2 --steps 5 --warmup 2 --profile-steps 2 > "$OUT/bench_b2.json" 2> "$OUT/bench_b2.err"
timeout 300 python bench.py --bf16-gemm --no-cpu-baseline > "$OUT/bench_b1_bf16.json" 2> "$OUT/bench_b1_bf16.err"
python - <<PY
import json
for f in ("bench_b2_bf16.json", "bench_b2.json", "bench_b1_bf16.json"):
    try:
        d=json.load(open("$OUT/"+f))
    except Exception as e:
        print(f, "failed", e); continue
    print(f, "value", d["value"], "ms/step", d["ms_per_step"])
    print({k:v["ms_per_step"] for k,v in d["kernels"].items()})
PY
