#!/bin/bash
# round 2, call A: parity tests, smoke, bench (single call step), RCCL world-size-1, self-spawn check, CU-mask census
TAG=${1:-r2a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
echo "=== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"; grep -E "^E  .*mismatch|Error|FAILED" "$OUT/pytest_gpu.log" | head -20
echo "=== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
echo "=== bench default"
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cut -c1-1500 "$OUT/bench_default.json"; tail -3 "$OUT/bench_default.err"
echo "=== bench BENCH_FORCE_DIST=1 (RCCL communicator over one rank)"
BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline > "$OUT/bench_forcedist.json" 2> "$OUT/bench_forcedist.err"; cut -c1-1200 "$OUT/bench_forcedist.json"; tail -5 "$OUT/bench_forcedist.err"
echo "=== bench --gpus 2 on a 1-GPU box (self-spawn; must fail cleanly, not hang)"
timeout 180 python bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/bench_gpus2.json" 2> "$OUT/bench_gpus2.err"; echo "rc=$?"; tail -4 "$OUT/bench_gpus2.err"
echo "=== CU mask census"
timeout 60 scripts/ubench/cumask > "$OUT/cumask.txt" 2>&1; echo "rc=$?"; cat "$OUT/cumask.txt"
echo "=== done"
