#!/bin/bash
# run a pytest selection on the GPU box:  gpurun -- 'bash scripts/gpu_one.sh TAG "<pytest args>"'
TAG=${1:-one}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 1500 python -m pytest -m gpu -q -s --durations=5 $@ > "$OUT/pytest.log" 2>&1; tail -40 "$OUT/pytest.log"
