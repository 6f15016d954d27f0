#!/bin/bash
TAG=${1:-p}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
CLSTM_HIP_VARIANT=prof timeout 300 python scripts/gpu_lstmprof.py > "$OUT/lstm_fwd_phase_cycles.txt" 2>&1; cat "$OUT/lstm_fwd_phase_cycles.txt"
