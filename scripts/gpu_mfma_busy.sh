#!/bin/bash
# MFMA-busy counters of the chip-filling step (2048 lines: batched-MFMA recurrences) -- separate PMC pass, kernel trace only
TAG=${1:-mfmabusy}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc" -o bench -- python "$ROOT/bench.py" --minibatch ${MB:-2048} --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0 > "$OUT/rocprof.log" 2>&1
for CNT in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc" $CNT > "$OUT/pmc_${CNT}_summary.txt" 2>&1; head -12 "$OUT/pmc_${CNT}_summary.txt"; done
find "$OUT" -type f \( -name "*.db" -o -name "*counter_collection*.csv" -o -name "*kernel_trace*.csv" \) -delete
