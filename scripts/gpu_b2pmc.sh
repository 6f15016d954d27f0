#!/bin/bash
# configs[4] (bench.py --config b2 --bf16): MFMA busy cycles, GUI-active cycles and HBM traffic per launch of every kernel of a step
# (separate rocprofv3 --pmc passes, kernel trace only).  Usage: gpurun --timeout 900 -- 'bash scripts/gpu_b2pmc.sh r04'
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --config b2 --bf16 --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_b2_MFMA" -o bench -- $CMD > "$OUT/rocprof_b2_MFMA.log" 2>&1
for CNT in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_b2_MFMA" $CNT > "$OUT/pmc_b2_${CNT}_summary.txt" 2>&1; done
for CNT in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_b2_$CNT" -o bench -- $CMD > "$OUT/rocprof_b2_$CNT.log" 2>&1
  python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_b2_$CNT" $CNT > "$OUT/pmc_b2_${CNT}_summary.txt" 2>&1
done
find "$OUT" -name "*.csv" -size +8M -delete
head -14 "$OUT"/pmc_b2_*_summary.txt
