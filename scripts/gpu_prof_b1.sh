#!/bin/bash
# The rocprofv3 half of scripts/gpu_round.sh on its own: kernel trace + stats and the PMC passes of the HEADLINE workload
# (bench.py --no-secondary: no other leg of the default line in the trace).  Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_prof_b1.sh r05'
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf "$OUT/prof" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_FETCH_SIZE_ov0 "$OUT"/pmc_WRITE_SIZE_ov0 "$OUT"/pmc_MFMA "$OUT"/pmc_SQ
echo "=== rocprofv3 kernel stats + one-step timeline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --profile-steps 0 > "$OUT/rocprof.log" 2>&1
tail -1 "$OUT/rocprof.log" | cut -c1-200
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do head -16 "$f"; done
F=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python "$ROOT/scripts/step_timeline.py" "$F" "$OUT/timeline.txt"
find "$OUT/prof" -name "*kernel_trace*" -size +20M -delete
echo "=== rocprofv3 PMC passes (separate runs): FETCH_SIZE, WRITE_SIZE, MFMA busy, SQ wave states"
for CNT in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_$CNT" -o bench -- python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0 > "$OUT/rocprof_$CNT.log" 2>&1
  python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_$CNT" $CNT > "$OUT/pmc_${CNT}_summary.txt" 2>&1; head -14 "$OUT/pmc_${CNT}_summary.txt"
  find "$OUT/pmc_$CNT" -name "*.csv" -size +8M -delete
done
# the PURE kernels (fusions off: recurrences alone, the batched gate GEMM): HBM bytes per launch
for CNT in FETCH_SIZE WRITE_SIZE; do
  CLSTM_OVERLAP=0 timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_${CNT}_ov0" -o bench -- python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0 > "$OUT/rocprof_${CNT}_ov0.log" 2>&1
  python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_${CNT}_ov0" $CNT > "$OUT/pmc_${CNT}_ov0_summary.txt" 2>&1; head -6 "$OUT/pmc_${CNT}_ov0_summary.txt"
  find "$OUT/pmc_${CNT}_ov0" -name "*.csv" -size +8M -delete
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_MFMA" -o bench -- python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0 > "$OUT/rocprof_MFMA.log" 2>&1
for CNT in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_MFMA" $CNT > "$OUT/pmc_${CNT}_summary.txt" 2>&1; head -8 "$OUT/pmc_${CNT}_summary.txt"
done
find "$OUT/pmc_MFMA" -name "*.csv" -size +8M -delete
SQC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
timeout 600 rocprofv3 --pmc $SQC --kernel-trace --output-format csv -d "$OUT/pmc_SQ" -o bench -- python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0 > "$OUT/rocprof_SQ.log" 2>&1
for CNT in $SQC; do python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_SQ" $CNT | head -6; done > "$OUT/pmc_SQ_summary.txt" 2>&1
head -12 "$OUT/pmc_SQ_summary.txt"
find "$OUT/pmc_SQ" -name "*.csv" -size +8M -delete
echo "=== done"
