#!/bin/bash
# SQ counters of the SATURATED form of the headline workload (256 lines per GPU: two recurrence workgroups per CU; the bench line's
# `saturated` leg) -- separate rocprofv3 --pmc passes with --kernel-trace only.  Usage: gpurun -- 'bash scripts/gpu_pmc_saturated.sh r05'
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
CMD="python $ROOT/bench.py --minibatch 256 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0"
P1="SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"
P2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
rm -rf "$OUT/pmc_sat1" "$OUT/pmc_sat2"
timeout 600 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d "$OUT/pmc_sat1" -o bench -- $CMD > "$OUT/rocprof_sat1.log" 2>&1
timeout 600 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d "$OUT/pmc_sat2" -o bench -- $CMD > "$OUT/rocprof_sat2.log" 2>&1
{
  echo "# SQ counters per launch, headline net at 256 lines per GPU (bench.py --minibatch 256; averages over the run's launches)"
  for CNT in $P1; do python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_sat1" $CNT | head -8; done
  for CNT in $P2; do python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_sat2" $CNT | head -8; done
} > "$OUT/pmc_SQ_saturated_summary.txt" 2>&1
head -40 "$OUT/pmc_SQ_saturated_summary.txt"
find "$OUT/pmc_sat1" "$OUT/pmc_sat2" -name "*.csv" -size +8M -delete
