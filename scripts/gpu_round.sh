#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench lines, rocprof kernel stats.
# Usage (from the build container):  gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh r01'
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
{ rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; } > "$OUT/host.txt" 2>&1
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1

echo "=== pytest -m gpu" 
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
RC=$?
tail -15 "$OUT/pytest_gpu.log"
if [ $RC -ne 0 ]; then
  echo "=== DPP variant failed; full run + shfl variant for diagnosis"
  timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu_all.log" 2>&1; tail -40 "$OUT/pytest_gpu_all.log"
  CLSTM_HIP_VARIANT=shfl timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu_shfl.log" 2>&1; tail -30 "$OUT/pytest_gpu_shfl.log"
fi

echo "=== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -3 "$OUT/smoke.log"

echo "=== bench (default = minibatch 64, T=200)"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cat "$OUT/bench_default.json"; tail -3 "$OUT/bench_default.err"
for MB in 1 16 256 1024; do
  echo "=== bench minibatch $MB"
  timeout 300 python bench.py --minibatch $MB --no-cpu-baseline --steps 20 > "$OUT/bench_mb$MB.json" 2> "$OUT/bench_mb$MB.err"; cat "$OUT/bench_mb$MB.json"; tail -2 "$OUT/bench_mb$MB.err"
done
echo "=== bench ragged"
timeout 300 python bench.py --ragged --no-cpu-baseline --steps 20 > "$OUT/bench_ragged.json" 2> "$OUT/bench_ragged.err"; cat "$OUT/bench_ragged.json"

echo "=== rocprofv3 kernel stats"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > "$OUT/rocprof.log" 2>&1
tail -3 "$OUT/rocprof.log"
find "$OUT/prof" -name "*kernel_stats*" | head -2 | while read f; do echo "--- $f"; head -25 "$f"; done
find "$OUT/prof" -name "*kernel_trace*" -size +20M -delete
echo "=== rocprofv3 PMC passes (HBM traffic): FETCH_SIZE, WRITE_SIZE in separate runs"
for CNT in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_$CNT" -o bench -- python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 > "$OUT/rocprof_$CNT.log" 2>&1
  tail -2 "$OUT/rocprof_$CNT.log"
  python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_$CNT" $CNT > "$OUT/pmc_${CNT}_summary.txt" 2>&1; cat "$OUT/pmc_${CNT}_summary.txt"
  find "$OUT/pmc_$CNT" -name "*.csv" -size +8M -delete
done
echo "=== rocprofv3 PMC pass: MFMA busy cycles (north star: MFMA utilisation of the batched gate GEMMs)"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_MFMA" -o bench -- python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 > "$OUT/rocprof_MFMA.log" 2>&1
tail -2 "$OUT/rocprof_MFMA.log"
for CNT in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_MFMA" $CNT > "$OUT/pmc_${CNT}_summary.txt" 2>&1; head -8 "$OUT/pmc_${CNT}_summary.txt"
done
find "$OUT/pmc_MFMA" -name "*.csv" -size +8M -delete
echo "=== rocprofv3 PMC pass: SQ wave-state counters (where the waves of each kernel spend their cycles)"
SQC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
timeout 600 rocprofv3 --pmc $SQC --kernel-trace --output-format csv -d "$OUT/pmc_SQ" -o bench -- python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 > "$OUT/rocprof_SQ.log" 2>&1
tail -2 "$OUT/rocprof_SQ.log"
for CNT in $SQC; do python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_SQ" $CNT | head -6; done > "$OUT/pmc_SQ_summary.txt" 2>&1
head -20 "$OUT/pmc_SQ_summary.txt"
find "$OUT/pmc_SQ" -name "*.csv" -size +8M -delete
echo "=== bench 2xBiLSTM(512) shape (f32)"
timeout 600 python "$ROOT/bench.py" --config b2 --steps 5 --warmup 2 --profile-steps 2 > "$OUT/bench_b2.json" 2> "$OUT/bench_b2.err"; cut -c1-300 "$OUT/bench_b2.json"
echo "=== done"
