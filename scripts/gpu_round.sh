#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench lines, rocprof kernel stats, PMC passes, diagnostics.
# Usage (from the build container):  gpurun --timeout 2400 -- 'bash scripts/gpu_round.sh r02'
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
{ rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; } > "$OUT/host.txt" 2>&1
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1

echo "=== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"
echo "=== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"

B() { timeout 600 python bench.py "$@"; }
echo "=== bench (default = minibatch 64, T=200)"
B > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cut -c1-400 "$OUT/bench_default.json"; tail -2 "$OUT/bench_default.err" | grep -v amdgpu
B --weights trained --no-cpu-baseline --no-secondary > "$OUT/bench_trained.json" 2>/dev/null; cut -c1-220 "$OUT/bench_trained.json"
for MB in 1 16 256 1024 2048; do
  B --minibatch $MB --no-cpu-baseline --no-secondary --steps 50 --warmup 10 > "$OUT/bench_mb$MB.json" 2> "$OUT/bench_mb$MB.err"; cut -c1-220 "$OUT/bench_mb$MB.json"
done
B --ragged --no-cpu-baseline --no-secondary > "$OUT/bench_ragged.json" 2>/dev/null; cut -c1-220 "$OUT/bench_ragged.json"
B --host-inputs --no-cpu-baseline --no-secondary > "$OUT/bench_host_inputs.json" 2>/dev/null; cut -c1-220 "$OUT/bench_host_inputs.json"
B --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2>/dev/null; cut -c1-220 "$OUT/bench_driver_cmd.json"
CLSTM_OVERLAP=0 B --no-cpu-baseline --no-secondary > "$OUT/bench_overlap0.json" 2>/dev/null; cut -c1-220 "$OUT/bench_overlap0.json"
BENCH_FORCE_DIST=1 B --no-cpu-baseline --no-secondary > "$OUT/bench_forcedist.json" 2>/dev/null; cut -c1-220 "$OUT/bench_forcedist.json"
echo "=== bench 2xBiLSTM(512) shape: f32 / bf16 hoisted GEMMs / bf16 MFMA everywhere"
B --config b2 --steps 10 --warmup 3 --profile-steps 2 > "$OUT/bench_b2.json" 2>/dev/null; cut -c1-200 "$OUT/bench_b2.json"
B --config b2 --steps 10 --warmup 3 --profile-steps 2 --bf16 > "$OUT/bench_b2_bf16.json" 2>/dev/null; cut -c1-200 "$OUT/bench_b2_bf16.json"

echo "=== one configs[4] step under rocprofv3 --kernel-trace"
bash "$ROOT/scripts/gpu_b2timeline.sh" "$TAG" > "$OUT/b2_timeline.log" 2>&1; tail -3 "$OUT/b2_timeline.log"
cd "$ROOT"
echo "=== the reference's own drivers over the INetwork adapter (test-ocr.sh scenario): rate of the literal drop-in"
timeout 300 python -m pytest tests/test_integration_shim.py -m gpu -q -s -k unmodified 2>&1 | grep -E "drop-in|passed|failed" | tee "$OUT/drop_in_rate.txt"
echo "=== this repo's driver from the same PNG files, batch=64"
bash "$ROOT/scripts/gpu_driver_rate.sh" "$TAG" 2>&1 | tee "$OUT/driver_rate.txt"
cd "$ROOT"
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
# the other legs of the default line, each with its own evidence: the strict_f32 step (f32-MFMA weight-gradient
# items: lstm_bwd_dw_kernel<7, 25, 0>) and the 256-line step -- kernel stats, then FETCH_SIZE / WRITE_SIZE in separate passes
for LEG in strict mb256; do
  if [ $LEG = strict ]; then LA="--strict-f32"; else LA="--minibatch 256"; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$LEG" -o bench -- python "$ROOT/bench.py" $LA --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --profile-steps 0 > "$OUT/rocprof_$LEG.log" 2>&1
  find "$OUT/prof_$LEG" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats_$LEG.csv"; head -5 "$f" | cut -c1-160; done
  find "$OUT/prof_$LEG" -name "*kernel_trace*" -size +20M -delete
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_${CNT}_$LEG" -o bench -- python "$ROOT/bench.py" $LA --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0 > "$OUT/rocprof_${CNT}_$LEG.log" 2>&1
    python "$ROOT/scripts/summarize_pmc.py" "$OUT/pmc_${CNT}_$LEG" $CNT > "$OUT/pmc_${CNT}_${LEG}_summary.txt" 2>&1; head -4 "$OUT/pmc_${CNT}_${LEG}_summary.txt"
    find "$OUT/pmc_${CNT}_$LEG" -name "*.csv" -size +8M -delete
  done
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
echo "=== diagnostics: per-phase cycles of the forward recurrence and of the CTC kernel"
cd "$ROOT"
CLSTM_FW_TRACE="$OUT/fw_trace.txt" timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0 > /dev/null 2>&1
python scripts/fwtrace_summary.py "$OUT/fw_trace.txt" > "$OUT/fwd_timeline.txt" 2>&1; head -3 "$OUT/fwd_timeline.txt"; tail -2 "$OUT/fwd_timeline.txt"
timeout 300 python scripts/gpu_ctcprof.py > "$OUT/ctc_phase_cycles.txt" 2>&1; tail -6 "$OUT/ctc_phase_cycles.txt"
# per-phase stamps of the persistent bf16 recurrences of wide layers (diagnostics build; built here if it is missing)
make -s -C clstm_amd/csrc ../lib/libclstm_hip_prof.so > /dev/null 2>&1   # (stale or missing: rebuilt here)
CLSTM_HIP_VARIANT=prof timeout 300 python scripts/gpu_xcdprof.py 2>&1 | grep -v amdgpu.ids > "$OUT/xcd_phase_cycles.txt"; head -11 "$OUT/xcd_phase_cycles.txt"
# gpurun copies back at most 64 MiB: keep the summaries, drop the raw traces / counter tables / databases
find "$OUT" -type f \( -name "*.db" -o -name "*.rocpd" -o -name "*.json" -size +2M -o -name "*.csv" -size +2M \) -delete
find "$OUT" -type f -name "*kernel_trace*.csv" -delete
find "$OUT" -type f -name "*counter_collection*.csv" -delete
du -sh "$OUT" | tail -1
echo "=== done"
