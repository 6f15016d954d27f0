#!/bin/bash
# one configs[4] training step (bf16) under rocprofv3 --kernel-trace --stats: start / end / duration of every launch of the median step
# (b2_timeline.txt) and the per-kernel averages over the whole run (b2_kernel_stats.csv)
# usage: bash scripts/gpu_b2timeline.sh TAG [bench args]
TAG=${1:-b2tl}; shift; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rm -rf "$OUT/trace"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python "$ROOT/bench.py" --config b2 --bf16 --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 "$@" > "$OUT/trace.log" 2>&1
tail -1 "$OUT/trace.log" | cut -c1-200
F=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python "$ROOT/scripts/step_timeline.py" "$F" "$OUT/b2_timeline.txt"
cat "$OUT/b2_timeline.txt"
S=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" "$OUT/b2_kernel_stats.csv" && head -12 "$OUT/b2_kernel_stats.csv" | cut -c1-160
rm -rf "$OUT/trace"
