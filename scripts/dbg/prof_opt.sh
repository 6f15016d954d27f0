#!/bin/bash
# per-kernel rocprofv3 averages of one bench configuration for each value of a CLSTM_DEBUG option: prof_opt.sh <option> [values...]
# KERNEL=<substring>: also the durations of that kernel's last launches, in launch order
OPT=$1; shift; VALS=${@:-0 1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/prof_$OPT"; mkdir -p "$OUT"
export TMPDIR=/tmp
for v in $VALS; do
  rm -rf /tmp/prof_$v
  CLSTM_DEBUG="$OPT=$v${EXTRA_DEBUG:+,$EXTRA_DEBUG}" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} > "$OUT/bench_$v.json" 2> "$OUT/err_$v.txt"
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  t=$(find /tmp/prof_$v -name "*kernel_trace.csv" | head -1)
  echo "== $OPT=$v"
  python scripts/dbg/prof_opt_print.py "$f" "$t" "${KERNEL:-}" | tee "$OUT/stats_$v.txt"
done
