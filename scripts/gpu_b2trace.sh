#!/bin/bash
# kernel-trace of the configs[4] step (bf16 lock-step recurrence): per-kernel durations and the gaps between the
# dependent per-step launches
# usage: bash scripts/gpu_b2trace.sh TAG [variant ...]   (variants: CLSTM_HIP_VARIANT libraries of `make variant`)
TAG=${1:-b2tr}; shift; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for V in "" "$@"; do
rm -rf "$OUT/trace"
CLSTM_HIP_VARIANT=$V timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python "$ROOT/bench.py" --config b2 --bf16 --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 $BENCH_ARGS > "$OUT/trace.log" 2>&1
tail -1 "$OUT/trace.log" | cut -c1-300
F=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python - "$F" > "$OUT/b2_step_gaps_${V:-base}.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i, r in enumerate(rows):
    n = r["Kernel_Name"].split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append(e - s)
    if i + 1 < len(rows) and rows[i + 1]["Kernel_Name"] == r["Kernel_Name"]:
        gap[n].append(int(rows[i + 1]["Start_Timestamp"]) - e)
print("%-62s %7s %9s %9s %9s %9s" % ("kernel", "calls", "avg us", "min us", "gap avg", "gap min"))
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    g = gap.get(n, [0])
    print("%-62s %7d %9.2f %9.2f %9.2f %9.2f" % (n, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, sum(g) / len(g) / 1e3, min(g) / 1e3))
PY
echo "== variant [${V:-base}]"; head -4 "$OUT/b2_step_gaps_${V:-base}.txt"
rm -rf "$OUT/trace"
done
