#!/bin/bash
# trimmed round evidence: full GPU suite, smoke, default / driver-command / ragged / host-fed bench lines,
# rocprofv3 kernel stats + one-step timeline.  (gpu_round.sh is the full version with the PMC passes.)
TAG=${1:-r3e}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 1200 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"; grep -E "^(FAILED|ERROR)|^E  " "$OUT/pytest_gpu.log" | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
B() { timeout 600 python bench.py "$@"; }
B > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -2 "$OUT/bench_default.err" | grep -v amdgpu
B --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_cmd.json" 2>/dev/null
B --ragged --no-cpu-baseline --no-secondary > "$OUT/bench_ragged.json" 2>/dev/null
B --host-inputs --no-cpu-baseline --no-secondary > "$OUT/bench_host_inputs.json" 2>/dev/null
B --minibatch 256 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 > "$OUT/bench_mb256.json" 2>/dev/null
python - <<PY
import json
for f in ("bench_default", "bench_driver_cmd", "bench_ragged", "bench_host_inputs", "bench_mb256"):
    try: d = json.load(open("$OUT/%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "repeats", d.get("repeats"), "enqueue ms", d.get("host_enqueue_ms_per_step"))
    print("  ", {k: v["ms_per_step"] for k, v in d["kernels"].items()})
    if d.get("secondary"): s = d["secondary"]; print("   secondary", s["value"], s["ms_per_step"], {k: v["ms_per_step"] for k, v in s["kernels"].items()})
    if d.get("roofline"): r = d["roofline"]; print("  ", {k: r.get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms")})
    if d.get("cpu_baseline"): print("  ", d["cpu_baseline"])
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --profile-steps 0 > "$OUT/rocprof.log" 2>&1
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do head -16 "$f" | cut -c1-200; done
F=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python - "$F" "$OUT/timeline.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_ingest_pack" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
out = open(sys.argv[2], "w")
out.write("one training step under rocprofv3 --kernel-trace (us from the start of k_ingest_pack): start  end  duration  kernel\n")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.write("%9.2f %9.2f %8.2f  %s\n" % (s / 1e3, e / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:80]))
out.write("step length %.2f us\n" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
out.close()
print(open(sys.argv[2]).read())
PY
find "$OUT/prof" -name "*kernel_trace*" -size +20M -delete
echo "=== done"
