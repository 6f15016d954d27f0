#!/bin/bash
# round 3, first GPU call: the whole GPU suite (incl. the configs[4] oracle-anchored tests) + the default bench line
TAG=${1:-r3a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 > "$OUT/pytest_gpu.log" 2>&1; tail -5 "$OUT/pytest_gpu.log"
grep -E "oracle f|configs\[4\]|softmax outputs vs|CTC argmax|minibatch gradient|max \|g" "$OUT/pytest_gpu.log" | head -20
grep -E "^(FAILED|ERROR)|^E  " "$OUT/pytest_gpu.log" | head -30
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; tail -3 "$OUT/bench_driver_cmd.err"
timeout 900 python bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - <<PY
import json
for f in ("bench_driver_cmd", "bench_default"):
    try:
        d = json.load(open("$OUT/%s.json" % f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "repeats", d["repeats"], d["timing"])
    print({k: v["ms_per_step"] for k, v in d["kernels"].items()})
    r = d["roofline"]; print({k: r[k] for k in ("kernel", "achieved", "frac", "algorithmic_bytes", "avg_launch_ms")}, list(r["others"]))
    if d.get("secondary"):
        s = d["secondary"]; print("secondary", s["value"], s["ms_per_step"], s["roofline"]["frac"], s["roofline"]["whole_step"], {k: v["ms_per_step"] for k, v in s["kernels"].items()})
    if d.get("cpu_baseline"): print(d["cpu_baseline"])
PY
