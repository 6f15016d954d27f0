#!/bin/bash
# full GPU suite + default bench line + host-fed rates (Python: pinned host frames; C++: clstmocrtrain batch=64 from PNGs)
TAG=${1:-r3d}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"; grep -E "^(FAILED|ERROR)|^E  " "$OUT/pytest_gpu.log" | head -20
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; tail -2 "$OUT/bench_driver_cmd.err" | grep -v amdgpu
timeout 600 python bench.py --no-cpu-baseline --no-secondary --host-inputs > "$OUT/bench_host_inputs.json" 2> "$OUT/bench_host_inputs.err"
python - <<PY
import json
for f in ("bench_driver_cmd", "bench_host_inputs"):
    try: d = json.load(open("$OUT/%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "repeats", d["repeats"], "enqueue ms", d["host_enqueue_ms_per_step"])
    print({k: v["ms_per_step"] for k, v in d["kernels"].items()})
    if d.get("secondary"): print("secondary", d["secondary"]["value"], d["secondary"]["ms_per_step"])
    if d.get("roofline"): r = d["roofline"]; print({k: r.get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms")})
PY
# C++ driver from PNGs: the fixture 256 times, minibatches of 64 (prefetch thread reads + normalises the next one)
W=$(mktemp -d); cp tests/golden/textline.bin.png "$W/l.bin.png"; cp tests/golden/textline.gt.txt "$W/l.gt.txt"
for i in $(seq 256); do echo "$W/l.bin.png"; done > "$W/list.txt"
( cd "$W"; time ( batch=64 ntrain=12800 report_every=6400 report_time=1 save_name="" test_every=1000000 hidden=100 lrate=1e-4 "$ROOT/clstm_amd/bin/clstmocrtrain" list.txt ) ) > "$OUT/driver_batch64.log" 2>&1
grep -E "steptime|real" "$OUT/driver_batch64.log"
