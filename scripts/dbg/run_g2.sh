python -c "from oracle.oracle import build; build()" >/dev/null 2>&1
timeout 300 python -m pytest tests/test_bench_ranks.py -m gpu -q -x 2>&1 | tail -5
mkdir -p gpurun_out/r6a; (time timeout 300 python bench.py --no-secondary > gpurun_out/r6a/bench_ns.json 2> gpurun_out/r6a/bench_ns.err); tail -3 gpurun_out/r6a/bench_ns.err
python - <<PY
import json
d=json.load(open("gpurun_out/r6a/bench_ns.json"))
print(d["value"], d["ms_per_step"])
print(d["cpu_baseline"])
PY
