#!/bin/bash
# launch-internal timeline of the fused backward + weight-gradient launch (CLSTM_DW_TRACE): when does the recurrence end,
# when do the items of each chunk become ready and finish
TAG=${1:-dwtr}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
CLSTM_DW_TRACE="$OUT/trace.txt" timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/trace.txt" <<'PY' | tee "$OUT/summary.txt"
import sys, collections
rows = [tuple(int(x) for x in l.split()) for l in open(sys.argv[1]) if not l.startswith("#")]
rec = [r for r in rows[:128] if r[2]]
t0 = min(r[0] for r in rec)
us = lambda t: (t - t0) / 100.0
print("recurrence: start %.1f..%.1f us, end %.1f..%.1f us" % (us(min(r[0] for r in rec)), us(max(r[0] for r in rec)), us(min(r[2] for r in rec)), us(max(r[2] for r in rec))))
items = [r for r in rows[128:] if r[2]]
by = collections.defaultdict(list)
for r in items: by[r[3]].append(r)
print("%8s %6s %28s %28s %28s %10s" % ("need_it", "items", "dispatched (min..max)", "ready (min..max)", "done (min..max)", "run avg"))
for k in sorted(by):
    v = by[k]
    f = lambda i: "%8.1f .. %8.1f" % (us(min(r[i] for r in v)), us(max(r[i] for r in v)))
    print("%8d %6d %28s %28s %28s %10.1f" % (k, len(v), f(0), f(1), f(2), sum(r[2] - r[1] for r in v) / len(v) / 100.0))
print("last item done %.1f us" % us(max(r[2] for r in items)))
PY
