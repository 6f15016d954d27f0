#!/bin/bash
# launch-internal timeline of the fused forward launch (CLSTM_FW_TRACE): producers, recurrence, softmax consumers
TAG=${1:-fwtr}; shift; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
CLSTM_FW_TRACE="$OUT/trace.txt" timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/trace.txt" <<'PY' | tee "$OUT/summary.txt"
import sys, collections
lines = open(sys.argv[1]).read().split("\n")
hdr = lines[0].split()
nrec, npi, nci = int(hdr[1]), int(hdr[7]), int(hdr[14])
rows = [tuple(int(x) for x in l.split()) for l in lines[1:] if l and not l.startswith("#")]
rec = [r for r in rows[:nrec] if r[2]]
t0 = min(r[0] for r in rec)
us = lambda t: (t - t0) / 100.0
print("recurrence workgroups: start %.1f..%.1f us, end %.1f..%.1f us" % (us(min(r[0] for r in rec)), us(max(r[0] for r in rec)), us(min(r[2] for r in rec)), us(max(r[2] for r in rec))))
prod = [r for r in rows[nrec:nrec + npi] if r[2]]
by = collections.defaultdict(list)
for r in prod: by[r[3]].append(r)
print("producer items by chunk: n, start min..max, done min..max, run avg (us)")
for k in sorted(by):
    v = by[k]
    print("  chunk %3d %4d  %7.1f..%7.1f  %7.1f..%7.1f  %5.1f" % (k, len(v), us(min(r[0] for r in v)), us(max(r[0] for r in v)), us(min(r[2] for r in v)), us(max(r[2] for r in v)), sum(r[2] - r[0] for r in v) / len(v) / 100.0))
cons = [r for r in rows[nrec + npi:] if r[2]]
by = collections.defaultdict(list)
for r in cons: by[(r[3] + 15) // 16 * 16].append(r)
print("consumer items by ready iteration (rounded up to 16): n, dispatched min..max, ready min..max, done min..max, run avg")
for k in sorted(by):
    v = by[k]
    f = lambda i: "%7.1f..%7.1f" % (us(min(r[i] for r in v)), us(max(r[i] for r in v)))
    print("  it %4d %4d  %s  %s  %s  %5.1f" % (k, len(v), f(0), f(1), f(2), sum(r[2] - r[1] for r in v) / len(v) / 100.0))
print("last consumer item done %.1f us" % us(max(r[2] for r in cons)))
PY
