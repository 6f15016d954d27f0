#!/bin/bash
# MFMA narrow recurrence: parity, phase stamps (prof build), the 256-line step
TAG=${1:-mfma2}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
python -c "from oracle.oracle import build; build()" > "$OUT/oracle_build.log" 2>&1
timeout 900 python -m pytest tests/test_mfma_recurrence.py -m gpu -q > "$OUT/pytest_mfma.log" 2>&1; tail -5 "$OUT/pytest_mfma.log"
if [ -f clstm_amd/lib/libclstm_hip_prof.so ]; then
  for n in 256 1024; do CLSTM_HIP_VARIANT=prof timeout 300 python scripts/gpu_mfmaprof.py $n 2>&1 | tee "$OUT/prof_$n.txt" | tail -9; done
fi
for mb in ${MBS:-256 512 1024}; do
  for mode in 1; do
    CLSTM_DEBUG="fwd_mfma=$mode,bwd_mfma=$mode" timeout 600 python bench.py --no-cpu-baseline --no-secondary --minibatch $mb --steps 20 --warmup 5 \
        > "$OUT/bench_mb${mb}_mfma${mode}.json" 2> "$OUT/bench_mb${mb}_mfma${mode}.err"
    python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_mb${mb}_mfma${mode}.json"))
    print("mb $mb mfma $mode: value", d["value"], "ms/step", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})
except Exception as e:
    print("mb $mb mfma $mode: FAILED", e); print(open("$OUT/bench_mb${mb}_mfma${mode}.err").read()[-1500:])
PY
  done
done
