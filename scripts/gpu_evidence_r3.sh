#!/bin/bash
# round-3 evidence that gpu_round.sh does not produce: width sweep of the pure recurrence kernels, per-phase stamps of the
# staggered forward kernel (diagnostics build), the configs[4] shape at 64 / 128 / 256 lines, the C++ driver's end-to-end rate
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/ev3"; mkdir -p "$OUT"
CLSTM_OVERLAP=0 timeout 300 python scripts/gpu_width_sweep.py 2>&1 | grep -v amdgpu.ids > "$OUT/width_sweep.txt"; cat "$OUT/width_sweep.txt"
CLSTM_OVERLAP=0 CLSTM_HIP_VARIANT=prof timeout 300 python scripts/gpu_lstmprof.py 2>&1 | grep -v amdgpu.ids > "$OUT/lstm_fwd_phase_cycles.txt"; cat "$OUT/lstm_fwd_phase_cycles.txt"
for MB in 64 128 256; do timeout 300 python bench.py --config b2 --bf16 --minibatch $MB --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('configs[4] shape, %d lines: %.0f lines/s, %.3f ms per minibatch' % ($MB, d['value'], d['ms_per_step']), {k: v['ms_per_step'] for k, v in d['kernels'].items()})"; done > "$OUT/b2_minibatch_sweep.txt"; cat "$OUT/b2_minibatch_sweep.txt"
bash scripts/gpu_driver_rate.sh ev3 > "$OUT/driver_rate.txt" 2>&1; cat "$OUT/driver_rate.txt"
