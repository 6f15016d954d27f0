#!/bin/bash
# The exact-f32 path of wide layers: the GPU parity tests that touch it, then the configs[4] f32 bench line.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"
python -c "from oracle.oracle import build; build()" >/dev/null 2>&1
timeout 600 python -m pytest tests -m gpu -x -q -k "wide or lock or stacked or configs4_full_shape_f32 or lazy" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --config b2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"; done
