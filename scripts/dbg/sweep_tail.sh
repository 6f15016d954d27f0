#!/bin/bash
# sweep of the tail plan of the in-launch weight-gradient items (dw_tail_parts x dw_tail_chunks) on the headline step
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"
for cfg in ${CFGS:-4,3 6,3 8,3 4,2 6,2 8,4 6,4 3,3 12,3 6,6}; do
  IFS=, read p c <<< "$cfg"
  CLSTM_DEBUG="dw_tail_parts=$p,dw_tail_chunks=$c" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('parts $p chunks $c:', d['value'], d['ms_per_step'], 'lstm_bwd', d['kernels']['lstm_bwd']['ms_per_step'])"
done
