#!/bin/bash
# the full-row form of the batched backward recurrence (option bwd_mfma_rows): parity selection, then timing against the default form
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"
python -c "from oracle.oracle import build; build()" >/dev/null 2>&1
CLSTM_DEBUG="bwd_mfma_rows=1" timeout 600 python -m pytest tests/test_mfma_recurrence.py -m gpu -q -x 2>&1 | tail -4
for mb in ${MBS:-256 1024 2048}; do for r in 0 1; do
CLSTM_DEBUG="fwd_mfma=1,bwd_mfma=2,bwd_mfma_rows=$r" python bench.py --no-cpu-baseline --no-secondary --minibatch $mb --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mb $mb rows $r:', d['value'], d['ms_per_step'], 'lstm_bwd', d['kernels']['lstm_bwd']['ms_per_step'])"
done; done
