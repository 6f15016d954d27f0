#!/bin/bash
# leave-out timings of the batched backward recurrence (lstm_mfma_bwd.h): mfma_bwd_dbg bits
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"
for mb in ${MBS:-1024}; do for d in ${DBGS:-0 1 2 4 8 12 15}; do
CLSTM_DEBUG="fwd_mfma=1,bwd_mfma=2,mfma_bwd_dbg=$d" python bench.py --no-cpu-baseline --no-secondary --minibatch $mb --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mb $mb dbg $d: lstm_bwd', d['kernels']['lstm_bwd']['ms_per_step'], 'fwd', d['kernels']['lstm_fwd']['ms_per_step'])"
done; done
