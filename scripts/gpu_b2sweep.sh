ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"
python -c "from oracle.oracle import build; build()" >/dev/null 2>&1
timeout 600 python -m pytest tests -m gpu -q -x -k "bf16 or configs4 or lockstep or stacked or lazy" 2>&1 | tail -3
for MB in 64 128 256; do timeout 300 python bench.py --config b2 --bf16 --minibatch $MB --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('mb=$MB', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"; done
