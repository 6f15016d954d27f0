for V in "" ""; do
  CLSTM_HIP_VARIANT=$V python bench.py --config b2 --bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('lstm_fwd','lstm_bwd','gemm_gates_x','gemm_gates_dw','gemm_gates_dx')})"
done
python bench.py --config b2 --bf16 --minibatch 256 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('256 lines', d['value'], d['ms_per_step'])"
CLSTM_HIP_VARIANT=prof timeout 200 python scripts/gpu_xcdprof.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests -m gpu -x -q -k "bf16 or wide or lock or configs4" 2>&1 | tail -3
