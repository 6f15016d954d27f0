#!/bin/bash
# debug: where does `bench.py --gpus 2` (two ranks sharing device 0) hang?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/gpus2"; mkdir -p "$OUT"
export CLSTM_BENCH_WATCHDOG_S=${WD:-40} PYTHONFAULTHANDLER=1 CLSTM_BENCH_BACKEND=gloo CLSTM_BENCH_SHARE_DEVICE=1 CLSTM_COMM_NO_RCCL=1 CLSTM_BENCH_MIN_TIMED_S=0.2 CLSTM_BENCH_MIN_WARMUP_S=0.1 CLSTM_PEER_TIMEOUT_S=10
run() {
  echo "=== $1"; shift
  env "$@" timeout 100 python bench.py --gpus 2 --steps 5 --warmup 2 --profile-steps 0 ${EXTRA} > "$OUT/out.json" 2> "$OUT/err.txt"; echo "rc $?"
  grep -v "amdgpu.ids\|socket.cpp\|^\*\*\*\|OMP_NUM_THREADS" "$OUT/err.txt" | grep -i "error\|time-out\|timeout\|bench.py\", line 2[0-9][0-9]" | head -8; head -c 200 "$OUT/out.json"; echo
}
EXTRA="--minibatch 8 --T 50" run "8 lines of 50 frames" A=1
EXTRA="--minibatch 32" run "32 lines" A=1
run "64 lines, fused launches off (CLSTM_OVERLAP=0)" CLSTM_OVERLAP=0
