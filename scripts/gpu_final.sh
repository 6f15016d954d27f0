#!/bin/bash
# end-of-round confirmation on the GPU box: the whole -m gpu suite, the default bench line, the configs[4] lines and a
# kernel trace of the configs[4] bf16 step; results under gpurun_out/fin/ (copied to profiles/ by hand)
OUT=$GRAFT_REPO_ROOT/gpurun_out/fin; mkdir -p $OUT
timeout 600 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-160 $OUT/bench_default.json
timeout 200 python bench.py --config b2 --bf16 > $OUT/bench_b2_bf16.json 2> $OUT/bench_b2_bf16.err; cut -c1-330 $OUT/bench_b2_bf16.json
if [ -z "$FAST" ]; then
timeout 200 python bench.py --config b2 --no-cpu-baseline > $OUT/bench_b2.json 2> $OUT/bench_b2.err; cut -c1-200 $OUT/bench_b2.json
timeout 200 python bench.py --config b2 --bf16-gemm --no-cpu-baseline > $OUT/bench_b2_bf16gemm.json 2> $OUT/bench_b2_bf16gemm.err; cut -c1-200 $OUT/bench_b2_bf16gemm.json
fi
timeout 300 bash scripts/gpu_b2trace.sh fin > $OUT/b2trace.log 2>&1; cp $GRAFT_REPO_ROOT/gpurun_out/fin/b2_step_gaps_base.txt $OUT/b2_step_kernels.txt 2>/dev/null; head -14 $OUT/b2_step_kernels.txt
