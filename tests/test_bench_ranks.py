"""bench.py's rank body at world size 2 on the CPU: `bench.py --gpus N` re-launches itself through
torch.distributed.run, creates the library communicator, runs clstm_net_train_step with the peer-read exchange on every
rank, barriers, max-reduces the block times and prints ONE JSON line on rank 0 -- none of which had ever executed with
N > 1 before the driver's first multi-GPU run.  Here the same file runs under its test hooks (CLSTM_BENCH_BACKEND=gloo,
CLSTM_BENCH_DEVICE=cpu, CLSTM_BENCH_LIB=the host-emulator build of the kernels): same control flow, same collectives."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("how", ["self_relaunch", "under_torch_distributed_run"])
def test_bench_rank_body_world2_on_the_emulator(how):
    from common import EMU_DIR, emu_lib
    emu_lib()
    env = dict(os.environ, CLSTM_BENCH_BACKEND="gloo", CLSTM_BENCH_DEVICE="cpu",
               CLSTM_BENCH_LIB=os.path.join(EMU_DIR, "build", "libclstm_emu.so"),
               CLSTM_BENCH_MIN_TIMED_S="0.05", CLSTM_BENCH_MIN_WARMUP_S="0.01", CLSTM_REPLICA_CHECK_EVERY="2",
               OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--minibatch", "2", "--T", "6", "--profile-steps", "0", "--no-cpu-baseline"]
    if how == "self_relaunch":      # the plain command: the script becomes the launcher
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:                           # the driver's command line
        port = 23000 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # ONE JSON line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["global_minibatch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and abs(out["value"] - 4 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-2   # whole-job lines/s (both figures are rounded in the JSON)
    ar = out["allreduce"]
    assert ar["ranks"] == 2 and ar["bytes"] == 4 * 135883
    assert "peer-read" in ar["impl"]                           # the one-call step ran the fused exchange, not a fallback
    assert out["secondary"] is None and out["cpu_baseline"] is None      # single-GPU legs stay out of a multi-rank line
