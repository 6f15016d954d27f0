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
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--minibatch", "2", "--T", "6", "--profile-steps", "2", "--no-cpu-baseline"]
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
    v = out["validity"]                                        # the timed steps updated the parameters (Workload.verify)
    assert v["device_errors"] == "none" and v["params_finite"] is True and v["max_param_change"] > 0 and v["legs"] == {}


def test_bench_refuses_steps_whose_updates_were_skipped():
    """bench.py's Workload.verify: after a sticky device error (a non-finite gradient, a fused launch that gave up) the library
    skips every later update, and the steps still run -- faster.  Found in round 6: the exact-f32 configs[4] leg reached a
    non-finite gradient at training step 43 of its synthetic trajectory and timed steps without updates from there on.
    verify() reads the parameters back (which raises a pending device error) and demands that they are finite and moved."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench

    class Net:
        def __init__(self, p):
            self.p = p

        def get_params(self):
            if self.p is None:
                raise RuntimeError("non-finite value (NaN or Inf) in the softmax logits or the gradient of training step 43")
            return self.p

    w = bench.Workload.__new__(bench.Workload)
    w.params0 = np.zeros(8, np.float32)
    w.net = Net(np.full(8, 0.5, np.float32))
    assert w.verify() == {"device_errors": "none", "params_finite": True, "max_param_change": 0.5}
    w.net = Net(np.zeros(8, np.float32))                       # nothing moved
    with pytest.raises(RuntimeError, match="did not update"):
        w.verify()
    w.net = Net(np.array([0.5, np.nan] * 4, np.float32))
    with pytest.raises(RuntimeError, match="did not update"):
        w.verify()
    w.net = Net(None)                                          # the library's own error comes through
    with pytest.raises(RuntimeError, match="non-finite"):
        w.verify()


@pytest.mark.gpu
def test_bench_gpus2_on_the_real_library_sharing_one_gpu():
    """the `--gpus 2` bench body on the REAL library before the driver's first multi-GPU run.  Two rank processes
    share device 0 (CLSTM_BENCH_SHARE_DEVICE=1; torch.distributed on gloo for the barriers and the max over ranks, because RCCL
    refuses two ranks on one device; CLSTM_COMM_NO_RCCL=1: the library communicator with the peer path only, as
    tests/test_distributed.py::test_two_processes_on_one_gpu_peer_read_allreduce).  What runs is what `--gpus 2` runs on two
    GPUs: the self-relaunch through torch.distributed.run, clstm_comm_create, HIP IPC mappings of the other rank's gradient
    slots, clstm_net_train_step with the all-reduce fused into the update kernel, the replica check, ONE JSON line -- except that
    ranks sharing a device run the forward / backward halves as separate launches (bench.py says why).
    (The first run of this test found a real bug: the per-kernel timing steps ran on rank 0 only, and with a communicator a step
    is a collective -- `bench.py --gpus N` would have hung in the driver's first multi-GPU run.)"""
    env = dict(os.environ, CLSTM_BENCH_BACKEND="gloo", CLSTM_BENCH_SHARE_DEVICE="1", CLSTM_COMM_NO_RCCL="1",
               CLSTM_BENCH_MIN_TIMED_S="0.2", CLSTM_BENCH_MIN_WARMUP_S="0.1", CLSTM_REPLICA_CHECK_EVERY="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CLSTM_BENCH_DEVICE", "CLSTM_BENCH_LIB"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_minibatch"] == 128 and out["config"]["minibatch_per_gpu"] == 64 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["value"] == out["value"] and out["value"] < 1e7          # finite, whole-job lines/s
    assert abs(out["value"] - 128 * 5 / (out["ms_per_step"] * 5e-3)) / out["value"] < 1e-2
    ar = out["allreduce"]
    assert ar["ranks"] == 2 and ar["bytes"] == 4 * 135883 and ar["peer_active"] is True
    assert "peer-read" in ar["impl"]
    assert r.stderr.count("clstm_comm_peer_active = 1") == 2       # every rank says at start-up which exchange it uses
    assert out["secondary"] is None and out["cpu_baseline"] is None and out["strict_f32"] is None


@pytest.mark.gpu
def test_bench_line_contract_on_one_gpu():
    """The driver's command on one GPU (short blocks): ONE JSON line with the contract's fields -- metric / value / unit / n_gpus /
    steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, the `roofline`
    and `cpu_baseline` objects, and `validity` (the timed steps updated the parameters); value = lines per step / ms_per_step."""
    env = dict(os.environ, CLSTM_BENCH_MIN_TIMED_S="0.2", CLSTM_BENCH_MIN_WARMUP_S="0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-secondary"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["metric"].startswith("text-line images/sec") and out["unit"] == "lines/s"
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["warmup"] == 2
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["data"] == "synthetic" and out["dtype"].startswith("f32")
    assert "workload" in out["config"] and out["config"]["minibatch_per_gpu"] == 64
    assert abs(out["value"] - 64 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-2
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf and "whole_step" in rf
    cb = out["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "lines/s" and cb["sample"]
    v = out["validity"]
    assert v["device_errors"] == "none" and v["params_finite"] is True and v["max_param_change"] > 0
    assert "clstm_net_train_step_next" in out["config"]["step_call"]
