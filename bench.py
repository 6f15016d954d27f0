#!/usr/bin/env python
"""bench.py -- text-line images/sec (fwd + CTC + bwd + all-reduce + update) of the MI355X hot path.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
  (or as a plain `python bench.py --gpus N`: without WORLD_SIZE in the environment the script re-executes
  itself through torch.distributed.run); one rank per GPU; ranks shard the minibatch (independent lines,
  no data-path collective) and all-reduce the 135,883-float gradient buffer over RCCL -- inside the
  library, on its own stream (clstm_allreduce_flat) -- before the identical update (weak scaling:
  64 lines per GPU).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2]/[3]): uw3-500 OCR shape -- BiLSTM(100) on 48-px lines,
83 classes, T=200 frames, transcripts of 25 labels, minibatch = 64 lines per GPU, synthetic
inputs (clip(N(0.2,0.3),0,1) smoothed along t), reference LCG init (seed 0.222, negbiased).
A step = one pass of the hot path over one minibatch whose frames are already resident in HBM.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NI, NH, NC = 48, 100, 83
LABELS = 25
# --config b2: BASELINE.json configs[4] shape (2 x BiLSTM(512), H=64, T~400, 100 classes, 50 labels), f32
CONFIGS = {"b1": dict(ni=48, nh=[100], nc=83, T=200, L=25),
           "b2": dict(ni=64, nh=[512, 512], nc=100, T=400, L=50)}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TFS = 157.3      # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
# algorithmic bytes per cell-step of the fused gate kernels
BYTES_PER_CELL_STEP = {"lstm_fwd": 44.0, "lstm_bwd": 56.0}


def synth_batch(rng, bs, T, ragged):
    Ts = [int(t) for t in (rng.integers(150, 251, bs) if ragged else [T] * bs)]
    xs = []
    for t in Ts:
        x = np.clip(rng.normal(0.2, 0.3, (t + 2, NI)), 0, 1)
        xs.append(((x[:-2] + x[1:-1] + x[2:]) / 3.0).astype(np.float32))
    labels = [rng.integers(1, NC, LABELS).astype(np.int32) for _ in Ts]
    return Ts, np.concatenate(xs, 0), labels


def cpu_baseline(params, seconds_target=12.0):
    """The oracle (CPU restatement of the reference's Eigen path, `kind: port`) timed on this box's
    host cores on a bounded sample of the same workload: fwd+CTC+bwd of T=200 lines, OpenMP over
    lines (the generous 'Eigen/OpenMP' figure) and single-threaded."""
    from oracle.oracle import Oracle, OracleNet
    ora = Oracle("f32")
    net = OracleNet(ora, NI, NH, NC, init=False)
    net.set_params(params)
    rng = np.random.default_rng(123)
    cores = os.cpu_count() or 1
    nlines = max(8, cores)
    Ts, x, labels = synth_batch(rng, nlines, 200, False)
    offs = np.concatenate([[0], np.cumsum(Ts)])
    loffs = np.concatenate([[0], np.cumsum([len(l) for l in labels])])
    lab = np.concatenate(labels)
    t1 = net.bench_lines(x, offs, lab, loffs, nthreads=1, reps=1)       # also warms the page cache
    single = nlines / t1
    reps = max(1, int(seconds_target * single * min(cores, 4) / nlines))
    tn = net.bench_lines(x, offs, lab, loffs, nthreads=cores, reps=reps)
    multi = nlines * reps / tn
    use_multi = multi >= single
    return {
        "value": round(multi if use_multi else single, 2), "unit": "lines/s",
        "cores": cores if use_multi else 1, "kind": "port",
        "sample": "%d lines x %d reps of T=200 fwd+CTC+bwd, OpenMP over lines on %d threads "
                  "(%.1f lines/s); single thread %.1f lines/s; gcc -O3 -march=native" %
                  (nlines, reps, cores, multi, single),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--minibatch", type=int, default=64, help="lines per GPU")
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="b1")
    ap.add_argument("--bf16-gemm", action="store_true",
                    help="hoisted gate GEMMs with bf16 inputs / f32 accumulation (not the parity path)")
    ap.add_argument("--bf16", action="store_true",
                    help="--bf16-gemm plus bf16 MFMA operands inside the lock-step recurrence of wide layers "
                         "(BASELINE configs[4]: '2 x BiLSTM(512), bf16 MFMA'; not the parity path)")
    ap.add_argument("--ragged", action="store_true", help="T ~ U{150..250} instead of fixed T")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=5, help="extra steps with per-kernel hipEvent timing")
    args = ap.parse_args()
    global NI, NH, NC, LABELS
    cfg = CONFIGS[args.config]
    NI, NH, NC, LABELS = cfg["ni"], cfg["nh"], cfg["nc"], cfg["L"]
    if args.T is None:
        args.T = cfg["T"]
    nh_list = list(NH)
    NH = NH[0] if len(NH) == 1 else NH
    if args.config != "b1":
        args.no_cpu_baseline = True     # the bounded CPU sample is defined for the headline workload only

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # invoked like the single-GPU command: become the launcher -- one rank per GPU through torch.distributed.run
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    # ONE JSON line on stdout, nothing else: RCCL prints a version banner on the process's stdout, so fd 1 is pointed
    # at stderr for the whole run and the result goes to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d (launch with torch.distributed.run --nproc-per-node %d, "
                 "or without WORLD_SIZE and let bench.py spawn the ranks)" % (world, args.gpus, args.gpus))
    if torch.cuda.device_count() <= local_rank:
        sys.exit("bench.py: rank %d needs GPU %d but only %d visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):   # BENCH_FORCE_DIST=1: exercise the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from clstm_amd import abi
    from clstm_amd.init import init_params
    from clstm_amd.net import Comm, Network
    from clstm_amd.parallel import Trainer

    lib = abi.load()   # raises if the HIP extension is missing -- there is no fallback path
    # a real (non-default) stream: the library replays launch-bound loops as hipGraphs, and the legacy
    # default stream cannot be captured.  Everything below -- kernels, RCCL all-reduce -- runs on it.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    lib.call("clstm_set_stream", stream.cuda_stream)
    params_h = init_params(NI, NH, NC, seed=0.222)
    nparams = params_h.size
    dev = torch.device("cuda", local_rank)
    params = torch.from_numpy(params_h).to(dev)
    derivs = torch.zeros(nparams, device=dev)
    grads = torch.zeros(nparams, device=dev)
    net = Network(NI, NH, NC, lib=lib, params=params, derivs=derivs, grads=grads)
    net.params_changed()
    net.setLearningRate(1e-4, 0.9)
    if args.bf16:
        net.set_gemm_precision(2)
    elif args.bf16_gemm:
        net.set_gemm_precision(1)
    # gradient exchange: the library's own RCCL communicator (all-reduce enqueued on the library stream right
    # before the update kernel, no cross-stream events); torch.distributed only carries the 128-byte id, the
    # barriers and the max-over-ranks of the timing.  If the communicator cannot be created the step falls back to
    # torch.distributed.all_reduce on the gradient tensor and the JSON line says so.
    allreduce_impl = None
    comm = None
    if dist is not None:
        def exchange(ident):
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        try:
            comm = Comm(rank, world, exchange, lib=lib)
            allreduce_impl = "clstm_allreduce_flat (RCCL ncclAllReduce on the library stream)"
        except Exception as e:     # noqa: BLE001 -- report and fall back, never silently
            sys.stderr.write("bench.py: library communicator unavailable (%s); falling back to torch.distributed\n" % e)
            allreduce_impl = "torch.distributed.all_reduce (fallback: %s)" % type(e).__name__
    trainer = Trainer(net, grads_tensor=grads if comm is None else None, comm=comm)

    # synthetic minibatches resident in HBM before the timed region (a small rotating pool)
    rng = np.random.default_rng(1000 + rank)
    pool = []
    for _ in range(4):
        Ts, x, labels = synth_batch(rng, args.minibatch, args.T, args.ragged)
        pool.append((Ts, torch.from_numpy(x).to(dev), labels, Network.prepare_step(Ts, labels)))
    one_call = trainer.dist is None     # single GPU or library communicator: clstm_net_train_step

    def step(i):
        Ts, xd, labels, prep = pool[i % len(pool)]
        if one_call:
            net.train_step_prepared(prep, xd)     # CLSTMOCR::train for the minibatch: one C-ABI call, no host sync
        else:
            trainer.step_device(Ts, xd, labels)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    lines_total = args.minibatch * world * args.steps
    value = lines_total / dt
    # host-side cost of issuing a step (diagnostic: is the loop host-bound?): a short burst on an idle stream, few
    # enough steps that the library's 8-slot pinned staging ring never makes the host wait for the GPU
    t1 = time.perf_counter()
    for i in range(4):
        step(i)
    t_enqueue = (time.perf_counter() - t1) / 4
    fence()

    # per-kernel device time (HIP events on the library's stream) for the roofline object
    kern = {}
    roofline = None
    if rank == 0 and args.profile_steps > 0:
        net.enable_timing(True)
        net.reset_timing()
        frames = 0
        for i in range(args.profile_steps):
            step(i)
            frames += sum(pool[i % len(pool)][0])
        torch.cuda.synchronize()
        for name in ("gemm_gates_x", "lstm_fwd", "gemm_softmax", "softmax_norm", "ctc_align",
                     "gemm_softmax_dw_dx", "gemm_softmax_dx", "gemm_softmax_dw", "lstm_bwd", "gemm_gates_dw", "gemm_gates_dx",
                     "allreduce_grads", "sgd_update"):
            ms, n = net.kernel_time_ms(name)
            if n:
                kern[name] = {"ms_per_step": round(ms / args.profile_steps, 4), "launches_per_step": n / args.profile_steps}
        net.enable_timing(False)
        dom = max(("lstm_fwd", "lstm_bwd"), key=lambda k: kern.get(k, {"ms_per_step": 0})["ms_per_step"])
        if "gemm_gates_dw" not in kern:
            dom = "lstm_fwd"     # the backward recurrence shares its launch with the weight-gradient GEMM (lstm_bwd_dw.h):
                                 # the pure fused gate kernel of the step is the forward recurrence
        traffic, traffic_src = None, None
        try:   # HBM bytes per launch from the newest committed rocprofv3 PMC passes (same workload only)
            import glob
            pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_r*.json")))[-1]
            pmc = json.load(open(pmc_file))
            if (pmc["workload"]["minibatch_per_gpu"] == args.minibatch and pmc["workload"]["T"] == args.T
                    and not args.ragged and dom in pmc["kernels"]):
                traffic = pmc["kernels"][dom]["hbm_bytes"]
                traffic_src = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                               "committed -- not re-measured inside this run" % os.path.basename(pmc_file))
        except Exception:
            traffic = None
        if dom in kern:
            frames_per_step = frames / args.profile_steps
            cells = 2 * sum(nh_list)                                           # directions x cells, all layers
            byts = BYTES_PER_CELL_STEP[dom] * cells * frames_per_step         # per minibatch, all layers
            sec = kern[dom]["ms_per_step"] * 1e-3                              # same scope
            nl = kern[dom]["launches_per_step"]
            ach = byts / sec / 1e9
            if max(nh_list) > 128:
                # lock-step recurrence of a wide layer: a (lines x no).(no x 4no) product per step and direction --
                # priced against the matrix peak of the operand type (north star: "MFMA utilisation ... against gfx950 peak")
                peak = 2500.0 if args.bf16 else F32_MFMA_PEAK_TFS
                fl = (8.0 if dom == "lstm_fwd" else 8.0) * sum(h * h for h in nh_list) * 2 * frames_per_step
                tf = fl / sec / 1e12
                roofline = {"kernel": dom, "bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                            "frac": round(tf / peak, 5), "traffic": None,
                            "algorithmic_flops": int(fl / nl), "avg_launch_ms": round(sec / nl * 1e3, 4),
                            "note": ("lock-step recurrence, ONE persistent launch per pass: a workgroup group per XCD, %d group-barrier-separated steps (latency-bound); "
                                     if os.environ.get("CLSTM_XCD_REC", "1") != "0" and os.environ.get("CLSTM_COOP", "0") == "0" else
                                     "lock-step recurrence, one launch per time step (latency-bound: %d dependent launches per pass); ") % args.T +
                                    "whole step: %.1f TFLOP/s of algorithmic flops (SURVEY 8d: 48 T sum no(ni+no) + 6 T nc 2no per line)"
                                    % ((48.0 * args.T * sum(o * (i + o) for i, o in zip([NI] + [2 * h for h in nh_list[:-1]], nh_list))
                                                + 6.0 * args.T * NC * 2 * nh_list[-1]) * args.minibatch / (dt / args.steps) / 1e12)}
            else:
              roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "traffic_source": traffic_src,
                        "algorithmic_bytes": int(byts / nl),
                        "avg_launch_ms": round(sec / nl * 1e3, 4),
                        "note": "latency-bound recurrence (%s); recurrent matmul rate %.2f TFLOP/s of %.1f f32 peak" %
                                ("persistent, %d workgroups (lines x directions) on 256 CUs" % (2 * args.minibatch)
                                 if max(nh_list) <= 128 else "lock-step, one MFMA launch per time step",
                                 16.0 * sum(h * h for h in nh_list) * frames_per_step / sec / 1e12
                                 * (1.0 if dom == "lstm_fwd" else 1.0), F32_MFMA_PEAK_TFS)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(params_h)

    if rank == 0:
        out = {
            "metric": "text-line images/sec (fwd+bwd+CTC), 100-unit BiLSTM H=48 T~200" if args.config == "b1" else
                      "text-line images/sec (fwd+bwd+CTC), 2xBiLSTM(512) H=64 T~400 (%s)" % ("bf16 MFMA" if args.bf16 else "f32"),
            "value": round(value, 2), "unit": "lines/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 MFMA operands (hoisted gate GEMMs and lock-step recurrence), f32 accumulate / state / softmax / CTC" if args.bf16
                      else "f32 (hoisted gate GEMMs: bf16 in, f32 accumulate)" if args.bf16_gemm else "f32"),
            "data": "synthetic",
            "config": {"workload": ("uw3-500 OCR shape: BiLSTM(100) H=48 nc=83, T=%s, L=25, minibatch=%d lines/GPU "
                                    "(BASELINE.json configs[2]; x%d GPUs = configs[3] sharding), fwd+CTC+bwd+allreduce+update"
                                    % ("U{150..250}" if args.ragged else args.T, args.minibatch, world))
                                   if args.config == "b1" else
                                   ("stacked 2xBiLSTM(512) H=64 nc=100, T=%s, L=50, minibatch=%d lines/GPU x%d GPUs "
                                    "(BASELINE.json configs[4] shape), fwd+CTC+bwd+allreduce+update"
                                    % (args.T, args.minibatch, world)),
                       "minibatch_per_gpu": args.minibatch, "global_minibatch": args.minibatch * world,
                       "parallelism": "dp%d" % world},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kern,
            "host_enqueue_ms_per_step": round(t_enqueue * 1e3, 4),   # host-side cost of issuing a step
            "allreduce": allreduce_impl,
        }
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        net.set_comm(None)
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
