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

Timing: W untimed warm-up steps (at least 0.3 s worth -- a 20-step command must not time clock ramp and first-touch
effects), then the block of EXACTLY K steps, bracketed by barrier + synchronize on both sides and max-reduced over the
ranks, is timed R times back to back (R such that the timed total is >= 0.5 s; `repeats` in the JSON) and the MEDIAN
block is reported: `ms_per_step` = median block / K, `value` = lines of one block / median block.

Test hooks (tests/test_bench_ranks.py runs this file's rank body at world size 2 on the CPU before the driver runs it on
eight GPUs): CLSTM_BENCH_BACKEND=gloo (process-group backend), CLSTM_BENCH_DEVICE=cpu (tensors in host memory, no streams),
CLSTM_BENCH_LIB=<path of a build of the library>, CLSTM_BENCH_MIN_TIMED_S / CLSTM_BENCH_MIN_WARMUP_S (bounds of the timing
protocol).  None is set in a measurement.

The default single-GPU line also carries `secondary`: BASELINE.json configs[4] (2 x BiLSTM(512), H = 64, T = 400,
bf16 MFMA) timed in the same process with the same protocol and its own `roofline` (MFMA, 2.5 PFLOP/s dense bf16).
"""
import argparse
import json
import os
import sys
import time

# thread placement of the cpu_baseline leg (the oracle's OpenMP loop over lines): fixed BEFORE numpy -- whose BLAS may start an
# OpenMP runtime at import -- so that libgomp reads it; set after, the figure moved 707...960 lines/s between runs of one box
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "threads")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# b1: BASELINE.json configs[1..3] (uw3 shape); b2: configs[4] (2 x BiLSTM(512), H=64, T~400, 100 classes, 50 labels)
CONFIGS = {"b1": dict(ni=48, nh=[100], nc=83, T=200, L=25),
           # lr: at the 1e-4 of b1 the f32 trajectory of this net on the synthetic (unlearnable) minibatches reaches a non-finite
           # gradient at training step 43 (exploding gradient through 2 x 400 steps: |g| 8e19 at step 39, scripts/dbg/b2_divergence.py),
           # where the reference aborts (clstm.cc:630-649) and the library skips every later update -- Workload.verify refuses that
           "b2": dict(ni=64, nh=[512, 512], nc=100, T=400, L=50, lr=1e-6)}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TFS = 157.3      # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
BF16_MFMA_PEAK_TFS = 2500.0    # dense bf16 MFMA
# algorithmic bytes per cell-step of the fused gate kernels
BYTES_PER_CELL_STEP = {"lstm_fwd": 44.0, "lstm_bwd": 56.0}
MIN_WARMUP_S, MIN_TIMED_S, MAX_REPEATS = 0.3, 2.0, 400   # (2 s timed: the driver's 5-s GPU-busy sampler sees the work)
if os.environ.get("CLSTM_BENCH_MIN_TIMED_S"):
    MIN_TIMED_S = float(os.environ["CLSTM_BENCH_MIN_TIMED_S"])
if os.environ.get("CLSTM_BENCH_MIN_WARMUP_S"):
    MIN_WARMUP_S = float(os.environ["CLSTM_BENCH_MIN_WARMUP_S"])
ON_CPU = os.environ.get("CLSTM_BENCH_DEVICE") == "cpu"      # test hook: the rank body on host memory (no GPU, no streams)


def device_sync(lib=None):
    if ON_CPU:
        return
    import torch
    torch.cuda.synchronize()
KERNEL_NAMES = ("ingest", "gemm_gates_x", "lstm_fwd", "gemm_softmax", "softmax_norm", "ctc_align", "gemm_softmax_dw_dx",
                "lstm_bwd", "gemm_gates_dw", "reduce_scatter", "gemm_gates_dx", "allreduce_grads", "sgd_update")


def synth_batch(rng, bs, T, ragged, ni, nc, L):
    Ts = [int(t) for t in (rng.integers(150, 251, bs) if ragged else [T] * bs)]
    xs = []
    for t in Ts:
        x = np.clip(rng.normal(0.2, 0.3, (t + 2, ni)), 0, 1)
        xs.append(((x[:-2] + x[1:-1] + x[2:]) / 3.0).astype(np.float32))
    labels = [rng.integers(1, nc, L).astype(np.int32) for _ in Ts]
    return Ts, np.concatenate(xs, 0), labels


def flops_per_line(cfg, T):
    """48 T sum_l no(ni+no) + 6 T nc 2no_last"""
    nis = [cfg["ni"]] + [2 * h for h in cfg["nh"][:-1]]
    return 48.0 * T * sum(o * (i + o) for i, o in zip(nis, cfg["nh"])) + 6.0 * T * cfg["nc"] * 2 * cfg["nh"][-1]


def cpu_topology():
    """(one logical CPU per physical core, all logical CPUs) among the CPUs this process may run on"""
    allowed = sorted(os.sched_getaffinity(0))
    seen, phys = set(), []
    for c in allowed:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
            pkg = open("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % c).read().strip()
            key = (pkg, sib)
        except OSError:
            key = c
        if key not in seen:
            seen.add(key)
            phys.append(c)
    return phys, allowed


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cfs quota), or None: threads beyond it are
    throttled, not run (the GPU box of this pool: 256 hardware threads visible, cpu.max = 16 CPUs -- 128 threads gave the same
    1.4k lines/s as 16)"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


CPU_TOPOLOGY = cpu_topology()     # at import: once libgomp is loaded (OMP_PROC_BIND above) the main thread is bound to one CPU


def cpu_baseline(params, cfg, seconds_target=12.0):
    """The oracle (CPU restatement of the reference's Eigen path, `kind: port`) timed on this box's host cores on a bounded
    sample of the same workload: fwd+CTC+bwd of T=200 lines -- single-threaded (what scons + Eigen executes: its tensor
    contractions on DefaultDevice do not thread) and OpenMP over lines (the generous 'Eigen/OpenMP' figure) with every thread
    pinned to its own CPU, on its own preallocated net, one untimed line first (oracle/clstm_oracle.c:ora_bench_lines_pinned),
    once on one thread per PHYSICAL core and once on every hardware thread; the best of the two is `value`."""
    from oracle.oracle import Oracle, OracleNet
    ora = Oracle("f32")
    net = OracleNet(ora, cfg["ni"], cfg["nh"], cfg["nc"], init=False)
    net.set_params(params)
    rng = np.random.default_rng(123)
    phys, logical = CPU_TOPOLOGY
    quota = cpu_quota()
    cap = len(logical) if quota is None else max(1, int(quota))
    nlines = 64
    Ts, x, labels = synth_batch(rng, nlines, 200, False, cfg["ni"], cfg["nc"], cfg["L"])
    offs = np.concatenate([[0], np.cumsum(Ts)])
    loffs = np.concatenate([[0], np.cumsum([len(l) for l in labels])])
    lab = np.concatenate(labels)
    t1 = net.bench_lines(x[:offs[16]], offs[:17], lab[:loffs[16]], loffs[:17], nthreads=1, reps=1, cpus=logical[:1])
    single = 16 / t1
    legs = {}
    for name, cpus in (("physical", phys[:cap]), ("logical", logical[:cap] if len(logical) > len(phys) and cap > len(phys) else None)):
        if cpus is None:
            continue
        n = len(cpus)
        # calibrate on 2 lines per thread, then two runs of ~seconds_target / 5 each (every thread at least 4 lines)
        r0 = max(1, int(np.ceil(2.0 * n / nlines)))
        rate0 = nlines * r0 / net.bench_lines(x, offs, lab, loffs, nthreads=n, reps=r0, cpus=cpus)
        reps = max(1, int(np.ceil(max(seconds_target / 5.0 * rate0, 4.0 * n) / nlines)))
        runs = sorted(nlines * reps / net.bench_lines(x, offs, lab, loffs, nthreads=n, reps=reps, cpus=cpus) for _ in range(2))
        legs[name] = {"threads": n, "lines_per_s": round(runs[-1], 1), "runs": [round(r, 1) for r in runs],
                      "parallel_efficiency": round(runs[-1] / (single * n), 3), "lines_timed": nlines * reps}
    best = max(legs, key=lambda k: legs[k]["lines_per_s"])
    multi = legs[best]["lines_per_s"]
    use_multi = multi >= single
    return {
        "value": round(multi if use_multi else single, 2), "unit": "lines/s",
        "cores": legs[best]["threads"] if use_multi else 1, "kind": "port",
        "single_thread_lines_per_s": round(single, 1),
        "parallel_efficiency": legs[best]["parallel_efficiency"] if use_multi else 1.0,
        "physical_cores": len(phys), "hardware_threads": len(logical), "cgroup_cpu_quota": quota, "legs": legs,
        "sample": "T=200 lines, fwd+CTC+bwd each (the reference's per-line work incl. its memset per Sequence resize): single thread "
                  "%.1f lines/s over 16 lines; OpenMP over lines on %s, threads pinned one per physical core, per-thread preallocated "
                  "nets, one untimed line per thread first, best of 2 runs of %d lines: %s; gcc -O3 -march=native" %
                  (single,
                   ("%d threads = the container's cgroup CPU quota (%.0f of the host's %d cores / %d hardware threads; threads beyond "
                    "the quota are throttled, not run)" % (cap, quota, len(phys), len(logical))) if quota is not None and cap < len(logical)
                   else "every core (%d physical, %d hardware threads)" % (len(phys), len(logical)),
                   legs[best]["lines_timed"],
                   "; ".join("%s: %d threads %.1f lines/s (parallel efficiency %.2f)" % (k, v["threads"], v["lines_per_s"], v["parallel_efficiency"])
                             for k, v in legs.items())),
    }


def trained_weights_and_lines(lib, cfg, rng, Ts_list):
    """SURVEY.md 8(d)'s "trained-like" regime WITHOUT the oracle (bench.py may use oracle/ for cpu_baseline only): the uw3 net is
    trained by the HIP path itself -- 500 online-SGD steps (CLSTMOCR::train, clstmhl.h:201-223; lr 1e-2) on the reference's OCR
    fixture line, the recipe tests/trained_weights.py runs on the oracle -- and the minibatches are jittered crops of that
    normalised line (clstm_amd/fixture.py), each with the transcript the trained net decodes on it.  Returns (parameters,
    [(Ts, x, labels)], info)."""
    from clstm_amd.fixture import GT, fixture_frames, fixture_transcript, jittered_crops
    from clstm_amd.init import init_params
    from clstm_amd.net import Network
    assert cfg["ni"] == 48 and len(cfg["nh"]) == 1, "the fixture is a 48-row line for the one-layer uw3 net"
    x, tr = fixture_frames(), fixture_transcript()
    net = Network(cfg["ni"], cfg["nh"][0], cfg["nc"], lib=lib)
    net.set_params(init_params(cfg["ni"], cfg["nh"][0], cfg["nc"], seed=0.222))
    net.setLearningRate(1e-2, 0.9)
    for _ in range(500):
        net.set_inputs([x]); net.forward(); net.ctc([tr]); net.backward(); net.update()
    net.set_inputs([x]); net.forward()
    reads = net.decode()[0].tolist() == tr.tolist()
    params = net.get_params().copy()
    batches, nlab = [], []
    for Ts in Ts_list:
        lines = jittered_crops(rng, Ts)
        net.set_inputs(lines); net.forward()
        labels = [np.asarray(d, np.int32) if len(d) and 2 * len(d) + 1 <= t else np.array([1], np.int32) for d, t in zip(net.decode(), Ts)]
        nlab += [len(l) for l in labels]
        batches.append((Ts, np.concatenate(lines, 0), labels))
    out = net.split(net.outputs())
    info = {"weights": "500 online-SGD steps of this library on tests/golden/textline.bin.png (lr 1e-2, momentum 0.9)",
            "reads_fixture": bool(reads), "ground_truth": GT, "max_abs_parameter": round(float(np.abs(params).max()), 3),
            "mean_max_posterior": round(float(np.mean([o.max(1).mean() for o in out])), 4),
            "labels_per_line_mean": round(float(np.mean(nlab)), 1),
            "inputs": "jittered crops of the normalised fixture line (clstm_amd/fixture.py), transcripts = the trained net's own decodes"}
    return params, batches, info


class Workload:
    """One network + a small rotating pool of synthetic minibatches resident in HBM."""

    def __init__(self, lib, cfg, minibatch, T, ragged, precision, dev, rank, comm=None, dist=None, host_inputs=False, strict_f32=False,
                 weights="init"):
        import torch
        from clstm_amd.init import init_params
        from clstm_amd.net import Network
        from clstm_amd.parallel import Trainer
        self.cfg, self.minibatch, self.T, self.ragged, self.precision = cfg, minibatch, T, ragged, precision
        nh = cfg["nh"][0] if len(cfg["nh"]) == 1 else cfg["nh"]
        self.params_h = init_params(cfg["ni"], nh, cfg["nc"], seed=0.222)
        rng = np.random.default_rng(1000 + rank)
        self.weights_info = None
        batches = None
        if weights == "trained":
            Ts_list = [[int(t) for t in (rng.integers(150, 251, minibatch) if ragged else [T] * minibatch)] for _ in range(4)]
            self.params_h, batches, self.weights_info = trained_weights_and_lines(lib, cfg, rng, Ts_list)
        n = self.params_h.size
        self.params0 = self.params_h.copy()     # (on a CPU device the tensor below aliases params_h)
        self.params = torch.from_numpy(self.params_h).to(dev)
        self.derivs = torch.zeros(n, device=dev)
        self.grads = torch.zeros(n, device=dev)
        self.net = Network(cfg["ni"], nh, cfg["nc"], lib=lib, params=self.params, derivs=self.derivs, grads=self.grads)
        self.net.params_changed()
        self.lr = cfg.get("lr", 1e-4)
        self.net.setLearningRate(self.lr, 0.9)
        if precision:
            self.net.set_gemm_precision(precision)
        if strict_f32:
            self.net.set_strict_f32(True)
        self.trainer = Trainer(self.net, grads_tensor=self.grads if (comm is None and dist is not None) else None, comm=comm)
        self.pool = []
        for k in range(4):
            Ts, x, labels = batches[k] if batches else synth_batch(rng, minibatch, T, ragged, cfg["ni"], cfg["nc"], cfg["L"])
            xt = torch.from_numpy(x).pin_memory() if host_inputs else torch.from_numpy(x).to(dev)
            self.pool.append((Ts, xt, labels, Network.prepare_step(Ts, labels)))
        self.host_inputs = host_inputs
        self.one_call = self.trainer.dist is None     # single GPU or library communicator: clstm_net_train_step
        self.declare_next = os.environ.get("CLSTM_BENCH_DECLARE_NEXT", "1") != "0"   # (--no-declare-next)

    def step(self, i):
        Ts, xd, labels, prep = self.pool[i % len(self.pool)]
        if self.host_inputs:
            self.net.train_step_host(prep, xd)         # frames in (pinned) host memory: copy stream + double-buffered device input
        elif self.one_call and self.declare_next:
            # ... and the loop knows its next minibatch (a loader one batch ahead): clstm_net_train_step_next -- the next
            # step's ingest rides this step's last launch; every step still ingests exactly one minibatch
            _, nxd, _, nprep = self.pool[(i + 1) % len(self.pool)]
            self.net.train_step_prepared(prep, xd, nprep, nxd)
        elif self.one_call:
            self.net.train_step_prepared(prep, xd)     # CLSTMOCR::train for the minibatch: one C-ABI call, no host sync
        else:
            self.trainer.step_device(Ts, xd, labels)

    def frames(self, i):
        return sum(self.pool[i % len(self.pool)][0])

    def verify(self):
        """The timed steps did their work.  A non-finite logit or gradient, or a fused launch that gave up, makes the library
        skip that update and every later one (sticky device error words, csrc/runtime.inc:check_device_errors) -- the steps
        would still be timed, and faster.  Reading the parameters back raises such an error; they must be finite and must
        have moved."""
        p = self.net.get_params()
        moved = float(np.abs(p - self.params0).max())
        if not np.isfinite(p).all() or not moved > 0.0:
            raise RuntimeError("bench: the timed steps did not update the parameters (finite: %s, largest change %g)" % (bool(np.isfinite(p).all()), moved))
        return {"device_errors": "none", "params_finite": True, "max_param_change": moved}


def timed_blocks(w, steps, warmup, fence, reduce_max, min_timed_s=None):
    """warm-up (>= `warmup` steps and >= MIN_WARMUP_S), then R blocks of exactly `steps` steps; returns the block times"""
    i = 0
    t0 = time.perf_counter()
    for _ in range(warmup):
        w.step(i)
        i += 1
    fence()
    # the time criterion looks at the max over ranks (a collective: every rank takes the same number of rounds)
    while reduce_max(time.perf_counter() - t0) < MIN_WARMUP_S:
        for _ in range(max(1, min(steps, 16))):
            w.step(i)
            i += 1
        fence()
    blocks = []
    repeats = 1
    k = 0
    while k < repeats:
        fence()
        t0 = time.perf_counter()
        for j in range(steps):
            w.step(i + j)
        fence()
        dt = reduce_max(time.perf_counter() - t0)
        i += steps
        blocks.append(dt)
        if k == 0:
            repeats = int(min(MAX_REPEATS, max(1, np.ceil((MIN_TIMED_S if min_timed_s is None else min_timed_s) / max(dt, 1e-9)))))   # identical on every rank: dt is max-reduced
        k += 1
    return blocks, i


def kernel_times(w, steps, first_step, ms_per_step):
    """per-kernel device time over `steps` extra steps: HIP events bound to each launch's own dispatch packet on the library's
    stream (hipExtLaunchKernel start / stop events, csrc/devintrin.h) -- the packet time stamps rocprofv3 --kernel-trace reads"""
    import torch
    net = w.net
    net.enable_timing(True)
    # lead-in with the timing on, not counted: ~25 ms of steps.  The first launches with profiled dispatch packets run slower
    # (profiles/README.md: the fused launches read 116 us, settling at the rocprofv3 figure of 107 us within ~40 steps), and
    # the lead-in fills the library's event pool, so the measured steps create no events
    for i in range(int(min(100, max(3, np.ceil(25.0 / max(ms_per_step, 1e-3)))))):
        w.step(first_step + i)
    device_sync()
    net.reset_timing()
    frames = 0
    for i in range(steps):
        w.step(first_step + i)
        frames += w.frames(first_step + i)
    device_sync()
    kern = {}
    for name in KERNEL_NAMES:
        ms, n = net.kernel_time_ms(name)
        if n:
            kern[name] = {"ms_per_step": round(ms / steps, 4), "launches_per_step": n / steps}
    net.enable_timing(False)
    return kern, frames / steps


def pmc_traffic(minibatch, T, ragged, strict=False):
    """HBM bytes per launch from the newest committed rocprofv3 PMC passes (same workload only: the default line, its strict_f32
    form -- `legs.strict` -- and the 256-line form -- `legs.mb256`)"""
    try:
        import glob
        pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_r*.json")))[-1]
        pmc = json.load(open(pmc_file))
        if T != pmc["workload"]["T"] or ragged:
            return {}, None
        leg = None
        if minibatch == pmc["workload"]["minibatch_per_gpu"]:
            leg = pmc.get("legs", {}).get("strict") if strict else pmc
        elif minibatch == 256 and not strict:
            leg = pmc.get("legs", {}).get("mb256")
        if leg:
            src = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command%s, committed -- "
                   "not re-measured inside this run" % (os.path.basename(pmc_file), " (--strict-f32)" if strict else
                                                        " (--minibatch 256)" if minibatch == 256 else ""))
            return {k.split(" ")[0]: v["hbm_bytes"] for k, v in leg["kernels"].items()}, src
    except Exception:
        pass
    return {}, None


def rocprof_avg_ms(kernel, minibatch, T, ragged, strict=False):
    """average duration (ms) of `kernel` in the newest committed rocprofv3 --kernel-trace --stats summary of this command
    (the default workload, its --strict-f32 form and its --minibatch 256 form; None otherwise) -- for the reader to hold against
    avg_launch_ms"""
    try:
        import csv
        import glob
        if T != 200 or ragged or (minibatch, strict) not in ((64, False), (64, True), (256, False)):
            return None
        suffix = "_strict" if strict else "_mb256" if minibatch == 256 else ""
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats%s.csv" % suffix)))[-1]
        for row in csv.DictReader(open(f)):
            if kernel in row["Name"]:
                return {"ms": round(float(row["AverageNs"]) * 1e-6, 4), "source": "profiles/" + os.path.basename(f),
                        "note": "COMMITTED rocprofv3 --kernel-trace --stats summary of this command from an earlier run -- not measured in this run"}
    except Exception:
        pass
    return None


def rocprof_b2_avg_ms(dominant):
    """average duration (ms) of the configs[4] step's dominant launch (`dominant`: "lstm_fwd" | "lstm_bwd" -> the persistent
    lstm_xcd_fwd_bf16* / lstm_xcd_bwd_bf16* kernels) in the newest committed rocprofv3 --kernel-trace --stats summary of
    `bench.py --config b2 --bf16` (profiles/r*_b2_kernel_stats.csv: call-weighted over the matching kernels), else in the one-step
    timeline of the same trace (profiles/r*_b2_timeline.txt)"""
    try:
        import csv
        import glob
        pat = {"lstm_fwd": "lstm_xcd_fwd_bf16", "lstm_bwd": "lstm_xcd_bwd_bf16"}.get(dominant)
        if not pat:
            return None
        note = "COMMITTED rocprofv3 summary of `bench.py --config b2 --bf16` from an earlier run -- not measured in this run"
        fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_b2_kernel_stats.csv")))
        if fs:
            tot = calls = 0.0
            for row in csv.DictReader(open(fs[-1])):
                if pat in row["Name"]:
                    tot += float(row["TotalDurationNs"]); calls += float(row["Calls"])
            if calls:
                return {"ms": round(tot / calls * 1e-6, 4), "source": "profiles/" + os.path.basename(fs[-1]), "note": note}
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_b2_timeline.txt")))[-1]
        d = [float(l.split()[2]) for l in open(f) if pat in l]
        return {"ms": round(sum(d) / len(d) * 1e-3, 4), "source": "profiles/" + os.path.basename(f), "note": note} if d else None
    except Exception:
        return None


def roofline_b1(w, kern, kern_unfused, frames_per_step, ms_per_step, strict=False):
    """Fused gate kernels against HBM (north star: 'achieved HBM GB/s for the fused gate kernel'), the batched gate GEMM
    against the f32 MFMA peak ('MFMA utilisation for the batched gate GEMM').  Algorithmic bytes:
    forward recurrence 44 B per cell-step, backward recurrence 56 B per cell-step.  In the default mode both recurrences
    share their launch with the products around them -- forward: the W_x GEMM producers (read x, write G) and the softmax
    consumers (read H rows, write Z), lstm_fwd_fused.h; backward: the weight-gradient items (read deltas and source rows
    once, for the top layer also the softmax layer's [1 | h] rows and output deltas), lstm_bwd_dw.h -- whose bytes are
    added for those launches.  `others` carries the PURE kernels, timed in a second short pass with the fusions off
    (clstm_net_set_overlap 0): the recurrences alone and the batched gate GEMM."""
    cfg = w.cfg
    ndir, N = 2, frames_per_step
    cells = ndir * sum(cfg["nh"])
    no, ni, nc = cfg["nh"][-1], cfg["ni"], cfg["nc"]
    M = ndir * 4 * cfg["nh"][0]
    lds = (1 + ni + no + 15) // 16 * 16
    ldh = (4 + ndir * no + 15) // 16 * 16
    traffic, traffic_src = pmc_traffic(w.minibatch, w.T, w.ragged, strict)
    entries = {}

    def hbm_entry(key, label, byts, k):
        sec = k["ms_per_step"] * 1e-3
        nl = k["launches_per_step"]
        ach = byts / sec / 1e9
        return {"kernel": label, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic.get(key),
                "algorithmic_bytes": int(byts / nl), "avg_launch_ms": round(sec / nl * 1e3, 4)}
    if "lstm_fwd" in kern:
        fused = "gemm_gates_x" not in kern
        byts = 44.0 * cells * N + (4.0 * N * (ni + M) + 4.0 * N * (ndir * no + nc) if fused else 0.0)
        entries["lstm_fwd_fused" if fused else "lstm_fwd"] = hbm_entry("lstm_fwd_fused" if fused else "lstm_fwd",
            "lstm_fwd_fused (W_x producers + recurrence + softmax consumers)" if fused else "lstm_fwd", byts, kern["lstm_fwd"])
    if "lstm_bwd" in kern:
        fused = "gemm_gates_dw" not in kern
        byts = 56.0 * cells * N + (4.0 * N * (ndir * 4 * no + ndir * lds) + 4.0 * N * (ldh + nc) if fused else 0.0)
        entries["lstm_bwd_dw" if fused else "lstm_bwd"] = hbm_entry("lstm_bwd_dw" if fused else "lstm_bwd",
            "lstm_bwd_dw (recurrence + weight-gradient items)" if fused else "lstm_bwd", byts, kern["lstm_bwd"])
    dom_keys = list(entries)
    for name, bpc in (("lstm_fwd", 44.0), ("lstm_bwd", 56.0)):      # the pure fused gate kernels
        if name in kern_unfused and name not in entries:
            entries[name] = hbm_entry(name, name + " (fusions off)", bpc * cells * N, kern_unfused[name])
    kx = kern.get("gemm_gates_x") or kern_unfused.get("gemm_gates_x")
    if kx:
        fl = 2.0 * N * M * (ni + 1)
        sec = kx["ms_per_step"] * 1e-3
        tf = fl / sec / 1e12
        entries["gemm_gates_x"] = {"kernel": "gemm_gates_x", "bound": "mfma", "achieved": round(tf, 2), "peak": F32_MFMA_PEAK_TFS,
                                   "unit": "TFLOP/s", "frac": round(tf / F32_MFMA_PEAK_TFS, 5), "traffic": traffic.get("gemm_gates_x"),
                                   "algorithmic_flops": int(fl), "avg_launch_ms": round(sec * 1e3, 4),
                                   "note": "batched gate GEMM W_x.x + b over every frame of the minibatch (f32 MFMA); "
                                           "its 4 B x N x M output stream (%.0f MB) is what bounds it" % (4e-6 * N * M)}
    if not dom_keys:
        return None
    dom = max(dom_keys, key=lambda k: entries[k]["avg_launch_ms"])     # the dominant launch of the step BY TIME
    out = dict(entries[dom])
    out["traffic_source"] = traffic_src if out["traffic"] is not None else None
    out["timing"] = ("avg_launch_ms: HIP start/stop events bound to the launch's own dispatch packet on the library's stream "
                     "(hipExtLaunchKernel), measured live in this run; rocprof_avg_launch_ms_from_committed_profile: the rocprofv3 "
                     "--kernel-trace --stats average of the same command in the named file under profiles/ (an earlier run)")
    out["rocprof_avg_launch_ms_from_committed_profile"] = rocprof_avg_ms({"lstm_bwd_dw": "lstm_bwd_dw_kernel", "lstm_fwd_fused": "lstm_fwd_fused_kernel",
                                                   "lstm_fwd": "lstm_fwd_kernel", "lstm_bwd": "lstm_bwd_kernel"}.get(dom, dom),
                                                  w.minibatch, w.T, w.ragged, strict)
    # the whole step against both roofs (3.62 MB + 162.0 MFLOP per line, 2.17 MB per minibatch)
    step_bytes = 3.62e6 * w.minibatch + 2.17e6
    step_flops = flops_per_line(cfg, w.T) * w.minibatch if not w.ragged else None
    sec = ms_per_step * 1e-3
    # the launches of THIS step that the PMC passes cover (the fused pair when the step ran fused; never both forms)
    in_step = [k for k in (("lstm_fwd_fused" if "gemm_gates_x" not in kern else "lstm_fwd"),
                           ("lstm_bwd_dw" if "gemm_gates_dw" not in kern else "lstm_bwd"), "ctc_align") if k in traffic]
    step_traffic = sum(traffic[k] for k in in_step) if len(in_step) == 3 else None
    out["whole_step"] = {
        "algorithmic_bytes": int(step_bytes), "achieved_GBps": round(step_bytes / sec / 1e9, 1), "hbm_frac": round(step_bytes / sec / 1e9 / HBM_PEAK_GBS, 4),
        "algorithmic_flops": None if step_flops is None else int(step_flops),
        "achieved_TFLOPs": None if step_flops is None else round(step_flops / sec / 1e12, 2),
        "f32_mfma_frac": None if step_flops is None else round(step_flops / sec / 1e12 / F32_MFMA_PEAK_TFS, 4),
        "traffic_all_profiled_kernels": step_traffic,
        "traffic_over_algorithmic": None if not step_traffic else round(step_traffic / step_bytes, 2),
        "traffic_kernels": in_step if step_traffic else None,
        "note": "SURVEY.md 8(d): 3.62 MB + 162.0 MFLOP per line (+ 2.17 MB per minibatch); traffic = per-launch HBM bytes of the "
                "committed PMC passes summed over the step's forward launch, CTC launch and backward launch (the small x.d / reduce / "
                "ingest launches are not in those passes' summary)"}
    out["note"] = ("latency-bound recurrence: %d workgroups (lines x directions) on 256 CUs, one dependent step per frame; "
                   "whole step %.1f GB/s of algorithmic bytes (SURVEY 8d: 3.62 MB/line + 2.17 MB/minibatch)"
                   % (2 * w.minibatch, (3.62e6 * w.minibatch + 2.17e6) / (ms_per_step * 1e-3) / 1e9))
    out["others"] = {k: v for k, v in entries.items() if k != dom}
    return out


def roofline_b2(w, kern, frames_per_step, ms_per_step):
    """Lock-step recurrence of wide layers: a (lines x no).(no x 4no) product per step and direction, priced against
    the matrix peak of the operand type (north star: 'MFMA utilisation ... against gfx950 peak')."""
    cfg = w.cfg
    bf16 = w.precision == 2
    peak = BF16_MFMA_PEAK_TFS if bf16 else F32_MFMA_PEAK_TFS
    cand = [k for k in ("lstm_fwd", "lstm_bwd") if k in kern]
    if not cand:
        return None
    dom = max(cand, key=lambda k: kern[k]["ms_per_step"])
    sec = kern[dom]["ms_per_step"] * 1e-3
    nl = kern[dom]["launches_per_step"]
    fl = 8.0 * sum(h * h for h in cfg["nh"]) * 2 * frames_per_step
    tf = fl / sec / 1e12
    whole = flops_per_line(cfg, w.T) * w.minibatch / (ms_per_step * 1e-3) / 1e12
    return {"kernel": dom, "bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 5),
            "traffic": None, "algorithmic_flops": int(fl / nl), "avg_launch_ms": round(sec / nl * 1e3, 4),
            "whole_step": {"achieved": round(whole, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(whole / peak, 5),
                           "algorithmic_flops_per_line": int(flops_per_line(cfg, w.T))},
            "note": "lock-step recurrence, one persistent launch per layer pass: a workgroup group per XCD, %d group-barrier-separated "
                    "steps (latency-bound); whole step: SURVEY 8d flops (48 T sum no(ni+no) + 6 T nc 2no per line)" % w.T}


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
    ap.add_argument("--host-inputs", action="store_true",
                    help="frames start in pinned HOST memory every step (clstm_net_train_step_h): the PCIe-inclusive rate, "
                         "not the headline `value` (bench contract: inputs resident in HBM)")
    ap.add_argument("--strict-f32", action="store_true",
                    help="every product on the f32 MFMA (clstm_net_set_strict_f32) as the MAIN workload -- what the rocprofv3 passes of "
                         "the strict leg run; the default line carries the same step as its `strict_f32` leg")
    ap.add_argument("--weights", choices=["init", "trained"], default="init",
                    help="trained: the SURVEY 8(d) 'trained-like' regime -- weights after 500 online-SGD steps on the reference's fixture line, "
                         "inputs = jittered crops of that line (config b1 only); the default line carries it as the `trained_weights` leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-declare-next", action="store_true",
                    help="plain clstm_net_train_step calls: every step starts with an ingest launch of its own (default: the loop "
                         "declares its next minibatch, clstm_net_train_step_next, and the ingest rides the previous step's last launch)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="the headline workload only: skip the other legs of the default line (strict_f32, saturated, configs[4] in both precisions) -- "
                         "what the rocprofv3 passes use, so that their per-kernel averages are the headline workload's")
    ap.add_argument("--profile-steps", type=int, default=None,
                    help="extra steps with per-kernel timing (events bound to each launch's dispatch packet); default 50 (b1) / 5 (b2)")
    args = ap.parse_args()
    if args.no_declare_next:
        os.environ["CLSTM_BENCH_DECLARE_NEXT"] = "0"
    cfg = CONFIGS[args.config]
    if args.T is None:
        args.T = cfg["T"]
    if args.profile_steps is None:
        args.profile_steps = 50 if args.config == "b1" else 5
    default_line = (args.config == "b1" and not args.bf16 and not args.bf16_gemm and not args.ragged
                    and args.minibatch == 64 and args.T == 200 and not args.host_inputs and not args.strict_f32)
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

    # a hang becomes a stack dump of every thread and a non-zero exit instead of the caller's time-out (multi-rank runs: the
    # collectives of a step wait for every rank) -- CLSTM_BENCH_WATCHDOG_S, default 20 minutes per rank process
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("CLSTM_BENCH_WATCHDOG_S", "1200")), exit=True)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d (launch with torch.distributed.run --nproc-per-node %d, "
                 "or without WORLD_SIZE and let bench.py spawn the ranks)" % (world, args.gpus, args.gpus))
    backend = os.environ.get("CLSTM_BENCH_BACKEND", "nccl")      # ("nccl" IS RCCL on ROCm; gloo: the CPU test of this rank body)
    # test hook (tests/test_bench_ranks.py, -m gpu): N rank processes SHARE device 0 -- the rank body, the library communicator's
    # peer-read exchange (CLSTM_COMM_NO_RCCL=1: RCCL refuses duplicate devices) and the JSON line on the real library with one GPU
    share_dev = os.environ.get("CLSTM_BENCH_SHARE_DEVICE") == "1"
    dev_index = 0 if share_dev else local_rank
    if not ON_CPU:
        if torch.cuda.device_count() <= dev_index:
            sys.exit("bench.py: rank %d needs GPU %d but only %d visible" % (rank, dev_index, torch.cuda.device_count()))
        torch.cuda.set_device(dev_index)
    dev = torch.device("cpu") if ON_CPU else torch.device("cuda", dev_index)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):   # BENCH_FORCE_DIST=1: exercise the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from clstm_amd import abi
    from clstm_amd.net import Comm

    lib = abi.load(os.environ.get("CLSTM_BENCH_LIB") or None)   # raises if the HIP extension is missing -- there is no fallback path
    if not ON_CPU:
        # a real (non-default) stream: the library replays launch-bound loops as hipGraphs, and the legacy
        # default stream cannot be captured.  Everything below -- kernels, RCCL all-reduce -- runs on it.
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        lib.call("clstm_set_stream", stream.cuda_stream)
    # gradient exchange: the library's own RCCL communicator (all-reduce enqueued on the library stream right
    # before the update kernel, no cross-stream events); torch.distributed only carries the 128-byte id, the
    # barriers and the max-over-ranks of the timing.  If the communicator cannot be created the step falls back to
    # torch.distributed.all_reduce on the gradient tensor and the JSON line says so.
    allreduce_impl = None
    allreduce_ranks = None
    comm = None
    if dist is not None:
        def exchange(ident):
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        try:
            comm = Comm(rank, world, exchange, lib=lib)
            allreduce_impl = "clstm_allreduce_flat (RCCL ncclAllReduce on the library stream)"
            allreduce_ranks = int(lib.dll.clstm_comm_size(comm.h))      # what the LIBRARY's communicator spans
        except Exception as e:     # noqa: BLE001 -- report and fall back, never silently
            sys.stderr.write("bench.py: library communicator unavailable (%s); falling back to torch.distributed\n" % e)
            allreduce_impl = "torch.distributed.all_reduce (fallback: %s)" % type(e).__name__
            allreduce_ranks = world

    def fence():
        device_sync()
        if dist is not None:
            dist.barrier()
        device_sync()

    def reduce_max(dt):
        if dist is None:
            return dt
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    leg_validity = {}

    def measure(w, steps, warmup, profile_steps, unfused_pass=False, min_timed_s=None, name="main"):
        blocks, nxt = timed_blocks(w, steps, warmup, fence, reduce_max, min_timed_s)
        dt = float(np.median(blocks))
        # host-side cost of issuing a step (diagnostic: is the loop host-bound?): a short burst on an idle stream, few
        # enough steps that the library's pinned staging ring never makes the host wait for the GPU
        t1 = time.perf_counter()
        for i in range(4):
            w.step(nxt + i)
        t_enq = (time.perf_counter() - t1) / 4
        fence()
        kern, fps, kern_unfused = {}, 0, {}
        if profile_steps > 0:
            # EVERY rank takes these steps (with a communicator a step is a collective: rank 0 stepping alone waited for its
            # peers forever -- found by tests/test_bench_ranks.py::test_bench_gpus2_on_the_real_library_sharing_one_gpu before
            # the first multi-GPU run); rank 0's figures are the ones reported
            kern, fps = kernel_times(w, profile_steps, nxt + 4, dt / steps * 1e3)
            fence()
            if unfused_pass and world == 1:      # the pure kernels: the same steps with the fused launches off
                w.net.set_overlap(0)
                kern_unfused, _ = kernel_times(w, profile_steps, nxt + 4 + profile_steps, dt / steps * 1e3)
                w.net.set_overlap(1)
            if rank != 0:
                kern, kern_unfused = {}, {}
        try:                    # (every leg: a step whose update was skipped is not a step)
            validity = w.verify()
        except Exception as e:  # the headline workload fails loudly; a leg reports it and leaves the line standing
            if name == "main":
                raise
            validity = {"error": str(e)[:400]}
        leg_validity[name] = validity
        return {"dt": dt, "blocks": blocks, "enqueue": t_enq, "kern": kern, "kern_unfused": kern_unfused, "frames_per_step": fps, "validity": validity}

    precision = 2 if args.bf16 else 1 if args.bf16_gemm else 0
    w = Workload(lib, cfg, args.minibatch, args.T, args.ragged, precision, dev, rank, comm=comm, dist=dist, host_inputs=args.host_inputs,
                 weights=args.weights, strict_f32=args.strict_f32)
    if share_dev and world > 1:
        # ranks SHARING a device (test hook only): the fused launches' role workgroups wait for each other inside one launch,
        # which needs the launch's workgroups co-resident -- true with one process per GPU (the only supported deployment: 128
        # recurrence workgroups + their helpers on 256 CUs), not with a second process filling the same CUs (measured: two
        # 64-line ranks on one MI355X stall in the first fused launch).  Separate launches there.
        w.net.set_overlap(0)
    if comm is not None:
        # which exchange this rank's steps use, said on every rank as soon as it is decided (the first training step sets the
        # peer path up, collectively): it needs every rank to have mapped every other rank's buffers (hipIpcOpenMemHandle) AND
        # the probe handshakes through the mappings to deliver; otherwise every rank falls back to RCCL
        w.step(0)
        device_sync()
        pa = int(lib.dll.clstm_comm_peer_active(comm.h))
        sys.stderr.write("bench.py: rank %d / %d on %s: clstm_comm_peer_active = %d (%s)\n" % (
            rank, world, dev, pa, "peer-read all-reduce over HIP IPC mappings, fused into the update kernel" if pa else
            "no peer mappings (ranks on different nodes, or hipIpcOpenMemHandle / the probe failed): RCCL ncclAllReduce"))
    m = measure(w, args.steps, args.warmup, args.profile_steps, unfused_pass=max(cfg["nh"]) <= 128)
    dt = m["dt"]
    ms_per_step = dt / args.steps * 1e3
    value = args.minibatch * world * args.steps / dt
    roofline = None
    if rank == 0 and m["kern"]:
        if max(cfg["nh"]) <= 128:
            roofline = roofline_b1(w, m["kern"], m["kern_unfused"], m["frames_per_step"], ms_per_step, strict=args.strict_f32)
        else:
            roofline = roofline_b2(w, m["kern"], m["frames_per_step"], ms_per_step)

    # the gradient exchange on its own: the library's all-reduce of the flat gradient buffer, back to back on the library stream
    allreduce = None
    if allreduce_impl is not None:
        ar_ms = None
        if comm is not None:
            n = int(w.grads.numel())
            for _ in range(5):
                comm.allreduce(w.grads, n)
            fence()
            if ON_CPU:
                t_ar = time.perf_counter()
                for _ in range(50):
                    comm.allreduce(w.grads, n)
                ar_ms = reduce_max((time.perf_counter() - t_ar) * 1e3 / 50.0)
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    comm.allreduce(w.grads, n)
                e1.record()
                torch.cuda.synchronize()
                ar_ms = reduce_max(e0.elapsed_time(e1) / 50.0)
        if comm is not None and int(lib.dll.clstm_comm_peer_active(comm.h)):
            allreduce_impl = ("one-shot peer-read all-reduce fused into the update kernel (HIP IPC mappings of the ranks' gradient buffers over "
                              "xGMI, flag handshake; ops.h:k_peer_allreduce_update) inside clstm_net_train_step; clstm_allreduce_flat = RCCL")
        allreduce = {"impl": allreduce_impl, "ranks": allreduce_ranks, "bytes": int(w.grads.numel()) * 4,
                     "peer_active": bool(comm is not None and int(lib.dll.clstm_comm_peer_active(comm.h))),
                     "ms_per_call_isolated": None if ar_ms is None else round(ar_ms, 4),
                     "note": "ranks = clstm_comm_size of the communicator the step all-reduces on; the isolated figure is 50 calls back "
                             "to back on the library stream (max over ranks) -- inside a step the call sits between the last reduction and the update"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w.params_h, cfg)

    # the same workload in the TRAINED regime: speed must not depend on the weights -- peaked posteriors and
    # saturated gates walk the same kernels (default single-GPU line only)
    trained = None
    if rank == 0 and world == 1 and default_line and args.weights == "init" and not args.no_secondary:
        wt = Workload(lib, cfg, args.minibatch, args.T, args.ragged, 0, dev, rank, weights="trained")
        mt_ = measure(wt, args.steps, 5, 0, min_timed_s=0.5, name="trained_weights")
        trained = dict(wt.weights_info, value=round(args.minibatch * args.steps / mt_["dt"], 2), unit="lines/s",
                       ms_per_step=round(mt_["dt"] / args.steps * 1e3, 4), repeats=len(mt_["blocks"]),
                       parity="tests/test_gpu_e2e.py::test_full_bench_shape_trained_weights_real_line_crops (same recipe on the oracle)")
        wt.net = wt.trainer = None
        del wt

    # the same workload with the CTC recursion's log_add on the float transcendentals (experiment option ctc_float, VERDICT r5 item
    # 3d: NOT the default -- posteriors then agree with the reference's to 5.6e-5 absolute / 3.2e-4 relative at T = 200 instead of
    # 7.3e-6 / 1.1e-4; profiles/r06_ctc_float_logadd.txt); default single-GPU line only
    ctc_float = None
    if rank == 0 and world == 1 and default_line and args.weights == "init" and not args.no_secondary:
        lib.call("clstm_debug_set_option", b"ctc_float", 1)
        try:
            wc = Workload(lib, cfg, args.minibatch, args.T, args.ragged, 0, dev, rank)
            mc_ = measure(wc, args.steps, 5, 20, min_timed_s=0.5, name="ctc_float_logadd")
            ctc_float = {"value": round(args.minibatch * args.steps / mc_["dt"], 2), "unit": "lines/s", "ms_per_step": round(mc_["dt"] / args.steps * 1e3, 4),
                         "repeats": len(mc_["blocks"]), "ctc_align_ms": mc_["kern"].get("ctc_align", {}).get("ms_per_step"),
                         "option": "CLSTM_DEBUG=ctc_float=1 (float-only log_add in the CTC lattice recursion; not the default)",
                         "parity": "reference known answer (test-ctc.cc:76-109) 4.6e-6; vs oracle at T=200, S=51: 5.6e-5 absolute, 3.2e-4 relative "
                                   "(exact form, the default: 7.3e-6 / 1.1e-4); tests/test_ops_parity.py::test_ctc_float_logadd_option"}
            wc.net = wc.trainer = None
            del wc
        finally:
            lib.call("clstm_debug_set_option", b"ctc_float", 0)

    # the same workload with EVERY product on the f32 MFMA (clstm_net_set_strict_f32: the default computes the backward
    # weight-gradient / softmax-backward products on the bf16 MFMA from f32 operands split exactly into three bf16 terms);
    # default single-GPU line only
    strict = None
    if rank == 0 and world == 1 and default_line and not args.no_secondary:
        ws = Workload(lib, cfg, args.minibatch, args.T, args.ragged, 0, dev, rank, strict_f32=True)
        ms_ = measure(ws, args.steps, 5, 20, min_timed_s=0.5, name="strict_f32")
        strict = {"value": round(args.minibatch * args.steps / ms_["dt"], 2), "unit": "lines/s", "ms_per_step": round(ms_["dt"] / args.steps * 1e3, 4),
                  "repeats": len(ms_["blocks"]), "dtype": "f32 (every product on the f32 MFMA: clstm_net_set_strict_f32)",
                  "roofline": roofline_b1(ws, ms_["kern"], {}, ms_["frames_per_step"], ms_["dt"] / args.steps * 1e3, strict=True) if ms_["kern"] else None,
                  "kernels": ms_["kern"]}
        ws.net = ws.trainer = None
        del ws

    # the same net with the chip FULL (default single-GPU line only): 256 lines per GPU = two recurrence workgroups per CU.  At 64
    # lines the step is one line's dependent chain on half of the CUs (the headline roofline says how far a latency-bound
    # recurrence sits from a bandwidth bound); here the fused gate kernel is VALU-throughput-bound, and its fraction of the
    # HBM roof and of the packed-FMA rate is the "how far from the machine" number for the kernel itself.
    saturated = None
    if rank == 0 and world == 1 and default_line and not args.no_secondary:
        wsat = Workload(lib, cfg, 256, args.T, False, 0, dev, rank)
        ssat = max(5, min(args.steps, 20))
        msat = measure(wsat, ssat, 3, 20, unfused_pass=True, min_timed_s=0.5, name="saturated")
        ms_sat = msat["dt"] / ssat * 1e3
        rsat = roofline_b1(wsat, msat["kern"], msat["kern_unfused"], msat["frames_per_step"], ms_sat) if msat["kern"] else None
        if rsat is not None:
            cells_steps = 2 * sum(cfg["nh"]) * msat["frames_per_step"]
            rec_flops = 8.0 * cfg["nh"][0] * cells_steps           # 4 gates x no MACs per cell-step of R.h
            pure = (msat["kern"] if "gemm_gates_x" in msat["kern"] else msat["kern_unfused"]).get("lstm_fwd")   # the recurrence alone
            if pure:
                tf = rec_flops / (pure["ms_per_step"] * 1e-3) / 1e12
                rsat["recurrent_matvec"] = {"kernel": "lstm_fwd (fusions off)", "achieved": round(tf, 2), "peak": F32_MFMA_PEAK_TFS, "unit": "TFLOP/s",
                                            "frac": round(tf / F32_MFMA_PEAK_TFS, 5),
                                            "note": "R.h of the forward recurrence as packed f32 FMAs (v_pk_fma_f32) against the f32 vector peak"}
        saturated = {"value": round(256 * ssat / msat["dt"], 2), "unit": "lines/s", "ms_per_step": round(ms_sat, 4), "steps": ssat,
                     "repeats": len(msat["blocks"]), "config": {"workload": "the headline net at minibatch = 256 lines per GPU (two recurrence workgroups per CU)",
                                                                "minibatch_per_gpu": 256},
                     "roofline": rsat, "kernels": msat["kern"]}
        wsat.net = wsat.trainer = None
        del wsat

    # the same net at 2048 lines per GPU (default single-GPU line only): the narrow recurrences batched over 16 lines per workgroup
    # on the matrix cores (lstm_mfma.h and lstm_mfma_bwd.h, both from 640 lines) -- north_star's "gate GEMMs batched across a
    # minibatch of text lines with MFMA"; 2048 lines = 256 workgroups of 16 lines x direction = one per CU.  The per-line kernels'
    # ceiling is ~359k lines/s (round 5, profiles/r05_bench_mb1024.json).  Here the recurrences are HBM-bound: their roofline
    # entries price the bytes the kernels must move (forward: x in, six saved values out = 28 B per cell-step, SURVEY 8(d)'s
    # "GEMM fused in" figure; backward: activations, c, dH in, four deltas out = 40 B per cell-step -- dc and dh_rec never
    # leave the chip) against the HBM peak.
    large = None
    if rank == 0 and world == 1 and default_line and not args.no_secondary:
        LB = 2048
        wl = Workload(lib, cfg, LB, args.T, False, 0, dev, rank)
        sl_ = max(5, min(args.steps, 10))
        ml_ = measure(wl, sl_, 2, 5, min_timed_s=0.3, name="large_minibatch")
        kl = ml_["kern"]
        cell_steps = 2.0 * cfg["nh"][0] * ml_["frames_per_step"]
        rl = {}
        for name, bpc in (("lstm_fwd", 28.0), ("lstm_bwd", 40.0)):
            if name in kl:
                gbs = bpc * cell_steps / (kl[name]["ms_per_step"] * 1e-3) / 1e9
                rl[name] = {"kernel": "lstm_fwd_mfma_kernel<100, 48> (+ k_pack_mfma)" if name == "lstm_fwd" else "lstm_bwd_mfma_rows_kernel<100, 2>", "bound": "hbm",
                            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                            "algorithmic_bytes": int(bpc * cell_steps), "bytes_per_cell_step": bpc, "avg_launch_ms": kl[name]["ms_per_step"], "traffic": None}
        # ... and their gate products against the matrix peak (north_star: "MFMA utilisation for the batched gate GEMM"): algorithmic
        # flops = 2 x 4 no x (no + ni + 1) per line, direction and step forward (R.h + W_x.x + b as ONE product), 2 x no x 4 no
        # backward (R^T.delta); the kernels issue three 16-bit MFMAs per product (split operands), so the pipe is ~3 x as busy
        # (profiles/r06_mfma_busy_mb2048.txt: SQ_VALU_MFMA_BUSY_CYCLES)
        no_, ni_ = cfg["nh"][0], cfg["ni"]
        for name, fl in (("lstm_fwd", 2.0 * 4 * no_ * (no_ + ni_ + 1)), ("lstm_bwd", 2.0 * no_ * 4 * no_)):
            if name in rl:
                tf = fl * 2.0 * ml_["frames_per_step"] / (kl[name]["ms_per_step"] * 1e-3) / 1e12
                rl[name]["recurrent_matvec"] = {"achieved": round(tf, 1), "peak": BF16_MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": round(tf / BF16_MFMA_PEAK_TFS, 4),
                                                "note": "algorithmic flops of the step's gate product; x3 MFMA issue (hi/lo split operands)"}
        large = {"value": round(LB * sl_ / ml_["dt"], 2), "unit": "lines/s", "ms_per_step": round(ml_["dt"] / sl_ * 1e3, 4), "steps": sl_,
                 "repeats": len(ml_["blocks"]), "config": {"workload": "the headline net at minibatch = %d lines per GPU (batched-MFMA recurrences, one 16-line workgroup per CU)" % LB,
                                                           "minibatch_per_gpu": LB},
                 "roofline": rl, "kernels": kl, "parity": "tests/test_mfma_recurrence.py",
                 "evidence": "profiles/r06_mfma_*.txt, r06_bench_mb1024.json, r06_bench_mb2048.json"}
        wl.net = wl.trainer = None
        del wl

    # BASELINE.json configs[4] in the same process (default single-GPU line only): 2 x BiLSTM(512), bf16 MFMA
    secondary = None
    if rank == 0 and world == 1 and default_line and not args.no_secondary:
        c2 = CONFIGS["b2"]
        if comm is not None:
            w.net.set_comm(None)
        w.net = w.trainer = None        # free the first workload's device arrays
        w2 = Workload(lib, c2, 64, c2["T"], False, 2, dev, rank)
        s2 = max(5, min(args.steps, 20))
        m2 = measure(w2, s2, 3, 2, name="secondary")
        ms2 = m2["dt"] / s2 * 1e3
        secondary = {
            "metric": "text-line images/sec (fwd+bwd+CTC), 2xBiLSTM(512) H=64 T~400 (bf16 MFMA)",
            "value": round(64 * s2 / m2["dt"], 2), "unit": "lines/s", "n_gpus": 1, "steps": s2, "warmup": 3,
            "repeats": len(m2["blocks"]), "ms_per_step": round(ms2, 4),
            "dtype": "bf16 MFMA operands (hoisted gate GEMMs and lock-step recurrence), f32 accumulate / state / softmax / CTC",
            "data": "synthetic", "learning_rate": w2.lr,
            "config": {"workload": "stacked 2xBiLSTM(512) H=64 nc=100, T=400, L=50, minibatch=64 lines on 1 GPU "
                                   "(BASELINE.json configs[4] shape), fwd+CTC+bwd+update", "minibatch_per_gpu": 64},
            "roofline": (lambda r: dict(r, rocprof_avg_launch_ms_from_committed_profile=rocprof_b2_avg_ms(r.get("kernel"))))(roofline_b2(w2, m2["kern"], m2["frames_per_step"], ms2)) if m2["kern"] else None,
            "kernels": m2["kern"],
            "parity": "stated tolerance against the f32 oracle at this size: tests/test_gpu_e2e.py::test_configs4_full_shape_bf16_vs_oracle",
        }
        del w2
    # ... and its parity-grade form: every gate activation inside 1e-4 of the oracle at this size
    # (tests/test_gpu_e2e.py::test_configs4_full_shape_f32_vs_oracle); 10 steps x >= 3 repeats
    secondary_f32 = None
    if rank == 0 and world == 1 and default_line and not args.no_secondary:
        c2 = CONFIGS["b2"]
        w3 = Workload(lib, c2, 64, c2["T"], False, 0, dev, rank)
        m3 = measure(w3, 10, 2, 0, min_timed_s=0.25, name="secondary_f32")        # 10 steps x >= 3 repeats
        fl3 = flops_per_line(c2, c2["T"]) * 64 / (m3["dt"] / 10) / 1e12
        secondary_f32 = {"metric": "text-line images/sec (fwd+bwd+CTC), 2xBiLSTM(512) H=64 T~400 (f32)", "value": round(64 * 10 / m3["dt"], 2),
                         "unit": "lines/s", "steps": 10, "repeats": len(m3["blocks"]), "ms_per_step": round(m3["dt"] / 10 * 1e3, 4), "learning_rate": w3.lr,
                         "whole_step_tflops": {"achieved": round(fl3, 2), "unit": "TFLOP/s",
                                               "note": "bf16x3-ASSISTED: the backward two thirds of these flops run as three bf16 MFMAs per product "
                                                       "(16x the f32 MFMA's rate), so this figure is NOT a fraction of the 157.3 TFLOP/s f32 MFMA peak"},
                         "dtype": "f32 (forward pass incl. its recurrences, CTC, decode: exact f32; BACKWARD products of the wide layers -- weight "
                                  "gradient, input deltas and the recurrent delta product R^T.delta inside the backward recurrence: f32-grade bf16x3 "
                                  "split, < 2^-16 per product; gradient 1.45e-5 of its largest entry from the float64 oracle at this size, "
                                  "the all-f32-MFMA path 1.44e-5)",
                         "parity": "tests/test_gpu_e2e.py::test_configs4_full_shape_f32_vs_oracle, ..._strict_at_reference_init"}
        del w3

    if rank == 0:
        b1 = args.config == "b1"
        out = {
            "metric": "text-line images/sec (fwd+bwd+CTC), 100-unit BiLSTM H=48 T~200" if b1 else
                      "text-line images/sec (fwd+bwd+CTC), 2xBiLSTM(512) H=64 T~400 (%s)" % ("bf16 MFMA" if args.bf16 else "f32"),
            "value": round(value, 2), "unit": "lines/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "repeats": len(m["blocks"]), "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 MFMA operands (hoisted gate GEMMs and lock-step recurrence), f32 accumulate / state / softmax / CTC" if args.bf16
                      else "f32 (hoisted gate GEMMs: bf16 in, f32 accumulate)" if args.bf16_gemm
                      else "f32 (every product on the f32 MFMA: clstm_net_set_strict_f32)" if args.strict_f32
                      else "f32 (operand-exact split): forward, recurrences, CTC, decode in f32 arithmetic; the backward weight-gradient / "
                           "softmax-backward products take their f32 operands split EXACTLY into three bf16 terms each (x1 + x2 + x3 = x) and "
                           "sum the six bf16-MFMA products of weight >= 2^-16 in f32 -- what is dropped is < 2^-23 |x||y| per product, the size of "
                           "an f32 multiply's own rounding; `strict_f32` carries the same step on the f32 MFMA" if b1
                      else "f32 (forward pass incl. its recurrences, CTC, decode: exact f32; BACKWARD products of the wide layers -- weight gradient, "
                           "input deltas and the recurrent delta product inside the backward recurrence: f32-grade bf16x3 split, < 2^-16 per product)"),
            "data": "synthetic" + (" (frames fed from pinned host memory every step: PCIe-inclusive)" if args.host_inputs else ""),
            "config": {"workload": ("uw3-500 OCR shape: BiLSTM(100) H=48 nc=83, T=%s, L=25, minibatch=%d lines/GPU "
                                    "(BASELINE.json configs[2]; x%d GPUs = configs[3] sharding), fwd+CTC+bwd+allreduce+update"
                                    % ("U{150..250}" if args.ragged else args.T, args.minibatch, world))
                                   if b1 else
                                   ("stacked 2xBiLSTM(512) H=64 nc=100, T=%s, L=50, minibatch=%d lines/GPU x%d GPUs "
                                    "(BASELINE.json configs[4] shape), fwd+CTC+bwd+allreduce+update"
                                    % (args.T, args.minibatch, world)),
                       "minibatch_per_gpu": args.minibatch, "global_minibatch": args.minibatch * world,
                       "step_call": ("clstm_net_train_step_next: the loop declares its next minibatch, whose ingest rides this step's last launch "
                                     "(one ingest per step, none in front of the forward launch)" if (w.one_call and w.declare_next and not w.host_inputs)
                                     else "clstm_net_train_step_h" if w.host_inputs else "clstm_net_train_step" if w.one_call else "separate calls + torch.distributed all-reduce"),
                       "parallelism": "dp%d" % world},
            "timing": {"protocol": "median of `repeats` blocks of exactly `steps` steps, each bracketed by barrier + synchronize "
                                   "and max-reduced over ranks; warm-up >= %.1f s" % MIN_WARMUP_S,
                       "block_ms_min": round(min(m["blocks"]) * 1e3, 3), "block_ms_max": round(max(m["blocks"]) * 1e3, 3)},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": m["kern"], "kernels_fusions_off": m["kern_unfused"],
            "host_enqueue_ms_per_step": round(m["enqueue"] * 1e3, 4),   # host-side cost of issuing a step
            # after the timed blocks of EVERY workload of this line (Workload.verify: it raises otherwise): no sticky device
            # error -- the library skips updates after one --, parameters finite and moved by the updates
            "validity": dict(m["validity"], learning_rate=w.lr,
                             legs={k: v for k, v in leg_validity.items() if k != "main"}),
            "allreduce": allreduce,
            "weights": (w.weights_info or "reference initialisation (rinit negbiased, seed 0.222)"),
            "trained_weights": trained,
            "ctc_float_logadd": ctc_float,
            "strict_f32": strict,
            "saturated": saturated,
            "large_minibatch": large,
            "secondary": secondary,
            "secondary_f32": secondary_f32,
        }
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        if w.net is not None:
            w.net.set_comm(None)
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
