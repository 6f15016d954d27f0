"""debug: how does the oracle's OpenMP-over-lines baseline scale with the thread count on this host?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from clstm_amd.init import init_params
from oracle.oracle import Oracle, OracleNet
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, "-")
print("nproc", os.cpu_count(), "affinity", len(bench.CPU_TOPOLOGY[1]), "physical", len(bench.CPU_TOPOLOGY[0]))
print(open("/proc/loadavg").read().strip())
cfg = bench.CONFIGS["b1"]
ora = Oracle("f32"); net = OracleNet(ora, 48, 100, 83, init=False); net.set_params(init_params(48, 100, 83, seed=0.222))
rng = np.random.default_rng(1)
Ts, x, labels = bench.synth_batch(rng, 64, 200, False, 48, 83, 25)
offs = np.concatenate([[0], np.cumsum(Ts)]); loffs = np.concatenate([[0], np.cumsum([len(l) for l in labels])]); lab = np.concatenate(labels)
phys = bench.CPU_TOPOLOGY[0]
for n in (1, 2, 4, 8, 16, 32, 64, 128):
    if n > len(phys): break
    reps = max(1, 4 * n // 64)
    t = net.bench_lines(x, offs, lab, loffs, nthreads=n, reps=reps, cpus=phys[:n])
    t = net.bench_lines(x, offs, lab, loffs, nthreads=n, reps=reps, cpus=phys[:n])
    print("threads %3d: %8.1f lines/s  (%.1f per thread)" % (n, 64 * reps / t, 64 * reps / t / n))
