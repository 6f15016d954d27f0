"""configs[4] training trajectory on the bench's synthetic minibatches: largest |gradient| / |parameter| / |output delta| per step.
usage: b2_divergence.py <precision 0|2> <lr> <steps> [strict]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from clstm_amd import abi

prec, lr, steps = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
strict = len(sys.argv) > 4
lib = abi.load()
dev = torch.device("cuda:0")
c2 = bench.CONFIGS["b2"]
w = bench.Workload(lib, c2, 64, c2["T"], False, prec, dev, 0, strict_f32=strict)
w.net.setLearningRate(lr, 0.9)
for i in range(steps):
    w.step(i)
    try:
        g = w.net.get_grads(); p = w.net.get_params()
    except Exception as e:
        print("step", i + 1, "ERROR", str(e)[:120]); break
    print("step %3d  max|g| %.4g  rms g %.4g  max|p| %.4g  finite %s" % (i + 1, np.abs(g).max(), np.sqrt((g.astype(np.float64) ** 2).mean()), np.abs(p).max(), bool(np.isfinite(g).all())), flush=True)
