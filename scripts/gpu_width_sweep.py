"""Per-step time of the register-resident recurrence kernels against the layer width (waves per workgroup = ceil(no/16)):
what a step costs with one wave per SIMD (<= 64 cells) and with two (65..128 cells).  Pure kernels (CLSTM_OVERLAP=0).
Run on the GPU box:  CLSTM_OVERLAP=0 python scripts/gpu_width_sweep.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from clstm_amd import abi
from clstm_amd.init import init_params
from clstm_amd.net import Network
lib = abi.load()
NI, NC, T, BS, L = 48, 83, 200, 64, 25
rng = np.random.default_rng(0)
lines = [np.clip(rng.normal(0.2, 0.3, (T, NI)), 0, 1).astype(np.float32) for _ in range(BS)]
trs = [rng.integers(1, NC, L).astype(np.int32) for _ in range(BS)]
print("cells waves  fwd us  bwd us   fwd ns/step  bwd ns/step")
for nh in [int(x) for x in os.environ.get("SWEEP_CELLS", "16,32,48,64,80,96,100,112,128").split(",")]:
    net = Network(NI, nh, NC, lib=lib)
    net.set_params(init_params(NI, nh, NC, seed=0.222))
    net.set_inputs(lines)
    for rep in range(2):
        if rep == 1:
            net.enable_timing(True); net.reset_timing()
        for _ in range(10):
            net.forward(); net.ctc(trs); net.backward()
    torch.cuda.synchronize()
    f, nf = net.kernel_time_ms("lstm_fwd"); b, nb = net.kernel_time_ms("lstm_bwd")
    print("%5d %5d %7.1f %7.1f %12.0f %12.0f" % (nh, (nh + 15) // 16, 1e3 * f / nf, 1e3 * b / nb, 1e6 * f / nf / T, 1e6 * b / nb / T))
    del net
