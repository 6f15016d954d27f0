"""Per-phase cycle stamps of the forward recurrence (diagnostics build libclstm_hip_prof.so).
Run on the GPU box:  make -C clstm_amd/csrc ../lib/libclstm_hip_prof.so && CLSTM_HIP_VARIANT=prof python scripts/gpu_lstmprof.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from clstm_amd import abi
from clstm_amd.init import init_params
from clstm_amd.net import Network
lib = abi.load()
NI, NH, NC, T, BS = 48, 100, 83, 200, 64
net = Network(NI, NH, NC, lib=lib)
net.set_params(init_params(NI, NH, NC, seed=0.222))
rng = np.random.default_rng(0)
lines = [np.clip(rng.normal(0.2, 0.3, (T, NI)), 0, 1).astype(np.float32) for _ in range(BS)]
net.set_inputs(lines)
for _ in range(3):
    net.forward()
out = (ctypes.c_longlong * 96)()
fn = lib.dll.clstm_debug_lstm_cycles
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
fn(net.h, out)
v = np.array(list(out), dtype=np.float64).reshape(8, 12) / T
names = ["loop/addr", "LDS read + FMA", "reduce + gx wait", "gate act + bcast", "c, tanh(c), h", "stores + LDS write", "barrier (end of step)",
         "A: reads, stores, FMAs < Y", "A: wait at Y", "B: wait at X (mid-tail)"]
print("cycles per step (workgroup 0, waves 0..6; s_memtime ticks):")
for k, n in enumerate(names):
    print("  %-28s" % n + "".join("%8.0f" % v[w, k] for w in range(7)))
print("  %-22s" % "total" + "".join("%8.0f" % v[w, :10].sum() for w in range(7)))
