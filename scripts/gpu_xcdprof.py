"""Per-phase cycle stamps of the persistent per-XCD bf16 recurrences of wide layers (diagnostics build libclstm_hip_prof.so).
Run on the GPU box:  make -C clstm_amd/csrc ../lib/libclstm_hip_prof.so && CLSTM_HIP_VARIANT=prof python scripts/gpu_xcdprof.py
One BiLSTM(512) layer, 64 lines x 400 frames (the configs[4] recurrence shape); prints, for the first and the last tile of
XCD 0's group and each of their four waves, the cycles per step spent in each phase (s_memtime ticks)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from clstm_amd import abi
from clstm_amd.init import init_params
from clstm_amd.net import Network
lib = abi.load()
NI, NH, NC, T, BS = int(os.environ.get("XCDPROF_NI", "64")), 512, 100, 400, 64   # XCDPROF_NI=1024: the input width of configs[4]'s second layer
net = Network(NI, NH, NC, lib=lib)
net.set_params(init_params(NI, NH, NC, seed=0.222))
net.set_gemm_precision(2)
rng = np.random.default_rng(0)
lines = [np.clip(rng.normal(0.2, 0.3, (T, NI)), 0, 1).astype(np.float32) for _ in range(BS)]
labels = [list(rng.integers(1, NC, 50)) for _ in range(BS)]
net.set_inputs(lines)
fn = lib.dll.clstm_debug_lstm_cycles
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
out = (ctypes.c_longlong * 96)()
def show(title, names):
    fn(net.h, out)
    v = np.array(list(out), dtype=np.float64).reshape(8, 12) / T
    print(title, "(cycles per step; columns: tile 0 waves 0-3 | last tile waves 0-3)")
    for k, n in enumerate(names):
        print("  %-44s" % n + "".join("%8.0f" % v[w, k] for w in range(8)))
    print("  %-44s" % "total" + "".join("%8.0f" % v[w, :len(names)].sum() for w in range(8)))
for _ in range(3):
    net.forward()
torch.cuda.synchronize()
show("forward", ["loop top", "group wait", "ring + gx loads issued", "loads returned + MFMAs + partials to LDS", "barrier", "reduce + nonlinearities + stores issued",
                 "stores acknowledged (drain)", "barrier + arrival", "(fused W_x) barrier + x fragments + MFMAs", "(fused W_x) x rows to LDS + next loads issued"])
net.ctc(labels)
for _ in range(2):
    net.backward()
torch.cuda.synchronize()
show("backward", ["loop top", "group wait", "ring + operand loads issued", "loads returned + MFMAs + partials to LDS", "barrier", "reduce + deltas + ring store issued",
                  "ring store acknowledged (drain)", "barrier + arrival", "per-frame stores issued"])
