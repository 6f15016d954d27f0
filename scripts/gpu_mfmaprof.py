"""Per-phase cycle stamps of the MFMA narrow recurrence (diagnostics build libclstm_hip_prof.so, lstm_mfma.h: MF_STAMP).
Run on the GPU box:  CLSTM_HIP_VARIANT=prof python scripts/gpu_mfmaprof.py [lines]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from clstm_amd import abi
from clstm_amd.init import init_params
from clstm_amd.net import Network
lib = abi.load()
NI, NH, NC, T = 48, 100, 83, 200
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib.call("clstm_debug_set_option", b"fwd_mfma", 2)
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
v = np.array(list(out), dtype=np.float64)[:64].reshape(8, 8) / T
names = ["barrier wait", "B frags, x, the pair's MFMAs", "pair epilogue | lone tile's MFMAs | rows", "-", "-", "lone (+ extra) tile's epilogue | rows"]
print("lines %d: cycles per step (workgroup 0, waves 0..3; s_memtime ticks = 100 MHz? see total vs wall)" % BS)
for k, n in enumerate(names):
    print("  %-42s" % n + "".join("%9.1f" % v[w, k] for w in range(8)))
print("  %-30s" % "total" + "".join("%9.1f" % v[w, :6].sum() for w in range(8)))

# forward launch time by what the memory pipeline is asked to do (mfma_dbg bits: 1 no activation rows, 2 no c / h / source rows, 4 no inputs)
for dbg in (0, 1, 2, 3, 7):
    lib.call("clstm_debug_set_option", b"mfma_dbg", dbg)
    net.forward(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.call("clstm_synchronize")
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        net.forward()
    lib.call("clstm_synchronize")
    print("  mfma_dbg %d: forward pass %.1f us (incl. ~%d us of softmax / ingest launches)" % (dbg, (time.perf_counter() - t0) / 5 * 1e6, 0))
