"""which steps of a declared-next loop find their minibatch prepared (path counters 19 / 20) -- on the GPU library"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from common import Backend, synth_lines
from clstm_amd.init import init_params
from clstm_amd.net import Network
be = Backend(sys.argv[1] if len(sys.argv) > 1 else "hip")
ni, nc, nh = 8, 7, [10]
rng = np.random.default_rng(23)
b = Network(ni, nh, nc, lib=be.lib)
b.set_params(init_params(ni, nh, nc, seed=0.222) * 20); b.setLearningRate(1e-2, 0.9)
def count(i):
    out = ctypes.c_longlong(0); be.lib.call("clstm_debug_path_count", i, ctypes.byref(out)); return out.value
batches = []
for k in range(7):
    T = [int(t) for t in rng.integers(3, 12, 2 + k % 3)]
    trs = [rng.integers(1, nc, max(1, t // 3)).astype(np.int32) for t in T]
    x = be.up(np.ascontiguousarray(np.concatenate(synth_lines(rng, T, ni), 0), np.float32))
    batches.append((Network.prepare_step(T, trs), x, T, trs))
for k in range(6):
    prep, x, T, trs = batches[k]
    if k < 5: b.train_step_prepared(prep, x, batches[k + 1][0], batches[k + 1][1])
    else: b.train_step_prepared(prep, x)
    print("step", k, "T", T, "tails", count(19), "used", count(20), "packs kept", count(10))
    b.get_params()
