"""Randomised stress of the fused backward launch (recurrence + monitor + weight-gradient items + softmax W.d items):
many batch geometries, each compared with the plain path (overlap 0) of the same library; no slab may time out."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from common import Backend, synth_lines
from clstm_amd.init import init_params
from clstm_amd.net import Network

be = Backend("hip")
rng = np.random.default_rng(int(os.environ.get("SEED", "7")))
ncase = int(os.environ.get("NCASE", "60"))
worst = 0.0
t0 = time.time()
for case in range(ncase):
    nh = int(rng.choice([100, 100, 100, 50, 33, 120]))
    ni, nc = int(rng.choice([48, 12])), int(rng.choice([83, 20]))
    bs = int(rng.integers(1, 90))
    tmax = int(rng.choice([8, 40, 70, 130, 200, 260]))
    T = [int(t) for t in rng.integers(max(1, tmax // 3), tmax + 1, bs)]
    params = init_params(ni, [nh], nc, seed=0.222) * 8.0
    lines = synth_lines(rng, T, ni)
    trs = [rng.integers(1, nc, max(1, t // 8)).astype(np.int32) for t in T]
    g = []
    for mode in (0, 2):
        net = Network(ni, [nh], nc, lib=be.lib)
        net.set_overlap(mode)
        net.set_params(params)
        for rep in range(2):
            net.set_inputs(lines); net.forward(); net.ctc(trs); net.backward()
        launches, timeouts = net.overlap_stats()
        assert timeouts == 0, (case, mode, timeouts)
        g.append(net.get_grads().copy())
        del net
    err = float(np.abs(g[1] - g[0]).max() / (np.abs(g[0]).max() + 1e-30))
    worst = max(worst, err)
    assert np.isfinite(g[1]).all() and err < 5e-5, (case, nh, ni, nc, bs, tmax, err)
print("stress ok: %d cases, worst |g_overlap - g_plain| / max|g| = %.3g, %.1f s" % (ncase, worst, time.time() - t0))
