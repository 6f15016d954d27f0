"""debug: where do the MFMA recurrence's saved activations differ from the oracle?  usage: mfma_mismatch.py [T...]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import Backend, oracle_minibatch, synth_lines
from oracle.oracle import OracleNet, Oracle
from clstm_amd.net import Network
T = [int(x) for x in sys.argv[1:]] or [31] * 16
be = Backend("hip")
ora32 = Oracle("f32")
be.lib.call("clstm_debug_set_option", b"fwd_mfma", 2)
ni, nh, nc = 48, 100, 83
rng = np.random.default_rng(1)
ref = OracleNet(ora32, ni, nh, nc, seed=0.222)
params = ref.get_params() * 10.0
lines = synth_lines(rng, T, ni)
trs = [rng.integers(1, nc, max(1, t // 3)).astype(np.int32) for t in T]
STATES = ("gi", "gf", "go", "ci", "state", "outputs")
skeys = [(0, d, w) for d in (0, 1) for w in STATES]
want = oracle_minibatch(ora32, OracleNet, params, ni, nh, nc, lines, trs, states=skeys, lr=1e-2, mom=0.9)
for rep in range(3):
    net = Network(ni, nh, nc, lib=be.lib)
    net.set_params(params)
    net.set_inputs(lines)
    net.forward()
    for k in skeys:
        s = net.split(net.state(*k))
        for b in range(len(T)):
            a = np.asarray(s[b], np.float64); w = np.asarray(want["states"][k][b], np.float64)
            bad = np.abs(a - w) > 2e-6 + 1e-4 * np.abs(w)
            if bad.any():
                idx = np.argwhere(bad)
                print("rep", rep, k, "line", b, "bad", len(idx), "first", idx[:6].tolist(), "got", a[bad][:4], "want", w[bad][:4])
print("done")
