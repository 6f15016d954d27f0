"""parity of the CTC kernel against the oracle for the library in use (CLSTM_HIP_VARIANT selects an experiment
build): the reference's known answer (test-ctc.cc:76-109), and max |aligned - oracle| over 64 lines of T = 200, S = 51 for
near-uniform and for peaked posteriors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import Backend
from oracle.oracle import Oracle
from test_ops_parity import ctc_via_abi
be = Backend("hip"); ora = Oracle("f32")
print("library variant:", os.environ.get("CLSTM_HIP_VARIANT", "(default)"))
o2 = np.array([[1, .5, 0, 0, 0, 0], [0, .5, .5, 0, 0, 0], [0, 0, .5, .5, 0, 0], [0, 0, 0, .5, .5, 0], [0, 0, 0, 0, .5, 1]], np.float32).T
e2 = np.array([[1., 0.12029, 0., 0., 0., 0.], [0., 0.87971, 0.40013, 0., 0., 0.], [0., 0., 0.59987, 0.59987, 0., 0.],
               [0., 0., 0., 0.40013, 0.87971, 0.], [0., 0., 0., 0., 0.12029, 1.]], np.float32).T
a2, _, _ = ctc_via_abi(be, [o2], [np.arange(5)])
print("known answer test-ctc.cc:76-109: max |aligned - expected| %.3g (the reference asserts 1e-4)" % np.abs(a2 - e2).max())
rng = np.random.default_rng(0)
nc, T, L, bs = 83, 200, 25, 64
for name, sharp in (("near-uniform posteriors (init-like)", 0.3), ("peaked posteriors (trained-like)", 6.0)):
    outs, trs = [], []
    for b in range(bs):
        z = rng.normal(0, sharp, (T, nc)).astype(np.float32)
        tr = rng.integers(1, nc, L).astype(np.int32)
        if sharp > 1:   # make the transcript likely: bump its classes along the line
            for t in range(T):
                z[t, tr[min(L - 1, t * L // T)] if (t % 8) < 5 else 0] += 8.0
        p = np.exp(z - z.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
        outs.append(p.astype(np.float32)); trs.append(tr)
    states = []
    for tr in trs:
        s = np.zeros(2 * L + 1, np.int32); s[1::2] = tr; states.append(s)
    al, _, loff = ctc_via_abi(be, outs, states)
    e = r = 0.0
    for b in range(bs):
        want = ora.ctc_align_classes(outs[b], states[b])
        g = al[loff[b]:loff[b + 1]]
        e = max(e, float(np.abs(g - want).max()))
        m = want > 1e-3
        r = max(r, float((np.abs(g - want)[m] / want[m]).max()))
    print("%s: max |aligned - oracle| %.3g, max relative error of entries > 1e-3: %.3g" % (name, e, r))
