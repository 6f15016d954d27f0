"""Shared helpers for the parity tests (oracle <-> HIP path)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")

# gate activations / states: 1e-4 relative (BASELINE.json north_star) with an absolute floor for
# values that are themselves ~0.
RTOL = 1e-4
ATOL = 2e-6


def assert_close(a, b, rtol=RTOL, atol=ATOL, what="", scale_atol=0.0):
    """|a-b| <= atol + scale_atol*max|b| + rtol*|b|.  `scale_atol` is for quantities that are sums
    with cancellation (deltas, gradients): float32 summation-order noise is relative to the
    magnitude of the terms (~ the largest entries), not of each small result."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if scale_atol:
        atol = atol + scale_atol * float(np.abs(b).max()) if b.size else atol
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    if not (err <= 0).all():
        i = np.unravel_index(np.argmax(err), err.shape)
        raise AssertionError("%s mismatch at %s: got %r want %r (max excess %g)" % (what, i, a[i], b[i], err[i]))


def emu_lib():
    """Build (if needed) and load the TEST-ONLY host emulation of the kernels."""
    subprocess.check_call(["make", "-C", EMU_DIR, "-s"])
    from clstm_amd import abi
    return abi.load(os.path.join(EMU_DIR, "build", "libclstm_emu.so"))


def synth_lines(rng, T_list, ni):
    """Normalised-line-like inputs: clip(N(0.2,0.3),0,1), 3-tap smoothed along t."""
    out = []
    for T in T_list:
        x = np.clip(rng.normal(0.2, 0.3, (T + 2, ni)), 0, 1)
        x = (x[:-2] + x[1:-1] + x[2:]) / 3.0
        out.append(x.astype(np.float32))
    return out


def synth_labels(rng, n, L, nc):
    return [rng.integers(1, nc, L).astype(np.int32) for _ in range(n)]


def oracle_minibatch(ora, OracleNet, params, ninput, nhidden, nclasses, lines, transcripts,
                     unidirectional=False, derivs0=None, lr=None, mom=None, states=()):
    """Reference semantics of one minibatch: every line is an independent bs=1 fwd/CTC/bwd
    (clstmhl.h:201-217) accumulating into the same Params.d.  Returns dict of results."""
    net = OracleNet(ora, ninput, nhidden, nclasses, unidirectional=unidirectional, init=False)
    net.set_params(params)
    if derivs0 is not None:
        net.set_derivs(derivs0)
    if lr is not None:
        net.set_lr(lr, mom)
    res = {"outputs": [], "aligned": [], "deltas": [], "decode": [], "states": {k: [] for k in states}}
    for x, tr in zip(lines, transcripts):
        net.set_inputs(x)
        out = net.forward()[:, 0, :]
        res["outputs"].append(out.copy())
        res["decode"].append(net.decode())
        if tr is not None:
            al = net.ctc_deltas(tr)
            res["aligned"].append(al.copy())
            res["deltas"].append(net.get_output_deltas()[:, 0, :].copy())
            net.backward()
        for k in states:
            layer, direction, which = k
            plane = 1 if which.startswith("d_") else 0
            s = net.state(layer, direction, which[2:] if plane else which, plane)[:, 0, :]
            if direction == 1:
                s = s[::-1]          # the NPLSTM inside Reversed runs on reversed frames
            res["states"][k].append(s.copy())
    res["derivs"] = net.get_derivs()
    res["net"] = net
    return res


class Backend:
    """Where the C ABI runs: 'emu' = host-thread emulator build of the kernel sources (CPU
    tests), 'hip' = the real libclstm_hip.so on an MI355X (-m gpu tests)."""

    def __init__(self, kind):
        self.kind = kind
        if kind == "emu":
            self.lib = emu_lib()
        else:
            import torch
            assert torch.cuda.is_available(), "the -m gpu tests need a GPU"
            from clstm_amd import abi
            self.lib = abi.load()          # raises loudly if the HIP extension is missing
            self.torch = torch

    def up(self, a, dtype=np.float32):
        a = np.ascontiguousarray(a, dtype=dtype)
        if self.kind == "emu":
            return a.copy()
        return self.torch.from_numpy(a).cuda()

    def zeros(self, shape, dtype=np.float32):
        return self.up(np.zeros(shape, dtype), dtype)

    def down(self, d):
        if self.kind == "emu":
            return np.array(d, copy=True)
        self.lib.call("clstm_synchronize")
        return d.cpu().numpy()

    def sync(self):
        self.lib.call("clstm_synchronize")
