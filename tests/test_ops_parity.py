"""Per-op C ABI (1:1 with the DEFGENERIC list of clstm_compute.h:72-103 on the BiLSTM path)
against the oracle's restatement of clstm_compute.cc, on the deterministic cos(3.7k) data of
test-cderiv.cc:29-36.  Runs on the emulator (CPU) and on the GPU (-m gpu)."""
import ctypes as C

import numpy as np
import pytest

from clstm_amd.abi import ptr
from common import assert_close

SIG, TANH, LIN, RELU, LOGMAG = 1, 2, 0, 3, 4


def cosdata(n, start):
    out, k = [], start
    while len(out) < n:
        x = np.cos(k * 3.7)
        k += 1
        if abs(x) > 0.1:
            out.append(x)
    return np.array(out, np.float32)


def P(o, a):
    return o.p(a)


class OraCall:
    def __init__(self, ora):
        self.o = ora

    def __getattr__(self, name):
        fn = getattr(self.o.lib, "ora_" + name)

        def call(*args):
            conv = []
            for a in args:
                if isinstance(a, np.ndarray):
                    conv.append(a.ctypes.data_as(C.c_void_p))
                else:
                    conv.append(a)
            return fn(*conv)
        return call


@pytest.mark.parametrize("nl", [SIG, TANH, LIN, RELU, LOGMAG])
def test_nonlin_ops(backend, ora32, nl):
    oc = OraCall(ora32)
    n = 77
    x = cosdata(n, 1) * 3
    yd = cosdata(n, 200)
    # forward_nonlin0 / backward_nonlin0 (in place)
    y_ref = x.copy(); oc.forward_nonlin0(y_ref, n, nl)
    y = backend.up(x); backend.lib.call("clstm_forward_nonlin0", ptr(y), n, nl)
    assert_close(backend.down(y), y_ref, what="forward_nonlin0")
    d_ref = yd.copy(); oc.backward_nonlin0(y_ref, d_ref, n, nl)
    d = backend.up(yd); yv = backend.up(y_ref)
    backend.lib.call("clstm_backward_nonlin0", ptr(yv), ptr(d), n, nl)
    assert_close(backend.down(d), d_ref, what="backward_nonlin0")
    # forward_nonlin / backward_nonlin (additive)
    y2 = backend.zeros(n); xv = backend.up(x)
    backend.lib.call("clstm_forward_nonlin", ptr(y2), ptr(xv), n, nl)
    assert_close(backend.down(y2), y_ref, what="forward_nonlin")
    xd0 = cosdata(n, 400)
    xd_ref = xd0.copy(); oc.backward_nonlin(y_ref, yd, xd_ref, n, nl)
    xd = backend.up(xd0); ydv = backend.up(yd)
    backend.lib.call("clstm_backward_nonlin", ptr(yv), ptr(ydv), ptr(xd), n, nl)
    assert_close(backend.down(xd), xd_ref, what="backward_nonlin")


@pytest.mark.parametrize("nl", [SIG, TANH])
def test_full1(backend, ora32, nl):
    # shapes of test-cderiv.cc:255-262: x 7x4, W 3x8
    oc = OraCall(ora32)
    n, m, bs = 3, 8, 4
    W = cosdata(n * m, 1); x = cosdata((m - 1) * bs, 50); yd0 = cosdata(n * bs, 90)
    y_ref = np.zeros(n * bs, np.float32); oc.forward_full1(y_ref, W, x, n, m, bs, nl)
    y = backend.zeros(n * bs); Wd_ = backend.up(W); xv = backend.up(x)
    backend.lib.call("clstm_forward_full1", ptr(y), ptr(Wd_), ptr(xv), n, m, bs, nl)
    assert_close(backend.down(y), y_ref, what="forward_full1")
    Wd0 = cosdata(n * m, 300); xd0 = cosdata((m - 1) * bs, 500)
    yd_ref = yd0.copy(); Wd_ref = Wd0.copy(); xd_ref = xd0.copy()
    oc.backward_full1(y_ref, yd_ref, W, Wd_ref, x, xd_ref, n, m, bs, nl)
    yv = backend.up(y_ref); yd = backend.up(yd0); Wd = backend.up(Wd0); xd = backend.up(xd0)
    backend.lib.call("clstm_backward_full1", ptr(yv), ptr(yd), ptr(Wd_), ptr(Wd), ptr(xv), ptr(xd), n, m, bs, nl)
    assert_close(backend.down(yd), yd_ref, what="y.d")
    assert_close(backend.down(Wd), Wd_ref, what="W.d")
    assert_close(backend.down(xd), xd_ref, what="x.d")
    # lin1 alone
    y1_ref = np.zeros(n * bs, np.float32); oc.forward_lin1(y1_ref, W, x, n, m, bs)
    y1 = backend.zeros(n * bs)
    backend.lib.call("clstm_forward_lin1", ptr(y1), ptr(Wd_), ptr(xv), n, m, bs)
    assert_close(backend.down(y1), y1_ref, what="forward_lin1")


def test_softmax(backend, ora32):
    oc = OraCall(ora32)
    n, m, bs = 5, 9, 3
    W = cosdata(n * m, 7) * 2; x = cosdata((m - 1) * bs, 77) * 2
    x[3] = 40.0                                   # drives one logit through the +-30 clamp
    z_ref = np.zeros(n * bs, np.float32); oc.forward_softmax(z_ref, W, x, n, m, bs)
    z = backend.zeros(n * bs); Wv = backend.up(W); xv = backend.up(x)
    backend.lib.call("clstm_forward_softmax", ptr(z), ptr(Wv), ptr(xv), n, m, bs)
    got = backend.down(z)
    assert_close(got, z_ref, what="forward_softmax")
    assert np.allclose(got.reshape(bs, n).sum(1), 1, atol=1e-5)      # check_normalized, batches.h:162
    zd = cosdata(n * bs, 123); Wd0 = cosdata(n * m, 321); xd0 = cosdata((m - 1) * bs, 555)
    Wd_ref = Wd0.copy(); xd_ref = xd0.copy()
    oc.backward_softmax(zd, W, Wd_ref, x, xd_ref, n, m, bs)
    Wd = backend.up(Wd0); xd = backend.up(xd0); zdv = backend.up(zd)
    backend.lib.call("clstm_backward_softmax", ptr(zdv), ptr(Wv), ptr(Wd), ptr(xv), ptr(xd), n, m, bs)
    assert_close(backend.down(xd), xd_ref, what="x.d (assign)")
    assert_close(backend.down(Wd), Wd_ref, what="W.d")


def test_stack_and_delay(backend, ora32):
    oc = OraCall(ora32)
    nx, ny, bs = 4, 3, 2
    x = cosdata(nx * bs, 1); y = cosdata(ny * bs, 30)
    z_ref = np.zeros((nx + ny) * bs, np.float32); oc.forward_stack(z_ref, x, y, nx, ny, bs)
    z = backend.zeros((nx + ny) * bs); xv = backend.up(x); yv = backend.up(y)
    backend.lib.call("clstm_forward_stack", ptr(z), ptr(xv), ptr(yv), nx, ny, bs)
    assert np.array_equal(backend.down(z), z_ref)
    backend.lib.call("clstm_forward_stack_delay", ptr(z), ptr(xv), None, nx, ny, bs)
    oc.forward_stack_delay(z_ref, x, None, nx, ny, bs)
    assert np.array_equal(backend.down(z), z_ref)                       # last < 0 -> zeros
    zd = cosdata((nx + ny) * bs, 60); xd0 = cosdata(nx * bs, 90); yd0 = cosdata(ny * bs, 120)
    xd_ref = xd0.copy(); yd_ref = yd0.copy(); oc.backward_stack(zd, xd_ref, yd_ref, nx, ny, bs)
    xd = backend.up(xd0); yd = backend.up(yd0); zdv = backend.up(zd)
    backend.lib.call("clstm_backward_stack", ptr(zdv), ptr(xd), ptr(yd), nx, ny, bs)
    assert_close(backend.down(xd), xd_ref); assert_close(backend.down(yd), yd_ref)
    xd = backend.up(xd0); xd_ref = xd0.copy(); oc.backward_stack_delay(zd, xd_ref, None, nx, ny, bs)
    backend.lib.call("clstm_backward_stack_delay", ptr(zdv), ptr(xd), None, nx, ny, bs)
    assert_close(backend.down(xd), xd_ref)


def test_reverse(backend):
    rows, bs, N = 3, 2, 4
    x = cosdata(rows * bs * 2 * N, 5)
    y = backend.zeros(x.size); xv = backend.up(x)
    backend.lib.call("clstm_forward_reverse", ptr(y), ptr(xv), rows, bs, N)
    want = x.reshape(N, 2, bs, rows)[::-1].reshape(-1)                  # y[N-1-i] = x[i], v and d
    assert np.array_equal(backend.down(y), want)
    yb = cosdata(x.size, 99); x2 = backend.up(x); ybv = backend.up(yb)
    backend.lib.call("clstm_backward_reverse", ptr(ybv), ptr(x2), rows, bs, N)
    w = x.reshape(N, 2, bs * rows).copy()
    w[:, 1] += yb.reshape(N, 2, bs * rows)[::-1][:, 1]                   # x[N-1-i].d += y[i].d
    assert_close(backend.down(x2), w.reshape(-1))


@pytest.mark.parametrize("rows,bs,N", [(3, 2, 4), (5, 7, 1), (1, 1, 6), (4, 3, 3)])
def test_btswitch(backend, ora32, rows, bs, N):
    """forward_btswitch / backward_btswitch (clstm_compute.cc:425-447) against the oracle's restatement of the
    Eigen chip + shuffle: only the value plane moves forward (y's derivative plane keeps what it held), the
    derivative plane accumulates backward."""
    oc = OraCall(ora32)
    x = cosdata(rows * bs * 2 * N, 5)
    y0 = cosdata(rows * N * 2 * bs, 700)                     # y is NOT cleared by forward: its d plane must survive
    y_ref = y0.copy(); oc.forward_btswitch(y_ref, x, rows, bs, N)
    y = backend.up(y0); xv = backend.up(x)
    backend.lib.call("clstm_forward_btswitch", ptr(y), ptr(xv), rows, bs, N)
    assert np.array_equal(backend.down(y), y_ref)
    # spot-check the restatement itself: y.v(i, t, b) = x.v(i, b, t)
    X = x.reshape(N, 2, bs, rows); Y = y_ref.reshape(bs, 2, N, rows)
    assert np.array_equal(Y[:, 0].transpose(1, 0, 2), X[:, 0])
    assert np.array_equal(Y[:, 1], y0.reshape(bs, 2, N, rows)[:, 1])
    yd = cosdata(y0.size, 99)
    x_ref = x.copy(); oc.backward_btswitch(yd, x_ref, rows, bs, N)
    x2 = backend.up(x); ydv = backend.up(yd)
    backend.lib.call("clstm_backward_btswitch", ptr(ydv), ptr(x2), rows, bs, N)
    assert np.array_equal(backend.down(x2), x_ref)
    assert np.array_equal(x_ref.reshape(N, 2, bs, rows)[:, 0], X[:, 0])       # value plane untouched


@pytest.mark.parametrize("d,bs,N,pre,post", [(3, 4, 2, 1, 1), (2, 5, 3, 2, 1), (4, 1, 2, 1, 1), (2, 3, 1, 0, 2), (3, 2, 2, 3, 3)])
def test_batchstack(backend, ora32, d, bs, N, pre, post):
    """forward_batchstack / backward_batchstack (clstm_compute.cc:451-500) against the oracle's slice-by-slice
    restatement: forward clears BOTH planes of y (:464) and fills the value plane block by block, backward
    accumulates the derivative plane into x without clearing it (:489 is commented out upstream); pre + post
    larger than the batch (crimp >= bs) leaves empty slices."""
    oc = OraCall(ora32)
    copies = pre + post + 1
    x = cosdata(d * bs * 2 * N, 11)
    y0 = cosdata(copies * d * bs * 2 * N, 300)
    y_ref = y0.copy(); oc.forward_batchstack(y_ref, x, d, bs, N, pre, post)
    y = backend.up(y0); xv = backend.up(x)
    backend.lib.call("clstm_forward_batchstack", ptr(y), ptr(xv), d, bs, N, pre, post)
    assert np.array_equal(backend.down(y), y_ref)
    assert not y_ref.reshape(N, 2, bs, copies * d)[:, 1].any()               # derivative plane cleared
    yd = cosdata(y0.size, 900)
    x_ref = x.copy(); oc.backward_batchstack(yd, x_ref, d, bs, N, pre, post)
    x2 = backend.up(x); ydv = backend.up(yd)
    backend.lib.call("clstm_backward_batchstack", ptr(ydv), ptr(x2), d, bs, N, pre, post)
    assert_close(backend.down(x2), x_ref, rtol=1e-6, atol=1e-6)             # (k summed in the reference's order: exact in practice)


@pytest.mark.parametrize("nl", [SIG, TANH, RELU, LIN])
def test_full_layer_network(backend, ora32, nl):
    """Stacked{Full<NONLIN>(ni -> nh), SoftmaxLayer(nh -> nc)} (clstm.cc:354-419, 421-455: LinearLayer / SigmoidLayer /
    TanhLayer / ReluLayer in front of a softmax) over a sequence of T batches: the per-op C ABI driven the way
    INetwork::forward / backward drive it (forward_full1 per step ascending, backward_full1 per step descending,
    W.d accumulating over the steps, Stacked's .d hand-off) against the oracle's layer-level restatement."""
    oc = OraCall(ora32)
    T, bs, ni, nh, nc = 5, 3, 6, 7, 4
    x = cosdata(T * bs * ni, 3)
    W1 = cosdata(nh * (ni + 1), 40) * 0.7; W2 = cosdata(nc * (nh + 1), 90)
    # oracle: layer-level
    h_ref = np.zeros(T * bs * nh, np.float32); oc.full_forward(h_ref, W1, x, T, nh, ni, bs, nl)
    z_ref = np.zeros(T * bs * nc, np.float32); oc.softmax_seq_forward(z_ref, W2, h_ref, T, nc, nh, bs)
    target = np.abs(cosdata(T * bs * nc, 500)); target /= target.reshape(T * bs, nc).sum(1).repeat(nc)
    zd = (target - z_ref).astype(np.float32)                                  # set_targets, clstm.cc:142-150
    W2d_ref = cosdata(W2.size, 600); W1d_ref = cosdata(W1.size, 650)
    hd_ref = np.zeros_like(h_ref); xd_ref = np.zeros_like(x)
    oc.softmax_seq_backward(zd, W2, W2d_ref, h_ref, hd_ref, T, nc, nh, bs)
    hd_in = hd_ref.copy()
    oc.full_backward(h_ref, hd_ref, W1, W1d_ref, x, xd_ref, T, nh, ni, bs, nl)
    # the C ABI, one operator call per step
    xv = backend.up(x); W1v = backend.up(W1); W2v = backend.up(W2)
    h = backend.zeros(T * bs * nh); z = backend.zeros(T * bs * nc)
    W1d = backend.up(cosdata(W1.size, 650)); W2d = backend.up(cosdata(W2.size, 600))
    hd = backend.zeros(T * bs * nh); xd = backend.zeros(T * bs * ni); zdv = backend.up(zd)
    fs = 4                                                                    # sizeof(float)

    def at(buf, t, n):
        return ptr(buf) + t * n * bs * fs
    for t in range(T):
        backend.lib.call("clstm_forward_full1", at(h, t, nh), ptr(W1v), at(xv, t, ni), nh, ni + 1, bs, nl)
    for t in range(T):
        backend.lib.call("clstm_forward_softmax", at(z, t, nc), ptr(W2v), at(h, t, nh), nc, nh + 1, bs)
    assert_close(backend.down(h), h_ref, what="Full outputs")
    assert_close(backend.down(z), z_ref, what="softmax outputs")
    for t in range(T - 1, -1, -1):
        backend.lib.call("clstm_backward_softmax", at(zdv, t, nc), ptr(W2v), ptr(W2d), at(h, t, nh), at(hd, t, nh), nc, nh + 1, bs)
    assert_close(backend.down(hd), hd_in, rtol=1e-4, atol=1e-6, what="softmax x.d")
    for t in range(T - 1, -1, -1):
        backend.lib.call("clstm_backward_full1", at(h, t, nh), at(hd, t, nh), ptr(W1v), ptr(W1d), at(xv, t, ni), at(xd, t, ni),
                         nh, ni + 1, bs, nl)
    assert_close(backend.down(W2d), W2d_ref, rtol=1e-4, atol=1e-5, what="softmax W.d")
    assert_close(backend.down(hd), hd_ref, rtol=1e-4, atol=1e-6, what="Full y.d after nonlin0 backward")
    assert_close(backend.down(W1d), W1d_ref, rtol=1e-4, atol=1e-5, what="Full W.d")
    assert_close(backend.down(xd), xd_ref, rtol=1e-4, atol=1e-6, what="Full x.d")


def test_statemem_nonlingate(backend, ora32):
    oc = OraCall(ora32)
    n = 24
    ci, gi, gf, last = (cosdata(n, s) for s in (1, 40, 80, 120))
    st_ref = np.zeros(n, np.float32); oc.forward_statemem(st_ref, ci, gi, last, gf, n)
    st = backend.zeros(n); a = [backend.up(v) for v in (ci, gi, last, gf)]
    backend.lib.call("clstm_forward_statemem", ptr(st), ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), n)
    assert_close(backend.down(st), st_ref)
    st0 = np.zeros(n, np.float32); oc.forward_statemem(st0, ci, gi, None, gf, n)
    backend.lib.call("clstm_forward_statemem", ptr(st), ptr(a[0]), ptr(a[1]), None, ptr(a[3]), n)
    assert_close(backend.down(st), st0)
    sd = cosdata(n, 160)
    acc0 = [cosdata(n, s) for s in (200, 240, 280, 320)]           # ci.d gi.d last.d gf.d
    ref = [v.copy() for v in acc0]
    oc.backward_statemem(sd, ci, ref[0], gi, ref[1], last, ref[2], gf, ref[3], n)
    dv = [backend.up(v) for v in acc0]; sdv = backend.up(sd)
    backend.lib.call("clstm_backward_statemem", ptr(sdv), ptr(a[0]), ptr(dv[0]), ptr(a[1]), ptr(dv[1]),
                     ptr(a[2]), ptr(dv[2]), ptr(a[3]), ptr(dv[3]), n)
    for g_, r_ in zip(dv, ref):
        assert_close(backend.down(g_), r_)
    # nonlingate
    go = np.abs(cosdata(n, 360))
    out_ref = np.zeros(n, np.float32); oc.forward_nonlingate(out_ref, st_ref, go, n, TANH)
    out = backend.zeros(n); stv = backend.up(st_ref); gov = backend.up(go)
    backend.lib.call("clstm_forward_nonlingate", ptr(out), ptr(stv), ptr(gov), n, TANH)
    assert_close(backend.down(out), out_ref)
    od = cosdata(n, 400); sd0 = cosdata(n, 440); gd0 = cosdata(n, 480)
    sd_ref = sd0.copy(); gd_ref = gd0.copy()
    oc.backward_nonlingate(od, st_ref, sd_ref, go, gd_ref, n, TANH)
    sdd = backend.up(sd0); gdd = backend.up(gd0); odv = backend.up(od)
    backend.lib.call("clstm_backward_nonlingate", ptr(odv), ptr(stv), ptr(sdd), ptr(gov), ptr(gdd), n, TANH)
    assert_close(backend.down(sdd), sd_ref); assert_close(backend.down(gdd), gd_ref)


def test_clip_and_sgd(backend, ora32):
    oc = OraCall(ora32)
    n = 50
    d0 = cosdata(n, 3) * 300; v0 = cosdata(n, 60)
    d_ref = d0.copy(); oc.clip_gradient(d_ref, n, C.c_float(100.0))
    d = backend.up(d0); backend.lib.call("clstm_clip_gradient", ptr(d), n, 100.0)
    assert np.array_equal(backend.down(d), d_ref)
    backend.lib.call("clstm_clip_gradient", ptr(d), n, 1e6)            # no-op branch (:554)
    assert np.array_equal(backend.down(d), d_ref)
    v_ref = v0.copy(); oc.sgd_update(v_ref, d_ref, n, C.c_float(1e-2), C.c_float(0.9))
    v = backend.up(v0); backend.lib.call("clstm_sgd_update", ptr(v), ptr(d), n, 1e-2, 0.9)
    assert_close(backend.down(v), v_ref, rtol=1e-6); assert_close(backend.down(d), d_ref, rtol=1e-6)


# ---- CTC through the batched ABI -------------------------------------------------------------------
def ctc_via_abi(backend, probs_list, states_list, want_aligned=True):
    nc = probs_list[0].shape[1]
    loff = np.concatenate([[0], np.cumsum([len(p) for p in probs_list])]).astype(np.int32)
    soff = np.concatenate([[0], np.cumsum([len(s) for s in states_list])]).astype(np.int32)
    states = np.concatenate(states_list).astype(np.int32)
    P_ = backend.up(np.concatenate(probs_list, 0))
    D_ = backend.zeros((loff[-1], nc)); A_ = backend.zeros((loff[-1], nc))
    backend.lib.call("clstm_ctc_align_batch", ptr(P_), ptr(D_), ptr(A_), nc, ptr(loff), ptr(states), ptr(soff),
                     len(probs_list))
    return backend.down(A_), backend.down(D_), loff


def test_ctc_reference_known_answers(backend):
    """test-ctc.cc:47-74 and :76-109 through the device kernel (tolerance 1e-4 as asserted there)."""
    o1 = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1]], np.float32).T
    o2 = np.array([[1, .5, 0, 0, 0, 0], [0, .5, .5, 0, 0, 0], [0, 0, .5, .5, 0, 0], [0, 0, 0, .5, .5, 0],
                   [0, 0, 0, 0, .5, 1]], np.float32).T
    e2 = np.array([[1., 0.12029, 0., 0., 0., 0.], [0., 0.87971, 0.40013, 0., 0., 0.],
                   [0., 0., 0.59987, 0.59987, 0., 0.], [0., 0., 0., 0.40013, 0.87971, 0.],
                   [0., 0., 0., 0., 0.12029, 1.]], np.float32).T
    a1, _, _ = ctc_via_abi(backend, [o1], [np.arange(3)])
    assert np.abs(a1 - o1).max() < 1e-4
    a2, d2, _ = ctc_via_abi(backend, [o2], [np.arange(5)])
    assert np.abs(a2 - e2).max() < 1e-4
    assert np.allclose(d2, a2 - o2, atol=1e-7)


@pytest.mark.parametrize("T,L,nc", [(12, 3, 6), (40, 12, 30), (7, 3, 5), (3, 1, 83), (60, 40, 30), (100, 62, 70), (200, 70, 90), (256, 25, 83),
                                    # 65..128 states: one wave per direction, two states per lane (round 4) -- the configs[4] transcript length
                                    # (101 / 99 / 97 states) on the tiled path, 65 states (the smallest), 127 / 125 / 123 (the largest)
                                    (400, 50, 100), (90, 32, 40), (150, 63, 70)])
def test_ctc_vs_oracle(backend, ora32, T, L, nc):
    # (12..40: short-line path, lattice resident in LDS; 60 x 81 states: the same with the wide recursion;
    #  100 x 125 states and 200 x 141: the tiled path through HBM; 256 x 51 = 13056 cells: the largest OCR-shaped lattice of
    #  the short-line path, 26 cells per thread)
    rng = np.random.default_rng(T)
    probs, states = [], []
    for b in range(3):
        Tb = max(1, T - 3 * b)
        p = rng.random((Tb, nc)).astype(np.float32) ** 4
        p /= p.sum(1, keepdims=True)
        p[0, 1] = 0.0                                     # exercises the 1e-5 floor (ctc.cc:68)
        tr = rng.integers(1, nc, max(1, L - b))
        probs.append(p.astype(np.float32)); states.append(ora32.mktargets(tr))
    al, dz, loff = ctc_via_abi(backend, probs, states)
    for b in range(3):
        want = ora32.ctc_align_classes(probs[b], states[b])
        assert_close(al[loff[b]:loff[b + 1]], want, rtol=1e-4, atol=1e-6, what="aligned line %d" % b)
        assert_close(dz[loff[b]:loff[b + 1]], want - probs[b], rtol=1e-4, atol=2e-6, what="delta line %d" % b)


@pytest.mark.parametrize("T,L,nc", [(40, 10, 30), (200, 25, 83), (256, 25, 83), (200, 70, 40), (400, 50, 100)])
def test_ctc_float_logadd_option(backend, ora32, T, L, nc):
    """Experiment option ctc_float=1 (NOT the default): log_add of the lattice recursion on the float
    transcendentals instead of the double evaluation that reproduces glibc's roundings (cr_math.h).  Its stated tolerance on the
    alignment posteriors: 1e-4 ABSOLUTE -- the bar of the reference's own test (test-ctc.cc:98-104) -- and 1e-3 relative, against
    1e-4 relative for the default; every lattice path (short line from LDS / HBM, two states per lane, tiled)."""
    from test_net_parity import set_opt
    set_opt(backend, "ctc_float", 1)
    try:
        rng = np.random.default_rng(T + L)
        probs, states = [], []
        for b in range(3):
            Tb = max(1, T - 3 * b)
            p = rng.random((Tb, nc)).astype(np.float32) ** 4
            p /= p.sum(1, keepdims=True)
            p[0, 1] = 0.0
            tr = rng.integers(1, nc, max(1, L - b))
            probs.append(p.astype(np.float32)); states.append(ora32.mktargets(tr))
        al, dz, loff = ctc_via_abi(backend, probs, states)
        for b in range(3):
            want = ora32.ctc_align_classes(probs[b], states[b])
            assert_close(al[loff[b]:loff[b + 1]], want, rtol=1e-3, atol=1e-4, what="aligned line %d" % b)
            assert_close(dz[loff[b]:loff[b + 1]], want - probs[b], rtol=1e-3, atol=1e-4, what="delta line %d" % b)
        # the reference's known answer, at its own tolerance
        o2 = np.array([[1, .5, 0, 0, 0, 0], [0, .5, .5, 0, 0, 0], [0, 0, .5, .5, 0, 0], [0, 0, 0, .5, .5, 0], [0, 0, 0, 0, .5, 1]], np.float32).T
        e2 = np.array([[1., 0.12029, 0., 0., 0., 0.], [0., 0.87971, 0.40013, 0., 0., 0.], [0., 0., 0.59987, 0.59987, 0., 0.],
                       [0., 0., 0., 0.40013, 0.87971, 0.], [0., 0., 0., 0., 0.12029, 1.]], np.float32).T
        a2, _, _ = ctc_via_abi(backend, [o2], [np.arange(5)])
        assert np.abs(a2 - e2).max() < 1e-4
    finally:
        set_opt(backend, "ctc_float", 0)


@pytest.mark.parametrize("L,nc", [(1, 5), (25, 83), (31, 40), (25, 60), (40, 30), (63, 70)])
def test_ctc_recursion_ring_boundaries(backend, ora32, L, nc):
    """The one-wave lattice recursions request their match scores CTC_PD = 8 frames ahead through a register ring: line lengths
    around the ring size and its multiples (the loop's rounds, its unrolled tail of up to 15 steps, lines shorter than the ring).
    (L, nc) pick the source of the scores: 3 / 51 / 63 states with the scores in LDS (ctc_lattice<true, true>), 51 states over
    60 classes with LDS lattices but the scores from HBM (the carve has no room behind the lattices), 81 / 127 states on two
    states per lane with everything in HBM."""
    rng = np.random.default_rng(100 * L + nc)
    probs, states = [], []
    for T in (1, 2, 3, 7, 8, 9, 15, 16, 17, 23, 24, 25, 33):
        p = rng.random((T, nc)).astype(np.float32) ** 3
        p /= p.sum(1, keepdims=True)
        tr = rng.integers(1, nc, L)
        tr[::3] = tr[0]                                   # repeated labels: the read-modify-write columns of phase E
        probs.append(p.astype(np.float32)); states.append(ora32.mktargets(tr))
    al, dz, loff = ctc_via_abi(backend, probs, states)
    for b in range(len(probs)):
        want = ora32.ctc_align_classes(probs[b], states[b])
        assert_close(al[loff[b]:loff[b + 1]], want, rtol=1e-4, atol=1e-6, what="aligned, T = %d" % len(probs[b]))
        assert_close(dz[loff[b]:loff[b + 1]], want - probs[b], rtol=1e-4, atol=2e-6, what="delta, T = %d" % len(probs[b]))


def test_ctc_ragged_ocr_minibatch(backend, ora32):
    """One launch over lines as a ragged OCR minibatch has them (83 classes): 256 frames -- the longest line whose phases C / D run
    with lane = state --, 257 and 261 frames (generic phases C / D, match scores still in LDS), lines around the bench length with
    25 / 31 / 10 labels, short lines, a one-label line; every line against the oracle."""
    rng = np.random.default_rng(83)
    nc = 83
    shapes = [(150, 25), (201, 31), (256, 25), (257, 25), (261, 25), (233, 10), (48, 5), (47, 5), (9, 1)]
    if backend.kind == "emu":
        shapes = [(256, 25), (257, 25), (261, 25), (48, 5), (9, 1)]
    probs, states = [], []
    for T, L in shapes:
        p = rng.random((T, nc)).astype(np.float32) ** 4
        p /= p.sum(1, keepdims=True)
        probs.append(p.astype(np.float32)); states.append(ora32.mktargets(rng.integers(1, nc, L)))
    al, dz, loff = ctc_via_abi(backend, probs, states)
    for b in range(len(probs)):
        want = ora32.ctc_align_classes(probs[b], states[b])
        assert_close(al[loff[b]:loff[b + 1]], want, rtol=1e-4, atol=1e-6, what="aligned, line %d (T = %d)" % (b, len(probs[b])))
        assert_close(dz[loff[b]:loff[b + 1]], want - probs[b], rtol=1e-4, atol=2e-6, what="delta, line %d" % b)


@pytest.mark.parametrize("case", ["no_blank", "one_state", "one_frame", "all_same", "many_classes", "huge_classes"])
def test_ctc_target_shapes(backend, ora32, case):
    """Targets as plain class lists (the Classes overload, ctc.cc:136-146): the short-line path classifies the
    states into blank / first of its class / repeat -- exercise each combination."""
    rng = np.random.default_rng(len(case))
    T, nc = 9, 7
    if case == "no_blank":
        st = np.array([3, 1, 4, 1, 5, 2, 6], np.int32)          # no class-0 state at all, one repeat
    elif case == "one_state":
        st = np.array([0], np.int32)
    elif case == "one_frame":
        T, st = 1, np.array([0, 2, 0], np.int32)
    elif case == "all_same":
        st = np.array([2, 2, 2, 2, 2], np.int32)                # four repeats of one label, no blank
    elif case == "many_classes":
        T, nc = 20, 600                                          # > 512 classes: the tiled path
        st = ora32.mktargets([17, 599, 17, 300])
    else:
        T, nc = 4, 25000                                         # one frame of posteriors per LDS tile
        st = ora32.mktargets([7, 24999])
    p = rng.random((T, nc)).astype(np.float32) ** 3
    p /= p.sum(1, keepdims=True)
    al, dz, _ = ctc_via_abi(backend, [p.astype(np.float32)], [st])
    want = ora32.ctc_align_classes(p.astype(np.float32), st)
    assert_close(al, want, rtol=1e-4, atol=1e-6, what="aligned " + case)
    assert_close(dz, want - p, rtol=1e-4, atol=2e-6, what="delta " + case)


@pytest.mark.parametrize("T,L", [(40, 300), (640, 280)])
def test_ctc_long_transcripts(backend, ora32, T, L):
    """More than 512 target states per line (text lines of clstmfiltertrain easily exceed 255 characters; the
    reference's ctc_align_targets has no limit): 601 states on a short lattice (no complete path, the per-frame
    normalisation still applies) and 561 states over 640 frames.  The limit is now 2048 states."""
    rng = np.random.default_rng(L)
    nc = 40
    p = rng.random((T, nc)).astype(np.float32) ** 2
    p /= p.sum(1, keepdims=True)
    st = ora32.mktargets(rng.integers(1, nc, L))
    assert st.size == 2 * L + 1 and st.size > 512
    al, dz, _ = ctc_via_abi(backend, [p.astype(np.float32)], [st])
    want = ora32.ctc_align_classes(p.astype(np.float32), st)
    assert_close(al, want, rtol=1e-4, atol=1e-6, what="aligned, %d states" % st.size)
    assert_close(dz, want - p, rtol=1e-4, atol=2e-6, what="delta, %d states" % st.size)


@pytest.mark.parametrize("T,L", [(16, 300), (10, 600), (20, 256)])
def test_ctc_many_states_on_a_short_lattice(backend, ora32, T, L):
    """T * S <= 12288 with S > 512: the lattice would fit the short-line path's LDS tile, but that path classifies one
    target state per thread -- such lines must take the tiled path (an earlier version did not: max error 0.43)."""
    rng = np.random.default_rng(T * 1000 + L)
    nc = 40
    p = rng.random((T, nc)).astype(np.float32) ** 2
    p /= p.sum(1, keepdims=True)
    st = ora32.mktargets(rng.integers(1, nc, L))
    assert T * st.size <= 12288 and st.size > 512
    al, dz, _ = ctc_via_abi(backend, [p.astype(np.float32)], [st])
    want = ora32.ctc_align_classes(p.astype(np.float32), st)
    assert_close(al, want, rtol=1e-4, atol=1e-6, what="aligned, %d states on %d frames" % (st.size, T))
    assert_close(dz, want - p, rtol=1e-4, atol=2e-6, what="delta")


@pytest.mark.parametrize("T,L", [(40, 1030), (7, 1500), pytest.param(2300, 1100, marks=pytest.mark.gpu)])
def test_ctc_unbounded_label_axis(backend, ora32, T, L):
    """More than 2048 target states per line (transcripts of more than 1023 labels): the reference's ctc_align_targets has
    no limit (ctc.cc:57-112).  Such lines take the kernel's round-by-round recursion with the per-state totals in memory;
    a minibatch mixes one with an ordinary line (the LDS carve is sized for the ordinary ones)."""
    if T > 1000 and backend.kind == "emu":
        pytest.skip("GPU-sized case")
    rng = np.random.default_rng(L)
    nc = 30
    ps, sts = [], []
    for (t, l) in ((T, L), (25, 6)):
        p = rng.random((t, nc)).astype(np.float32) ** 2
        p /= p.sum(1, keepdims=True)
        ps.append(p.astype(np.float32)); sts.append(ora32.mktargets(rng.integers(1, nc, l)))
    assert sts[0].size > 2048
    al, dz, _ = ctc_via_abi(backend, ps, sts)
    want = np.concatenate([ora32.ctc_align_classes(p, st) for p, st in zip(ps, sts)], 0)
    assert_close(al, want, rtol=1e-4, atol=1e-6, what="aligned, %d states" % sts[0].size)
    assert_close(dz, want - np.concatenate(ps, 0), rtol=1e-4, atol=2e-6, what="delta")


def test_ctc_more_states_than_frames(backend, ora32):
    # S > T: no complete path exists; the reference still returns per-frame-normalised values
    rng = np.random.default_rng(9)
    p = rng.random((3, 5)).astype(np.float32); p /= p.sum(1, keepdims=True)
    st = ora32.mktargets([1, 2, 3, 4])
    al, _, _ = ctc_via_abi(backend, [p], [st])
    assert_close(al, ora32.ctc_align_classes(p, st), rtol=1e-4, atol=1e-6)


def test_decode_batch(backend, ora32):
    rng = np.random.default_rng(11)
    nc = 7
    Ts = [1, 9, 70, 130]
    probs = []
    for T in Ts:
        p = rng.random((T, nc)).astype(np.float32)
        p[:, 0] *= 3.0                                    # plenty of blank frames
        p /= p.sum(1, keepdims=True)
        probs.append(p)
    probs[1][4, 2] = probs[1][4, 5] = probs[1][4].max() + 0.1   # an exact tie -> last index wins
    loff = np.concatenate([[0], np.cumsum(Ts)]).astype(np.int32)
    P_ = backend.up(np.concatenate(probs, 0))
    cls = np.zeros(loff[-1], np.int32); loc = np.zeros(loff[-1], np.int32); cnt = np.zeros(len(Ts), np.int32)
    backend.lib.call("clstm_trivial_decode_batch", ptr(P_), nc, ptr(loff), len(Ts), ptr(cls), ptr(loc), ptr(cnt))
    for b, T in enumerate(Ts):
        wc, wl = ora32.trivial_decode(probs[b])
        assert cls[loff[b]:loff[b] + cnt[b]].tolist() == wc.tolist()
        assert loc[loff[b]:loff[b] + cnt[b]].tolist() == wl.tolist()


def test_mktargets(backend):
    st = np.zeros(7, np.int32); tr = np.array([5, 7, 7], np.int32)
    backend.lib.call("clstm_mktargets", ptr(st), ptr(tr), 3)
    assert st.tolist() == [0, 5, 0, 7, 0, 7, 0]
