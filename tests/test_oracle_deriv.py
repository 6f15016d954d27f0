"""Pin the oracle's backward passes with the reference's own finite-difference method
(test-deriv.cc:100-177: forward-difference quotient over h in 1e-6..1e-1, best-h relative
error < 0.1 with the delta convention d = target - output, derivative = -2 * delta for the
squared loss) -- re-hosted on the double build (run-tests:16-17 builds test-*deriv with double=1).

On top of the reference's loose acceptance we assert a tight central-difference agreement.
The Softmax head is checked with the cross-entropy loss, for which `aligned - output` IS the
negative logit gradient (the convention CLSTMOCR::fwdbwd relies on, clstmhl.h:211-212)."""
import numpy as np
import pytest
from oracle.oracle import OracleNet


def cosrand(n, start=[1]):
    # test-deriv.cc:29-36: deterministic data cos(3.7*k) with |x| > 0.1
    out = []
    while len(out) < n:
        x = np.cos(start[0] * 3.7)
        start[0] += 1
        if abs(x) > 0.1:
            out.append(x)
    return np.array(out)


def ce_loss(p, y):
    return -(y * np.log(p)).sum()


@pytest.mark.parametrize("uni,bs", [(True, 1), (False, 1), (False, 2)])
def test_net_gradients_fd(ora64, uni, bs):
    ni, nh, nc, T = 7, 5, 3, 4
    net = OracleNet(ora64, ni, nh, nc, unidirectional=uni, seed=0.222)
    params = cosrand(net.nparams) * 0.5
    net.set_params(params)
    x = cosrand(T * bs * ni).reshape(T, bs, ni)
    y = np.abs(cosrand(T * bs * nc)).reshape(T, bs, nc)
    y /= y.sum(-1, keepdims=True)

    def loss_at(xv, pv):
        net.set_params(pv)
        net.set_inputs(xv)
        return ce_loss(net.forward(), y)

    net.set_params(params)
    net.clear_derivs()
    net.set_inputs(x)
    net.forward()
    net.set_targets(y)          # outputs.d = y - p  (clstm.cc:142-150)
    net.backward()
    din = net.input_deltas()    # = -dL/dx
    dpar = net.get_derivs()     # = -dL/dparam
    # inputs: the reference's acceptance (forward difference, best h, < 0.1) + tight central check
    rng = np.random.default_rng(0)
    for idx in rng.choice(x.size, 12, replace=False):
        t, b, i = np.unravel_index(idx, x.shape)
        best = np.inf
        for h in [1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1]:
            xp = x.copy(); xp[t, b, i] += h
            num = (loss_at(xp, params) - loss_at(x, params)) / h
            best = min(best, abs(1.0 - num / -din[t, b, i]))
        assert best < 0.1
        h = 1e-5
        xp = x.copy(); xp[t, b, i] += h
        xm = x.copy(); xm[t, b, i] -= h
        num = (loss_at(xp, params) - loss_at(xm, params)) / (2 * h)
        assert abs(num + din[t, b, i]) <= 1e-6 * max(1.0, abs(num))
    # parameters (every block of the flat buffer is sampled: all gates, both dirs, softmax)
    for idx in np.linspace(0, net.nparams - 1, 40).astype(int):
        h = 1e-5
        pp = params.copy(); pp[idx] += h
        pm = params.copy(); pm[idx] -= h
        num = (loss_at(x, pp) - loss_at(x, pm)) / (2 * h)
        assert abs(num + dpar[idx]) <= 1e-6 * max(1.0, abs(num)), idx


def test_bidi2_gradients_fd(ora64):
    ni, nc, T, bs = 5, 4, 3, 2
    net = OracleNet(ora64, ni, [4, 3], nc, seed=0.222)
    params = cosrand(net.nparams) * 0.5
    x = cosrand(T * bs * ni).reshape(T, bs, ni)
    y = np.abs(cosrand(T * bs * nc)).reshape(T, bs, nc)
    y /= y.sum(-1, keepdims=True)

    def loss_at(pv):
        net.set_params(pv)
        net.set_inputs(x)
        return ce_loss(net.forward(), y)

    net.set_params(params)
    net.clear_derivs()
    net.set_inputs(x)
    net.forward()
    net.set_targets(y)
    net.backward()
    dpar = net.get_derivs()
    for idx in np.linspace(0, net.nparams - 1, 60).astype(int):
        h = 1e-5
        pp = params.copy(); pp[idx] += h
        pm = params.copy(); pm[idx] -= h
        num = (loss_at(pp) - loss_at(pm)) / (2 * h)
        assert abs(num + dpar[idx]) <= 1e-6 * max(1.0, abs(num)), idx


def test_param_count_and_flat_order(ora32):
    # bidi(48,100,83) = 8 x 14,900 + 16,683 = 135,883 floats
    net = OracleNet(ora32, 48, 100, 83, seed=0.222)
    assert net.nparams == 135883
    # init order = fwd LSTM (WGI,WGF,WGO,WCI), rev LSTM, softmax (clstm.cc:587-590, prefab :52-68)
    # while the flat order is alphabetical (WCI,WGF,WGI,WGO): the very first LCG draw must land
    # at WGI(0,0) = flat block 2 of the forward LSTM.
    import ctypes as C
    ora32.lib.ora_seed(0.222)
    u = ora32.lib.ora_randu()
    p = net.get_params()
    blk = 100 * 149
    assert np.isclose(p[2 * blk], np.float32(3 * 0.01 * u - 2 * 0.01), rtol=0, atol=1e-9)
    assert (p >= -0.02 - 1e-7).all() and (p <= 0.01 + 1e-7).all()   # negbiased: 3su - 2s


@pytest.mark.parametrize("kind,ni,nh,nc", [("bidi", 48, [100], 83), ("bidi2", 9, [7, 5], 11), ("lstm1", 7, [3], 4)])
def test_init_matches_oracle_rinit(ora32, kind, ni, nh, nc, tmp_path):
    """a29: `rinit` + the LCG (batches.cc:11-52, clstm.cc:30-36).  The Python mirror (clstm_amd/init.py) and the
    C++ host mirror (clstm_amd/host/model.h, through `clstm_hosttool init-model`) must be BIT-equal to the
    oracle's ora_rinit for all three prefabs -- parity tests and the drivers start from these weights."""
    import os
    import subprocess
    from clstm_amd.init import init_params
    uni = kind == "lstm1"
    want = OracleNet(ora32, ni, nh, nc, unidirectional=uni, seed=0.222).get_params()
    got = init_params(ni, nh, nc, unidirectional=uni, seed=0.222)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "init.py differs from ora_rinit"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "clstm_amd", "bin", "clstm_hosttool")
    if not os.path.exists(tool):
        pytest.skip("host tools not built (they link libclstm_hip.so)")
    model, raw = str(tmp_path / "m.clstm"), str(tmp_path / "p.f32")
    subprocess.check_call([tool, "init-model", kind, str(ni), str(nh[0]), str(nh[1] if len(nh) > 1 else 0),
                           str(nc), "0.222", model])
    subprocess.check_call([tool, "params", model, raw])
    host = np.fromfile(raw, np.float32)
    assert np.array_equal(host.view(np.uint32), want.view(np.uint32)), "host/model.h init differs from ora_rinit"
