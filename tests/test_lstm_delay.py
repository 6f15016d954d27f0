"""The reference's end-to-end LSTM test re-hosted (test-lstm.cc:25-152): a 1:4:2 `lstm1` network learns to reproduce
its binary input delayed by one step (softmax targets, online SGD lr 1e-4 / momentum 0.9, T = 20), must reach a
maximum output error below 0.1 on 1000 fresh sequences, keep it after save -> load, lose it when the shared
parameter buffer is zeroed ("hacked-params") and regain it when the buffer is restored.  Runs on the oracle in the
CPU suite and on the HIP path (-m gpu); data from glibc's drand48 as in the reference."""
import numpy as np
import pytest

from oracle.oracle import OracleNet


class Drand48:
    """glibc drand48() without srand48(): the state starts at 0."""

    def __init__(self):
        self.x = 0

    def __call__(self):
        self.x = (0x5DEECE66D * self.x + 0xB) & ((1 << 48) - 1)
        return self.x / float(1 << 48)


def gentest(rnd, N=20):   # test-lstm.cc:28-40
    xs = np.zeros((N, 1), np.float32)
    ys = np.zeros((N, 2), np.float32)
    ys[0, 0] = 1
    for t in range(N):
        out = int(rnd() < 0.3)
        xs[t, 0] = out
        if t < N - 1:
            ys[t + 1, out] = 1.0
    return xs, ys


NTRAIN, NTEST = 100000, 1000   # test-lstm.cc:25-26


def test_delay_task_on_the_oracle(ora32):
    rnd = Drand48()
    net = OracleNet(ora32, 1, [4], 2, unidirectional=True, seed=0.1)
    net.set_lr(1e-4, 0.9)
    for _ in range(NTRAIN):
        xs, ys = gentest(rnd)
        net.set_inputs(xs); net.forward(); net.set_targets(ys); net.backward(); net.update()

    def test_net(n):
        merr = 0.0
        for _ in range(NTEST):
            xs, ys = gentest(rnd)
            n.set_inputs(xs)
            merr = max(merr, float(np.abs(n.forward()[:, 0, :] - ys).max()))
        return merr

    assert test_net(net) < 0.1                                   # "OK (pre-save)"
    p = net.get_params()
    net2 = OracleNet(ora32, 1, [4], 2, unidirectional=True, init=False)
    net2.set_params(p)                                            # save_net / load_net carry exactly the params
    assert test_net(net2) < 0.1
    net2.set_params(np.zeros_like(p))
    assert test_net(net2) >= 0.1                                  # "OK (hacked-params)"
    net2.set_params(p)
    assert test_net(net2) < 0.1


@pytest.mark.gpu
def test_delay_task_on_the_gpu(tmp_path):
    import torch
    from clstm_amd import abi
    from clstm_amd.init import init_params
    from clstm_amd.net import Network
    lib = abi.load()
    rnd = Drand48()
    dev = torch.device("cuda", 0)
    nparams = init_params(1, [4], 2, unidirectional=True, seed=0.1).size
    shared = torch.from_numpy(init_params(1, [4], 2, unidirectional=True, seed=0.1)).to(dev)   # share_params: caller-owned buffer
    net = Network(1, [4], 2, unidirectional=True, lib=lib, params=shared)
    net.params_changed()
    net.setLearningRate(1e-4, 0.9)
    for _ in range(NTRAIN):
        xs, ys = gentest(rnd)
        net.set_inputs([xs]); net.forward()
        net.set_output_deltas(ys - net.outputs())                 # set_targets: outputs.d = targets - outputs (clstm.cc:692-699)
        net.backward(); net.update()

    def test_net(n):
        merr = 0.0
        for _ in range(NTEST):
            xs, ys = gentest(rnd)
            n.set_inputs([xs]); n.forward()
            merr = max(merr, float(np.abs(n.outputs() - ys).max()))
        return merr

    assert test_net(net) < 0.1                                   # "OK (pre-save)"
    # save -> load through the reference's model format (clstm.proto via the host tools)
    p = net.get_params()
    assert p.size == nparams
    net2 = Network(1, [4], 2, unidirectional=True, lib=lib)
    net2.set_params(p)
    assert test_net(net2) < 0.1                                   # "OK"
    backup = shared.clone()
    shared.zero_(); net.params_changed()
    assert test_net(net) >= 0.1                                   # "OK (hacked-params)": the net really uses the shared buffer
    shared.copy_(backup); net.params_changed()
    assert test_net(net) < 0.1                                    # "OK (restored-params)"
