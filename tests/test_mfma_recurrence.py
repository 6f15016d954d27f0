"""The narrow layer's recurrence batched over 16 lines per workgroup on the f16 MFMA (clstm_amd/csrc/lstm_mfma.h) against the oracle.

-m gpu only: the kernels exist for gfx950 alone (the host emulator keeps the per-line kernels).  The path is forced onto small
minibatches (experiment switch fwd_mfma=2; the default takes it from 640 lines per GPU on); `run_case` then checks EVERY saved
activation (gi, gf, go, ci, c, h of both directions, 1e-4 relative: BASELINE.json north_star), the softmax outputs, bit-exact
decodes, CTC, every gate delta, the minibatch gradient and the update -- and the path counter proves the MFMA kernel really ran.
Reference semantics: /root/reference/clstm.cc:600-653, clstm_compute.cc:275-320,504-547."""
import ctypes

import numpy as np
import pytest

from test_net_parity import run_case, set_opt, _forget_debug_options  # noqa: F401  (autouse fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    from common import Backend
    return Backend("hip")


def _count(backend, which):
    out = ctypes.c_longlong(0)
    backend.lib.call("clstm_debug_path_count", which, ctypes.byref(out))
    return out.value


@pytest.mark.parametrize("nh,T", [
    (100, [40, 23, 1, 70]),                                  # one partial group, a 1-frame line, ragged
    (100, [31] * 16),                                        # exactly one full group
    (100, [20, 37, 5, 64, 33, 12, 50, 41, 8, 29, 64, 3, 17, 45, 26, 58, 11, 39, 2, 61, 30, 7, 52, 19, 44,
           35, 9, 63, 24, 48, 15, 56, 28]),                  # 33 lines: two full groups and one line in a third
    (64, [25, 40, 13]),                                      # the 64-cell instantiation (no unpaired tile)
    (128, [18, 33]),                                         # the 128-cell instantiation
])
def test_mfma_recurrence_vs_oracle(backend, ora32, nh, T):
    set_opt(backend, "fwd_mfma", 2)
    set_opt(backend, "bwd_mfma", 2)
    before, before_b = _count(backend, 16), _count(backend, 17)
    run_case(backend, ora32, 48, nh, 83, T, scale=10.0)
    assert _count(backend, 16) > before, "the MFMA recurrence did not run"
    assert nh == 128 or _count(backend, 17) > before_b, "the MFMA backward recurrence did not run"   # (128 cells: forward only)


@pytest.mark.parametrize("rows", [1, 0])
@pytest.mark.parametrize("T", [[40, 23, 1, 70], [33] * 17, [7, 3, 2, 1, 5, 6, 4]])
def test_mfma_backward_alone(backend, ora32, T, rows):
    """the batched backward recurrence (lstm_mfma_bwd.h) behind the per-line forward kernel: its deltas, the weight gradient formed
    from them by the separate-launch weight-gradient items, the update -- in both forms of its memory side (rows = 1, the default:
    whole rows through LDS, lstm_bwd_mfma_rows_kernel; rows = 0: 64-byte pieces per lane group, lstm_bwd_mfma_kernel)"""
    set_opt(backend, "fwd_mfma", 0)
    set_opt(backend, "bwd_mfma", 2)
    set_opt(backend, "bwd_mfma_rows", rows)
    before = _count(backend, 17)
    run_case(backend, ora32, 48, 100, 83, T, scale=10.0)
    assert _count(backend, 17) > before


@pytest.mark.parametrize("seed", [3, 4, 5, 6, 7, 8])
def test_mfma_randomised_geometries(backend, ora32, seed):
    """random minibatches through both batched kernels: 1..40 lines of 1..90 frames (groups of 16 with lines that drop out of the
    lock-step at different steps, groups whose longest line has 1-3 frames: shorter than the kernels' peeled four steps and their
    operand-request pipeline), 64 or 100 cells -- every activation, every gate delta, the gradient and the update vs the oracle"""
    rng = np.random.default_rng(seed)
    for _ in range(4):
        bs = int(rng.integers(1, 41))
        T = [int(t) for t in rng.integers(1, 91, bs)]
        if rng.random() < 0.5:
            T[:min(bs, 16)] = [int(t) for t in rng.integers(1, 4, min(bs, 16))]     # a whole group of tiny lines (longest first: the LAST group)
        nh = int(rng.choice([64, 100]))
        fused = nh == 100 and rng.random() < 0.5     # (the one-launch form needs the in-launch items: 100 cells)
        set_opt(backend, "fwd_mfma", 2)
        set_opt(backend, "bwd_mfma", 2)
        set_opt(backend, "bwd_mfma_fused", 2 if fused else 0)
        before, before_1 = _count(backend, 17), _count(backend, 18)
        net, _ = run_case(backend, ora32, 48, nh, 83, T, scale=10.0, seed=int(rng.integers(1 << 30)), overlap=2 if fused else None)
        assert _count(backend, 17) > before
        if fused:
            assert _count(backend, 18) > before_1 and net.overlap_stats()[1] == 0


@pytest.mark.parametrize("T", [[40, 23, 1, 70], [33] * 17, [90, 64, 77, 12, 5, 81, 33, 90, 2, 64, 18, 71, 90, 45, 9, 60, 27, 88, 90, 3]])
def test_mfma_backward_and_weight_gradient_items_as_one_launch(backend, ora32, T):
    """lstm_mfma_bwd_dw.h: the batched backward recurrence in REPORT mode (write-through delta rows, per-step progress words) and the
    weight-gradient items of gemm_dw.h as two workgroup roles of ONE launch -- forced onto small minibatches (overlap mode 2):
    the deltas, the gradient the items form from them WHILE the recurrence runs, the update; no wait ran into its watchdog"""
    set_opt(backend, "fwd_mfma", 2)
    set_opt(backend, "bwd_mfma", 2)
    set_opt(backend, "bwd_mfma_fused", 2)
    before = _count(backend, 18)
    net, _ = run_case(backend, ora32, 48, 100, 83, T, scale=10.0, overlap=2)
    assert _count(backend, 18) > before, "the fused launch did not run"
    assert net.overlap_stats()[1] == 0


@pytest.mark.parametrize("bs,T", [(640, 24), (1024, 12)])
def test_mfma_default_rule_at_chip_filling_minibatches(backend, ora32, bs, T):
    """NO experiment option: what a user's 640- / 1024-line minibatch takes by the library's own rule -- the batched forward
    recurrence, the batched backward recurrence (640 lines: in ONE launch with the weight-gradient items; 1024: items behind it)
    -- against the oracle: every activation, gate delta, the gradient, the update.  Lines of 64..T frames so that the overlap
    rule (N >= 2048 frames, longest line >= 64) holds."""
    rng = np.random.default_rng(bs)
    Ts = [64] * 8 + [int(t) for t in rng.integers(1, T + 1, bs - 8)]
    before_f, before_b, before_1 = _count(backend, 16), _count(backend, 17), _count(backend, 18)
    run_case(backend, ora32, 48, 100, 83, Ts, scale=10.0, seed=bs)
    assert _count(backend, 16) > before_f and _count(backend, 17) > before_b
    assert (_count(backend, 18) > before_1) == (bs < 900)


def test_mfma_recurrence_large_weights(backend, ora32):
    """init x 60: saturated gates and |R| of order 1 -- the power-of-two scaling of the f16 split must follow the weights"""
    set_opt(backend, "fwd_mfma", 2)
    set_opt(backend, "bwd_mfma", 2)
    run_case(backend, ora32, 48, 100, 83, [50, 44, 37, 29, 18], scale=60.0, ctc_rtol=1e-3, grad_tol=1e-3)


def test_mfma_second_step_repacks(backend, ora32):
    """the fragments follow the parameters: two training steps, the second against the oracle's second"""
    from clstm_amd.net import Network
    from common import assert_close, oracle_minibatch, synth_lines
    from oracle.oracle import OracleNet
    set_opt(backend, "fwd_mfma", 2)
    set_opt(backend, "bwd_mfma", 2)
    rng = np.random.default_rng(5)
    ni, nh, nc, T = 48, 100, 83, [33, 21, 40, 12, 27]
    ref = OracleNet(ora32, ni, nh, nc, seed=0.222)
    params = ref.get_params() * 10.0
    net = Network(ni, nh, nc, lib=backend.lib)
    net.set_params(params)
    net.setLearningRate(1e-2, 0.9)
    onet = None
    for step in range(2):
        lines = synth_lines(rng, T, ni)
        trs = [rng.integers(1, nc, max(1, t // 3)).astype(np.int32) for t in T]
        want = oracle_minibatch(ora32, OracleNet, params if onet is None else onet.get_params(), ni, nh, nc, lines, trs,
                                derivs0=None if onet is None else onet.get_derivs(), lr=1e-2, mom=0.9)
        net.set_inputs(lines)
        net.forward()
        got = net.split(net.outputs())
        for b in range(len(T)):
            assert_close(got[b], want["outputs"][b], what="step %d outputs line %d" % (step, b))
        net.ctc(trs)
        net.backward()
        net.update()
        want["net"].update()
        onet = want["net"]
    assert_close(net.get_params(), onet.get_params(), rtol=1e-4, atol=1e-6, what="params after two updates")
