"""integration/clstm_compute_hip.cc -- the translation unit a clstm maintainer would add (every DEFGENERIC operator
of clstm_compute.h:72-103 forwarded to the C ABI) -- must COMPILE and RUN: the reference's per-operator derivative
test (test-cderiv.cc:140-430) is re-hosted on it in integration/test_cderiv_hip.cc.  CPU suite: linked against the
host-emulator build of the kernels, every 9th element (a launch costs milliseconds on the emulator); -m gpu: the real
library, every element, as the reference does."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTEG = os.path.join(ROOT, "integration")
CASES = ["TestBatchstack", "TestFull1Sigmoid", "TestFull1Tanh", "TestFull1Logmag", "TestStack", "TestStackDelay",
         "TestReverse", "TestBtswitch", "TestStatemem", "TestNonlingate", "TestSoftmaxCrossEntropy"]


def _run(target, env_extra):
    subprocess.check_call(["make", "-C", INTEG, "-s", target])
    exe = os.path.join(INTEG, "build", "test_cderiv_" + target)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=1500, env=dict(os.environ, **env_extra))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    for c in CASES:
        assert "testing " + c in r.stdout
    assert r.stdout.count("\nOK ") == len(CASES) and "ALL OK" in r.stdout and "FAIL" not in r.stdout


def test_shim_lists_every_defgeneric_operator():
    """27 DEFGENERIC declarations in clstm_compute.h:72-103; `fill` is declared on Tensor2& but defined on TensorMap2&
    and never linked -- the other 26 must be forwarded."""
    src = open(os.path.join(INTEG, "clstm_compute_hip.cc")).read()
    ops = ["forward_nonlin", "backward_nonlin", "forward_nonlin0", "backward_nonlin0", "forward_lin1", "backward_lin1",
           "forward_full1", "backward_full1", "forward_stack", "backward_stack", "forward_stack_delay",
           "backward_stack_delay", "forward_reverse", "backward_reverse", "forward_btswitch", "backward_btswitch",
           "forward_batchstack", "backward_batchstack", "forward_softmax", "backward_softmax", "forward_statemem",
           "backward_statemem", "forward_nonlingate", "backward_nonlingate", "clip_gradient", "sgd_update"]
    assert len(ops) == 26
    for op in ops:
        assert "void %s(HipDevice*" % op in src, op
        assert "clstm_%s(" % op in src, op


def test_cderiv_through_the_shim_on_the_emulator():
    _run("emu", {"CDERIV_STRIDE": "9"})


@pytest.mark.gpu
def test_cderiv_through_the_shim_on_the_gpu():
    _run("hip", {})
