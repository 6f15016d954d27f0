"""Pin the oracle's CTC against the reference's own known-answer tests
(test-ctc.cc:47-74 identity case, :76-109 5x6 case, tolerance 1e-4 as asserted there)."""
import numpy as np
import pytest


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_ctc_kat1(prec, ora32, ora64):
    ora = ora32 if prec == "f32" else ora64
    # test-ctc.cc:50-54 fills (3x4) then transposes -> outputs is T=4 x nc=3
    outputs = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1]], float).T
    targets = np.eye(3).T
    post = ora.ctc_align_targets(outputs, targets)
    expected = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1]], float).T
    assert np.abs(post - expected).max() < 1e-4


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_ctc_kat2(prec, ora32, ora64):
    ora = ora32 if prec == "f32" else ora64
    outputs = np.array([
        [1, .5, 0, 0, 0, 0],
        [0, .5, .5, 0, 0, 0],
        [0, 0, .5, .5, 0, 0],
        [0, 0, 0, .5, .5, 0],
        [0, 0, 0, 0, .5, 1]], float).T          # T=6 x nc=5   (test-ctc.cc:79-86)
    targets = np.eye(5).T                        # S=5 x nc=5   (test-ctc.cc:87-94)
    expected = np.array([
        [1., 0.12029, 0., 0., 0., 0.],
        [0., 0.87971, 0.40013, 0., 0., 0.],
        [0., 0., 0.59987, 0.59987, 0., 0.],
        [0., 0., 0., 0.40013, 0.87971, 0.],
        [0., 0., 0., 0., 0.12029, 1.]], float).T  # test-ctc.cc:97-104
    post = ora.ctc_align_targets(outputs, targets)
    assert np.abs(post - expected).max() < 1e-4
    # the Classes overload (ctc.cc:136-146) must agree with explicit one-hot targets
    post2 = ora.ctc_align_classes(outputs, [0, 1, 2, 3, 4])
    assert np.array_equal(post, post2)


def test_mktargets_and_decode(ora32):
    st = ora32.mktargets([5, 7, 7])
    assert st.tolist() == [0, 5, 0, 7, 0, 7, 0]          # ctc.cc:148-157
    # trivial_decode (ctc.cc:159-190): strongest class per non-blank run; a trailing run
    # not closed by a blank frame is dropped.
    nc = 4
    frames = [0, 1, 1, 0, 2, 3, 0, 1]
    probs = np.full((len(frames), nc), 0.1, np.float32)
    vals = [0.9, 0.6, 0.8, 0.9, 0.5, 0.7, 0.9, 0.9]
    for t, (c, v) in enumerate(zip(frames, vals)):
        probs[t, c] = v
    cs, locs = ora32.trivial_decode(probs)
    assert cs.tolist() == [1, 3]
    assert locs.tolist() == [2, 5]


def test_argmax_ties_last(ora32):
    import ctypes as C
    a = np.array([0.2, 0.5, 0.5, 0.1], np.float32)
    assert ora32.lib.ora_argmax(a.ctypes.data_as(C.c_void_p), 4) == 2   # tensor.h:357-366
