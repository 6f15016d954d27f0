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


def _per_op_rate(kind, args, env_extra):
    subprocess.check_call(["make", "-C", INTEG, "-s", kind])
    exe = os.path.join(INTEG, "build", "per_op_rate_" + kind)
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env_extra))
    assert r.returncode == 0 and "per-op drop-in:" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1]


def test_literal_per_operator_drop_in_runs_on_the_emulator():
    """integration/per_op_rate.cc: the reference's layer loops (NPLSTM / Parallel / Reversed / softmax, clstm.cc:405-653) over the
    per-operator shim -- the literal drop-in of INTEGRATION.md 1 -- run one training pass of a small BiLSTM."""
    _per_op_rate("emu", [12, 8, 6, 5, 1], {"CLSTM_EMU_CUS": "8"})


@pytest.mark.gpu
def test_literal_per_operator_drop_in_rate_on_the_gpu():
    """VERDICT r3 'next round' 2: the rate of the LITERAL drop-in (one launch per operator and time step, bs = 1, the fixture
    line's 447 frames) for INTEGRATION.md, next to the fused adapter's and the fused ABI's."""
    print(_per_op_rate("hip", [447, 48, 50, 83, 5], {}))


# ---- the reference's OWN drivers over the fused-level INetwork adapter (integration/inetwork/) ------------------------------
REF = "/root/reference"
REFBIN = os.path.join(INTEG, "_ref")


def _build_drop_in():
    """integration/Makefile ref_drop_in: clstmocrtrain.cc, clstmocr.cc, clstmhl.h, extras.h, utils.h, pstring.h compiled
    UNMODIFIED from /root/reference (a directory of symbolic links shadows only clstm.h and tensor.h).  The reference tree
    exists in the build container only: on the GPU box the prebuilt binaries under integration/_ref/ are used as shipped."""
    if os.path.isdir(REF):
        subprocess.check_call(["make", "-C", INTEG, "-s", "ref_drop_in"])
    for name in ("clstmocrtrain_emu", "clstmocr_emu", "clstmocrtrain_hip", "clstmocr_hip"):
        if not os.path.exists(os.path.join(REFBIN, name)):
            pytest.skip("integration/_ref/%s not built (needs /root/reference at build time: __graft_entry__.build())" % name)


def test_reference_drivers_unmodified_over_the_inetwork_adapter_on_the_emulator(tmp_path):
    """VERDICT r3 'Missing 2': the reference's own clstmocrtrain.cc / clstmhl.h, unmodified, train through
    make_net("bidi") -> INetwork::forward()/backward() -> sgd_update(net) of integration/inetwork (one clstm_net_forward /
    clstm_net_backward / clstm_net_update each) and must leave the model this repo's own driver leaves after the same
    draws: same report lines, parameters equal to float noise (both go through the same kernels; the adapter forms
    outputs.d = aligned - outputs on the host, as clstmhl.h:211 writes it).  Then the reference's clstmocr.cc loads that
    model file and prints one line per image."""
    import numpy as np
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_distributed import _driver_fixture, _model_params
    _build_drop_in()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "clstm_amd", "host"), "-s", "all"])
    emu_dir = os.path.join(ROOT, "tests", "hipemu", "build")
    mine = tmp_path / "clstmocrtrain_mine_emu"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(mine), os.path.join(ROOT, "clstm_amd", "host", "clstmocrtrain.cc"),
                           "-L" + emu_dir, "-lclstm_emu", "-Wl,-rpath," + emu_dir, "-lz", "-pthread"])
    lst = _driver_fixture(tmp_path)
    tool = os.path.join(ROOT, "clstm_amd", "bin", "clstm_hosttool")
    out = {}
    for tag, exe in (("ref", os.path.join(REFBIN, "clstmocrtrain_emu")), ("mine", str(mine))):
        env = dict(os.environ, ntrain="6", nhidden="6", target_height="12", lrate="1e-2", report_every="2", save_every="1000",
                   save_name=str(tmp_path / tag), seed="0.222")
        r = subprocess.run([exe, str(lst)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
        assert "#: ntrain = 6" in r.stderr
        model = tmp_path / (tag + "-5.clstm")
        assert model.exists(), r.stdout[-800:]
        out[tag] = (_model_params(tool, model, tmp_path, tag), [l for l in r.stdout.splitlines() if l[:4] in ("TRU ", "ALN ", "OUT ")])
    assert out["ref"][1] == out["mine"][1] and len(out["ref"][1]) == 12, (out["ref"][1], out["mine"][1])
    assert out["ref"][0].size == out["mine"][0].size and np.abs(out["ref"][0]).max() > 0
    assert np.allclose(out["ref"][0], out["mine"][0], rtol=1e-5, atol=1e-7), np.abs(out["ref"][0] - out["mine"][0]).max()
    r2 = subprocess.run([os.path.join(REFBIN, "clstmocr_emu"), str(lst)], env=dict(os.environ, load=str(tmp_path / "ref-5.clstm")),
                        capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-1500:]
    assert r2.stdout.count("\t") == 2            # one "file<TAB>text" line per image (clstmocr.cc:95)


@pytest.mark.gpu
def test_reference_test_ocr_scenario_through_the_unmodified_reference_drivers(tmp_path):
    """test-ocr.sh:4-8 with the REFERENCE'S OWN clstmocrtrain.cc / clstmocr.cc (compiled unmodified against
    integration/inetwork, linked with libclstm_hip.so): 201 updates on misc/textline.bin.png at lrate 1e-2, then
    load=_ocrtest-200.clstm clstmocr must print 'performance analysis'."""
    _build_drop_in()
    fixture = os.path.join(ROOT, "tests", "golden", "textline.bin.png")
    png = tmp_path / "textline.bin.png"
    png.write_bytes(open(fixture, "rb").read())
    (tmp_path / "textline.gt.txt").write_text("performance analysis\n", encoding="utf-8")
    lst = tmp_path / "_ocrtest.txt"
    lst.write_text(str(png) + "\n")
    env = dict(os.environ, ntrain="201", hidden="50", lrate="1e-2", save_name=str(tmp_path / "_ocrtest"), seed="0.222", report_time="1",
               CLSTM_ADAPTER_TIMING="1")
    r = subprocess.run([os.path.join(REFBIN, "clstmocrtrain_hip"), str(lst)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2000:])
    assert "TRU performance analysis" in r.stdout and "saving" in r.stdout and "steptime" in r.stdout
    model = tmp_path / "_ocrtest-200.clstm"
    assert model.exists()
    r2 = subprocess.run([os.path.join(REFBIN, "clstmocr_hip"), str(lst)], env=dict(os.environ, load=str(model)), capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "performance analysis" in r2.stdout
    assert (tmp_path / "textline.bin.txt").read_text().strip() == "performance analysis"
    # the literal drop-in's rate, for INTEGRATION.md (one line per update, every Sequence through host memory)
    st = [float(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("steptime")]
    print("drop-in (reference drivers + INetwork adapter): steptime %.2f ms per line (T = 447 frames)" % (1e3 * sorted(st)[len(st) // 2]))
    # ... and where it goes (CLSTM_ADAPTER_TIMING=1): host preparation of the sample against the calls behind INetwork
    split = [l for l in r.stderr.splitlines() if l.startswith("adapter_time")]
    assert any("forward" in l for l in split) and any("normalizer" in l for l in split), r.stderr[-1500:]
    for l in split: print("drop-in", l)
