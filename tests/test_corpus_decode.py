"""Decode identity on an OCR corpus: the stand-in for BASELINE.json's "CTC decode identical to reference on uw3-500".

UW3-500 is a download (/root/reference/run-uw3-500:5) and there is no network, so the corpus is RENDERED where the test runs
(scripts/make_corpus.py: 512 distinct lines of English running text, six DejaVu faces, 26-44 px, shear, speckle).  The drop-in driver trains the
uw3 architecture on it exactly as a user would (`clstmocrtrain batch=64`, 1,000 updates, PNG files -> CenterNormalizer ->
device), saves the model through the clstm.proto writer, and then EVERY line of the corpus is decoded three ways on that saved
model:
  * `clstmocr` (the drop-in CLI, clstmocr.cc:56-111) -- text files next to the images,
  * the C ABI's network (default arithmetic and clstm_net_set_strict_f32) on the normalised frames,
  * the oracle (oracle/clstm_oracle.c: forward of clstm.cc:600-621 + trivial_decode of ctc.cc:159-190) on the same frames
    and the same parameters.
Required: 0 mismatching decodes (class sequences bit-identical, BASELINE.json north_star), and a model that is really in the
trained regime (character error rate against the ground truth well below chance: peaked posteriors, saturated gates -- the
regime the noise inputs of the other bench-shape tests do not reach)."""
import glob
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "clstm_amd", "bin")
sys.path.insert(0, os.path.join(ROOT, "scripts"))

N_LINES = int(os.environ.get("CORPUS_LINES", "512"))
UPDATES = int(os.environ.get("CORPUS_UPDATES", "1000"))
LRATE = os.environ.get("CORPUS_LRATE", "1e-4")


def levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def normalized_frames(png, tmp):
    subprocess.run([os.path.join(BIN, "clstm_hosttool"), "normalize", png, tmp, "48"], check=True, capture_output=True)
    data = open(tmp, "rb").read()
    w, h = struct.unpack("<ii", data[:8])
    return np.frombuffer(data[8:], np.float32).reshape(w, h).copy()


@pytest.fixture(scope="module")
def trained_corpus(tmp_path_factory):
    """(directory, image names, ground truth, model file) after `clstmocrtrain batch=64` on the rendered corpus"""
    from make_corpus import make_corpus
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "clstm_amd", "host"), "-s", "all"])
    d = str(tmp_path_factory.mktemp("corpus"))
    names, texts = make_corpus(d, n=N_LINES, seed=0)
    env = dict(os.environ, batch="64", ntrain=str(64 * UPDATES), lrate=LRATE, nhidden="100", seed="0.222",
               save_name=os.path.join(d, "_corpus"), save_every=str(64 * UPDATES), report_every=str(64 * max(1, UPDATES // 10)),
               test_every="100000000")
    r = subprocess.run([os.path.join(BIN, "clstmocrtrain"), os.path.join(d, "list.txt")], env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    sys.stderr.write("\n".join(l for l in r.stdout.split("\n") if l.startswith(("TRU", "OUT")))[-1200:] + "\n")
    models = sorted(glob.glob(os.path.join(d, "_corpus-*.clstm")))
    assert models, r.stdout[-2000:]
    return d, names, texts, models[-1]


@pytest.mark.gpu
def test_corpus_decodes_identical_to_oracle(trained_corpus, ora32, tmp_path):
    from clstm_amd.net import Network
    from oracle.oracle import OracleNet
    d, names, texts, model = trained_corpus
    # the saved model: flat parameters (clstm.cc:894-905 order) and the codec, through the drop-in's own proto reader
    pfile = str(tmp_path / "params.f32")
    subprocess.run([os.path.join(BIN, "clstm_hosttool"), "params", model, pfile], check=True)
    params = np.fromfile(pfile, np.float32)
    chars = sorted(set("".join(texts)))
    nc = len(chars) + 1                                  # Codec::build: class 0 + the sorted code points (clstmhl.h / extras.cc)
    ni, nh = 48, 100
    assert params.size == 2 * 4 * nh * (1 + ni + nh) + nc * (1 + 2 * nh)
    codec = [0] + [ord(c) for c in chars]

    # 1. the drop-in CLI
    r = subprocess.run([os.path.join(BIN, "clstmocr"), os.path.join(d, "list.txt")], env=dict(os.environ, load=model),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    cli_text = [open(n[:-len(".png")] + ".txt", encoding="utf-8").read().rstrip("\n") for n in names]

    # 2./3. the C ABI network (64 lines per launch) and the oracle, same frames, same parameters
    frames = [normalized_frames(n, str(tmp_path / "n.raw")) for n in names]
    ref = OracleNet(ora32, ni, nh, nc, init=False)
    ref.set_params(params)
    want = []
    for x in frames:
        ref.set_inputs(x)
        ref.forward()
        want.append(ref.decode().tolist())
    got = {}
    for mode in ("default", "strict_f32"):
        net = Network(ni, nh, nc)
        if mode == "strict_f32":
            net.set_strict_f32(True)
        net.set_params(params)
        out = []
        for i in range(0, len(frames), 64):
            net.set_inputs(frames[i:i + 64])
            net.forward()
            out += [c.tolist() for c in net.decode()]
        got[mode] = out

    mism = {m: [i for i in range(len(names)) if got[m][i] != want[i]] for m in got}
    cli_mism = [i for i in range(len(names)) if cli_text[i] != "".join(chr(codec[c]) for c in want[i])]
    errs = sum(levenshtein("".join(chr(codec[c]) for c in want[i]), texts[i]) for i in range(len(names)))
    cer = errs / float(sum(len(t) for t in texts))
    nonempty = sum(1 for w in want if w)
    sys.stderr.write("corpus: %d lines (%d frames, T %d..%d), %d updates x 64 lines; mismatching decodes vs oracle: default %d, "
                     "strict_f32 %d, clstmocr CLI %d; %d non-empty decodes; character error rate of the trained model %.4f\n"
                     % (len(names), sum(len(x) for x in frames), min(len(x) for x in frames), max(len(x) for x in frames), UPDATES,
                        len(mism["default"]), len(mism["strict_f32"]), len(cli_mism), nonempty, cer))
    assert mism["default"] == [] and mism["strict_f32"] == [], mism
    assert got["default"] == got["strict_f32"]
    assert cli_mism == [], [(cli_text[i], want[i]) for i in cli_mism[:3]]
    assert nonempty >= 0.9 * len(names) and cer < 0.1, (nonempty, cer)     # trained regime, not the near-uniform start
