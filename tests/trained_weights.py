"""The "trained-like" weight set of SURVEY.md section 8(d): the uw3 architecture (48 -> BiLSTM(100) -> 83 classes) after 500
steps of the ORACLE's online SGD on the reference's OCR fixture (misc/textline.bin.png, "performance analysis"), so that gates
are not all ~0.5 and the posteriors are peaked -- plus real-line inputs for it: crops of the normalised fixture with jitter.

Test infrastructure (it drives oracle/): only tests/ imports this.  Nothing generated here is committed -- the generator is
(VERDICT r5 item 1a); 500 steps take the oracle ~10 s, cached per process."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "textline.bin.png")
GT = open(os.path.join(ROOT, "tests", "golden", "textline.gt.txt"), encoding="utf-8").read().rstrip("\n")
NI, NH, NC = 48, 100, 83
_cache = {}


def fixture_frames():
    """[T][48] normalised fixture line (clstmocrtrain.cc:73 inversion + CenterNormalizer, through the drop-in's host tool)"""
    if "x" not in _cache:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "clstm_amd", "host"), "-s", "all"])
        out = "/tmp/_fixture_%d.raw" % os.getpid()
        subprocess.run([os.path.join(ROOT, "clstm_amd", "bin", "clstm_hosttool"), "normalize", FIXTURE, out, "48"], check=True,
                       capture_output=True)
        data = open(out, "rb").read()
        os.unlink(out)
        w, h = struct.unpack("<ii", data[:8])
        _cache["x"] = np.frombuffer(data[8:], np.float32).reshape(w, h).copy()
    return _cache["x"]


def fixture_transcript():
    chars = sorted(set(GT))                       # the fixture's own codec, classes 1..14 of the 83 (the rest stay unused)
    return np.array([1 + chars.index(c) for c in GT], np.int32)


def trained_like_params(ora32, steps=500, lr=1e-2, mom=0.9, seed=0.222):
    """parameters after `steps` online-SGD steps (CLSTMOCR::train, clstmhl.h:201-223) of the oracle on the fixture line"""
    key = (steps, lr, mom, seed)
    if key not in _cache:
        from clstm_amd.init import init_params
        from oracle.oracle import OracleNet
        x, tr = fixture_frames(), fixture_transcript()
        ref = OracleNet(ora32, NI, NH, NC, init=False)
        ref.set_params(init_params(NI, NH, NC, seed=seed))
        ref.set_lr(lr, mom)
        dec = None
        for _ in range(steps):
            dec = ref.train_line(x, tr)
        _cache[key] = (ref.get_params().copy(), np.asarray(dec).tolist() == tr.tolist())
    return _cache[key]


def fixture_crops(rng, T_list, ora_net=None):
    """real-line inputs: windows of the normalised fixture (cyclically extended when a window runs past its end), each with its
    own jitter -- a sub-frame shift along t (linear interpolation), a vertical shift of up to 2 px, gain and a little noise.
    With `ora_net` (an OracleNet holding the trained parameters) the transcript of a crop is what the oracle decodes on it
    (at least one label), i.e. a target the alignment is confident about; without, random labels."""
    x = fixture_frames()
    Tx = len(x)
    lines, trs = [], []
    for T in T_list:
        s = float(rng.uniform(0, Tx))
        pos = (s + np.arange(T)) % (Tx - 1)
        i0 = np.floor(pos).astype(int)
        f = (pos - i0)[:, None].astype(np.float32)
        w = (1 - f) * x[i0] + f * x[i0 + 1]
        w = np.roll(w, int(rng.integers(-2, 3)), axis=1)
        w = np.clip(w * float(rng.uniform(0.85, 1.15)) + rng.normal(0, 0.02, w.shape), 0, 1).astype(np.float32)
        lines.append(w)
        if ora_net is not None:
            ora_net.set_inputs(w)
            ora_net.forward()
            d = np.asarray(ora_net.decode(), np.int32)
            trs.append(d if len(d) and 2 * len(d) + 1 <= T else np.array([1], np.int32))
        else:
            trs.append(rng.integers(1, 15, max(1, T // 8)).astype(np.int32))
    return lines, trs
