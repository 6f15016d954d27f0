"""Real-line inputs from the one OCR line the reference ships (misc/textline.bin.png, "performance analysis"; a copy is
tests/golden/textline.bin.png): the normalised frames through the drop-in's own host tool (read_png + inversion of
clstmocrtrain.cc:73 + CenterNormalizer, extras.cc:227-285), and jittered crops of them -- what `bench.py --weights trained`
and the trained-regime parity tests feed instead of smoothed noise.  No oracle here: this is host-side plumbing."""
import os
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "textline.bin.png")
GT = open(os.path.join(ROOT, "tests", "golden", "textline.gt.txt"), encoding="utf-8").read().rstrip("\n")
_frames = None


def fixture_frames():
    """[T][48] float32"""
    global _frames
    if _frames is None:
        tool = os.path.join(ROOT, "clstm_amd", "bin", "clstm_hosttool")
        if not os.path.exists(tool):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "clstm_amd", "host"), "-s", "all"])
        with tempfile.NamedTemporaryFile(suffix=".raw") as f:
            subprocess.run([tool, "normalize", FIXTURE, f.name, "48"], check=True, capture_output=True)
            data = open(f.name, "rb").read()
        w, h = struct.unpack("<ii", data[:8])
        _frames = np.frombuffer(data[8:], np.float32).reshape(w, h).copy()
    return _frames


def fixture_transcript():
    """classes of the ground truth under the fixture's own codec (sorted distinct characters -> 1..14; Codec::build, clstm.cc:246-267)"""
    chars = sorted(set(GT))
    return np.array([1 + chars.index(c) for c in GT], np.int32)


def jittered_crops(rng, T_list):
    """windows of the normalised fixture (cyclically extended when a window runs past its end), each with its own jitter: a
    sub-frame shift along t (linear interpolation), a vertical shift of up to 2 px, gain and a little noise"""
    x = fixture_frames()
    Tx = len(x)
    lines = []
    for T in T_list:
        s = float(rng.uniform(0, Tx))
        pos = (s + np.arange(T)) % (Tx - 1)
        i0 = np.floor(pos).astype(int)
        f = (pos - i0)[:, None].astype(np.float32)
        w = (1 - f) * x[i0] + f * x[i0 + 1]
        w = np.roll(w, int(rng.integers(-2, 3)), axis=1)
        lines.append(np.clip(w * float(rng.uniform(0.85, 1.15)) + rng.normal(0, 0.02, w.shape), 0, 1).astype(np.float32))
    return lines
