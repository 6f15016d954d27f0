"""GPU-free host pieces of the drivers (clstm_amd/host): PNG reading without libpng, the
CenterNormalizer restatement and the clstm.proto model-file codec, checked against independent
implementations (PIL, a numpy restatement of extras.cc:227-285, python-protobuf with a descriptor
built from clstm.proto:1-26)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "clstm_amd", "bin", "clstm_hosttool")
FIXTURE = os.path.join(ROOT, "tests", "golden", "textline.bin.png")


@pytest.fixture(scope="module", autouse=True)
def build_tools():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "clstm_amd", "csrc"), "-s", "all"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "clstm_amd", "host"), "-s", "all"])


def run(*args):
    return subprocess.run([TOOL] + [str(a) for a in args], check=True, capture_output=True, text=True).stdout


def read_raw(path):
    data = open(path, "rb").read()
    w, h = struct.unpack("<ii", data[:8])
    return np.frombuffer(data[8:], np.float32).reshape(w, h)      # image(x, y)


def test_png_fixture_matches_pil(tmp_path):
    from PIL import Image
    out = tmp_path / "a.raw"
    run("png2raw", FIXTURE, out)
    got = read_raw(out)
    im = np.asarray(Image.open(FIXTURE).convert("RGBA")).astype(np.float64)
    want = (im[..., 0] + im[..., 1] + im[..., 2]) / (3 * 255.0)   # extras.cc:540-541
    assert got.shape == (im.shape[1], im.shape[0])
    assert np.array_equal(got, want.T.astype(np.float32))


@pytest.mark.parametrize("mode", ["L", "1", "P", "RGB", "I;16", "LA"])
def test_png_modes(tmp_path, mode):
    from PIL import Image
    rng = np.random.default_rng(3)
    w, h = 37, 11
    if mode == "RGB":
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8); im = Image.fromarray(arr, "RGB")
        want = arr.astype(np.float64).sum(-1) / (3 * 255.0)
    elif mode == "L":
        arr = rng.integers(0, 256, (h, w), dtype=np.uint8); im = Image.fromarray(arr, "L")
        want = arr.astype(np.float64)                     # grey stays 0..255: reference quirk (extras.cc:537-538)
    elif mode == "LA":
        arr = rng.integers(0, 256, (h, w, 2), dtype=np.uint8); im = Image.fromarray(arr, "LA")
        want = arr[..., 0].astype(np.float64)             # STRIP_ALPHA
    elif mode == "1":
        arr = rng.integers(0, 2, (h, w), dtype=np.uint8); im = Image.fromarray(arr * 255, "L").convert("1")
        want = arr.astype(np.float64) * 255
    elif mode == "I;16":
        arr = rng.integers(0, 65536, (h, w), dtype=np.uint16); im = Image.fromarray(arr, "I;16")
        want = (arr >> 8).astype(np.float64)              # STRIP_16
    else:
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        im = Image.fromarray(arr, "RGB").quantize(16)
        rgb = np.asarray(im.convert("RGB")).astype(np.float64)
        want = rgb.sum(-1) / (3 * 255.0)                  # EXPAND: palette -> RGB
    path = tmp_path / "x.png"
    im.save(path)
    out = tmp_path / "x.raw"
    run("png2raw", path, out)
    assert np.array_equal(read_raw(out), want.T.astype(np.float32))


def test_write_png_roundtrip(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(5)
    img = rng.random((23, 9)).astype(np.float32)          # image(x, y)
    raw = tmp_path / "p.raw"
    open(raw, "wb").write(struct.pack("<ii", 23, 9) + img.tobytes())
    run("writepng", raw, tmp_path / "p.png")
    got = np.asarray(Image.open(tmp_path / "p.png").convert("RGB"))
    want = np.floor(np.clip(img.astype(np.float64) * 256, 0, 255.999999)).astype(np.uint8).T
    assert got.shape == (9, 23, 3) and np.array_equal(got[..., 0], want) and np.array_equal(got[..., 2], want)


# ---- numpy restatement of the CenterNormalizer (extras.cc:58-131, 205-285) ------------------------
def np_gauss1d(v, sigma):
    sigma = np.float32(sigma)
    rng_ = 1 + int(3.0 * sigma)
    i = np.arange(rng_ + 1)
    half = np.exp(-i * i / 2.0 / float(sigma) / float(sigma)).astype(np.float32)
    mask = np.concatenate([half[:0:-1], half]).astype(np.float32)
    total = np.float32(0)
    for m in mask:
        total = np.float32(total + m)
    mask = (mask / total).astype(np.float32)
    n = len(v)
    idx = np.clip(np.arange(n)[:, None] + np.arange(len(mask))[None, :] - rng_, 0, n - 1)
    return (v[idx].astype(np.float64) * mask[None, :].astype(np.float64)).cumsum(1)[:, -1].astype(np.float32)


def np_normalize(line, target_height=48):
    w, h = line.shape
    smooth = line.copy()
    for i in range(w):
        smooth[i, :] = np_gauss1d(smooth[i, :], h * 0.5)
    for j in range(h):
        smooth[:, j] = np_gauss1d(smooth[:, j], np.float32(h * np.float32(1.0)))
    for j in range(h):
        v = 0.0
        for i in range(w):
            v = v * 0.9 + float(line[i, j])
            smooth[i, j] = np.float32(smooth[i, j] + np.float32(min(1.0, v) * 1e-3))
    a = np.zeros(w, np.float32)
    for i in range(w):
        a[i] = h - 1 - np.argmax(smooth[i, ::-1])          # ties -> last index
    center = np_gauss1d(a, np.float32(h * np.float32(0.3)))
    s1 = np.float32(0); sy = np.float32(0)
    for i in range(w):
        for j in range(h):
            s1 = np.float32(s1 + line[i, j])
            sy = np.float32(sy + np.float32(line[i, j] * np.float32(abs(np.float32(j) - center[i]))))
    mad = np.float32(sy / s1)
    r = np.float32(int(np.float32(4.0) * mad + 1))
    scale = np.float32((2.0 * float(r)) / target_height)
    tw = max(int(np.float32(w) / scale), 1)
    out = np.zeros((tw, target_height), np.float32)
    for i in range(tw):
        x = np.float32(scale * np.float32(i))
        for j in range(target_height):
            y = np.float32(np.float32(scale * np.float32(j - target_height // 2)) + center[int(x)])
            i0, j0 = int(np.floor(x)), int(np.floor(y))
            l, m = np.float32(x - i0), np.float32(y - j0)
            c = lambda ii, jj: float(line[min(max(ii, 0), w - 1), min(max(jj, 0), h - 1)])
            out[i, j] = np.float32((1.0 - float(l)) * ((1.0 - float(m)) * c(i0, j0) + float(m) * c(i0, j0 + 1)) +
                                   float(l) * ((1.0 - float(m)) * c(i0 + 1, j0) + float(m) * c(i0 + 1, j0 + 1)))
    return out, float(r)


def test_center_normalizer_on_fixture(tmp_path):
    out = tmp_path / "n.raw"
    msg = run("normalize", FIXTURE, out, 48)
    got = read_raw(out)
    raw = tmp_path / "r.raw"
    run("png2raw", FIXTURE, raw)
    line = (1.0 - read_raw(raw)).astype(np.float32)       # ink = 1 (clstmocrtrain.cc:73)
    want, r = np_normalize(line)
    assert got.shape == want.shape and got.shape[1] == 48
    assert ("r %g" % r) in msg
    assert np.abs(got - want).max() < 1e-5
    assert 300 < got.shape[0] < 450                        # T ~ 370 for this line


# ---- model files vs python-protobuf -----------------------------------------------------------------
def clstm_pb2():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto(name="clstm.proto", package="clstm", syntax="proto2")
    F = descriptor_pb2.FieldDescriptorProto

    def field(m, name, num, typ, label, tn=None):
        fd = m.field.add(name=name, number=num, type=typ, label=label)
        if tn:
            fd.type_name = tn
    kv = f.message_type.add(name="KeyValue")
    field(kv, "key", 1, F.TYPE_STRING, F.LABEL_REQUIRED); field(kv, "value", 2, F.TYPE_STRING, F.LABEL_REQUIRED)
    ar = f.message_type.add(name="Array")
    field(ar, "name", 1, F.TYPE_STRING, F.LABEL_OPTIONAL); field(ar, "dim", 2, F.TYPE_INT32, F.LABEL_REPEATED)
    field(ar, "value", 3, F.TYPE_FLOAT, F.LABEL_REPEATED)
    n = f.message_type.add(name="NetworkProto")
    field(n, "kind", 1, F.TYPE_STRING, F.LABEL_REQUIRED); field(n, "name", 2, F.TYPE_STRING, F.LABEL_OPTIONAL)
    field(n, "ninput", 10, F.TYPE_INT32, F.LABEL_REQUIRED); field(n, "noutput", 11, F.TYPE_INT32, F.LABEL_REQUIRED)
    field(n, "icodec", 12, F.TYPE_INT32, F.LABEL_REPEATED); field(n, "codec", 13, F.TYPE_INT32, F.LABEL_REPEATED)
    field(n, "attribute", 20, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".clstm.KeyValue")
    field(n, "weights", 30, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".clstm.Array")
    field(n, "sub", 40, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".clstm.NetworkProto")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("clstm.NetworkProto"))


def test_model_file_is_protobuf_compatible(tmp_path):
    from clstm_amd.init import init_params
    NetworkProto = clstm_pb2()
    path = tmp_path / "m.clstm"
    run("init-model", "bidi", 48, 100, 0, 83, 0.222, path)
    data = open(path, "rb").read()
    msg = NetworkProto()
    msg.ParseFromString(data)
    assert msg.SerializeToString() == data                  # byte-for-byte what libprotobuf emits
    assert msg.kind == "Stacked" and msg.ninput == 48 and msg.noutput == 83 and len(msg.codec) == 83
    assert {a.key: a.value for a in msg.attribute}["kind"] == "bidi"
    par, sm = msg.sub
    assert par.kind == "Parallel" and [s.kind for s in par.sub] == ["NPLSTM", "Reversed"]
    assert [w.name for w in par.sub[0].weights] == ["WCI", "WGF", "WGI", "WGO"]
    assert sm.kind == "SoftmaxLayer" and list(sm.weights[0].dim) == [83, 201]
    # weights: row-major in the file (clstm_proto.cc:43-44), column-major in the flat vector
    flat = init_params(48, 100, 83, seed=0.222)
    blk = 100 * 149
    wgi = np.array(par.sub[0].weights[2].value, np.float32).reshape(100, 149)
    assert np.array_equal(wgi, flat[2 * blk:3 * blk].reshape(149, 100).T)
    rev_wci = np.array(par.sub[1].sub[0].weights[0].value, np.float32).reshape(100, 149)
    assert np.array_equal(rev_wci, flat[4 * blk:5 * blk].reshape(149, 100).T)
    w1 = np.array(sm.weights[0].value, np.float32).reshape(83, 201)
    assert np.array_equal(w1, flat[8 * blk:].reshape(201, 83).T)
    # the tool's own reader gives the same flat vector back
    run("params", path, tmp_path / "p.bin")
    assert np.array_equal(np.fromfile(tmp_path / "p.bin", np.float32), flat)


def test_reads_files_written_by_protobuf(tmp_path):
    """A model written by libprotobuf-compatible code (python-protobuf) -- including a `name` field the
    reference never sets -- loads and re-saves to an equivalent message."""
    NetworkProto = clstm_pb2()
    rng = np.random.default_rng(0)
    msg = NetworkProto(kind="Stacked", ninput=5, noutput=4, name="x")
    msg.codec.extend([0, 97, 98, 99])
    msg.attribute.add(key="kind", value="bidi2"); msg.attribute.add(key="trial", value="77")

    def lstm(parent, ni, no):
        n = parent.add(kind="NPLSTM", ninput=ni, noutput=no)
        for nm in ("WCI", "WGF", "WGI", "WGO"):
            w = n.weights.add(name=nm)
            w.dim.extend([no, ni + no + 1]); w.value.extend(rng.normal(size=no * (ni + no + 1)).astype(np.float32).tolist())
    ni = 5
    for no in (3, 2):
        par = msg.sub.add(kind="Parallel", ninput=ni, noutput=2 * no)
        lstm(par.sub, ni, no)
        rev = par.sub.add(kind="Reversed", ninput=ni, noutput=no)
        lstm(rev.sub, ni, no)
        ni = 2 * no
    sm = msg.sub.add(kind="SoftmaxLayer", ninput=ni, noutput=4)
    w = sm.weights.add(name="W1"); w.dim.extend([4, ni + 1]); w.value.extend(rng.normal(size=4 * (ni + 1)).astype(np.float32).tolist())
    src = tmp_path / "in.clstm"
    open(src, "wb").write(msg.SerializeToString())
    out = run("roundtrip", src, tmp_path / "out.clstm")
    assert "bidi2 ninput 5 nhidden 3 nclasses 4" in out
    back = NetworkProto()
    back.ParseFromString(open(tmp_path / "out.clstm", "rb").read())
    assert [list(w.value) for w in back.sub[1].sub[1].sub[0].weights] == [list(w.value) for w in msg.sub[1].sub[1].sub[0].weights]
    assert list(back.sub[2].weights[0].value) == list(sm.weights[0].value)
    assert {a.key: a.value for a in back.attribute}["trial"] == "77" and list(back.codec) == [0, 97, 98, 99]


def test_unsupported_model_is_rejected(tmp_path):
    NetworkProto = clstm_pb2()
    msg = NetworkProto(kind="Stacked", ninput=3, noutput=2)
    msg.sub.add(kind="Btswitch", ninput=3, noutput=3)
    msg.sub.add(kind="SigmoidLayer", ninput=3, noutput=2)
    src = tmp_path / "bad.clstm"
    open(src, "wb").write(msg.SerializeToString())
    r = subprocess.run([TOOL, "roundtrip", str(src), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode != 0 and "FATAL" in r.stderr


def test_cpu_emulated_cli_smoke(tmp_path):
    """The drivers run against the C ABI; without a GPU they fail loudly (no CPU fallback)."""
    import shutil
    if shutil.which("rocminfo") and os.path.exists("/dev/kfd"):
        pytest.skip("GPU present: covered by tests/test_gpu_e2e.py")
    lst = tmp_path / "l.txt"
    lst.write_text(FIXTURE + "\n")
    r = subprocess.run([os.path.join(ROOT, "clstm_amd", "bin", "clstmocr"), str(lst)],
                       env=dict(os.environ, load=str(tmp_path / "missing.clstm")), capture_output=True, text=True)
    assert r.returncode != 0 and "FATAL" in r.stderr
