"""ctypes binding for the CPU oracle (oracle/clstm_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (clstm_amd/) must never import it.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_stamp():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                import hashlib
                return hashlib.md5(line.encode()).hexdigest()
    except OSError:
        pass
    return "unknown"


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile).  Built with -march=native, so a library
    that travelled from another host (the build container -> the GPU box) is rebuilt."""
    stamp_file = os.path.join(_HERE, "build", ".cpu_stamp")
    stamp = _cpu_stamp()
    if not (os.path.exists(stamp_file) and open(stamp_file).read() == stamp):
        force = True
    need = force or not all(
        os.path.exists(os.path.join(_HERE, "build", f"liboracle_{s}.so")) for s in ("f32", "f64"))
    if not need:
        src = os.path.getmtime(os.path.join(_HERE, "clstm_oracle.c"))
        need = any(os.path.getmtime(os.path.join(_HERE, "build", f"liboracle_{s}.so")) < src
                   for s in ("f32", "f64"))
    if need:
        if force:
            subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
        with open(stamp_file, "w") as f:
            f.write(stamp)


class Oracle:
    """One precision of the oracle: Oracle('f32') or Oracle('f64')."""

    def __init__(self, prec="f32"):
        build()
        self.prec = prec
        self.dtype = np.float32 if prec == "f32" else np.float64
        self.cF = C.c_float if prec == "f32" else C.c_double
        self.lib = C.CDLL(os.path.join(_HERE, "build", f"liboracle_{prec}.so"))
        L = self.lib
        assert L.ora_sizeof_float() == np.dtype(self.dtype).itemsize
        P = C.c_void_p
        I = C.c_int
        F = self.cF
        sig = {
            "ora_seed": (None, [C.c_double]),
            "ora_get_seed": (C.c_double, []),
            "ora_randu": (C.c_double, []),
            "ora_limexp": (F, [F]),
            "ora_log_add": (F, [F, F]),
            "ora_rinit": (None, [P, I, I, F, C.c_char_p, F]),
            "ora_forward_nonlin0": (None, [P, I, I]),
            "ora_backward_nonlin0": (None, [P, P, I, I]),
            "ora_forward_nonlin": (None, [P, P, I, I]),
            "ora_backward_nonlin": (None, [P, P, P, I, I]),
            "ora_forward_lin1": (None, [P, P, P, I, I, I]),
            "ora_backward_lin1": (None, [P, P, P, P, P, I, I, I]),
            "ora_forward_full1": (None, [P, P, P, I, I, I, I]),
            "ora_backward_full1": (None, [P, P, P, P, P, P, I, I, I, I]),
            "ora_forward_softmax": (None, [P, P, P, I, I, I]),
            "ora_backward_softmax": (None, [P, P, P, P, P, I, I, I]),
            "ora_forward_stack": (None, [P, P, P, I, I, I]),
            "ora_backward_stack": (None, [P, P, P, I, I, I]),
            "ora_forward_stack_delay": (None, [P, P, P, I, I, I]),
            "ora_backward_stack_delay": (None, [P, P, P, I, I, I]),
            "ora_forward_statemem": (None, [P, P, P, P, P, I]),
            "ora_backward_statemem": (None, [P] * 9 + [I]),
            "ora_forward_nonlingate": (None, [P, P, P, I, I]),
            "ora_backward_nonlingate": (None, [P, P, P, P, P, I, I]),
            "ora_clip_gradient": (None, [P, I, F]),
            "ora_sgd_update": (None, [P, P, I, F, F]),
            "ora_argmax": (I, [P, I]),
            "ora_ctc_align_targets": (None, [P, P, P, I, I, I]),
            "ora_ctc_align_classes": (None, [P, P, P, I, I, I]),
            "ora_mktargets_classes": (I, [P, P, I]),
            "ora_trivial_decode": (I, [P, P, P, I, I]),
            "ora_set_nan_asserts": (None, [I]),
            "ora_net_nparams_for": (I, [I, I, I, P, I]),
            "ora_net_create": (P, [I, I, I, P, I, I]),
            "ora_net_free": (None, [P]),
            "ora_net_nparams": (I, [P]),
            "ora_net_get_params": (None, [P, P]),
            "ora_net_set_params": (None, [P, P]),
            "ora_net_get_derivs": (None, [P, P]),
            "ora_net_set_derivs": (None, [P, P]),
            "ora_net_set_lr": (None, [P, F, F]),
            "ora_net_set_inputs": (None, [P, P, I, I]),
            "ora_net_forward": (None, [P]),
            "ora_net_backward": (None, [P]),
            "ora_net_get_outputs": (None, [P, P]),
            "ora_net_set_output_deltas": (None, [P, P]),
            "ora_net_get_output_deltas": (None, [P, P]),
            "ora_net_get_input_deltas": (None, [P, P]),
            "ora_net_set_targets": (None, [P, P]),
            "ora_net_ctc_deltas": (None, [P, P, I, P]),
            "ora_net_decode": (I, [P, P, P, I]),
            "ora_net_update": (None, [P]),
            "ora_net_clear_derivs": (None, [P]),
            "ora_net_get_state": (I, [P, I, I, I, I, P]),
            "ora_net_train_line": (I, [P, P, I, P, I, P, I]),
            "ora_bench_lines": (C.c_double, [P, P, P, P, P, I, I, I]),
            "ora_bench_lines_pinned": (C.c_double, [P, P, P, P, P, I, I, I, P]),
            "ora_minibatch_lines": (None, [P, P, P, P, P, I, I, P, P, P, P, P, I, P]),
            "ora_forward_btswitch": (None, [P, P, I, I, I]),
            "ora_backward_btswitch": (None, [P, P, I, I, I]),
            "ora_forward_batchstack": (None, [P, P, I, I, I, I, I]),
            "ora_backward_batchstack": (None, [P, P, I, I, I, I, I]),
            "ora_full_forward": (None, [P, P, P, I, I, I, I, I]),
            "ora_full_backward": (None, [P, P, P, P, P, P, I, I, I, I, I]),
            "ora_softmax_seq_forward": (None, [P, P, P, I, I, I, I]),
            "ora_softmax_seq_backward": (None, [P, P, P, P, P, I, I, I, I]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args

    # -- helpers ---------------------------------------------------------
    def arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    @staticmethod
    def p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    @staticmethod
    def ints(a):
        return np.ascontiguousarray(a, dtype=np.int32)

    # -- CTC ---------------------------------------------------------------
    def ctc_align_targets(self, outputs, targets):
        """outputs (T x nc), targets (S x nc) -> posteriors (T x nc). ctc.cc:57-112"""
        o, t = self.arr(outputs), self.arr(targets)
        post = np.zeros_like(o)
        self.lib.ora_ctc_align_targets(self.p(post), self.p(o), self.p(t), o.shape[0], t.shape[0], o.shape[1])
        return post

    def ctc_align_classes(self, outputs, classes):
        o, c = self.arr(outputs), self.ints(classes)
        post = np.zeros_like(o)
        self.lib.ora_ctc_align_classes(self.p(post), self.p(o), self.p(c), o.shape[0], len(c), o.shape[1])
        return post

    def mktargets(self, transcript):
        tr = self.ints(transcript)
        st = np.zeros(2 * len(tr) + 1, np.int32)
        self.lib.ora_mktargets_classes(self.p(st), self.p(tr), len(tr))
        return st

    def trivial_decode(self, outputs):
        o = self.arr(outputs)
        cs = np.zeros(o.shape[0] + 1, np.int32)
        locs = np.zeros(o.shape[0] + 1, np.int32)
        n = self.lib.ora_trivial_decode(self.p(cs), self.p(locs), self.p(o), o.shape[0], o.shape[1])
        return cs[:n].copy(), locs[:n].copy()


class OracleNet:
    """Stacked{BiLSTM x nlayers, Softmax} ('bidi'/'bidi2') or 'lstm1' on the oracle."""

    def __init__(self, ora, ninput, nhidden, nclasses, unidirectional=False, init=True, seed=None):
        self.o = ora
        self.nhidden = list(nhidden) if isinstance(nhidden, (list, tuple)) else [nhidden]
        self.ninput, self.nclasses, self.uni = ninput, nclasses, unidirectional
        if seed is not None:
            ora.lib.ora_seed(seed)
        nh = Oracle.ints(self.nhidden)
        self.h = ora.lib.ora_net_create(len(self.nhidden), int(unidirectional), ninput, Oracle.p(nh),
                                        nclasses, int(init))
        self.nparams = ora.lib.ora_net_nparams(self.h)
        self.T = self.bs = 0

    def __del__(self):
        try:
            self.o.lib.ora_net_free(self.h)
        except Exception:
            pass

    def get_params(self):
        a = np.zeros(self.nparams, self.o.dtype)
        self.o.lib.ora_net_get_params(self.h, Oracle.p(a))
        return a

    def set_params(self, a):
        a = self.o.arr(a)
        assert a.size == self.nparams
        self.o.lib.ora_net_set_params(self.h, Oracle.p(a))

    def get_derivs(self):
        a = np.zeros(self.nparams, self.o.dtype)
        self.o.lib.ora_net_get_derivs(self.h, Oracle.p(a))
        return a

    def set_derivs(self, a):
        a = self.o.arr(a)
        self.o.lib.ora_net_set_derivs(self.h, Oracle.p(a))

    def clear_derivs(self):
        self.o.lib.ora_net_clear_derivs(self.h)

    def set_lr(self, lr, mom):
        self.o.lib.ora_net_set_lr(self.h, lr, mom)

    def set_inputs(self, x):
        """x: [T][bs][ni] or [T][ni]"""
        x = self.o.arr(x)
        if x.ndim == 2:
            x = x[:, None, :]
        self.T, self.bs = x.shape[0], x.shape[1]
        assert x.shape[2] == self.ninput
        self.o.lib.ora_net_set_inputs(self.h, Oracle.p(np.ascontiguousarray(x)), self.T, self.bs)

    def forward(self):
        self.o.lib.ora_net_forward(self.h)
        out = np.zeros((self.T, self.bs, self.nclasses), self.o.dtype)
        self.o.lib.ora_net_get_outputs(self.h, Oracle.p(out))
        return out

    def set_output_deltas(self, d):
        d = self.o.arr(d).reshape(self.T, self.bs, self.nclasses)
        self.o.lib.ora_net_set_output_deltas(self.h, Oracle.p(d))

    def get_output_deltas(self):
        d = np.zeros((self.T, self.bs, self.nclasses), self.o.dtype)
        self.o.lib.ora_net_get_output_deltas(self.h, Oracle.p(d))
        return d

    def set_targets(self, t):
        t = self.o.arr(t).reshape(self.T, self.bs, self.nclasses)
        self.o.lib.ora_net_set_targets(self.h, Oracle.p(t))

    def ctc_deltas(self, transcript):
        tr = Oracle.ints(transcript)
        al = np.zeros((self.T, self.nclasses), self.o.dtype)
        self.o.lib.ora_net_ctc_deltas(self.h, Oracle.p(tr), len(tr), Oracle.p(al))
        return al

    def backward(self):
        self.o.lib.ora_net_backward(self.h)

    def input_deltas(self):
        d = np.zeros((self.T, self.bs, self.ninput), self.o.dtype)
        self.o.lib.ora_net_get_input_deltas(self.h, Oracle.p(d))
        return d

    def update(self):
        self.o.lib.ora_net_update(self.h)

    def decode(self, batch=0):
        cs = np.zeros(self.T + 1, np.int32)
        locs = np.zeros(self.T + 1, np.int32)
        n = self.o.lib.ora_net_decode(self.h, Oracle.p(cs), Oracle.p(locs), batch)
        return cs[:n].copy()

    STATES = {"gi": 0, "gf": 1, "go": 2, "ci": 3, "state": 4, "outputs": 5, "source": 6}

    def state(self, layer, direction, which, plane=0):
        w = self.STATES[which]
        n = self.o.lib.ora_net_get_state(self.h, layer, direction, w, plane, None)
        a = np.zeros(n, self.o.dtype)
        self.o.lib.ora_net_get_state(self.h, layer, direction, w, plane, Oracle.p(a))
        return a.reshape(self.T, self.bs, -1)

    def train_line(self, x, transcript, update=True):
        x = self.o.arr(x)
        tr = Oracle.ints(transcript)
        self.T, self.bs = x.shape[0], 1
        cs = np.zeros(x.shape[0] + 1, np.int32)
        n = self.o.lib.ora_net_train_line(self.h, Oracle.p(x), x.shape[0], Oracle.p(tr), len(tr),
                                          Oracle.p(cs), int(update))
        return cs[:n].copy()

    def minibatch(self, lines, transcripts, nthreads=None, keep=()):
        """Reference semantics of a minibatch (every line an independent fwd/CTC/bwd accumulating into Params.d,
        clstmhl.h:201-217) with OpenMP over lines -- for sizes where the sequential Python loop of
        tests/common.py::oracle_minibatch takes minutes.  Returns a dict: outputs / aligned (lists of [T][nc]),
        decode (list of int arrays), derivs (net.derivs after accumulating), kept {line: OracleNet view} whose
        .state() reads that line's internals."""
        Ts = [len(l) for l in lines]
        offs = np.concatenate([[0], np.cumsum(Ts)]).astype(np.int32)
        x = self.o.arr(np.concatenate(lines, 0))
        lab = Oracle.ints(np.concatenate(transcripts))
        loffs = np.concatenate([[0], np.cumsum([len(t) for t in transcripts])]).astype(np.int32)
        N, nc = int(offs[-1]), self.nclasses
        outputs = np.zeros((N, nc), self.o.dtype)
        aligned = np.zeros((N, nc), self.o.dtype)
        dec = np.zeros(N + 1, np.int32)
        decn = np.zeros(len(lines), np.int32)
        keep = Oracle.ints(list(keep))
        kept = (C.c_void_p * max(1, len(keep)))()
        if nthreads is None:
            nthreads = min(len(lines), os.cpu_count() or 1)
        self.o.lib.ora_minibatch_lines(self.h, Oracle.p(x), Oracle.p(offs), Oracle.p(lab), Oracle.p(loffs), len(lines),
                                       nthreads, Oracle.p(outputs), Oracle.p(aligned), Oracle.p(dec), Oracle.p(decn),
                                       Oracle.p(keep), len(keep), kept)
        views = {}
        for j, b in enumerate(keep.tolist()):
            v = OracleNet.__new__(OracleNet)
            v.o, v.nhidden, v.ninput, v.nclasses, v.uni = self.o, self.nhidden, self.ninput, self.nclasses, self.uni
            v.h, v.nparams, v.T, v.bs = kept[j], self.nparams, Ts[b], 1
            views[b] = v
        sp = lambda a: [a[offs[b]:offs[b + 1]] for b in range(len(lines))]
        return {"outputs": sp(outputs), "aligned": sp(aligned),
                "decode": [dec[offs[b]:offs[b] + decn[b]].copy() for b in range(len(lines))],
                "derivs": self.get_derivs(), "kept": views}

    def bench_lines(self, x, offs, labels, loffs, nthreads=1, reps=1, cpus=None):
        """seconds for `reps` passes over the lines (fwd + CTC + bwd each) on `nthreads` threads, every thread on its own
        preallocated net, one untimed line first; `cpus`: the logical CPU each thread pins itself to"""
        x = self.o.arr(x)
        offs, labels, loffs = Oracle.ints(offs), Oracle.ints(labels), Oracle.ints(loffs)
        cp = None
        if cpus is not None:
            assert len(cpus) >= nthreads
            cp = Oracle.ints(list(cpus))
        return self.o.lib.ora_bench_lines_pinned(self.h, Oracle.p(x), Oracle.p(offs), Oracle.p(labels), Oracle.p(loffs),
                                                 len(offs) - 1, nthreads, reps, Oracle.p(cp) if cp is not None else None)
