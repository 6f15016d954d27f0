"""Host-side mirror of the reference's network interface for the hot path.

Names follow the reference: `make_net("bidi"|"bidi2"|"lstm1", ...)` (clstm_prefab.cc:163-173),
`INetwork.forward()/backward()`, `sgd_update(net)` (clstm.cc:201-217), `set_inputs`,
`ctc_align_targets`, `trivial_decode` (clstm.h:310-320), `get_params/set_params/get_derivs`
(clstm.cc:872-917) and `CLSTMOCR` (clstmhl.h:146-272).  All arithmetic happens in
libclstm_hip.so; this file only moves pointers and small host arrays.
"""
import ctypes as C

import numpy as np

from . import abi
from .abi import NetDesc, f32, i32, ptr

STATE_CODES = {"gi": 0, "gf": 1, "go": 2, "ci": 3, "state": 4, "outputs": 5,
               "d_gi": 6, "d_gf": 7, "d_go": 8, "d_ci": 9}


class Network:
    """Stacked{Parallel{NPLSTM, Reversed{NPLSTM}} x L, SoftmaxLayer} on one MI355X.

    A minibatch is a list of text lines (line b has T_b frames of `ninput` features);
    frames are packed line after line ("frame-major", feature contiguous).
    """

    def __init__(self, ninput, nhidden, nclasses, unidirectional=False, lib=None,
                 params=None, derivs=None, grads=None):
        self.lib = lib or abi.load()
        self.nhidden = list(nhidden) if isinstance(nhidden, (list, tuple)) else [int(nhidden)]
        self.ninput, self.nclasses, self.unidirectional = int(ninput), int(nclasses), bool(unidirectional)
        d = NetDesc()
        d.nlayers = len(self.nhidden)
        d.unidirectional = int(self.unidirectional)
        d.ninput = self.ninput
        d.nclasses = self.nclasses
        for i, h in enumerate(self.nhidden):
            d.nhidden[i] = int(h)
        self.desc = d
        self.nparams = self.lib.call("clstm_net_nparams_for", C.byref(d))
        for buf in (params, derivs, grads):
            if buf is not None:
                assert buf.numel() == self.nparams and buf.is_contiguous()
        self._keep = (params, derivs, grads)    # caller-owned device buffers stay alive
        h = C.c_void_p()
        self.lib.call("clstm_net_create", C.byref(h), C.byref(d), ptr(params), ptr(derivs), ptr(grads))
        self.h = h
        self.T = []
        self.N = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.call("clstm_net_destroy", self.h)
                self.h = None
        except Exception:
            pass

    # -- parameters: get_params / set_params / get_derivs (clstm.cc:872-917) -------------
    def set_params(self, a):
        a = f32(a)
        assert a.size == self.nparams
        self.lib.call("clstm_net_set_params_h", self.h, ptr(a))

    def get_params(self):
        a = np.empty(self.nparams, np.float32)
        self.lib.call("clstm_net_get_params_h", self.h, ptr(a))
        return a

    def set_derivs(self, a):
        a = f32(a)
        assert a.size == self.nparams
        self.lib.call("clstm_net_set_derivs_h", self.h, ptr(a))

    def get_derivs(self):
        a = np.empty(self.nparams, np.float32)
        self.lib.call("clstm_net_get_derivs_h", self.h, ptr(a))
        return a

    def get_grads(self):
        a = np.empty(self.nparams, np.float32)
        self.lib.call("clstm_net_get_grads_h", self.h, ptr(a))
        return a

    def params_changed(self):
        self.lib.call("clstm_net_params_changed", self.h)

    def setLearningRate(self, lr, momentum):          # INetwork::setLearningRate clstm.cc:158
        self.lib.call("clstm_net_set_learning_rate", self.h, float(lr), float(momentum))

    def set_gemm_precision(self, mode):
        """0: exact f32 MFMA (default, the parity path); 1: bf16 inputs / f32 accumulation for the hoisted
        gate GEMMs (BASELINE config '2 x BiLSTM(512), bf16 MFMA')."""
        self.lib.call("clstm_net_set_gemm_precision", self.h, int(mode))

    def set_gradient_clip(self, clip):
        self.lib.call("clstm_net_set_gradient_clip", self.h, float(clip))

    # -- data ------------------------------------------------------------------------------
    def set_batch(self, T):
        self._declared = None
        self.T = [int(t) for t in T]
        self.N = int(sum(self.T))
        t = i32(self.T)
        self.lib.call("clstm_net_set_batch", self.h, ptr(t), len(self.T))

    def set_inputs(self, lines):
        """lines: list of [T_b, ninput] arrays (set_inputs, clstm.cc:684-690), or a packed [N, ninput]
        array after an explicit set_batch()."""
        if isinstance(lines, (list, tuple)):
            self.set_batch([len(x) for x in lines])
            x = f32(np.concatenate([f32(x).reshape(-1, self.ninput) for x in lines], 0))
        else:
            x = f32(lines)
        assert x.shape == (self.N, self.ninput)
        self.lib.call("clstm_net_set_inputs_h", self.h, ptr(x))

    def set_inputs_device(self, x_dev):
        self.lib.call("clstm_net_set_inputs_d", self.h, ptr(x_dev))

    def forward(self):
        if getattr(self, "_declared", None):      # a minibatch declared by the last train step becomes the current one
            self.T, self.N = self._declared
            self._declared = None
        self.lib.call("clstm_net_forward", self.h)

    def outputs(self):
        p = np.empty((self.N, self.nclasses), np.float32)
        self.lib.call("clstm_net_get_outputs_h", self.h, ptr(p))
        return p

    def split(self, packed):
        out, o = [], 0
        for t in self.T:
            out.append(packed[o:o + t])
            o += t
        return out

    def set_output_deltas(self, d):
        d = f32(d)
        assert d.shape == (self.N, self.nclasses)
        self.lib.call("clstm_net_set_output_deltas_h", self.h, ptr(d))

    def ctc(self, transcripts, want_aligned=False):
        """mktargets + ctc_align_targets + `outputs.d = aligned - outputs.v` for every line
        (clstmhl.h:207-212)."""
        assert len(transcripts) == len(self.T)
        L = i32([len(t) for t in transcripts])
        flat = [int(c) for t in transcripts for c in t]
        labels = i32(flat if flat else [0])
        al = np.empty((self.N, self.nclasses), np.float32) if want_aligned else None
        self.lib.call("clstm_net_ctc", self.h, ptr(labels), ptr(L), ptr(al))
        return al

    def backward(self):
        self.lib.call("clstm_net_backward", self.h)

    def enable_input_deltas(self, on=True):
        self.lib.call("clstm_net_enable_input_deltas", self.h, int(on))

    def input_deltas(self):
        d = np.empty((self.N, self.ninput), np.float32)
        self.lib.call("clstm_net_get_input_deltas_h", self.h, ptr(d))
        return d

    def update(self):                                   # sgd_update(Network) clstm.cc:201-217
        self.lib.call("clstm_net_update", self.h)

    def train_step(self, T, x_dev, transcripts):
        """CLSTMOCR::train (clstmhl.h:201-223) for a minibatch resident in HBM, ONE library call:
        set_batch + set_inputs + forward + CTC + backward + [all-reduce] + update, no host sync."""
        self.train_step_prepared(self.prepare_step(T, transcripts), x_dev)

    @staticmethod
    def prepare_step(T, transcripts):
        """Host arrays of one minibatch in the form the C ABI takes (line lengths, packed transcripts, their
        lengths) -- build once per minibatch, outside any timed loop."""
        Tl = [int(t) for t in T]
        assert len(transcripts) == len(Tl)
        L = i32([len(tr) for tr in transcripts])
        flat = np.concatenate([i32(tr).reshape(-1) for tr in transcripts]) if len(transcripts) else i32([])
        return Tl, i32(Tl), i32(flat if flat.size else [0]), L

    def train_step_prepared(self, prep, x_dev, next_prep=None, next_x_dev=None):
        """clstm_net_train_step; with the NEXT minibatch given (same forms), clstm_net_train_step_next: its ingest rides this
        step's last launch and the next call -- which must pass that very minibatch -- starts with its forward launch."""
        Tl, t, labels, L = prep
        self.T, self.N = Tl, int(sum(Tl))
        self._declared = None
        if next_prep is None:
            self.lib.call("clstm_net_train_step", self.h, ptr(t), len(Tl), ptr(x_dev), ptr(labels), ptr(L))
        else:
            nTl, nt, nlabels, nL = next_prep
            self._declared = (nTl, int(sum(nTl)))
            self.lib.call("clstm_net_train_step_next", self.h, ptr(t), len(Tl), ptr(x_dev), ptr(labels), ptr(L),
                          ptr(nt), len(nTl), ptr(next_x_dev), ptr(nlabels), ptr(nL))

    def train_step_host(self, prep, x_host):
        """The same step fed from host memory (numpy array or pinned torch tensor [sum T, ninput]): the frames travel on a
        copy stream while the previous step computes (clstm_net_train_step_h)."""
        Tl, t, labels, L = prep
        self.T, self.N = Tl, int(sum(Tl))
        self.lib.call("clstm_net_train_step_h", self.h, ptr(t), len(Tl), ptr(x_host), ptr(labels), ptr(L))

    def replica_check(self):
        """Enqueue the replica-consistency check (parameter checksum all-reduced over the attached communicator; collective).
        A mismatch surfaces at the next synchronisation point as 'replicas diverged'.  The library also runs it by itself
        every CLSTM_REPLICA_CHECK_EVERY updates.  Reference: distribute_weights (clstm.cc:718-729) re-syncs instead."""
        self.lib.call("clstm_net_replica_check", self.h)

    def set_training(self, on):
        """False: forward passes of this net belong to no training step (predict): a non-finite logit does not arm the
        device NaN flag that blocks updates (the reference asserts in backward only, clstm.cc:630-649)."""
        self.lib.call("clstm_net_set_training", self.h, int(bool(on)))

    def set_comm(self, comm):
        """Attach a `Comm` (RCCL): update()/train_step() all-reduce the fresh gradient first."""
        self._comm = comm
        self.lib.call("clstm_net_set_comm", self.h, comm.h if comm is not None else None)

    # -- state externalisation: n_states / get_states / set_states (clstm.cc:762-811) --------
    def n_states(self):
        n = C.c_longlong()
        self.lib.call("clstm_net_n_states", self.h, C.byref(n))
        return n.value

    def get_states(self):
        a = np.empty(self.n_states(), np.float32)
        self.lib.call("clstm_net_get_states_h", self.h, ptr(a), a.size)
        return a

    def set_states(self, a):
        a = f32(a)
        self.lib.call("clstm_net_set_states_h", self.h, ptr(a), a.size)
        self.T = [int(a[1])] * int(a[3])
        self.N = sum(self.T)

    def decode(self):
        """trivial_decode (ctc.cc:159-190) of every line -> list of int arrays."""
        cls = np.zeros(self.N, np.int32)
        loc = np.zeros(self.N, np.int32)
        cnt = np.zeros(len(self.T), np.int32)
        self.lib.call("clstm_net_decode", self.h, ptr(cls), ptr(loc), ptr(cnt))
        out, o = [], 0
        for b, t in enumerate(self.T):
            out.append(cls[o:o + cnt[b]].copy())
            o += t
        return out

    def state(self, layer, direction, which):
        a = np.empty((self.N, self.nhidden[layer]), np.float32)
        self.lib.call("clstm_net_get_state_h", self.h, layer, direction, STATE_CODES[which], ptr(a))
        return a

    def device_buffers(self):
        v, d, g = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.lib.call("clstm_net_buffers", self.h, C.byref(v), C.byref(d), C.byref(g))
        return v.value, d.value, g.value

    def device_outputs(self):
        p, d = C.c_void_p(), C.c_void_p()
        self.lib.call("clstm_net_outputs", self.h, C.byref(p), C.byref(d))
        return p.value, d.value

    def set_overlap(self, mode):
        """0: weight-gradient GEMM after the backward recurrence; 1: beside it where it pays (default); 2: always."""
        self.lib.call("clstm_net_set_overlap", self.h, int(mode))

    def set_strict_f32(self, on=True):
        """every product of the step on the exact f32 MFMA (the default computes the backward weight-gradient / softmax-backward
        products as f32-grade bf16 x 3 split products)"""
        self.lib.call("clstm_net_set_strict_f32", self.h, int(bool(on)))

    def overlap_stats(self):
        n, t = C.c_longlong(), C.c_int()
        self.lib.call("clstm_net_overlap_stats", self.h, C.byref(n), C.byref(t))
        return n.value, t.value

    # -- timing (bench.py) ---------------------------------------------------------------------
    def enable_timing(self, on=True):
        self.lib.call("clstm_net_enable_timing", self.h, int(on))

    def kernel_time_ms(self, name):
        ms, n = C.c_double(), C.c_int()
        self.lib.call("clstm_net_kernel_time_ms", self.h, name.encode(), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def reset_timing(self):
        self.lib.call("clstm_net_reset_timing", self.h)


class Comm:
    """RCCL communicator of the data-parallel ranks (clstm_comm_*): one per process / GPU.
    `exchange(id_bytes_or_None) -> id_bytes` ships rank 0's 128-byte id to every rank."""

    ID_BYTES = 128

    def __init__(self, rank, nranks, exchange, lib=None):
        self.lib = lib or abi.load()
        buf = C.create_string_buffer(self.ID_BYTES)
        if rank == 0:
            self.lib.call("clstm_comm_unique_id", buf)
        ident = exchange(bytes(buf.raw) if rank == 0 else None)
        assert len(ident) == self.ID_BYTES
        h = C.c_void_p()
        self.lib.call("clstm_comm_create", C.byref(h), C.create_string_buffer(ident, self.ID_BYTES), int(rank), int(nranks))
        self.h = h
        self.rank, self.nranks = int(rank), int(nranks)

    def allreduce(self, buf, n):
        self.lib.call("clstm_allreduce_flat", self.h, ptr(buf), int(n))

    def close(self):
        if getattr(self, "h", None):
            self.lib.call("clstm_comm_destroy", self.h)
            self.h = None


def make_net(kind, ninput, noutput, nhidden, nhidden2=None, lib=None, **bufs):
    """make_net(kind, {ninput, noutput, nhidden[, nhidden2]}) -- clstm_prefab.cc:151-173."""
    if kind == "bidi":
        return Network(ninput, [nhidden], noutput, lib=lib, **bufs)
    if kind == "bidi2":
        return Network(ninput, [nhidden, nhidden2], noutput, lib=lib, **bufs)
    if kind == "lstm1":
        return Network(ninput, [nhidden], noutput, unidirectional=True, lib=lib, **bufs)
    raise ValueError("no such network or layer: %s (on the MI355X path: bidi, bidi2, lstm1)" % kind)


def sgd_update(net):
    net.update()


def mktargets(transcript):
    """ctc.cc:148-157 as a list of state classes."""
    lib = abi.load()
    tr = i32(transcript)
    st = np.zeros(2 * len(tr) + 1, np.int32)
    lib.call("clstm_mktargets", ptr(st), ptr(tr), len(tr))
    return st
