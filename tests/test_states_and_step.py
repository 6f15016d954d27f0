"""State externalisation (n_states/get_states/set_states, clstm.cc:762-811 -- upstream test
test-lstm2.cc:79-142), the one-call training step (CLSTMOCR::train, clstmhl.h:201-223) and the
communicator entry points of the C ABI.  Each test runs on the host emulator (CPU suite) and on the
MI355X (-m gpu) through the same ABI."""
import numpy as np
import pytest

from common import assert_close, synth_lines
from oracle.oracle import OracleNet


def _walk(kind, ni, nh, nc):
    """(name, rows, oracle accessor) in walk_states(net, f, "", io=true) order, clstm.cc:63-70."""
    out = [("Stacked.inputs", ni, None), ("Stacked.outputs", nc, None)]
    lin = ni
    for l, no in enumerate(nh):
        def nplstm(d, lin=lin, no=no, l=l):
            return [("inputs", lin, None), ("outputs", no, (l, d, "outputs")), ("ci", no, (l, d, "ci")),
                    ("gf", no, (l, d, "gf")), ("gi", no, (l, d, "gi")), ("go", no, (l, d, "go")),
                    ("source", lin + no, (l, d, "source")), ("state", no, (l, d, "state"))]
        if kind == "lstm1":
            out += nplstm(0)
            lin = no
        else:
            out += [("Parallel.inputs", lin, None), ("Parallel.outputs", 2 * no, None)] + nplstm(0)
            out += [("Reversed.inputs", lin, None), ("Reversed.outputs", no, None)] + nplstm(1)
            lin = 2 * no
    out += [("Softmax.inputs", lin, None), ("Softmax.outputs", nc, None)]
    return out


@pytest.mark.parametrize("kind,ni,nh,nc,T,bs", [("bidi", 5, [6], 4, 7, 2), ("lstm1", 1, [4], 2, 9, 1),
                                                ("bidi2", 4, [5, 3], 6, 4, 3)])
def test_states_roundtrip_reference_format(backend, ora32, kind, ni, nh, nc, T, bs):
    from clstm_amd.net import Network
    uni = kind == "lstm1"
    rng = np.random.default_rng(3)
    ref = OracleNet(ora32, ni, nh, nc, unidirectional=uni, seed=0.222)
    params = ref.get_params() * 30.0
    ref.set_params(params)
    x = np.stack(synth_lines(rng, [T] * bs, ni), 1)            # [T][bs][ni]
    ref.set_inputs(x)
    zref = ref.forward()
    net = Network(ni, nh, nc, unidirectional=uni, lib=backend.lib)
    net.set_params(params)
    net.set_inputs([x[:, b, :] for b in range(bs)])
    net.forward()
    walk = _walk(kind, ni, nh, nc)
    total = sum(T * rows * bs + 4 for _, rows, _ in walk)        # n_states, clstm.cc:762-769
    assert net.n_states() == total
    st = net.get_states()
    assert st.size == total
    pos = 0
    for name, rows, acc in walk:
        assert st[pos:pos + 4].tolist() == [999999.0, T, rows, bs], name    # get_states header, clstm.cc:776-779
        blk = st[pos + 4:pos + 4 + T * rows * bs].reshape(T, rows, bs)
        pos += 4 + T * rows * bs
        if name == "Stacked.inputs":
            assert np.array_equal(blk, np.transpose(x, (0, 2, 1)))
        elif name in ("Stacked.outputs", "Softmax.outputs"):
            assert_close(np.transpose(blk, (0, 2, 1)), zref, what=name)
        elif acc is not None:   # NPLSTM states, in the NPLSTM's own time order (the one inside Reversed runs reversed)
            assert_close(np.transpose(blk, (0, 2, 1)), ref.state(*acc), what=name + str(acc))
    assert pos == total
    # test-lstm2.cc:96-121: a FRESH network given the states and the weights must produce the same backward pass
    dz = rng.normal(0, 0.1, (T * bs, nc)).astype(np.float32)
    net.set_output_deltas(dz)
    net.backward()
    g1 = net.get_grads()
    net2 = Network(ni, nh, nc, unidirectional=uni, lib=backend.lib)
    net2.set_states(st)
    net2.set_params(params)
    assert np.array_equal(net2.get_states(), st)
    net2.set_output_deltas(dz)
    net2.backward()
    assert np.array_equal(net2.get_grads(), g1), "backward from externalised states differs"
    with pytest.raises(Exception):
        net2.set_states(st[:-1])                                  # "size mismatch in set_states", clstm.cc:801-809


def test_states_need_rectangular_batch(backend):
    from clstm_amd.net import Network
    net = Network(3, [4], 3, lib=backend.lib)
    net.set_inputs([np.zeros((4, 3), np.float32), np.zeros((2, 3), np.float32)])
    net.forward()
    with pytest.raises(Exception, match="equal-length"):
        net.n_states()


@pytest.mark.parametrize("nh", [[10], [7, 5]])
def test_train_step_is_the_sequence_of_calls(backend, nh):
    """clstm_net_train_step == set_batch; set_inputs_d; forward; ctc; backward; update -- bit for bit,
    over several steps (momentum carried in derivs).  (Without a communicator train_step applies the update inside the
    slab reductions of its backward pass, layer by layer: same arithmetic, one launch less.)"""
    from clstm_amd.init import init_params
    from clstm_amd.net import Network
    ni, nc = 6, 7
    rng = np.random.default_rng(5)
    p0 = init_params(ni, nh, nc, seed=0.222) * 20
    a = Network(ni, nh, nc, lib=backend.lib)
    b = Network(ni, nh, nc, lib=backend.lib)
    for n in (a, b):
        n.set_params(p0)
        n.setLearningRate(1e-2, 0.9)

    def kept_current():   # times the fused update rewrote the packed copies of the parameters it moved (ops.h: PackDst)
        import ctypes
        out = ctypes.c_longlong(0)
        backend.lib.call("clstm_debug_path_count", 10, ctypes.byref(out))
        return out.value
    before = kept_current()
    for step in range(3):
        T = [int(t) for t in rng.integers(3, 9, 3)]
        lines = synth_lines(rng, T, ni)
        trs = [rng.integers(1, nc, max(1, t // 3)).astype(np.int32) for t in T]
        xd = backend.up(np.concatenate(lines, 0))
        a.set_batch(T)
        a.set_inputs_device(xd)
        a.forward()
        a.ctc(trs)
        a.backward()
        a.update()
        b.train_step(T, xd, trs)
        assert np.array_equal(a.get_params(), b.get_params()), step
        assert np.array_equal(a.get_derivs(), b.get_derivs()), step
        assert [d.tolist() for d in a.decode()] == [d.tolist() for d in b.decode()]
    # a single narrow layer: the update of every one-call step also rewrote the layer's packed parameter copies -- the next
    # step's ingest launch repacked nothing, and the steps above still equal the sequence of calls, which repacks from scratch
    assert kept_current() - before == (3 if len(nh) == 1 else 0)


def test_allreduce_single_rank_is_identity(backend):
    """World size 1 through the communicator entry points: on the MI355X this is a real RCCL communicator
    (ncclCommInitRank + ncclAllReduce on the library stream); update() with it attached must equal update()
    without it bit for bit."""
    from clstm_amd.init import init_params
    from clstm_amd.net import Comm, Network
    ni, nh, nc = 5, [8], 6
    rng = np.random.default_rng(9)
    p0 = init_params(ni, nh, nc, seed=0.222) * 20
    comm = Comm(0, 1, lambda ident: ident, lib=backend.lib)
    assert backend.lib.call("clstm_comm_size", comm.h) == 1 and backend.lib.call("clstm_comm_rank", comm.h) == 0
    nets = [Network(ni, nh, nc, lib=backend.lib) for _ in range(2)]
    nets[1].set_comm(comm)
    T = [6, 4]
    lines = synth_lines(rng, T, ni)
    trs = [rng.integers(1, nc, 2).astype(np.int32) for _ in T]
    for n in nets:
        n.set_params(p0)
        n.setLearningRate(1e-2, 0.9)
        for _ in range(2):
            n.set_inputs(lines)
            n.forward()
            n.ctc(trs)
            n.backward()
            n.update()
    assert np.array_equal(nets[0].get_params(), nets[1].get_params())
    assert np.array_equal(nets[0].get_derivs(), nets[1].get_derivs())
    # the flat all-reduce on its own: sum over one rank = identity
    v = backend.up(np.arange(1000, dtype=np.float32))
    comm.allreduce(v, 1000)
    assert np.array_equal(backend.down(v), np.arange(1000, dtype=np.float32))
    nets[1].set_comm(None)
    comm.close()


@pytest.mark.parametrize("nh,T", [([9], [21, 13, 17]), ([7, 5], [18, 11]), ([100], [70, 33, 64, 9])])
def test_overlapped_weight_gradient_gemm(backend, ora32, nh, T):
    """The chunked weight-gradient GEMM that runs beside the backward recurrence (gemm_dw.h) against the plain
    path and the oracle: same gradient sums (different slab order), no slab may have given up waiting."""
    from clstm_amd.net import Network
    from common import oracle_minibatch
    ni, nc = 6, 5
    rng = np.random.default_rng(11)
    ref = OracleNet(ora32, ni, nh, nc, seed=0.222)
    params = ref.get_params() * (30.0 if max(nh) < 50 else 8.0)
    lines = synth_lines(rng, T, ni)
    trs = [rng.integers(1, nc, max(1, t // 3)).astype(np.int32) for t in T]
    want = oracle_minibatch(ora32, OracleNet, params, ni, nh, nc, lines, trs)["derivs"]
    grads = []
    for mode in (0, 2):
        net = Network(ni, nh, nc, lib=backend.lib)
        net.set_overlap(mode)
        net.set_params(params)
        for _ in range(2):          # twice: progress words of the first pass must not satisfy the second
            net.set_inputs(lines)
            net.forward()
            net.ctc(trs)
            net.backward()
        launches, timeouts = net.overlap_stats()
        assert timeouts == 0
        assert (launches > 0) == (mode != 0)
        grads.append(net.get_grads())
        assert_close(grads[-1], want, rtol=1e-4, atol=1e-9, scale_atol=1e-4, what="gradient, overlap mode %d" % mode)
    for g in grads[1:]:
        assert_close(g, grads[0], rtol=1e-5, atol=1e-9, scale_atol=1e-5, what="overlapped vs plain")


def test_update_is_skipped_while_a_device_error_is_pending(backend, ora32):
    """A failed persistent recurrence launch (or a timed-out weight-gradient item) is reported to the host
    asynchronously; until then the host keeps enqueueing minibatches.  None of them may be APPLIED: k_update looks at
    the device error words and leaves parameters and momentum untouched, the next synchronisation point raises, and
    training resumes afterwards."""
    from clstm_amd.net import Network
    rng = np.random.default_rng(5)
    ni, nh, nc, T = 5, 6, 4, [7, 4]
    params = OracleNet(ora32, ni, nh, nc, seed=0.222).get_params() * 20.0
    lines = synth_lines(rng, T, ni)
    trs = [rng.integers(1, nc, 2).astype(np.int32) for _ in T]
    net = Network(ni, nh, nc, lib=backend.lib)
    net.set_params(params)
    net.setLearningRate(1e-2, 0.9)

    def step():
        net.set_inputs(lines); net.forward(); net.ctc(trs); net.backward(); net.update()
    for which in (0, 1):
        backend.lib.call("clstm_debug_set_device_error", which, 2)
        net.set_inputs(lines); net.forward(); net.ctc(trs); net.backward()
        backend.lib.dll.clstm_net_update(net.h)            # enqueued; must not touch v or d
        with pytest.raises(Exception, match="NOT applied"):
            backend.lib.call("clstm_synchronize")
        assert np.array_equal(net.get_params(), params.astype(np.float32))
        assert not net.get_derivs().any()
    step()                                                  # the words were cleared by the report: updates resume
    assert not np.array_equal(net.get_params(), params.astype(np.float32))


@pytest.mark.parametrize("nh,precision", [([10], 0), ([7, 5], 0), ([136], 0), ([136, 132], 2)])
def test_train_step_next_equals_train_step(backend, nh, precision):
    """clstm_net_train_step_next: the NEXT minibatch's front half -- batch geometry, alignment metadata,
    the ingest of its frames -- rides this step's last launch (ops.h: k_reduce_scatter_ingest), and the next call starts with its
    forward launch.  Bit for bit the sequence of plain clstm_net_train_step calls, over minibatches of changing geometry; a
    call that passes another minibatch than the one declared takes the ordinary path (and is still right); between the two
    calls the per-batch accessors refuse; the path counters say the tail ran and was used."""
    import ctypes
    from clstm_amd.init import init_params
    from clstm_amd.net import Network
    ni, nc = 8, 7
    rng = np.random.default_rng(23)
    p0 = init_params(ni, nh, nc, seed=0.222) * (20 if max(nh) < 100 else 1)
    a, b = Network(ni, nh, nc, lib=backend.lib), Network(ni, nh, nc, lib=backend.lib)
    for n in (a, b):
        n.set_params(p0)
        n.setLearningRate(1e-2 if max(nh) < 100 else 1e-4, 0.9)
        if precision:       # wide layers with bf16 MFMA operands (the persistent lock-step recurrences on the GPU)
            n.set_gemm_precision(precision)

    def count(i):
        out = ctypes.c_longlong(0)
        backend.lib.call("clstm_debug_path_count", i, ctypes.byref(out))
        return out.value
    batches = []
    for k in range(7):
        T = [int(t) for t in rng.integers(3, 12, 2 + k % 3)]
        trs = [rng.integers(1, nc, max(1, t // 3)).astype(np.int32) for t in T]
        x = backend.up(np.ascontiguousarray(np.concatenate(synth_lines(rng, T, ni), 0), np.float32))
        batches.append((Network.prepare_step(T, trs), x, T, trs))
    tails0, used0 = count(19), count(20)
    for k in range(6):
        prep, x, T, trs = batches[k]
        a.train_step_prepared(prep, x)
        if k == 3:      # declares batch 6, but the next call brings batch 4: the ordinary path, and the declaration is dropped
            b.train_step_prepared(prep, x, batches[6][0], batches[6][1])
        elif k < 5:
            b.train_step_prepared(prep, x, batches[k + 1][0], batches[k + 1][1])
        else:
            b.train_step_prepared(prep, x)
        assert np.array_equal(a.get_params(), b.get_params()), k
        assert np.array_equal(a.get_derivs(), b.get_derivs()), k
        if k < 5:       # the net's minibatch is the declared one now: nothing of step k is addressable
            with pytest.raises(Exception, match="no current minibatch"):
                b.decode()
    assert [d.tolist() for d in a.decode()] == [d.tolist() for d in b.decode()]
    assert np.array_equal(a.outputs(), b.outputs())
    # a declared minibatch can also be adopted by an explicit forward pass (its outputs are then addressable), and frames set
    # by hand replace it: neither may leave a stale declaration behind
    prep6, x6, T6, trs6 = batches[6]
    a.train_step_prepared(batches[5][0], batches[5][1])
    b.train_step_prepared(batches[5][0], batches[5][1], prep6, x6)
    a.set_batch(T6); a.set_inputs_device(x6); a.forward()
    b.forward()
    assert np.array_equal(a.outputs(), b.outputs())
    a.train_step_prepared(prep6, x6)
    b.train_step_prepared(prep6, x6)
    assert np.array_equal(a.get_params(), b.get_params())
    tails0 += 1                             # (the step above carried batch 6; the explicit forward adopted it, nobody "used" it)
    assert count(19) - tails0 == 5          # five steps carried a next minibatch in their last launch ...
    assert count(20) - used0 == 4           # ... four of which the next call used (one was declared and not brought)


def test_train_step_from_host_memory(backend, ora32):
    """clstm_net_train_step_h: the step fed from host frames (pageable numpy memory here; on the GPU the frames go through
    the library's pinned staging buffer and a copy stream, double-buffered) must equal the device-resident
    clstm_net_train_step bit for bit over several steps with changing batch shapes -- the slot hand-off (a step may only
    overwrite the input buffer of the step two before it once that one has ended) included."""
    from clstm_amd.init import init_params
    from clstm_amd.net import Network
    ni, nh, nc = 6, 9, 5
    rng = np.random.default_rng(17)
    p0 = init_params(ni, nh, nc, seed=0.222) * 20
    a, b = Network(ni, nh, nc, lib=backend.lib), Network(ni, nh, nc, lib=backend.lib)
    for n in (a, b):
        n.set_params(p0)
        n.setLearningRate(1e-2, 0.9)
    for step in range(6):
        T = [int(t) for t in rng.integers(3, 12, 2 + step % 3)]
        lines = synth_lines(rng, T, ni)
        trs = [rng.integers(1, nc, max(1, t // 3)).astype(np.int32) for t in T]
        x = np.ascontiguousarray(np.concatenate(lines, 0), np.float32)
        prep = Network.prepare_step(T, trs)
        a.train_step_prepared(prep, backend.up(x))
        b.train_step_host(prep, x)
        x[:] = -7.0            # pageable source: free for reuse as soon as the call returns
    assert np.array_equal(a.get_params(), b.get_params())
    assert np.array_equal(a.get_derivs(), b.get_derivs())


def test_failed_host_steps_do_not_block_the_next_one(backend, ora32):
    """a clstm_net_train_step_h call that fails on the host side (label out of range) used to burn its step
    number; after two such calls the next valid call waited for ever for a step that was never enqueued.  A step number is
    committed only when the step's last kernel is enqueued: failed calls leave no trace and the valid step equals the
    device-resident clstm_net_train_step bit for bit."""
    from clstm_amd.init import init_params
    from clstm_amd.net import Network
    ni, nh, nc = 6, 9, 5
    rng = np.random.default_rng(23)
    p0 = init_params(ni, nh, nc, seed=0.222) * 20
    a, b = Network(ni, nh, nc, lib=backend.lib), Network(ni, nh, nc, lib=backend.lib)
    for n in (a, b):
        n.set_params(p0)
        n.setLearningRate(1e-2, 0.9)
    for step in range(5):
        T = [int(t) for t in rng.integers(3, 12, 3)]
        lines = synth_lines(rng, T, ni)
        trs = [rng.integers(1, nc, max(1, t // 3)).astype(np.int32) for t in T]
        x = np.ascontiguousarray(np.concatenate(lines, 0), np.float32)
        bad = [t.copy() for t in trs]
        bad[1][0] = nc + 3                                   # target class out of range
        for _ in range(3):                                   # three failures in a row
            with pytest.raises(Exception, match="out of range"):
                b.train_step_host(Network.prepare_step(T, bad), x)
        prep = Network.prepare_step(T, trs)
        a.train_step_prepared(prep, backend.up(x))
        b.train_step_host(prep, x)                           # must return (pytest-timeout guards the old behaviour)
    assert np.array_equal(a.get_params(), b.get_params())
    assert np.array_equal(a.get_derivs(), b.get_derivs())


@pytest.mark.parametrize("fused", [True, False])
def test_non_finite_gradient_is_never_applied(backend, ora32, fused):
    """The reference asserts on NaN in every backward step (clstm.cc:630-649).  Here the update path checks every gradient
    entry it touches anyway: a non-finite minibatch gradient is not applied (parameters and momentum stay finite and
    unchanged), the device error word takes the step number, later updates are skipped until the host has reported it
    (clstm_last_error names the step), and training resumes afterwards.  fused: clstm_net_train_step (update inside the slab
    reduction); not fused: the sequence of calls (k_update)."""
    from clstm_amd.net import Network
    rng = np.random.default_rng(5)
    ni, nh, nc, T = 5, 6, 4, [7, 4]
    params = OracleNet(ora32, ni, nh, nc, seed=0.222).get_params() * 20.0
    lines = synth_lines(rng, T, ni)
    trs = [rng.integers(1, nc, 2).astype(np.int32) for _ in T]
    net = Network(ni, nh, nc, lib=backend.lib)
    net.set_params(params)
    net.setLearningRate(1e-2, 0.9)
    prep = Network.prepare_step(T, trs)

    def step(ls):
        x = np.ascontiguousarray(np.concatenate(ls, 0), np.float32)
        if fused:
            net.train_step_prepared(prep, backend.up(x))
        else:
            net.set_inputs(ls); net.forward(); net.ctc(trs); net.backward(); net.update()
    step(lines)                                              # step 1: fine
    backend.sync()
    p1, d1 = net.get_params(), net.get_derivs()
    assert np.isfinite(p1).all() and not np.array_equal(p1, params.astype(np.float32))
    poisoned = [l.copy() for l in lines]
    poisoned[1][2, 3] = np.nan
    step(poisoned)                                           # step 2: NaN reaches every gradient entry of the layer
    step(lines)                                              # step 3: enqueued behind it, must be skipped too
    with pytest.raises(Exception, match="non-finite value .* training step 2"):
        backend.sync()
    assert np.array_equal(net.get_params(), p1) and np.array_equal(net.get_derivs(), d1)
    step(lines)                                              # reported: updates resume
    backend.sync()
    p4 = net.get_params()
    assert np.isfinite(p4).all() and not np.array_equal(p4, p1)
