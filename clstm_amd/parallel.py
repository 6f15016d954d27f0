"""Data-parallel training step: each GPU owns an independent shard of the minibatch's text lines
and the fresh gradient sum is all-reduced (RCCL over xGMI; gloo in the CPU tests) before the
identical update on every rank.

Reference precedent: share_deltas (clstm.cc:731-744) sums Params.d over in-process replicas.
Params.d also carries the momentum (clstm_compute.cc:560-563), so summing it over R replicas
would count mom*d_prev R times; here only the separate fresh-gradient buffer `grads` crosses
GPUs and `derivs += grads` happens inside the update kernel, which keeps the single-process
semantics  d = mom*d_prev + sum_b g_b  for any world size.
"""
import numpy as np


class Trainer:
    """CLSTMOCR::train (clstmhl.h:201-223) for a minibatch of lines, optionally data-parallel."""

    def __init__(self, net, grads_tensor=None, process_group=None, comm=None):
        """comm: a clstm_amd.net.Comm (RCCL inside the library, all-reduce on the library's own stream --
        the production path); otherwise, with grads_tensor, torch.distributed's all_reduce (gloo in the CPU
        tests; also the fallback if the library communicator cannot be created)."""
        self.net = net
        self.grads = grads_tensor          # torch tensor aliasing the net's grads buffer
        self.pg = process_group
        self.dist = None
        self.comm = comm
        if comm is not None:
            net.set_comm(comm)             # clstm_net_update / clstm_net_train_step all-reduce `grads` themselves
        elif grads_tensor is not None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist = dist

    def world_size(self):
        return self.dist.get_world_size(self.pg) if self.dist else 1

    def allreduce_grads(self):
        if self.dist is not None and self.world_size() > 1:
            self.dist.all_reduce(self.grads, op=self.dist.ReduceOp.SUM, group=self.pg)

    def fwdbwd(self, lines, transcripts):
        self.net.set_inputs(lines)
        self.net.forward()
        self.net.ctc(transcripts)
        self.net.backward()

    def step_device(self, T, x_dev, transcripts):
        """One training step on inputs already resident in HBM (x_dev: [sum T, ninput])."""
        net = self.net
        if self.dist is None:              # single GPU, or the library's own communicator: ONE call
            net.train_step(T, x_dev, transcripts)
            return
        net.set_batch(T)
        net.set_inputs_device(x_dev)
        net.forward()
        net.ctc(transcripts)
        net.backward()
        self.allreduce_grads()
        net.update()

    def train(self, lines, transcripts):
        self.fwdbwd(lines, transcripts)
        out = self.net.decode()
        self.allreduce_grads()
        self.net.update()
        return out


def shard(items, rank, world):
    """Contiguous shard of a minibatch for this rank: n // world lines each, the first n % world ranks
    take one more -- so every rank owns at least one line whenever n >= world (a rank with no lines would
    sit out the collective and hang the others)."""
    n = len(items)
    if n < world:
        raise ValueError("minibatch of %d lines cannot be sharded over %d ranks (every rank joins the all-reduce)" % (n, world))
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return items[lo:lo + base + (1 if rank < extra else 0)]


def shard_by_length(lengths, rank, world):
    """Indices of the minibatch lines rank `rank` trains on, balanced by LENGTH: the lines are dealt longest first,
    round-robin (stable for equal lengths), so every rank's longest line -- which sets the length of its recurrence and
    CTC launches, one workgroup per (line, direction) -- is the same to within one position of the sorted order, and the
    all-reduce does not wait for the rank that drew the long lines (ragged T ~ U{150..250} costs a single GPU 18 %).
    Every rank computes the same deal from the same `lengths`.  Returned in ascending index order."""
    n = len(lengths)
    if n < world:
        raise ValueError("minibatch of %d lines cannot be sharded over %d ranks (every rank joins the all-reduce)" % (n, world))
    order = sorted(range(n), key=lambda i: -int(lengths[i]))     # (sorted() is stable)
    return sorted(order[rank::world])
