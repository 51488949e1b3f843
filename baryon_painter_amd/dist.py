"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-device (scripts/CVAE_single_scale.py:23).  Sharding the minibatch
changes nothing arithmetically if (a) batch-norm uses GLOBAL batch statistics -- forward
{sum x, sum x^2} and backward {sum g, sum g*x} are all-reduced per layer (tiny messages) -- and
(b) parameter gradients are averaged across ranks (each rank normalises its loss by its local
batch, cvae.py:129,144).  Gradients travel as ONE flat buffer (all parameters are views of a
single allocation, see CVAE._flatten_parameters): 6.65 MB per step for the fiducial network.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


class PeerAllReduce:
    """The batch-norm statistics' all-reduce as ONE kernel over peer memory (csrc/peer_comm.hip, include/bp_hip.h
    ``bp_peer_*``): every rank's fine-grained exchange buffer is mapped into every other rank's address space through IPC
    handles gathered once over ``group``; a collective is then a single launch on the caller's stream (peer stores, flags,
    rank-order sum) instead of 16-18 us of torch.distributed / RCCL plumbing per 2 KB message, 44 times per step.

    ``create`` returns None when the path is not available (no GPU, handles cannot be opened -- ranks on different nodes
    --, or the start-up self-test with known data fails or times out on ANY rank): the caller then keeps the process
    group's all-reduce.  The decision is collective: every rank takes the same branch."""

    def __init__(self, lib, comm, rank, world):
        self.lib, self.comm, self.rank, self.world = lib, comm, rank, world
        self.max_doubles = int(lib.bp_peer_max_doubles())
        self.slots = int(lib.bp_peer_slots())

    @staticmethod
    def create(group, device):
        from . import _lib as L
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if not torch.cuda.is_available():
            return None
        lib = L.load()
        nb = int(lib.bp_peer_handle_bytes())
        comm, handle = C.c_void_p(), (C.c_char * nb)()
        with torch.cuda.device(device):
            ok = lib.bp_peer_create(rank, world, C.byref(comm), handle) == L.BP_OK
            # every rank learns every rank's handle (and whether its creation worked) through the process group
            mine = torch.tensor(list(bytes(handle)) + [1 if ok else 0], dtype=torch.uint8)
            backend = dist.get_backend(group)
            if backend == "nccl":
                mine = mine.to(device)
            gathered = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine, group=group)
            gathered = [g.cpu() for g in gathered]
            all_ok = all(int(g[-1]) == 1 for g in gathered)
            peer = None
            if ok and all_ok:
                blob = b"".join(bytes(g[:-1].tolist()) for g in gathered)
                if lib.bp_peer_open(comm, blob) == L.BP_OK:
                    peer = PeerAllReduce(lib, comm, rank, world)
            # (any exception of the self-test counts as a failed test on this rank: every rank must reach the vote below)
            try:
                good = peer is not None and peer._self_test(device)
            except Exception:
                good = False
            flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) != 1:
                if comm.value:
                    torch.cuda.synchronize(device)
                    lib.bp_peer_destroy(comm)
                return None
        return peer

    def _self_test(self, device, rounds=None):
        """Known contributions, once around the ring of slots and a little further: the sum must be exact on this rank."""
        rounds = self.slots + 8 if rounds is None else rounds
        n = 257
        j = torch.arange(n, dtype=torch.float64, device=device)
        for r in range(rounds):
            t = (self.rank + 1.0) * (j + 1.0) + r
            self.all_reduce_sum(t)
            want = (self.world * (self.world + 1) / 2.0) * (j + 1.0) + r * self.world
            if not torch.equal(t, want):
                return False
        # ... and the per-channel exchange that the batch-norm finalize kernels run (bp_peer_bind): 128 channels x 2 sums
        from . import _lib as L
        c = 128
        j = torch.arange(2 * c, dtype=torch.float64, device=device)
        for r in range(4):
            t = (self.rank + 1.0) * (j + 1.0) - r
            L.check(self.lib.bp_peer_exchange_check(self.comm, t.data_ptr(), c,
                                                    C.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "peer exchange check")
            want = (self.world * (self.world + 1) / 2.0) * (j + 1.0) - r * self.world
            if not torch.equal(t, want):
                return False
        return self.timeouts() == 0

    def all_reduce_sum(self, t):
        from . import _lib as L
        L.check(self.lib.bp_peer_all_reduce(self.comm, t.data_ptr(), t.numel(), 0,
                                            C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)),
                "peer all-reduce")

    def usable(self, t):
        return t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and 0 < t.numel() <= self.max_doubles

    def bind(self, on):
        """Bind (unbind) the communicator to this host thread: the fused batch-norm finalize launches issued while it is
        bound exchange their channel sums with the other ranks themselves (``bp_peer_bind``)."""
        from . import _lib as L
        L.check(self.lib.bp_peer_bind(self.comm if on else None), "peer bind")

    def timeouts(self):
        return int(self.lib.bp_peer_status(self.comm))


class Sync:
    """Collectives the launch plan needs.  ``sync_bn=False`` keeps batch-norm statistics local
    (throughput mode: different arithmetic from the single-device reference)."""

    def __init__(self, group=None, sync_bn=True, grad_group=None):
        """``group``: communicator of every collective (batch-norm statistics: latency-bound, on the critical path, main
        stream; the flat gradient buffer: once, after the backward pass).  All collectives are issued by one host
        thread in program order on ONE communicator, so their order is the same on every rank by construction.

        ``grad_group``: opt-in SECOND communicator for the gradient buffers ("new": created here, collectively, by
        every rank; also selected by BP_EARLY_ALLREDUCE=1).  With it the generator trunk's gradients are reduced
        early, on the weight-gradient stream, while statistics keep flowing on the main stream's communicator
        (cvae._Plan._reduce_trunk_gradients).  That puts two communicators' collectives in flight concurrently, whose
        relative order across ranks RCCL does not guarantee; it has only been exercised with gloo and with one rank
        (profiles/r02_rccl_one_rank.txt), so it is OFF by default until a multi-GPU RCCL run has validated it
        (tests/test_gpu_painter.py::test_two_gpu_rccl_data_parallel, skipped on one-GPU boxes).  The flat all-reduce
        it replaces costs ~0.1-0.2 ms of exposed latency per step (6.65 MB over xGMI)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.sync_bn = sync_bn
        # BP_SYNC_FORCE=1: issue every collective even with one rank (a one-GPU box then drives the whole data-parallel
        # schedule -- the float64 statistics buffers, the gradient buffer -- through RCCL itself)
        self.active = self.world_size > 1 or os.environ.get("BP_SYNC_FORCE") == "1"
        if grad_group is None and (os.environ.get("BP_EARLY_ALLREDUCE", "0") == "1"
                                   or os.environ.get("BP_GRAD_COMM", "0") == "1"):      # (=1: experiments)
            grad_group = "new"
        self.grad_group = dist.new_group() if (grad_group == "new" and self.active) else \
            (group if grad_group in ("new", None) else grad_group)
        self.overlap = self.grad_group is not group and self.active
        self.n_small = self.n_grad = self.n_fused = 0
        self.bytes_grad = 0
        self.timing = None           # bench.py: list of (start event, end event, kind) while enabled
        # batch-norm statistics over peer memory (one kernel per collective) where that path validates itself at start-up
        # on every rank; BP_PEER_SYNC=0 keeps every collective on the process group (RCCL / gloo)
        self.peer = None
        self._peer_tried = os.environ.get("BP_PEER_SYNC", "1") == "0" or not (self.sync_bn and self.active)

    def _peer_for(self, t):
        """Lazily (the device is only known at the first collective) set up the peer-memory path; None: process group."""
        if not self._peer_tried:
            self._peer_tried = True
            if t.is_cuda:
                self.peer = PeerAllReduce.create(self.group, t.device)
        return self.peer if self.peer is not None and self.peer.usable(t) else None

    def fused(self, device):
        """True when batch-norm statistics are exchanged INSIDE the kernels that finalize them (peer memory validated on
        every rank, BP_PEER_FUSED != 0): such layers issue no separate collective.  Collective at its first call (sets up
        the peer path): every rank calls it at the same point of the program (the first batch-norm layer of a step)."""
        if not (self.sync_bn and self.active):
            return False
        if not self._peer_tried:
            self._peer_tried = True
            self.peer = PeerAllReduce.create(self.group, torch.device(device))
        return self.peer is not None and os.environ.get("BP_PEER_FUSED", "1") != "0"

    def check(self):
        """After a step (synchronises): raise if a peer-memory collective timed out (its sums were then wrong)."""
        if self.peer is not None and self.peer.timeouts() != 0:
            raise RuntimeError("a peer-memory all-reduce timed out: the step's batch-norm statistics are invalid")

    def _timed(self, kind):
        import contextlib
        import torch.cuda as tc
        if self.timing is None or not tc.is_available():
            return contextlib.nullcontext()
        sync = self

        class _T:
            def __enter__(self):
                self.e0 = tc.Event(enable_timing=True)
                self.e0.record()

            def __exit__(self, *a):
                e1 = tc.Event(enable_timing=True)
                e1.record()
                sync.timing.append((self.e0, e1, kind))
        return _T()

    def all_reduce_sum(self, t):
        """Batch-norm statistics (float64 vector of 2*C entries per layer of a level)."""
        if self.sync_bn and self.active:
            peer = self._peer_for(t)
            if peer is None and self.peer is not None and os.environ.get("BP_PEER_FUSED", "1") != "0":
                # (the plan put independent branches on streams of their own because statistics travel over peer memory:
                #  a process-group collective from there would break the communicator's ordering)
                raise RuntimeError("batch-norm statistics that the peer-memory all-reduce cannot carry "
                                   f"({tuple(t.shape)} {t.dtype}): set BP_PEER_SYNC=0")
            with self._timed("bn"):
                if peer is not None:
                    peer.all_reduce_sum(t)
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self.n_small += 1

    def all_reduce_mean(self, flat):
        """A gradient buffer (or a contiguous slice of one), on the CURRENT stream."""
        if self.peer is not None:
            # once per step: the peer-memory ring must hold more than a step's statistics exchanges (csrc/peer_dev.hpp)
            done = self.n_small + self.n_fused
            if done - getattr(self, "_stats_mark", done) >= self.peer.slots // 2:
                raise RuntimeError("more batch-norm statistics exchanges per step than the peer-memory ring allows: "
                                   "set BP_PEER_SYNC=0")
            self._stats_mark = done
        if self.active:
            with self._timed("grad"):
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.grad_group)
                flat.mul_(1.0 / self.world_size)
            self.n_grad += 1
            self.bytes_grad += flat.numel() * flat.element_size()


def shard_indices(permutation, rank, world_size, batch_size):
    """Slice a GLOBAL sample permutation (the reference's DataLoader(shuffle=True) order,
    painter.py:87-91) into this rank's share of every global batch of ``batch_size*world_size``
    samples: rank r takes positions [r*batch_size, (r+1)*batch_size) of each global batch, so
    the union over ranks reproduces the single-device batches exactly."""
    gb = batch_size * world_size
    out = []
    for start in range(0, len(permutation) - gb + 1, gb):
        out.append(list(permutation[start + rank * batch_size:start + (rank + 1) * batch_size]))
    return out
