"""Data-parallel sharding of trajectory batches: one process per GPU, one exchange per step.

The reference has no distributed code; the path shards trivially because
every (x, x+) pair is independent given the (tiny, replicated) parameters.  Rank ``r`` of ``N`` owns
the contiguous rows ``shard_bounds(B, r, N)``; after the fused loss+gradient kernels each rank holds
``[sum_i w_i loss_i | sum_i w_i dloss_i/dparams]`` for its rows in ONE contiguous buffer, and a
single SUM all-reduce of that buffer (16 numbers for the cube, 30 for the elbow) makes every rank hold
the global mean loss and gradient, so identical optimizer steps keep the replicas in sync.

The message is pure latency, so the default transport is :class:`PeerAllReduce` -- every rank
stores its vector straight into every peer's IPC-shared receive buffer over xGMI and sums what arrived
(``csrc/dpll_allreduce.hpp``, one ~2 us kernel inside the step's hipGraph) -- verified against
``torch.distributed.all_reduce`` at start-up and replaced by it (RCCL) if the self-test fails or the
message is too long for it (mesh systems: 67 k gradients are bandwidth, not latency, bound).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import _capi

_DTYPES = {torch.float32: _capi.F32, torch.float64: _capi.F64}
PEER_MAX_BYTES = 1024  # dpll_allreduce.hpp kMaxWords * 4


def shard_bounds(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of ``batch`` rows: the first ``batch % world`` ranks get one extra."""
    base, extra = divmod(batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class PeerAllReduce:
    """One-shot all-reduce over peer memory for vectors of at most 1 KiB.  Construction is collective
    (handles travel through ``all_gather_object`` of the given process group)."""

    def __init__(self, group=None) -> None:
        lib = _capi.library()
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        size = lib.dpll_ar_handle_bytes()
        handle = (ctypes.c_char * size)()
        self._ar = ctypes.c_void_p()
        # every rank goes through the same collectives whether or not its own set-up worked, and all of them raise
        # together: a rank that bailed out alone would leave the others waiting in the next collective
        error = ''
        try:
            _capi.check(lib.dpll_ar_create(self.rank, self.world, ctypes.cast(handle, ctypes.c_void_p),
                                           ctypes.byref(self._ar)))
        except _capi.DpllError as exc:
            error = str(exc) or 'dpll_ar_create failed'
            self._ar = None
        gathered = [None] * self.world
        dist.all_gather_object(gathered, (error, bytes(handle)), group=group)
        errors = [e for e, _ in gathered if e]
        if not errors:
            blob = b''.join(h for _, h in gathered)
            try:
                _capi.check(lib.dpll_ar_connect(self._ar, ctypes.cast(ctypes.c_char_p(blob), ctypes.c_void_p)))
            except _capi.DpllError as exc:
                error = str(exc) or 'dpll_ar_connect failed'
            connected = [None] * self.world
            dist.all_gather_object(connected, error, group=group)
            errors = [e for e in connected if e]
        if errors:
            self.close()
            raise _capi.DpllError('peer-memory all-reduce set-up failed on some rank: ' + errors[0])

    def all_reduce(self, tensor: torch.Tensor) -> None:
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.numel() * tensor.element_size() <= PEER_MAX_BYTES
        _capi.check(_capi.library().dpll_ar_allreduce(self._ar, _DTYPES[tensor.dtype], tensor.data_ptr(), tensor.numel(),
                                                      torch.cuda.current_stream().cuda_stream))

    def healthy(self) -> bool:
        """Synchronises; False once any call hit its spin limit."""
        return _capi.library().dpll_ar_status(self._ar) == 0

    def self_test(self, rounds: int = 4) -> bool:
        """A few reductions of rank-dependent data checked against the closed-form sum; collective."""
        ok = True
        for step in range(rounds):
            probe = torch.arange(16, dtype=torch.float32, device='cuda') * (self.rank + 1) + step
            self.all_reduce(probe)
            expect = torch.arange(16, dtype=torch.float32, device='cuda') * (self.world * (self.world + 1) / 2) + \
                step * self.world
            ok = ok and self.healthy() and bool(torch.equal(probe, expect))
        flag = [None] * self.world
        dist.all_gather_object(flag, ok, group=self.group)
        return all(flag)

    def close(self) -> None:
        if self._ar is not None and _capi._lib is not None:
            _capi._lib.dpll_ar_destroy(self._ar)
        self._ar = None


class GradientAllReduce:
    """Binds a :class:`MultibodyLearnableSystem` to a process group.

    ``system.contactnets_loss_and_grad`` then scales every item by ``1 / (local_batch * world)``
    so that the SUM all-reduce yields the global batch mean directly (no extra scaling kernel);
    with unequal shards pass ``global_batch`` to weight by ``1 / global_batch`` instead.

    ``transport``: ``'auto'`` = the backend's collective (RCCL: ``north_star``'s single all-reduce of loss and gradients) --
    unless the environment carries ``DPLL_PEER_EXCHANGE=1``, which lets ``'auto'`` take the peer-memory one-shot kernel when
    the buffer fits and its self-test passes.  The hand-written exchange has only ever run with several processes on ONE
    device; until it has crossed real xGMI once it is opt-in (``'peer'`` asks for it explicitly).  ``'collective'`` forces
    RCCL.  ``fuse`` (peer transport): the exchange is done by the loss launch's own finalize kernel instead of a kernel after
    it."""

    def __init__(self, system, group=None, global_batch: int = 0, transport: str = 'auto', fuse: bool = True) -> None:
        self.system = system
        self.group = group
        self.world = dist.get_world_size(group)
        system.grad_world = self.world
        system.global_batch = global_batch
        self.peer: Optional[PeerAllReduce] = None
        buf = system.grad_buffer()
        fits = buf.is_cuda and buf.numel() * buf.element_size() <= PEER_MAX_BYTES
        if transport not in ('auto', 'peer', 'collective'):
            raise ValueError(transport)
        if transport == 'peer' and not fits:
            raise ValueError('gradient buffer too long for the peer-memory all-reduce')
        if transport == 'auto' and os.environ.get('DPLL_PEER_EXCHANGE', '0') != '1':
            transport = 'collective'
        if transport != 'collective' and fits and self.world > 1:
            try:
                peer = PeerAllReduce(group)
                if peer.self_test():
                    self.peer = peer
                else:
                    peer.close()
            except _capi.DpllError:
                self.peer = None
            decided = [None] * self.world  # every rank must take the same route
            dist.all_gather_object(decided, self.peer is not None, group=group)
            if not all(decided) and self.peer is not None:
                self.peer.close()
                self.peer = None
            if transport == 'peer' and self.peer is None:
                raise _capi.DpllError('peer-memory all-reduce unavailable')
        self.transport = 'peer' if self.peer is not None else 'collective'
        # fuse: with the peer transport the exchange runs inside the finalize kernel of the loss launch
        # (dpll_contactnets_loss_allreduce); all_reduce_mean() then has nothing left to launch
        self.fused = bool(fuse and self.peer is not None and getattr(system, '_mesh', lambda: None)() is None and
                          system.spec.is_fast())  # the general build reduces with a kernel of its own after the launch
        system._fused_ar = self.peer._ar if self.fused else None

    def check_healthy(self) -> None:
        """Synchronises and raises if a peer-memory exchange hit its spin limit (the affected rows were replaced by NaN,
        never by partial sums).  Collective in effect: every rank of a timed-out exchange sees its own error word."""
        if self.peer is not None and not self.peer.healthy():
            raise _capi.DpllError('peer-memory all-reduce timed out (a rank did not arrive within 50 ms); '
                                  'use GradientAllReduce(transport=\'collective\')')

    def all_reduce_mean(self) -> torch.Tensor:
        buf = self.system.grad_buffer()
        if self.fused and self.system._grad_reduced:
            self.system._grad_reduced = False  # already summed by the loss launch
            return buf
        if self.peer is not None:
            self.peer.all_reduce(buf)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        self.system.after_grad_reduce()  # (rows of welded links: chained from the summed buffer)
        return buf
