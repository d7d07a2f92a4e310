"""Data parallelism for the hot path: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's single-process `DataParallelImbalance` (utils/data_parallel.py:91-132),
which every step scatters inputs from GPU0, re-broadcasts all 250 M parameters to every replica,
runs the replicas from Python threads and reduce-adds the gradients back onto GPU0.  Here each rank
owns a persistent replica and its own shard of the batch; the only exchange is the gradient
all-reduce (average), issued per arena bucket on a side stream as soon as backward has finished
that bucket (one contiguous slice per encoder block: 28-80 MB fp32 at the full config), so it
overlaps with the rest of backward.  Losses stay per-rank means and gradients are averaged, which
reproduces the reference's `lm_loss.mean()` over replicas (train.py:164-166).

The wrapper is device-agnostic: with CPU tensors and the gloo backend the same bucket logic runs
synchronously (that is what the multi-process CPU tests exercise)."""
from __future__ import annotations

from contextlib import contextmanager

import torch
import torch.distributed as dist
from torch import nn

from .arena import FlatArena


class DataParallelRCCL(nn.Module):
    def __init__(self, module: nn.Module, process_group=None, device=None, broadcast_params=True, reduce_when_single=False):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._single_ok = reduce_when_single and dist.is_initialized()   # tests: run the exchange with 1 rank too
        self._sync = True
        self._pending = []
        self._comm_stream = None
        self._done = set()
        eng = self._engine()
        if eng is not None:
            if device is None:
                device = next(module.parameters()).device
            eng.ensure(device)
            self.arena = eng.arena
            eng.grad_bucket_hook = self._on_bucket
            if self.arena.flat.is_cuda:
                # high priority: the exchange's workgroups take CUs as they free up instead of queueing behind the
                # backward kernels that fill the chip (the point of issuing buckets early is to finish them under backward)
                self._comm_stream = torch.cuda.Stream(device=self.arena.flat.device, priority=-1)
        else:
            named = dict(module.named_parameters())
            groups = [("all", [(n, tuple(p.shape)) for n, p in named.items()])]
            self.arena = FlatArena(named, groups)
        self._ranges = {g: (lo, hi) for g, lo, hi in self.arena.buckets}
        if broadcast_params and self.world > 1:
            dist.broadcast(self.arena.flat, src=0, group=self.group)   # one-time replica sync

    def _engine(self):
        m = self.module
        for cand in (m, getattr(m, "bert_pretrained", None)):
            if cand is not None and hasattr(cand, "engine"):
                return cand.engine
        return None

    def forward(self, *inputs, **kwargs):
        self._done.clear()
        return self.module(*inputs, **kwargs)

    # -- gradient exchange ---------------------------------------------------------------------
    @contextmanager
    def no_sync(self):
        """Skip the all-reduce (gradient accumulation micro-steps, train.py:451-455 `batch_multiply`)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _reduce_slice(self, lo, hi):
        t = self.arena.grad_flat[lo:hi]
        if self._comm_stream is not None:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                t.mul_(1.0 / self.world)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.mul_(1.0 / self.world)

    def _on_bucket(self, group):
        """Called by the engine when backward has finished every gradient of one arena group."""
        if not self._sync or (self.world == 1 and not self._single_ok):
            return
        lo, hi = self._ranges[group]
        self._reduce_slice(lo, hi)
        self._done.add(group)
        if len(self._done) == len(self._ranges) and self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)   # optimizer sees reduced grads

    def sync_gradients(self):
        """Explicit reduction of whatever has not been reduced yet (generic modules without the engine
        hook call this after backward)."""
        if (self.world == 1 and not self._single_ok) or not self._sync:
            return
        for g, (lo, hi) in self._ranges.items():
            if g not in self._done:
                self._reduce_slice(lo, hi)
                self._done.add(g)
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)


def shard_range(n_rows: int, rank: int, world: int):
    """Even split of a batch over ranks: rows [lo, hi).  (The reference's uneven table,
    utils/data_parallel.py:16-57, only relieves GPU0 of the gather / optimizer memory that the
    process-per-GPU design does not have.)"""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
