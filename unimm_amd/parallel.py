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
    def __init__(self, module: nn.Module, process_group=None, device=None, broadcast_params=True, reduce_when_single=False,
                 wire_dtype="fp32", algorithm="allreduce", wgrad_group_rounds="auto"):
        """wire_dtype: "fp32" (default: the gradients travel as they are) or "bf16" (opt-in: every bucket is cast to
        bf16 for the exchange and back, halving the 1.0 GB per step on the xGMI links; each rank's contribution is
        pre-divided by the world size so the bf16 sum cannot overflow where the fp32 one would not).
        algorithm: "allreduce" or "rs_ag" (reduce-scatter + all-gather on the bucket: the same bytes as a ring
        all-reduce but as two collectives RCCL can place on direct xGMI links; buckets are multiples of 64 elements,
        so every world size up to 64 that divides them splits evenly - others fall back to all-reduce per bucket).
        wgrad_group_rounds: how many rounds of the chip the engine lets its weight-gradient queue grow to before a grouped
        launch (Engine.wgrad_group_rounds; buckets are handed over per launch).  "auto" = 2 with more than one rank (4 is
        the single-GPU optimum: +2 % kernel time at 30 sequences per GPU, but the LAST launch's buckets -- text blocks
        0..5 and the embeddings, 270 MB -- are exchanged after backward has ended, and halving that launch takes ~0.7 ms off
        the exposed tail of an 8-rank ring at 300 GB/s), an int = that value, None = leave the engine's."""
        super().__init__()
        if wire_dtype not in ("fp32", "bf16") or algorithm not in ("allreduce", "rs_ag"):
            raise ValueError(f"DataParallelRCCL: wire_dtype {wire_dtype!r} / algorithm {algorithm!r}")
        self.wire_dtype, self.algorithm = wire_dtype, algorithm
        self._stats = dict(buckets=0, bytes_wire=0, calls=0)
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._single_ok = reduce_when_single and dist.is_initialized()   # tests: run the exchange with 1 rank too
        self._sync = True
        self._pending = []
        self._comm_stream = None
        self._done = set()
        self._run = None                 # open run of adjacent buckets: [lo, hi, count]
        self._wire16 = None              # persistent exchange buffers (see _exchange_buffers)
        self._shard = None
        eng = self._engine()
        if eng is not None:
            if device is None:
                device = next(module.parameters()).device
            eng.ensure(device)
            self.arena = eng.arena
            eng.register_arena_user(self)
            eng.grad_bucket_hook = self._on_bucket
            if wgrad_group_rounds == "auto":
                wgrad_group_rounds = 2 if self.world > 1 else None
            if wgrad_group_rounds is not None:
                eng.wgrad_group_rounds = int(wgrad_group_rounds)
            # every replica draws its own dropout masks (the reference's replicas each consume torch's per-device
            # generator): fold the rank into the counter-based seed
            rank = dist.get_rank(process_group) if dist.is_initialized() else 0
            eng.seed = (eng.seed ^ (0x9E3779B1 * (rank + 1))) & 0xFFFFFFFF if rank else eng.seed
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
            if eng is not None:
                eng.invalidate_weights()                               # the bf16 copies follow the broadcast values

    def _engine(self):
        m = self.module
        for cand in (m, getattr(m, "bert_pretrained", None)):
            if cand is not None and hasattr(cand, "engine"):
                return cand.engine
        return None

    def forward(self, *inputs, **kwargs):
        self._done.clear()
        self._run = None
        return self.module(*inputs, **kwargs)

    def forward_backward(self, *inputs, **kwargs):
        """The wrapped module's one-call training step (BertForMultiModalPreTraining.forward_backward): the gradient buckets are
        handed over from inside its backward exactly as under `loss.backward()`."""
        self._done.clear()
        self._run = None
        return self.module.forward_backward(*inputs, **kwargs)

    # -- gradient exchange ---------------------------------------------------------------------
    @contextmanager
    def no_sync(self):
        """Skip the all-reduce (gradient accumulation micro-steps, train.py:451-455 `batch_multiply`)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _exchange_buffers(self, lo, hi):
        """Persistent exchange buffers (allocated once, on first use): a bf16 mirror of the gradient arena for the bf16
        wire (a bucket travels as the same [lo, hi) slice of it) and one reduce-scatter shard sized for the largest
        exchange.  A step allocates nothing."""
        wire = None
        if self.wire_dtype == "bf16":
            if self._wire16 is None:
                self._wire16 = torch.empty(self.arena.grad_flat.numel(), dtype=torch.bfloat16, device=self.arena.grad_flat.device)
            wire = self._wire16[lo:hi]
        shard = None
        if self.algorithm == "rs_ag" and self.world > 1 and (hi - lo) % self.world == 0:
            need = (hi - lo) // self.world
            dt = torch.bfloat16 if self.wire_dtype == "bf16" else torch.float32
            if self._shard is None or self._shard.numel() < need:
                self._shard = torch.empty(max(need, self.arena.grad_flat.numel() // self.world + 64), dtype=dt,
                                          device=self.arena.grad_flat.device)
            shard = self._shard[:need]
        return wire, shard

    def _exchange(self, lo, hi):
        """Average grad_flat[lo:hi] (a contiguous fp32 slice of the gradient arena) over the ranks, in place."""
        t = self.arena.grad_flat[lo:hi]
        inv = 1.0 / self.world
        wire, shard = self._exchange_buffers(lo, hi)
        if wire is None:
            wire = t
        else:
            torch.mul(t, inv, out=wire)                  # pre-divided: the bf16 sum stays in range
        n = wire.numel()
        if shard is not None:
            dist.reduce_scatter_tensor(shard, wire, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(wire, shard, group=self.group)
        else:
            dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group)
        if wire is t:
            t.mul_(inv)
        else:
            t.copy_(wire)
        self._stats["bytes_wire"] += n * wire.element_size()
        self._stats["calls"] += 1

    def _reduce_slice(self, lo, hi, n_buckets=1):
        if self._comm_stream is not None:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                self._exchange(lo, hi)
        else:
            self._exchange(lo, hi)
        self._stats["buckets"] += n_buckets

    def comm_stats(self, reset=False):
        """Buckets exchanged, bytes put on the wire per rank (payload, before the algorithm's factor), collective calls."""
        out = dict(self._stats, n_buckets_expected=len(self._ranges), wire_dtype=self.wire_dtype, algorithm=self.algorithm,
                   bucket_bytes=[(g, (hi - lo) * (2 if self.wire_dtype == "bf16" else 4)) for g, (lo, hi) in self._ranges.items()])
        if reset:
            self._stats = dict(buckets=0, bytes_wire=0, calls=0)
        return out

    def _on_bucket(self, group, more=False):
        """Called by the engine when every gradient of one arena group is final (its kernels are enqueued).  The engine
        launches the weight gradients of several blocks together, so buckets arrive in runs (`more` = the next one
        follows at once): adjacent slices of a run travel as ONE collective."""
        if not self._sync or (self.world == 1 and not self._single_ok):
            return
        if group in self._done:
            raise RuntimeError(f"gradient bucket {group!r} was handed over twice in one step")
        lo, hi = self._ranges[group]
        self._done.add(group)
        if self._run and (self._run[0] == hi or self._run[1] == lo):          # extends the open run
            self._run = [min(lo, self._run[0]), max(hi, self._run[1]), self._run[2] + 1]
        else:
            if self._run:
                self._reduce_slice(*self._run)
            self._run = [lo, hi, 1]
        if not more:
            self._reduce_slice(*self._run)
            self._run = None
        if len(self._done) == len(self._ranges):
            assert self._run is None
            if self._comm_stream is not None:
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
