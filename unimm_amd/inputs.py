"""Compact inputs for the hot path.

The reference's dataloader materialises, per sequence, a dense [256, 256] int64 text mask and a [256] int64
co-attention mask (512 KiB, utils/data_utils.py:199-210 / :391-396) and train.py:413-432 expands one image's
region features to every round and sample (303 KB of fp32 per sequence).  Both are functions of a few
integers: `DialogMaskSpec` carries (mode, L, n) per sequence and the engine synthesises the packed masks on
the device (`unimm_mask_synth`); region features can stay per image with an `image_index` per sequence.
With a spec the engine also knows every sequence's valid length on the host, so the one device->host
synchronisation per step that the dense-mask path needs for the unpadded schedule disappears."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

DISCRIMINATIVE, GENERATIVE = 0, 1


@dataclass
class DialogMaskSpec:
    """Per-sequence mask descriptors (host int arrays of length B).
    mode: 0 discriminative / 1 generative;  length: L, tokens up to and including the answer's [SEP];
    answer: n = answer length + 1 (size of the [MASK]-copy block; 0 / ignored for discriminative rows)."""
    mode: np.ndarray
    length: np.ndarray
    answer: np.ndarray

    def __post_init__(self):
        self.mode, self.length, self.answer = (np.asarray(a, dtype=np.int32).reshape(-1) for a in (self.mode, self.length, self.answer))
        if not (self.mode.shape == self.length.shape == self.answer.shape):
            raise ValueError("DialogMaskSpec: mode / length / answer must have one entry per sequence")
        if ((self.mode != 0) & (self.answer >= self.length)).any() or (self.length < 1).any():
            raise ValueError("DialogMaskSpec: need 1 <= L and n < L for generative sequences")

    def __len__(self):
        return int(self.mode.shape[0])

    def valid_lengths(self, T: int) -> np.ndarray:
        """Rows that exist for the text stream: L (+ n copy rows in the generative regime), capped at T."""
        return np.minimum(self.length + np.where(self.mode != 0, self.answer, 0), T).astype(np.int64)

    def to_device(self, device):
        return tuple(torch.from_numpy(a).to(device) for a in (self.mode, self.length, self.answer))

    # -- dense equivalents (tests, and callers that still want the reference's tensors) ------------
    def dense(self, T: int):
        """(txt_attention_mask [B, T, T], co_attention_mask [B, T]) as the reference's encoders emit them."""
        B = len(self)
        txt = np.zeros((B, T, T), dtype=bool)
        co = np.zeros((B, T), dtype=bool)
        ids = np.arange(T)
        for b, (m, L, n) in enumerate(zip(self.mode, self.length, self.answer)):
            if m == 0:
                txt[b, :L, :L] = True
                co[b, :L] = True
                continue
            c = L - n
            txt[b, 0, :L + n] = True
            txt[b, 1:c, 1:c] = True
            rows = np.arange(c, L)
            txt[b, c:L, 1:L] = ids[None, 1:L] <= rows[:, None]
            k = min(n, T - L)
            txt[b, L:L + k, 1:L] = ids[None, 1:L] < rows[:k, None]
            txt[b, np.arange(L, L + k), np.arange(L, L + k)] = True
            co[b, 1:c] = True
        return torch.from_numpy(txt), torch.from_numpy(co)


class DevicePrefetcher:
    """Host-resident batches -> device-resident batches, one batch ahead, on a copy stream of its own.

    The reference's callers hand CPU tensors to `forward` (train.py:113-129: the `.to(device)` lines are commented out; the
    DataParallel wrapper scatters them, utils/data_parallel.py:123-124) -- ~270 MB per 240 sequences in the reference's layout
    (int64 [256, 256] text masks, per-sequence copies of the region features), ~25 MB with compact inputs.  Passing CPU
    tensors to this build's `forward` works too, but then the copies sit in front of the step on the compute stream.  This
    iterator pins each host batch once, issues the host->device copies of batch i+1 on a side stream while step i computes,
    and makes the consumer's stream wait for exactly those copies (an event, no host synchronisation).

        for batch in DevicePrefetcher(loader, device):       # dict values: tensors, DialogMaskSpec, or anything else (passed through)
            loss = step(batch)
    """

    def __init__(self, batches, device, pin=True, cache_pinned=False, ring=3):
        """pin: stage pageable host tensors through pinned memory (asynchronous copies).  cache_pinned=False (default, what a
        real DataLoader needs: fresh tensors every step, buffers possibly reused): every batch is copied into one of `ring`
        reusable pinned staging sets (re-allocated only when a tensor's shape / dtype changes) -- pinned memory stays bounded by
        ring x one batch, nothing the caller hands over is kept alive, and a loader that mutates or reuses its host buffers is
        read afresh every time; a staging set is reused only after the host->device copies issued from it have completed
        (its event).  cache_pinned=True: the caller promises a CYCLED list of IMMUTABLE batches (bench.py --host-inputs
        prefetch): each distinct host tensor is pinned once and the pinned copy is re-sent."""
        self.it = iter(batches)
        self.device = torch.device(device)
        self.pin = pin
        self.cache_pinned = cache_pinned
        self.stream = torch.cuda.Stream(device=self.device)
        self._pinned = {}                        # cache_pinned: id(host tensor) -> (host tensor, pinned copy)
        self._ring = [dict(bufs={}, ev=None) for _ in range(max(2, int(ring)))]   # staging sets: key -> pinned buffer; ev = last copy-out
        self._turn = 0
        self._next = self._issue()

    def _host(self, key, t, slot):
        if not self.pin or t.is_pinned():
            return t
        if self.cache_pinned:
            p = self._pinned.get(id(t))
            if p is None:
                p = self._pinned[id(t)] = (t, t.pin_memory())      # keep `t` alive: its id is the key
            return p[1]
        buf = slot["bufs"].get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = slot["bufs"][key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t)                               # host -> pinned staging (the slot's previous copies have completed: _issue waited)
        return buf

    def _issue(self):
        try:
            hb = next(self.it)
        except StopIteration:
            return None
        db = {}
        slot = self._ring[self._turn % len(self._ring)]
        self._turn += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()               # the copies that last read this staging set (ring - 1 batches ago) are done
        self.stream.wait_stream(torch.cuda.current_stream(self.device))      # buffers freed by the consumer may be reused here
        with torch.cuda.stream(self.stream):
            for k, v in hb.items():
                if torch.is_tensor(v) and not v.is_cuda:
                    db[k] = self._host(k, v, slot).to(self.device, non_blocking=True)
                else:
                    db[k] = v
            ev = torch.cuda.Event()
            ev.record(self.stream)
        slot["ev"] = ev
        return db, ev

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        db, ev = self._next
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for v in db.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)              # allocated on the copy stream, consumed on the compute stream
        self._next = self._issue()
        return db


class PackedMask:
    """A 0/1 mask that exists only as bit-packed words on the device: `words` int32 [..., ceil(T/32)] (the layout of
    unimm_mask_pack) standing for a dense mask of `shape` [..., T].  What `HostStager` hands the engine instead of the
    caller's dense CPU mask."""
    __slots__ = ("words", "shape")

    def __init__(self, words, shape):
        self.words, self.shape = words, tuple(shape)

    def dim(self):
        return len(self.shape)


class HostStager:
    """CPU tensors handed straight to `forward()` -> device tensors, without the caller adopting anything.

    The reference's scripts call `model.forward` with HOST tensors (train.py:113-161; the .to(device) lines are commented out
    and `DataParallelImbalance` scatters them, utils/data_parallel.py:123-124).  Taken literally -- `.to(device)` of pageable
    memory on the compute stream, int64 masks converted on one host thread -- that call ran 44 % slower than the step on
    resident inputs (3,398 against 5,927 sequences/s, profiles/r5z_bench_host_direct_dense.json).  This is the same staging
    `DevicePrefetcher` does, but INSIDE the call: every CPU tensor of the step is copied into a reusable PINNED staging set
    (a ring of `ring` sets, re-allocated only when a shape or dtype changes: pinned memory stays bounded by ring x one batch)
    and sent from there with asynchronous copies on a copy stream of the engine's own; dense attention masks are bit-packed on
    the host side of the copy (unimm_host_mask_pack: 8 KiB instead of 512 KiB per sequence on the wire) and reach the engine
    as `PackedMask`.  The compute stream waits for the copies through an event, never the host.  Why this is enough without a
    one-step-ahead prefetch: the host runs AHEAD of the GPU (it enqueues a 240-sequence step in ~15 of the step's 40 ms), so
    when forward(k) is called the GPU still has most of step k-1's backward in front of it; the host-side staging and the
    PCIe copies of step k happen under that backward, and the step's own host sync (the plan header) comes after them."""

    MASK_KEYS = ("attention_mask", "co_attention_mask", "image_attention_mask")
    # (Host-side cost of a 240-sequence batch in the reference's layout, 16-CPU share of the GPU box: 1.3 ms for the masks,
    # 2.5 ms for the copies into pinned memory; a process whose torch intra-op pool is larger than its cgroup CPU quota can
    # see these stall for a whole scheduling period when some parallel CPU op exhausts the quota: torch.set_num_threads().)

    def __init__(self, device, ring=3, pack_threads=0):
        self.device = torch.device(device)
        self.stream = None                       # the copy stream, created on first use (see `stage`: never under the step executor)
        self._ring = [dict(bufs={}, ev=None) for _ in range(max(2, int(ring)))]
        self._turn = 0
        self.pack_threads = pack_threads
        self.stats = dict(steps=0, bytes_h2d=0, realloc=0)

    def pinned_bytes(self):
        return sum(b.numel() * b.element_size() for s in self._ring for b in s["bufs"].values())

    def _buf(self, slot, key, shape, dtype):
        buf = slot["bufs"].get(key)
        if buf is None or buf.shape != tuple(shape) or buf.dtype != dtype:
            buf = slot["bufs"][key] = torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
            self.stats["realloc"] += 1
        return buf

    def stage(self, inp, keys, pack=None, own_stream=True):
        """Replaces, in place, every CPU tensor among inp[k] for k in keys by its device copy (dense masks among `pack`,
        default MASK_KEYS: PackedMask).  Returns True when something was staged.
        own_stream=False: the copies are issued on the CURRENT stream and no stream is created (A/B: under the step executor at 30
        sequences the copy stream is worth 3 %: 9.5 against 9.8 ms per step)."""
        pack = self.MASK_KEYS if pack is None else pack
        from . import lib as L
        todo = [k for k in keys if torch.is_tensor(inp.get(k)) and not inp[k].is_cuda]
        if not todo:
            return False
        import time
        t0 = time.perf_counter()
        slot = self._ring[self._turn % len(self._ring)]
        self._turn += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()               # the copies that last read this staging set (ring - 1 steps ago) are done
        t1 = time.perf_counter()
        self.stats["wait_ms"] = self.stats.get("wait_ms", 0.0) + (t1 - t0) * 1e3
        cur = torch.cuda.current_stream(self.device)
        if own_stream and self.stream is None:
            self.stream = torch.cuda.Stream(device=self.device)
        cstream = self.stream if own_stream else cur
        # NO wait on the compute stream here: the copies of step k+1 must run while step k still computes (a
        # `stream.wait_stream(cur)` would queue them behind the whole backward pass: +5 ms per 240-sequence step).  The device
        # tensors come from the caching allocator on the copy stream and are handed to the consumers with record_stream(), so a
        # block the consumer has freed is only handed out again after the consumer's work on it has completed.
        staged = {}
        # small tensors and masks first (the text stream's first kernels need them), the region features and targets last
        todo.sort(key=lambda k: inp[k].numel() * inp[k].element_size() if k not in pack else 0)
        with torch.cuda.stream(cstream):
            for k in todo:
                v = inp[k]
                if k in pack and v.dim() in (2, 3):
                    if v.dtype not in (torch.bool, torch.uint8, torch.int32, torch.int64, torch.float32):
                        v = v.float()
                    nw = (v.shape[-1] + 31) // 32
                    words = self._buf(slot, k, v.shape[:-1] + (nw,), torch.int32)
                    tp = time.perf_counter()
                    L.host_mask_pack(v, out=words, threads=self.pack_threads)
                    self.stats["pack_ms"] = self.stats.get("pack_ms", 0.0) + (time.perf_counter() - tp) * 1e3
                    staged[k] = PackedMask(words.to(self.device, non_blocking=True), v.shape)
                    self.stats["bytes_h2d"] += words.numel() * 4
                    continue
                if v.is_pinned():
                    src = v
                else:
                    src = self._buf(slot, k, v.shape, v.dtype)
                    tp = time.perf_counter()
                    L.host_copy(src, v)            # pageable -> pinned staging (a threaded memcpy of its own: independent of torch's pool size)
                    self.stats["copy_ms"] = self.stats.get("copy_ms", 0.0) + (time.perf_counter() - tp) * 1e3
                staged[k] = src.to(self.device, non_blocking=True)
                self.stats["bytes_h2d"] += v.numel() * v.element_size()
            ev = torch.cuda.Event()
            ev.record(cstream)
        slot["ev"] = ev
        if cstream is not cur:
            cur.wait_event(ev)
        for k, v in staged.items():
            t = v.words if isinstance(v, PackedMask) else v
            if cstream is not cur:
                t.record_stream(cur)              # allocated on the copy stream, consumed on the compute stream(s)
            inp[k] = v
        self.stats["steps"] += 1
        self.stats["host_ms"] = self.stats.get("host_ms", 0.0) + (time.perf_counter() - t0) * 1e3
        return True
