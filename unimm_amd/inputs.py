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
