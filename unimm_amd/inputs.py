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
