"""Training-loop shell around the hot path: what train.py:350-507 does between the
dataloader and the checkpoint file, without the data plane.

* `train_step`       - one iteration of the loop body (train.py:411-463): forward through the harness, the
                       reference's loss combination, `/ batch_multiply`, backward, optimizer step + zero_grad
                       on accumulation boundaries only, scheduler step every iteration.  bf16 needs no
                       GradScaler; with the data-parallel wrapper the gradient exchange is skipped on
                       non-boundary micro-steps (`no_sync`).
* `save_checkpoint`  - the dict of train.py:503-505 (`model_state_dict` with the `bert_pretrained.` prefix,
                       `scheduler_state_dict`, `optimizer_state_dict`, `iter_id`).
* `load_checkpoint`  - warm start by key intersection (train.py:352-364) or `-continue` (train.py:366-389).
* `expand_image_fields` - train.py:413-432 (one image's features repeated per round and per sample)."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import harness


def expand_image_fields(batch: dict) -> dict:
    """[n_img, ...] image tensors -> [n_img, rounds, samples, ...] as `forward` flattens them (train.py:413-432)."""
    rounds, samples = batch["tokens"].shape[1], batch["tokens"].shape[2]
    out = dict(batch)
    for k in ("image_feat", "image_loc", "image_target", "image_label", "image_mask"):
        v = batch[k]
        out[k] = v.unsqueeze(1).unsqueeze(1).expand(v.shape[0], rounds, samples, *v.shape[1:]).contiguous()
    return out


def train_step(dialog_encoder, optimizer, scheduler, batch, params, iter_id, sample_size=None):
    """One loop iteration (iter_id is the 1-based counter of train.py:411).  Returns (loss, lm, nsp, img) floats."""
    bm = int(params.get("batch_multiply", 1))
    boundary = iter_id % bm == 0
    dialog_encoder.train()
    sync_ctx = dialog_encoder.no_sync() if (hasattr(dialog_encoder, "no_sync") and not boundary) else _null()
    with sync_ctx:
        loss, lm_loss, nsp_loss, img_loss = harness.forward(dialog_encoder, batch, params, sample_size=sample_size)
        (loss / bm).backward()
    if boundary:
        if hasattr(dialog_encoder, "sync_gradients"):
            dialog_encoder.sync_gradients()           # no-op when every bucket was reduced inside backward
        optimizer.step()
        optimizer.zero_grad()
    scheduler.step()
    return float(loss.detach()), lm_loss, nsp_loss, img_loss


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _unwrap(m):
    return m.module if hasattr(m, "module") else m


def save_checkpoint(path, dialog_encoder, optimizer, scheduler, iter_id):
    torch.save({"model_state_dict": _unwrap(dialog_encoder).state_dict(), "scheduler_state_dict": scheduler.state_dict(),
                "optimizer_state_dict": optimizer.state_dict(), "iter_id": iter_id}, path)
    return path


def load_checkpoint(path_or_dict, dialog_encoder, optimizer=None, scheduler=None, resume=False):
    """resume=False: copy every tensor whose key exists in the model (accepts a bare state_dict or the
    training dict); returns the number of keys transferred.  resume=True: also restore optimizer and
    scheduler state and return the saved iter_id."""
    ck = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
    model = _unwrap(dialog_encoder)
    model_dict = model.state_dict()
    if not resume:
        sd = ck["model_state_dict"] if "model_state_dict" in ck else ck
        sd = {k: v for k, v in sd.items() if k in model_dict}
        if not sd:
            raise ValueError("load_checkpoint: no key of the checkpoint exists in the model")
        model_dict.update(sd)
        model.load_state_dict(model_dict)
        return len(sd)
    model_dict.update({k: v for k, v in ck["model_state_dict"].items() if k in model_dict})
    model.load_state_dict(model_dict)
    if optimizer is not None:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    if scheduler is not None:
        scheduler.load_state_dict(ck["scheduler_state_dict"])
    return ck["iter_id"]
