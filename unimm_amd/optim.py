"""Optimizer step of the hot path's caller: fused AdamW over the flat arenas and the
reference's learning-rate schedule.

`FusedAdamW` takes what train.py:329-347 builds -- a list of per-parameter groups
`{"params": [p], "lr": lr or image_lr, "weight_decay": 0.01 or 0}` -- and keeps the
`torch.optim.Optimizer` surface the script touches: `param_groups` (the scheduler writes `group["lr"]`),
`step()`, `zero_grad()`, `state_dict()` / `load_state_dict()` in torch's layout (`exp_avg`, `exp_avg_sq`,
`step` per parameter index, train.py:369-385, :505).  The update itself is ONE kernel launch over the
parameter arena (`unimm_adamw_step`), which also rewrites the bf16 GEMM-operand copy of the weights; the
per-tensor groups collapse to the distinct (lr, weight_decay) pairs of the step (4 in the reference setup).

`WarmupLinearScheduleNonZero` restates utils/optim_utils.py:8-26 (pinned by tests/golden/sched.npz)."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
from torch.optim.lr_scheduler import _LRScheduler

from . import lib as L
from . import params as P
from .arena import ALIGN


class WarmupLinearScheduleNonZero(_LRScheduler):
    """Linear warm-up from 0 over `warmup_steps`, then linear decay to 0 at `t_total`, floored at `min_lr`
    (utils/optim_utils.py:8-26; train.py:348 uses warmup_steps=10000, t_total=200000)."""

    def __init__(self, optimizer, warmup_steps, t_total, min_lr=1e-5, last_epoch=-1):
        self.warmup_steps, self.t_total, self.min_lr = warmup_steps, t_total, min_lr
        super().__init__(optimizer, last_epoch=last_epoch)

    def get_lr(self):
        step = self.last_epoch
        if step < self.warmup_steps:
            f = float(step) / float(max(1, self.warmup_steps))
        else:
            f = max(0, float(self.t_total - step) / float(max(1.0, self.t_total - self.warmup_steps)))
        return [b * f if b * f > self.min_lr else self.min_lr for b in self.base_lrs]


def default_language_weights(model):
    """The entries of the reference's config/language_weights.json that exist in this model: the parameters
    inherited from BERT (text embeddings, the 12 text layers, the MLM head's bias and transform).  The json
    also lists `bert.pooler.*`, `cls.seq_relationship.*` and `inconsistency_head.*`, which this model does not
    have (its pooler is `t_pooler`, its NSP head `bi_seq_relationship`), so those train at image_lr.  Pass the
    json's own list to `reference_param_groups` when it is available."""
    keep = ("bert.embeddings.", "bert.encoder.layer.", "cls.predictions.bias", "cls.predictions.transform.")
    return [n for n, _ in model.named_parameters() if any(k in n for k in keep)]


def reference_param_groups(model, lr, image_lr, language_weights, weight_decay=0.01):
    """The grouping of train.py:322-345: one group per parameter, lr by membership in
    config/language_weights.json, no decay for bias / LayerNorm parameters."""
    no_decay = ("bias", "LayerNorm.bias", "LayerNorm.weight")
    language_weights = set(language_weights)
    groups = []
    for key, value in dict(model.named_parameters()).items():
        if not value.requires_grad:
            continue
        groups.append({"params": [value], "lr": lr if key in language_weights else image_lr,
                       "weight_decay": 0.0 if any(nd in key for nd in no_decay) else weight_decay})
    return groups


class FusedAdamW(torch.optim.Optimizer):
    """AdamW of `pytorch_transformers` (betas (0.9, 0.999), eps 1e-6, correct_bias True, decay after the
    update) as one fused launch.  `engine`: the unimm_amd.engine.Engine whose arena holds the parameters."""

    def __init__(self, params, engine, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        params = list(params)
        first = params[0]["params"] if isinstance(params[0], dict) else params
        first = first[0] if isinstance(first, (list, tuple)) else next(iter(first))
        self.engine = engine
        engine.ensure(first.device)         # (re)builds the arena if the model was moved since
        self.arena = A = engine.arena
        engine.register_arena_user(self)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        # every parameter -> its chunk range in the arena
        by_ptr = {A.flat.data_ptr() + 4 * o: (name, o, shape) for name, (o, shape) in A.offsets.items()}
        self._ranges = []           # per param_group: list of (chunk_lo, chunk_hi, unused)
        seen = set()
        for g in self.param_groups:
            rs = []
            for p in g["params"]:
                hit = by_ptr.get(p.data_ptr())
                if hit is None or p.device != A.flat.device:
                    raise ValueError("FusedAdamW: parameter does not live in the engine's arena")
                name, o, _ = hit
                if name in seen:
                    raise ValueError(f"FusedAdamW: parameter {name} appears in two groups")
                seen.add(name)
                rs.append((o // ALIGN, (o + p.numel() + ALIGN - 1) // ALIGN, P.is_unused(name), name))
            self._ranges.append(rs)
        self.exp_avg = torch.zeros_like(A.flat)
        self.exp_avg_sq = torch.zeros_like(A.flat)
        self._group_host = np.full(A.numel // ALIGN, L.ADAMW_SKIP, dtype=np.uint8)
        self._group_dev = torch.from_numpy(self._group_host.copy()).to(A.flat.device)
        self._combo_key = None
        self.step_count = 0
        self.grad_scale = 1.0       # set to 1 / batch_multiply (or 1 / loss scale) by the caller when needed

    # -- torch.optim.Optimizer surface ------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False):
        """One memset of the gradient arena (the parameters' .grad stay attached views of it)."""
        self.arena.zero_grads()

    def _combos(self):
        combos: Dict[tuple, int] = {}
        ids = []
        for g in self.param_groups:
            k = (float(g["lr"]), float(g["weight_decay"]))
            ids.append(combos.setdefault(k, len(combos)))
        if len(combos) > L.ADAMW_MAX_GROUPS:
            raise ValueError(f"FusedAdamW: {len(combos)} distinct (lr, weight_decay) pairs in one step "
                             f"(the kernel takes {L.ADAMW_MAX_GROUPS})")
        return combos, ids

    @torch.no_grad()
    def step(self, closure=None):
        combos, ids = self._combos()
        # the chunk -> group table only changes when the PARTITION changes, not when the lr values do
        part = tuple(ids)
        if part != self._combo_key:
            self._group_host.fill(L.ADAMW_SKIP)
            for gi, rs in zip(ids, self._ranges):
                for lo, hi, unused, _ in rs:
                    if not unused:
                        self._group_host[lo:hi] = gi
            self._group_dev.copy_(torch.from_numpy(self._group_host), non_blocking=False)
            self._combo_key = part
        self.step_count += 1
        b1, b2 = self.defaults["betas"]
        A = self.arena
        L.adamw_step(A.flat, A.grad_flat, self.exp_avg, self.exp_avg_sq, self._group_dev,
                     [k[0] for k in combos], [k[1] for k in combos], self.step_count, beta1=b1, beta2=b2,
                     eps=self.defaults["eps"], w16=self.engine.w16, grad_scale=self.grad_scale,
                     correct_bias=self.defaults["correct_bias"])
        self.engine.refresh_weights(force=True, cast=False)     # transposed copies; the cast is already done

    # -- checkpoint format of torch.optim (train.py:369-385, :505) ---------------------------------
    def state_dict(self):
        state, groups, idx = {}, [], 0
        for g, rs in zip(self.param_groups, self._ranges):
            ids = []
            for p, (lo, hi, unused, name) in zip(g["params"], rs):
                if self.step_count > 0 and not unused:
                    o, shape = self.arena.offsets[name]
                    state[idx] = {"step": self.step_count,
                                  "exp_avg": self.exp_avg[o:o + p.numel()].view(shape).clone(),
                                  "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view(shape).clone()}
                ids.append(idx)
                idx += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        flat = [(p, r) for g, rs in zip(self.param_groups, self._ranges) for p, r in zip(g["params"], rs)]
        steps = set()
        for idx, st in sd["state"].items():
            p, (lo, hi, unused, name) = flat[int(idx)]
            o, shape = self.arena.offsets[name]
            self.exp_avg[o:o + p.numel()].view(shape).copy_(st["exp_avg"])
            self.exp_avg_sq[o:o + p.numel()].view(shape).copy_(st["exp_avg_sq"])
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("FusedAdamW: per-parameter step counts differ; the fused step keeps one")
        self.step_count = steps.pop() if steps else 0
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            for k, v in saved.items():
                if k != "params":
                    g[k] = v
        self._combo_key = None
