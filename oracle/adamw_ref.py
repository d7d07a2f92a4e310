"""TEST INFRASTRUCTURE (oracle): CPU restatement of the optimizer step of the reference's training loop.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

* `adamw_step` restates `pytorch_transformers.optimization.AdamW.step` as train.py:347 uses it (defaults
  betas (0.9, 0.999), eps 1e-6, weight_decay from the group, correct_bias True).  That library is an
  un-vendored dependency of the reference (imported at train.py:20, no version pinned anywhere in the
  repository; its last release is 1.2.0) and is NOT in this image, so this half is **parity unpinned**:
  it follows the published algorithm of that class --
      exp_avg.mul_(b1).add_(1 - b1, grad); exp_avg_sq.mul_(b2).addcmul_(1 - b2, grad, grad)
      denom = exp_avg_sq.sqrt().add_(eps)
      step_size = lr * sqrt(1 - b2 ** t) / (1 - b1 ** t)        (correct_bias)
      p.addcdiv_(-step_size, exp_avg, denom)
      if weight_decay > 0: p.add_(-lr * weight_decay, p)         (after the update, on the updated p)
  -- and is anchored on the reference's call sites (train.py:322-347 grouping, :458 step, :460 zero_grad).
* `warmup_linear_nonzero` restates utils/optim_utils.py:8-26 and is PINNED by tests/golden/sched.npz,
  generated from the reference's own class (oracle/make_goldens.py, group "sched")."""
from __future__ import annotations

import math

import numpy as np


def adamw_step(p, g, m, v, lr, weight_decay, step, beta1=0.9, beta2=0.999, eps=1e-6, correct_bias=True):
    """In-place fp32 numpy update of one parameter tensor; returns nothing."""
    f = np.float32
    m *= f(beta1); m += f(1.0 - beta1) * g
    v *= f(beta2); v += f(1.0 - beta2) * g * g
    denom = np.sqrt(v) + f(eps)
    step_size = lr
    if correct_bias:
        step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p += f(-step_size) * (m / denom)
    if weight_decay > 0.0:
        p += f(-lr * weight_decay) * p


def warmup_linear_nonzero(step, base_lr, warmup_steps, t_total, min_lr=1e-5):
    """Learning rate at scheduler step `step` (utils/optim_utils.py:18-26)."""
    if step < warmup_steps:
        f = float(step) / float(max(1, warmup_steps))
    else:
        f = max(0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))
    return base_lr * f if base_lr * f > min_lr else min_lr
