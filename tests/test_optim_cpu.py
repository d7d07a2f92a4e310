"""CPU tests of the optimizer row: the schedule against the golden learning rates taken from
the reference's own class, the parameter grouping of train.py:322-345, and the oracle's AdamW restatement
against an independent float64 evaluation of the published formula."""
import math
import os

import numpy as np
import torch

from oracle import adamw_ref as AR


def test_schedule_oracle_and_port_match_reference_goldens(golden_dir):
    from unimm_amd.optim import WarmupLinearScheduleNonZero
    g = np.load(os.path.join(golden_dir, "sched.npz"))
    for tag in ("train", "short"):
        w, t, mn, *bases = g[tag + "_cfg"].tolist()
        steps, lrs = g[tag + "_steps"], g[tag + "_lrs"]
        for s, row in zip(steps, lrs):
            for b, want in zip(bases, row):
                assert AR.warmup_linear_nonzero(int(s), b, int(w), int(t), mn) == want      # bit-exact (float64)
        ps = [torch.nn.Parameter(torch.zeros(1)) for _ in bases]
        opt = torch.optim.SGD([{"params": [p], "lr": b} for p, b in zip(ps, bases)], lr=bases[0])
        sch = WarmupLinearScheduleNonZero(opt, warmup_steps=int(w), t_total=int(t), min_lr=mn)
        k = 0
        for s in range(int(steps.max()) + 1):
            if s == steps[k]:
                assert [gr["lr"] for gr in opt.param_groups] == lrs[k].tolist()
                k += 1
                if k == len(steps):
                    break
            opt.step()
            sch.step()


def test_reference_param_groups_follow_train_py():
    from unimm_amd.optim import reference_param_groups

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dense = torch.nn.Linear(4, 4)
            self.LayerNorm = torch.nn.LayerNorm(4)
            self.v_dense = torch.nn.Linear(4, 4)
            self.frozen = torch.nn.Parameter(torch.zeros(2), requires_grad=False)

    groups = reference_param_groups(M(), lr=2e-5, image_lr=1e-4, language_weights=["dense.weight", "dense.bias", "LayerNorm.weight"])
    got = {tuple(g["params"][0].shape) + (g["lr"], g["weight_decay"]) for g in groups}
    assert len(groups) == 6                                          # one group per trainable parameter
    by = {n: g for (n, _), g in zip([(n, p) for n, p in M().named_parameters() if p.requires_grad], groups)}
    assert by["dense.weight"]["lr"] == 2e-5 and by["dense.weight"]["weight_decay"] == 0.01
    assert by["dense.bias"]["lr"] == 2e-5 and by["dense.bias"]["weight_decay"] == 0.0
    assert by["LayerNorm.weight"]["weight_decay"] == 0.0 and by["LayerNorm.bias"]["lr"] == 1e-4
    assert by["v_dense.weight"]["lr"] == 1e-4 and by["v_dense.weight"]["weight_decay"] == 0.01
    assert got


def test_oracle_adamw_matches_float64_formula():
    rng = np.random.default_rng(3)
    p = rng.standard_normal(1000).astype(np.float32)
    m = np.zeros_like(p); v = np.zeros_like(p)
    p64, m64, v64 = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    lr, wd, b1, b2, eps = 3e-4, 0.01, 0.9, 0.999, 1e-6
    for t in range(1, 8):
        g = rng.standard_normal(1000).astype(np.float32)
        AR.adamw_step(p, g, m, v, lr, wd, t)
        m64 = b1 * m64 + (1 - b1) * g
        v64 = b2 * v64 + (1 - b2) * g.astype(np.float64) ** 2
        p64 = p64 - lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m64 / (np.sqrt(v64) + eps)
        p64 = p64 - lr * wd * p64
    assert np.abs(p - p64).max() < 5e-6 and np.abs(v - v64).max() < 1e-6
