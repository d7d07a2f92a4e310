"""GPU parity of the fused AdamW step (through the C ABI) against the oracle restatement, and of the
optimizer wrapper on the small model: parameters, moments, the bf16 / transposed weight copies the engine
computes with, the checkpoint layout."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_adamw_step_matches_oracle_over_several_steps():
    from oracle import adamw_ref as AR
    from unimm_amd import lib
    rng = np.random.default_rng(11)
    n = 64 * 4000
    group_h = rng.integers(0, 4, size=n // 64).astype(np.uint8)
    group_h[rng.random(n // 64) < 0.05] = lib.ADAMW_SKIP
    lrs, wds = [2e-5, 2e-5, 1e-4, 1e-4], [0.01, 0.0, 0.01, 0.0]
    p_h = rng.standard_normal(n).astype(np.float32)
    m_h, v_h = np.zeros(n, np.float32), np.zeros(n, np.float32)
    p, m, v = (torch.from_numpy(a.copy()).to(DEV) for a in (p_h, m_h, v_h))
    w16 = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    group = torch.from_numpy(group_h).to(DEV)
    elem_group = np.repeat(group_h, 64)
    for t in range(1, 6):
        g_h = (rng.standard_normal(n) * 10.0 ** rng.uniform(-4, 0, size=n)).astype(np.float32)
        scale = 0.25 if t == 3 else 1.0
        g = torch.from_numpy(g_h).to(DEV)
        lib.adamw_step(p, g, m, v, group, [l * (1 + 0.1 * t) for l in lrs], wds, t, w16=w16, grad_scale=scale, zero_grad=(t == 5))
        for gi in range(4):
            sel = elem_group == gi
            pp, mm, vv = p_h[sel], m_h[sel], v_h[sel]
            AR.adamw_step(pp, g_h[sel] * np.float32(scale), mm, vv, lrs[gi] * (1 + 0.1 * t), wds[gi], t)
            p_h[sel], m_h[sel], v_h[sel] = pp, mm, vv
        torch.cuda.synchronize()
        # fp32, one rounding per operation on both sides: the moments are bit-exact, the parameter within an
        # ulp or two (sqrt / divide roundings), accumulated over the steps
        assert np.array_equal(m.cpu().numpy(), m_h)
        assert np.array_equal(v.cpu().numpy(), v_h)
        assert np.allclose(p.cpu().numpy(), p_h, rtol=1e-6, atol=1e-9)
        upd = elem_group != lib.ADAMW_SKIP
        assert torch.equal(w16.cpu()[torch.from_numpy(upd)], p.cpu().to(torch.bfloat16)[torch.from_numpy(upd)])
        if t == 5:
            assert float(g[torch.from_numpy(upd).to(DEV)].abs().max()) == 0.0
    skipped = elem_group == lib.ADAMW_SKIP
    assert skipped.any() and np.array_equal(p.cpu().numpy()[skipped], rng_p0(skipped))


def rng_p0(mask):
    rng = np.random.default_rng(11)
    n = 64 * 4000
    rng.integers(0, 4, size=n // 64); rng.random(n // 64)
    return rng.standard_normal(n).astype(np.float32)[mask]


def test_adamw_rejects_bad_arguments():
    from unimm_amd import lib
    p = torch.zeros(100, device=DEV)            # not a multiple of 64
    gr = torch.zeros(2, dtype=torch.uint8, device=DEV)
    with pytest.raises(lib.UnimmHipError):
        lib.adamw_step(p, p, p, p, gr, [1e-3], [0.0], 1)
    p = torch.zeros(128, device=DEV)
    with pytest.raises(lib.UnimmHipError):
        lib.adamw_step(p, p, p, p, gr, [1e-3] * 9, [0.0] * 9, 1)
    with pytest.raises(lib.UnimmHipError):
        lib.adamw_step(p, p, p, p, gr, [1e-3], [0.0], 0)       # step counts from 1


def test_fused_adamw_on_the_small_model(golden_dir):
    """train.py's optimizer setup on the small config: grouping, schedule, two steps.  Parameters and moments
    must equal the oracle applied tensor by tensor to the gradients the engine produced, the engine must
    compute the next step with the UPDATED weights (bf16 and transposed copies), and state_dict round-trips."""
    from oracle import adamw_ref as AR
    from tests.test_gpu_model import build_small, kwargs_from
    from unimm_amd import params as P
    from unimm_amd.optim import FusedAdamW, WarmupLinearScheduleNonZero, reference_param_groups
    model, _, _ = build_small(golden_dir)
    model.eval()
    g = np.load(os.path.join(golden_dir, "small_mixed.npz"))
    args, kw = kwargs_from(g)
    names = [n for n, _ in model.named_parameters()]
    lang = [n for i, n in enumerate(names) if i % 3 != 0]
    groups = reference_param_groups(model, lr=2e-3, image_lr=1e-2, language_weights=lang)
    opt = FusedAdamW(groups, model.engine, lr=2e-3)
    sch = WarmupLinearScheduleNonZero(opt, warmup_steps=2, t_total=10, min_lr=1e-5)
    ref = {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    mom = {n: (np.zeros_like(v), np.zeros_like(v)) for n, v in ref.items()}
    p0 = {n: v.copy() for n, v in ref.items()}
    losses = []
    for t in (1, 2, 3):                 # scheduler steps 0 (factor 0 -> min_lr floor), 1, 2 (= warm-up end)
        opt.zero_grad()
        lm, img, nsp_l, _, _, _ = model(*args, **kw, _want_lm_scores=False)
        loss = (lm + img + nsp_l).sum()
        loss.backward()
        losses.append(float(loss.detach()))
        grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters() if p.grad is not None}
        lrs = [gr["lr"] for gr in opt.param_groups]
        opt.step(); sch.step()
        for n, gr, lr in zip(names, opt.param_groups, lrs):
            if P.is_unused(n):
                continue
            AR.adamw_step(ref[n], grads[n], *mom[n], lr, gr["weight_decay"], t)
    torch.cuda.synchronize()
    for n, p in model.named_parameters():
        got = p.detach().cpu().numpy()
        if P.is_unused(n):
            assert np.array_equal(got, p0[n])                          # never touched, not even by the decay
        else:
            assert np.allclose(got, ref[n], rtol=1e-5, atol=1e-7), n
    assert losses[2] < losses[1]                                        # and the steps went downhill
    # the engine computes with the updated copies: a fresh model loaded with the updated weights gives the same loss
    lm, img, nsp_l, _, _, _ = model(*args, **kw, _want_lm_scores=False)
    fresh, _, _ = build_small(golden_dir)
    fresh.load_state_dict(model.state_dict())
    fresh.eval()
    lm2, img2, nsp2, _, _, _ = fresh(*args, **kw, _want_lm_scores=False)
    assert abs(float((lm + img + nsp_l).sum()) - float((lm2 + img2 + nsp2).sum())) <= 1e-3 * abs(float((lm2 + img2 + nsp2).sum()))
    # checkpoint layout of torch.optim (train.py:505) and a round trip through it
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["param_groups"]) == len(names)
    k0 = next(iter(sd["state"]))
    assert set(sd["state"][k0]) == {"step", "exp_avg", "exp_avg_sq"} and sd["state"][k0]["step"] == 3
    opt2 = FusedAdamW(reference_param_groups(fresh, lr=2e-3, image_lr=1e-2, language_weights=lang), fresh.engine, lr=2e-3)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 3 and torch.equal(opt2.exp_avg_sq.cpu(), opt.exp_avg_sq.cpu())
