"""GPU parity of the fp32-accuracy mode (compute_dtype="fp32x3": unimm_amd/engine_x3.py, csrc/x3ops.hip) behind the drop-in
API, against the goldens produced by the REFERENCE's own modules (fp32 CPU) and against the CPU oracle.

This is north_star's fp32 gate: outputs within 1e-3 in allclose form, |got - want| <= 1e-3 + 1e-3 |want| -- NSP logits, MLM
logits, hidden states, losses, candidate log-likelihoods (and identical ranks); gradients within 1e-3 of each tensor's scale.
The reference runs this arithmetic in dense_annotation_finetuning.py:253 (no autocast)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_PATH = os.path.join(ROOT, "unimm_amd", "config", "bert_base_6layer_6conect.json")


def T_(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def close(got, want, tol=TOL, what=""):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    want = want.astype(np.float64)
    diff = np.abs(got.astype(np.float64) - want)
    ratio = float((diff / (tol + tol * np.abs(want))).max())
    print(f"  {what}: max |err| {diff.max():.3e} on scale {np.abs(want).max():.3g}; worst |err| / ({tol:g} + {tol:g} |want|) = {ratio:.3f}")
    assert ratio <= 1.0, f"{what}: max |err| {diff.max():.4g} (tol {tol}, worst ratio {ratio:.2f})"
    return ratio


def build_small(golden_dir):
    from oracle import vilbert_ref as R
    from unimm_amd import BertConfig, BertForMultiModalPreTraining
    cfgd = json.load(open(os.path.join(golden_dir, "small_config.json")))
    model = BertForMultiModalPreTraining(BertConfig.from_dict(cfgd), compute_dtype="fp32x3")
    ocfg = R.make_config(cfgd)
    sd = R.init_state_dict(ocfg, seed=11)
    model.load_state_dict(sd, strict=True)
    return model.cuda(), ocfg, sd


def kwargs_from(g, train=True, use_lm_weight=True, device=None):
    i = lambda k: (T_(g["in::" + k]).to(device) if device else T_(g["in::" + k]))
    kw = dict(token_type_ids=i("token_type_ids"), position_ids=i("position_ids"), attention_mask=i("attention_mask"),
              image_attention_mask=i("image_attention_mask"), co_attention_mask=i("co_attention_mask"))
    if train:
        kw.update(masked_lm_labels=i("masked_lm_labels"), image_label=i("image_label"), image_target=i("image_target"),
                  next_sentence_label=i("next_sentence_label"), nsp_weight=i("nsp_weight"),
                  lm_weight=i("lm_weight") if use_lm_weight else None)
    return (i("input_ids"), i("image_feat"), i("image_loc")), kw


@pytest.fixture(scope="module")
def small(golden_dir):
    return build_small(golden_dir)


@pytest.mark.parametrize("case", ["dis", "genpos", "genneg", "mixed"])
def test_small_config_matches_reference_golden_fp32(golden_dir, small, case):
    model, ocfg, _ = small
    assert type(model.engine).__name__ == "EngineX3"
    model.eval()
    g = np.load(os.path.join(golden_dir, f"small_{case}.npz"))
    args, kw = kwargs_from(g)
    print()
    with torch.no_grad():
        lm, img, nsp_l, seq_t, pred_t, nsp = model(*args, **kw)
    close(lm, g["lm_loss"], what="lm_loss"); close(img, g["img_loss"], what="img_loss"); close(nsp_l, g["nsp_loss"], what="nsp_loss")
    close(nsp, g["nsp"], what="nsp scores")
    V = ocfg.vocab_size
    close(pred_t.reshape(-1, V)[T_(g["pred_rows"]).cuda()], g["pred_t_rows"], what="pred_t")
    valid = g["in::attention_mask"].reshape(-1, g["in::attention_mask"].shape[-1]).any(-1) if g["in::attention_mask"].ndim == 3 else None
    st = seq_t.reshape(-1, seq_t.shape[-1]).cpu().numpy()
    want = g["seq_out_t"].reshape(st.shape)
    close(st[valid], want[valid], what="seq_out_t")
    assert np.isfinite(st).all()
    args, kw = kwargs_from(g, train=False, device="cuda")
    with torch.no_grad():
        p_t, p_v, nsp2, _, _ = model(*args, **kw)
    close(p_v, g["inf_pred_v"], what="pred_v"); close(nsp2, g["inf_nsp"], what="inference nsp")
    args, kw = kwargs_from(g, use_lm_weight=False)
    with torch.no_grad():
        lm_ce = model(*args, **kw, _want_lm_scores=False)[0]
    close(lm_ce, g["lm_loss_ce"], what="CE fallback")


def test_small_config_gradients_match_reference_golden_fp32(golden_dir, small):
    model, _, _ = small
    model.eval()
    g = np.load(os.path.join(golden_dir, "small_mixed.npz"))
    args, kw = kwargs_from(g)
    model.zero_grad(set_to_none=True)
    lm, img, nsp_l, _, _, _ = model(*args, **kw, _want_lm_scores=False)
    (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    params = dict(model.named_parameters())
    names, norms = [str(n) for n in g["grad_names"]], g["grad_norms"]
    worst_n = worst_e = 0.0
    for n, want in zip(names, norms):
        p = params[n]
        if want < 0:
            assert p.grad is None, f"{n} is never used by forward: grad must stay None"
            continue
        got = float(p.grad.double().norm())
        rel = abs(got - want) / max(want, 1e-4)
        worst_n = max(worst_n, rel)
        assert rel < 1e-3, (n, got, want)
    for k in g.files:
        if k.startswith("grad::"):
            want = g[k]
            got = params[k[6:]].grad.cpu().numpy()
            err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-5)
            worst_e = max(worst_e, err)
            assert err <= 1e-3, (k, err)
    want = g["grad_rows::word_embeddings"]
    got = params["bert.embeddings.word_embeddings.weight"].grad[:64].cpu().numpy()
    assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max()
    print(f"\nfp32x3 small-config gradients vs the reference: worst norm error {worst_n:.2e}, worst element error / scale {worst_e:.2e}")


def test_training_mode_dropout_matches_oracle_with_replayed_masks_fp32(golden_dir, small):
    """Train mode: the oracle re-plays the kernels' counter-based dropout masks site by site (same sites and counters as the
    bf16 engine)."""
    import zlib
    from oracle import vilbert_ref as R
    from unimm_amd import dropout as DR
    model, ocfg, sd = small
    g = np.load(os.path.join(golden_dir, "small_mixed.npz"))
    args, kw = kwargs_from(g)
    model.train()
    model.set_dropout_seed(77, step=4)
    model.zero_grad(set_to_none=True)
    lm, img, nsp_l, _, _, nsp = model(*args, **kw, _want_lm_scores=False)
    plan = model.engine.last_plan
    (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    assert plan is not None and plan["Mv"] < g["in::input_ids"].size
    rows = plan["rows"].cpu()

    def drop_fn(site, x, p):
        key = DR.make_key(77, 5, zlib.crc32(site.encode()) & 0xFFFFFFFF)
        _, thr, scale = DR.drop_arg(p, key)
        text_rowwise = site == "emb_t" or (site.startswith("bert.encoder.layer.") and site.endswith((".so", ".out"))) \
            or site.endswith((".bo2", ".tout"))
        if text_rowwise:
            n = x.shape[-1]
            keep_p = torch.from_numpy(DR.keep_mask2d(key, thr, rows.numel(), n))
            keep = torch.ones((x.shape[0] * x.shape[1], n), dtype=torch.bool)
            keep[rows] = keep_p
            keep = keep.view(x.shape)
        else:
            keep = torch.from_numpy(DR.keep_mask_nd(key, thr, tuple(x.shape)))
        return x * keep * scale

    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != R.TIED[0]}
    leaves[R.TIED[0]] = leaves[R.TIED[1]]
    out = R.forward(leaves, ocfg, *args, **kw, drop_fn=drop_fn)
    (out["lm_loss"] + out["img_loss"] + out["nsp_loss"]).sum().backward()
    print()
    close(lm, out["lm_loss"].detach(), what="lm_loss (train)")
    close(img, out["img_loss"].detach(), what="img_loss (train)")
    close(nsp_l, out["nsp_loss"].detach(), what="nsp_loss (train)")
    close(nsp, out["nsp"].detach(), what="nsp (train)")
    worst = 0.0
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        want = leaves[n].grad
        err = float((p.grad.cpu() - want).abs().max() / want.abs().max().clamp_min(1e-6))
        worst = max(worst, err)
        assert err <= 2e-3, (n, err)
    print(f"  train-mode gradients vs the oracle with replayed masks: worst element error / scale {worst:.2e}")
    model.eval()


def test_unpadded_equals_padded_fp32(golden_dir, small):
    model, _, _ = small
    model.eval()
    eng = model.engine
    g = np.load(os.path.join(golden_dir, "small_mixed.npz"))
    args, kw = kwargs_from(g)
    res = {}
    try:
        for unpad in (True, False):
            eng.unpad = unpad
            model.zero_grad(set_to_none=True)
            lm, img, nsp_l, _, _, nsp = model(*args, **kw, _want_lm_scores=False)
            (lm + img + nsp_l).sum().backward()
            torch.cuda.synchronize()
            res[unpad] = (torch.stack([lm, img, nsp_l]).flatten().detach().clone(), nsp.detach().clone(), eng.arena.grad_flat.clone())
    finally:
        eng.unpad = True
    a, b = res[True], res[False]
    assert (a[0] - b[0]).abs().max() <= 2e-6 and (a[1] - b[1]).abs().max() <= 2e-6
    d = float((a[2] - b[2]).abs().max() / a[2].abs().max())
    assert d <= 2e-5, d


def _build_full(seed):
    from oracle import vilbert_ref as R
    from unimm_amd import BertConfig, BertForMultiModalPreTraining
    model = BertForMultiModalPreTraining(BertConfig.from_json_file(CFG_PATH), compute_dtype="fp32x3")
    model.load_state_dict(R.init_state_dict(R.make_config(CFG_PATH), seed=seed), strict=True)
    return model.cuda().eval()


def _full_b6_call(g):
    i = lambda k: T_(g["in::" + k])
    n = g["in::input_ids"].shape[0]
    rep = lambda x: x.expand(n, *x.shape[1:])
    args = (i("input_ids"), rep(i("image_feat")), rep(i("image_loc")))
    kw = dict(token_type_ids=i("token_type_ids"), position_ids=i("position_ids"), attention_mask=i("attention_mask"),
              image_attention_mask=i("image_attention_mask"), co_attention_mask=i("co_attention_mask").expand(n, 37, 256),
              masked_lm_labels=i("masked_lm_labels"), image_label=i("image_label"), image_target=rep(i("image_target")),
              next_sentence_label=i("next_sentence_label"), nsp_weight=i("nsp_weight"), lm_weight=i("lm_weight"))
    return args, kw


@pytest.fixture(scope="module")
def full5():
    return _build_full(seed=5)


def test_full_config_b6_matches_reference_golden_fp32(golden_dir, full5):
    """BASELINE config 1 on the GPU in the fp32-accuracy mode: full model, 1 image x 6 sequences x 256 tokens x 37 regions,
    every output in allclose form at 1e-3 (north_star's fp32 tolerance)."""
    from oracle import vilbert_ref as R
    g = np.load(os.path.join(golden_dir, "full_b6.npz"))
    model = full5
    args, kw = _full_b6_call(g)
    with torch.no_grad():
        lm, img, nsp_l, seq_t, pred_t, nsp = model(*args, **kw)
        kw2 = {k: kw[k] for k in ("token_type_ids", "position_ids", "attention_mask", "image_attention_mask", "co_attention_mask")}
        scores, _ = model.sequence_log_likelihood(*args, kw["masked_lm_labels"], **kw2)
    print()
    close(lm, g["lm_loss"], what="lm_loss"); close(img, g["img_loss"], what="img_loss"); close(nsp_l, g["nsp_loss"], what="nsp_loss")
    close(nsp, g["nsp"], what="nsp")
    rows = T_(g["rows"]).cuda()
    close(pred_t.reshape(-1, pred_t.shape[-1])[rows][:, ::16], g["pred_t_rows"], what="MLM logits")
    close(seq_t.reshape(-1, 768)[rows], g["seq_out_t_rows"], what="seq_out_t")
    close(scores, g["seq_loglik"], what="candidate log-likelihoods")
    from unimm_amd.harness import scores_to_ranks
    assert torch.equal(scores_to_ranks(scores.view(1, 1, -1)).cpu(), R.scores_to_ranks(T_(g["seq_loglik"]).view(1, 1, -1)))


def test_full_config_b6_backward_matches_reference_golden_fp32(golden_dir, full5):
    """The whole model's backward against the REFERENCE's gradients (tests/golden/full_b6_grads.npz), with NO exceptions for
    the poolers / the top connection block: in this mode the ReLU units of the pooled rows no longer switch on bf16 noise."""
    from oracle.cases import grad_sample_index
    g = np.load(os.path.join(golden_dir, "full_b6.npz"))
    gg = np.load(os.path.join(golden_dir, "full_b6_grads.npz"))
    model = full5
    args, kw = _full_b6_call(g)
    model.zero_grad(set_to_none=True)
    lm, img, nsp_l, _, _, _ = model(*args, **kw, _want_lm_scores=False)
    (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    for name, got in (("lm_loss", lm), ("img_loss", img), ("nsp_loss", nsp_l)):
        assert abs(float(got.detach()) - float(gg[name].item())) <= 1e-3 * (1 + abs(float(gg[name].item()))), name
    params = dict(model.named_parameters())
    names = [str(n) for n in gg["grad_names"]]
    norms, amax = gg["grad_norms"], gg["grad_absmax"]
    gmax = float(amax.max())
    worst_n, checked = 0.0, 0
    for n, want, am in zip(names, norms, amax):
        p = params[n]
        if want < 0:
            assert p.grad is None, n
            continue
        got = float(p.grad.double().norm())
        if am < 1e-6 * gmax:
            assert got <= 1e-3 * float(norms.max()), (n, got)
            continue
        rel = abs(got - want) / want
        worst_n = max(worst_n, rel)
        assert rel <= 1e-3, (n, got, want, rel)
        checked += 1
    assert checked > 450
    worst_l2 = worst_max = 0.0
    worst_name = ""
    nslices = 0
    for k in gg.files:
        if not k.startswith("grad::"):
            continue
        n = k[6:]
        want = gg[k]
        gr = params[n].grad
        if n.endswith("word_embeddings.weight"):
            got = gr[T_(gg["grad_rowidx::" + n]).cuda()][:, ::4]
        elif gr.dim() == 1:
            got = gr[::4]
        else:
            got = gr[T_(grad_sample_index(tuple(gr.shape))[0]).cuda()][:, ::4]
        if np.abs(want).max() < 1e-6 * gmax:
            continue
        gn = got.detach().double().cpu().numpy()
        r = float(np.abs(gn - want).max() / np.abs(want).max())
        l2 = float(np.linalg.norm(gn - want) / max(np.linalg.norm(want), 1e-30))
        if l2 > worst_l2:
            worst_name = n
        worst_l2, worst_max = max(worst_l2, l2), max(worst_max, r)
        assert l2 <= 1e-3 and r <= 1e-3, (n, l2, r)          # <= 1e-3 of the tensor's scale, poolers and the top connection block included
        nslices += 1
    assert nslices > 100
    print(f"\nfp32x3 full-config backward vs the reference: worst norm error {worst_n:.2e}; sampled slices: worst L2 {worst_l2:.2e} "
          f"({worst_name}), worst element / scale {worst_max:.2e}")


def test_optimizer_step_and_ranking_gradient_in_fp32_mode(golden_dir):
    """The rest of the dense fine-tune iteration around the fp32-accuracy engine: a gradient arriving through the returned NSP
    scores (the NeuralNDCG^T term of dense_annotation_finetuning.py:286-293) and a FusedAdamW step whose updated weights must
    reach the split weight copies (the next forward equals the oracle's forward on the updated state dict)."""
    from oracle import vilbert_ref as R
    from unimm_amd.optim import FusedAdamW
    model, ocfg, sd = build_small(golden_dir)
    model.eval()
    g = np.load(os.path.join(golden_dir, "small_dis.npz"))
    args, kw = kwargs_from(g)
    eng = model.engine
    eng.ensure(torch.device("cuda", 0))
    opt = FusedAdamW([dict(params=list(model.parameters()), lr=1e-3, weight_decay=0.01)], eng, lr=1e-3)
    opt.zero_grad()
    lm, img, nsp_l, _, _, nsp = model(*args, **kw, _want_lm_scores=False)
    wvec = torch.linspace(-1.0, 1.0, nsp.numel(), device=nsp.device).view_as(nsp)
    (lm.sum() + (torch.softmax(nsp, -1) * wvec).sum()).backward()          # a loss on the scores + the LM loss, no NSP / image term
    torch.cuda.synchronize()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != R.TIED[0]}
    leaves[R.TIED[0]] = leaves[R.TIED[1]]
    out = R.forward(leaves, ocfg, *args, **kw)
    (out["lm_loss"].sum() + (torch.softmax(out["nsp"], -1) * wvec.cpu()).sum()).backward()
    worst = 0.0
    for n, p in model.named_parameters():
        if p.grad is None:
            assert leaves[n].grad is None or float(leaves[n].grad.abs().max()) == 0.0, n
            continue
        want = leaves[n].grad
        if want is None:
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        worst = max(worst, float((p.grad.cpu() - want).abs().max() / want.abs().max().clamp_min(1e-6)))
    assert worst <= 1e-3, worst
    opt.step()
    with torch.no_grad():
        lm2, _, _, _, _, nsp2 = model(*args, **kw, _want_lm_scores=False)
        new_sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        new_sd[R.TIED[0]] = new_sd[R.TIED[1]]
        ref2 = R.forward(new_sd, ocfg, *args, **kw)
    assert abs(float(lm2) - float(lm)) > 1e-3                 # the step really changed the model
    print()
    close(lm2, ref2["lm_loss"], what="lm_loss after one AdamW step")
    close(nsp2, ref2["nsp"], what="nsp after one AdamW step")


@pytest.mark.parametrize("compute,tol", [("bf16", 1e-2), ("fp32x3", 1e-3)])
def test_config_switches_sum_fusion_and_predict_feature(golden_dir, compute, tol):
    """fusion_method='sum' (models/vilbert_dialog.py:1062-1063) and predict_feature=True (masked-region MSE, :1562-1566) against
    the REFERENCE's output on the small config (tests/golden/small_sumfeat.npz), both engines."""
    from oracle import vilbert_ref as R
    from unimm_amd import BertConfig, BertForMultiModalPreTraining
    cfgd = dict(json.load(open(os.path.join(golden_dir, "small_config.json"))), fusion_method="sum", predict_feature=True)
    model = BertForMultiModalPreTraining(BertConfig.from_dict(cfgd), compute_dtype=compute)
    model.load_state_dict(R.init_state_dict(R.make_config(cfgd), seed=11), strict=True)
    model = model.cuda().eval()
    g = np.load(os.path.join(golden_dir, "small_sumfeat.npz"))
    args, kw = kwargs_from(g)
    model.zero_grad(set_to_none=True)
    lm, img, nsp_l, _, _, nsp = model(*args, **kw, _want_lm_scores=False)
    (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    print()
    close(lm, g["lm_loss"], tol=tol, what="lm_loss"); close(img, g["img_loss"], tol=tol, what="img_loss (MSE)")
    close(nsp_l, g["nsp_loss"], tol=tol, what="nsp_loss"); close(nsp, g["nsp"], tol=tol, what="nsp (sum fusion)")
    params = dict(model.named_parameters())
    gate = 4e-2 if compute == "bf16" else 1e-3
    for n, want in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        if want < 0:
            assert params[n].grad is None, n
            continue
        got = float(params[n].grad.double().norm())
        assert abs(got - want) <= gate * max(want, 1e-4), (n, got, want)
    for k in g.files:
        if k.startswith("grad::"):
            want = g[k]
            err = np.abs(params[k[6:]].grad.cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-6)
            # (bf16 engine: ReLU units of the 4 pooled rows switch on operand noise -- the pooler gate of test_gpu_fullsize.py)
            assert err <= (0.35 if (compute == "bf16" and "pooler" in k) else gate), (k, err)
    args, kw = kwargs_from(g, train=False, device="cuda")
    with torch.no_grad():
        _, p_v, nsp2, _, _ = model(*args, **kw)
    close(p_v, g["inf_pred_v"], tol=tol, what="pred_v"); close(nsp2, g["inf_nsp"], tol=tol, what="inference nsp")


def test_two_streams_equal_one_stream_fp32(golden_dir, small):
    """The fp32-accuracy engine on the base engine's two-stream schedule (image side beside the text side) against everything on
    one stream: same kernels, same order within each stream -> identical losses / scores, gradients equal up to the order of
    the fp32 atomics of the weight-gradient launches; train mode (dropout keys are per site, not per launch order)."""
    model, _, _ = small
    eng = model.engine
    g = np.load(os.path.join(golden_dir, "small_mixed.npz"))
    args, kw = kwargs_from(g)
    res = {}
    model.train()
    try:
        for dual in (True, False, True):
            eng.dual_stream = dual
            model.set_dropout_seed(5, step=0)
            model.zero_grad(set_to_none=True)
            lm, img, nsp_l, _, _, nsp = model(*args, **kw, _want_lm_scores=False)
            (lm + img + nsp_l).sum().backward()
            torch.cuda.synchronize()
            res.setdefault(dual, []).append((torch.stack([lm, img, nsp_l]).flatten().detach().clone(), nsp.detach().clone(),
                                             eng.arena.grad_flat.clone()))
    finally:
        eng.dual_stream = True
        model.eval()
    a, b, a2 = res[True][0], res[False][0], res[True][1]
    for x, y in ((a, b), (a, a2)):
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1])
        d = float((x[2] - y[2]).abs().max() / x[2].abs().max())
        assert d <= 2e-6, d


@pytest.mark.parametrize("tag,extra", [("frozen", dict(fixed_t_layer=2)), ("nocoatt", dict(with_coattention=False))])
def test_encoder_options_frozen_text_layers_and_no_coattention(golden_dir, tag, extra):
    """fixed_t_layer (the first text layers run under no_grad: models/vilbert_dialog.py:864-869) and with_coattention=False (:901)
    on the bf16 engine against the REFERENCE's output on the small config (tests/golden/small_frozen.npz / small_nocoatt.npz):
    losses, NSP scores, hidden states, gradient norms of every tensor, sampled gradients -- and exactly the reference's set of
    parameters whose .grad stays None (an optimizer must not touch them, not even with weight decay).  Under a data-parallel
    hook every bucket is still reported once; the graph executor replays the same step; the fp32x3 engine refuses the options."""
    from oracle import vilbert_ref as R
    from unimm_amd import BertConfig, BertForMultiModalPreTraining
    cfgd = dict(json.load(open(os.path.join(golden_dir, "small_config.json"))), **extra)
    model = BertForMultiModalPreTraining(BertConfig.from_dict(cfgd))
    model.load_state_dict(R.init_state_dict(R.make_config(cfgd), seed=11), strict=True)
    model = model.cuda().eval()
    g = np.load(os.path.join(golden_dir, f"small_{tag}.npz"))
    args, kw = kwargs_from(g)
    model.engine.ensure(torch.device("cuda", 0))
    calls = []
    model.engine.grad_bucket_hook = lambda group, more=False: calls.append(group)
    model.zero_grad(set_to_none=True)
    lm, img, nsp_l, _, _, nsp = model(*args, **kw, _want_lm_scores=False)
    (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    model.engine.grad_bucket_hook = None
    print()
    tol = 1e-2
    close(lm, g["lm_loss"], tol=tol, what="lm_loss"); close(img, g["img_loss"], tol=tol, what="img_loss")
    close(nsp_l, g["nsp_loss"], tol=tol, what="nsp_loss"); close(nsp, g["nsp"], tol=tol, what="nsp")
    assert sorted(calls) == sorted(gname for gname, _, _ in model.engine.arena.buckets) and calls[-1] == "text_embeddings"
    params = dict(model.named_parameters())
    n_none = 0
    for n, want in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        if want < 0:
            assert params[n].grad is None, n
            n_none += 1
            continue
        assert params[n].grad is not None, n
        got = float(params[n].grad.double().norm())
        assert abs(got - want) <= 4e-2 * max(want, 1e-4), (n, got, want)
    assert n_none == (46 if tag == "frozen" else 73)
    for k in g.files:
        if k.startswith("grad::"):
            want = g[k]
            err = np.abs(params[k[6:]].grad.cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-6)
            assert err <= 4e-2, (k, err)
    werr = np.abs(params["bert.embeddings.word_embeddings.weight"].grad[:64].cpu().numpy() - g["grad_rows::word_embeddings"]).max()
    assert werr <= 4e-2 * np.abs(g["grad_rows::word_embeddings"]).max()          # the tied decoder still trains the word table
    with torch.no_grad():
        seq = model(*args, **kw)[3]
    am = g["in::attention_mask"]
    valid = torch.from_numpy((am.reshape(am.shape[0], am.shape[1], -1) != 0).any(-1))      # (padding rows: zeros here, garbage there)
    close(seq.cpu()[valid], g["seq_out_t"][valid.numpy()], tol=tol, what="seq_out_t (valid rows)")
    # the step executor replays the same step
    model.engine.enable_graphs(row_bucket=16, lm_bucket=8, capture_after=0)
    dargs = [a.cuda() if torch.is_tensor(a) else a for a in args]
    dkw = {k: (v.cuda() if torch.is_tensor(v) and k != "nsp_weight" else v) for k, v in kw.items()}
    want_g = model.engine.arena.grad_flat.clone()
    for _ in range(2):
        model.engine.arena.zero_grads()
        lm2, img2, nsp_l2, _, _, _ = model(*dargs, **dkw, _want_lm_scores=False)
        (lm2 + img2 + nsp_l2).sum().backward()
        torch.cuda.synchronize()
    assert model.engine.graphs.stats["replays"] >= 1
    assert abs(float(lm2) - float(lm)) <= 2e-6 * max(1.0, abs(float(lm)))
    assert float((model.engine.arena.grad_flat - want_g).abs().max()) <= 2e-5 * float(want_g.abs().max())
    with pytest.raises(NotImplementedError):
        m3 = BertForMultiModalPreTraining(BertConfig.from_dict(cfgd), compute_dtype="fp32x3").cuda().eval()
        m3(*args, **kw, _want_lm_scores=False)
