"""Training-loop shell on the small config: batch_multiply accumulation + optimizer /
scheduler cadence against a hand-rolled replay with the oracle's AdamW, and the reference's checkpoint dict
through a save -> fresh process state -> resume round trip."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _encoder(golden_dir, tmp_path, state=None):
    from unimm_amd import VisualDialogEncoder
    cfg = json.load(open(os.path.join(golden_dir, "small_config.json")))
    for k in list(cfg):
        if k.endswith("dropout_prob"):
            cfg[k] = 0.0                      # the replay below must see the same function on every call
    path = os.path.join(tmp_path, "small_nodrop.json")
    json.dump(cfg, open(path, "w"))
    torch.manual_seed(5)
    enc = VisualDialogEncoder(path).to("cuda")
    if state is not None:
        enc.load_state_dict(state)
    return enc


def _optim(enc):
    from unimm_amd.optim import FusedAdamW, WarmupLinearScheduleNonZero, default_language_weights, reference_param_groups
    groups = reference_param_groups(enc, lr=1e-3, image_lr=4e-3, language_weights=default_language_weights(enc))
    opt = FusedAdamW(groups, enc.bert_pretrained.engine, lr=1e-3)
    return opt, WarmupLinearScheduleNonZero(opt, warmup_steps=2, t_total=20, min_lr=1e-5)


def test_train_step_accumulation_and_checkpoint_resume(golden_dir, tmp_path):
    from oracle import adamw_ref as AR
    from unimm_amd import params as P, synth, trainer
    enc = _encoder(golden_dir, tmp_path)
    cfg = enc.bert_pretrained.config
    init = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    opt, sch = _optim(enc)
    batches = []
    for i in range(4):
        b, nsp_w = synth.make_loader_batch(n_img=2, rounds=1, samples=3, T=64, cfg=cfg, seed=100 + i)
        batches.append(trainer.expand_image_fields(b))
    params = dict(lm_loss_coeff=1.0, nsp_loss_coeff=1.0, img_loss_coeff=1.0, nsp_weight=nsp_w, batch_multiply=2)
    assert batches[0]["image_feat"].shape[:3] == (2, 1, 3)

    # ---- per-micro-batch gradients from a second copy of the model (same initial weights)
    rep = _encoder(golden_dir, tmp_path, init)
    names = [n for n, _ in rep.named_parameters()]
    rep.train()
    micro = []
    for it in (1, 2):
        rep.zero_grad(set_to_none=True)
        loss, *_ = trainer.harness.forward(rep, batches[it - 1], params)
        loss.backward()
        micro.append({n: p.grad.detach().cpu().numpy().copy() for n, p in rep.named_parameters() if p.grad is not None})

    # ---- the shell itself; snapshot what each optimizer step consumed
    ref = {n: p.detach().cpu().numpy().copy() for n, p in enc.named_parameters()}
    mom = {n: (np.zeros_like(v), np.zeros_like(v)) for n, v in ref.items()}
    seen = []
    inner = opt.step

    def spy(*a, **k):
        seen.append(({n: p.grad.detach().cpu().numpy().copy() for n, p in enc.named_parameters() if p.grad is not None},
                     [g["lr"] for g in opt.param_groups]))
        return inner(*a, **k)

    opt.step = spy
    losses = []
    for it in range(1, 3):
        losses.append(trainer.train_step(enc, opt, sch, batches[it - 1], params, it)[0])
    ck = trainer.save_checkpoint(os.path.join(tmp_path, "visdial_dialog_encoder_2.ckpt"), enc, opt, sch, 2)
    res = _encoder(golden_dir, tmp_path)
    opt2, sch2 = _optim(res)
    assert trainer.load_checkpoint(ck, res, opt2, sch2, resume=True) == 2
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq) and opt2.step_count == 1
    assert torch.equal(res.bert_pretrained.engine.arena.flat, enc.bert_pretrained.engine.arena.flat)
    assert [g["lr"] for g in opt2.param_groups] == [g["lr"] for g in opt.param_groups] and sch2.last_epoch == 2
    # the fused-representation dropout (p = 0.1, hard-coded in the model as in models/vilbert_dialog.py:1064) is the
    # one stochastic site left; like torch's RNG state it is not part of the reference's checkpoint, so line the
    # counter-based stream up by hand
    enc.bert_pretrained.set_dropout_seed(9, step=2)
    res.bert_pretrained.set_dropout_seed(9, step=2)
    for it in range(3, 5):
        losses.append(trainer.train_step(enc, opt, sch, batches[it - 1], params, it)[0])
    torch.cuda.synchronize()
    assert opt.step_count == 2 and len(seen) == 2                   # 4 iterations, batch_multiply 2
    # cadence: the scheduler has stepped once before the first optimizer step and three times before the second
    from oracle.adamw_ref import warmup_linear_nonzero
    assert seen[0][1][0] == warmup_linear_nonzero(1, 1e-3, 2, 20) and seen[1][1][0] == warmup_linear_nonzero(3, 1e-3, 2, 20)
    # accumulation: the first step consumed (g1 + g2) / batch_multiply
    for n in micro[0]:
        want = (micro[0][n] + micro[1][n]) / 2
        assert np.abs(seen[0][0][n] - want).max() <= 2e-3 * max(1e-6, np.abs(want).max()), n
    # update: oracle AdamW on exactly those gradients and learning rates
    for t, (grads, lrs) in enumerate(seen, start=1):
        for n, gr, lr in zip(names, opt.param_groups, lrs):
            if n in grads and not P.is_unused(n.replace("bert_pretrained.", "", 1)):
                AR.adamw_step(ref[n], grads[n], *mom[n], lr, gr["weight_decay"], t)
    for n, p in enc.named_parameters():
        assert np.allclose(p.detach().cpu().numpy(), ref[n], rtol=1e-5, atol=1e-7), n

    # ---- checkpoint: reference layout, then resume in a fresh model / optimizer / scheduler
    d = torch.load(ck, map_location="cpu")
    assert set(d) == {"model_state_dict", "scheduler_state_dict", "optimizer_state_dict", "iter_id"} and d["iter_id"] == 2
    assert all(k.startswith("bert_pretrained.") for k in d["model_state_dict"])
    for it in range(3, 5):
        trainer.train_step(res, opt2, sch2, batches[it - 1], params, it)
    torch.cuda.synchronize()
    # same kernels from the same state; Adam turns atomics-order noise into +-lr where the true gradient is ~0,
    # so compare in the mean (the step-by-step parity is the oracle check above)
    for (n, p), (_, q) in zip(enc.named_parameters(), res.named_parameters()):
        assert float((p - q).detach().abs().mean()) <= 1e-4 * max(1e-3, float(p.detach().abs().mean())), n
    # warm start by key intersection (train.py:352-364): a checkpoint with extra / missing keys still loads
    sd = dict(d["model_state_dict"])
    sd["not.in.the.model"] = torch.zeros(3)
    dropped = next(iter(sd))
    sd.pop(dropped)
    fresh = _encoder(golden_dir, tmp_path)
    assert trainer.load_checkpoint({"model_state_dict": sd}, fresh) == len(d["model_state_dict"]) - 1


def test_harness_forward_evaluation_and_score_outputs(golden_dir, tmp_path):
    """The evaluation calls of the reference (train.py:180-290, val_lm.py:104-136): no labels / targets, NSP scores
    and dense LM scores on request, sample_size subsampling in training."""
    from unimm_amd import synth, trainer
    from unimm_amd.harness import forward, generative_scores
    enc = _encoder(golden_dir, tmp_path)
    cfg = enc.bert_pretrained.config
    b, nsp_w = synth.make_loader_batch(n_img=2, rounds=1, samples=3, T=64, cfg=cfg, seed=7)
    b = trainer.expand_image_fields(b)
    params = dict(lm_loss_coeff=1.0, nsp_loss_coeff=1.0, img_loss_coeff=1.0, nsp_weight=nsp_w)
    enc.eval()
    with torch.no_grad():
        out = forward(enc, b, params, output_nsp_scores=True, output_lm_scores=True, evaluation=True)
    loss, lm, nsp, img, nsp_scores, lm_scores = out
    assert loss is None and lm is None and nsp is None and img is None
    assert nsp_scores.shape == (6, 2) and lm_scores.shape == (6, 64, cfg.vocab_size)
    assert torch.isfinite(nsp_scores).all() and torch.isfinite(lm_scores[:, 0]).all()
    ll = generative_scores(lm_scores, b["mask"].reshape(6, 64))
    assert ll.shape == (6,) and torch.isfinite(ll).all()
    # the fused path gives the same per-sequence log-likelihoods without the dense scores
    flat = lambda k, keep: b[k].reshape((-1,) + tuple(b[k].shape[-keep:]))
    s2, _ = enc.bert_pretrained.sequence_log_likelihood(
        flat("tokens", 1), flat("image_feat", 2), flat("image_loc", 2), flat("mask", 1), token_type_ids=flat("segments", 1),
        position_ids=flat("positions", 1), attention_mask=flat("txt_attention_mask", 2), image_attention_mask=flat("image_mask", 1),
        co_attention_mask=flat("co_attention_mask", 2))
    assert (ll.cpu() - s2.cpu()).abs().max() <= 2e-2 * max(1.0, float(ll.abs().max()))
    enc.train()
    res = forward(enc, b, params, sample_size=4)            # training call with subsampling (train.py:138-145)
    assert len(res) == 4 and torch.isfinite(res[0])


def test_dense_finetune_step_matches_oracle_objective(golden_dir, tmp_path):
    """Row F4: one dense-annotation step (dense_annotation_finetuning.py:146-301) on the HIP path against the
    oracle encoder + the golden-pinned ranking loss under autograd: the NeuralNDCG^T gradient has to arrive
    through the returned NSP scores."""
    import zlib
    from oracle import vilbert_ref as R
    from unimm_amd import dropout as DR, ranking, synth, trainer
    enc = _encoder(golden_dir, tmp_path)
    cfg = enc.bert_pretrained.config
    sd = {k[len("bert_pretrained."):]: v.detach().float().cpu().clone() for k, v in enc.state_dict().items()}
    opt, sch = _optim(enc)
    n_opt = 12
    b, nsp_w = synth.make_loader_batch(n_img=1, rounds=1, samples=n_opt, T=64, cfg=cfg, seed=31)
    rng = np.random.Generator(np.random.PCG64(5))
    b["gt_option"] = torch.tensor([4])
    b["gt_relevance"] = torch.from_numpy(rng.choice(np.array([0, 0, 0.2, 0.6, 1.0], np.float32), size=(1, n_opt)))
    b["gt_relevance"][0, 4] = 1.0
    params = dict(lm_loss_coeff=1.0, nsp_loss_coeff=0.5, img_loss_coeff=1.0, nsp_weight=nsp_w, batch_multiply=1)
    order = torch.tensor([4, 7, 0, 11, 2, 9, 1, 3, 10, 5, 8, 6])
    grads = {}
    real_step = opt.step
    opt.step = lambda *a, **k: grads.update({n: p.grad.detach().float().cpu().clone() for n, p in enc.named_parameters()
                                             if p.grad is not None}) or real_step(*a, **k)
    enc.bert_pretrained.set_dropout_seed(13, step=2)
    lr0 = [g["lr"] for g in opt.param_groups]
    loss, parts = trainer.dense_finetune_step(enc, opt, sch, b, params, iter_id=1, num_options=n_opt, option_indices=order)
    torch.cuda.synchronize()
    assert grads and [g["lr"] for g in opt.param_groups] != lr0          # optimizer and scheduler both stepped

    # ---- oracle replay: same option order, the fused-vector dropout mask of step 3 re-played
    sel = trainer.expand_image_fields(trainer.select_options(b, order))
    flat = lambda k, keep: sel[k].reshape((-1,) + tuple(sel[k].shape[-keep:])) if keep else sel[k].reshape(-1)

    def drop_fn(site, x, p):
        if site != "fuse":
            return x
        key = DR.make_key(13, 3, zlib.crc32(site.encode()) & 0xFFFFFFFF)
        _, thr, scale = DR.drop_arg(p, key)
        return x * torch.from_numpy(DR.keep_mask_nd(key, thr, tuple(x.shape))) * scale

    ocfg = R.make_config(json.load(open(os.path.join(tmp_path, "small_nodrop.json"))))
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != R.TIED[0]}
    leaves[R.TIED[0]] = leaves[R.TIED[1]]
    out = R.forward(leaves, ocfg, flat("tokens", 1), flat("image_feat", 2), flat("image_loc", 2),
                    token_type_ids=flat("segments", 1), position_ids=flat("positions", 1),
                    attention_mask=flat("txt_attention_mask", 2), image_attention_mask=flat("image_mask", 1),
                    co_attention_mask=flat("co_attention_mask", 2), masked_lm_labels=flat("mask", 1),
                    image_label=flat("image_label", 1), image_target=flat("image_target", 2),
                    next_sentence_label=flat("next_sentence_labels", 0), nsp_weight=nsp_w, lm_weight=flat("weights", 1),
                    drop_fn=drop_fn)
    want, wparts = ranking.dense_finetune_loss(out["nsp"], flat("next_sentence_labels", 0), b["gt_relevance"][:, order],
                                               out["lm_loss"], 0.5, num_options=n_opt)
    want.backward()
    assert abs(float(parts["target"]) - float(wparts["target"].detach())) <= 1e-2
    assert abs(loss - float(want)) <= 2e-2 * max(1.0, abs(float(want)))
    bad = []
    for n, g in grads.items():
        ref = leaves[n[len("bert_pretrained."):]].grad
        if ref is None:                     # image-prediction head: the masked-region loss is not in this objective
            assert float(g.abs().max()) == 0.0, n
            continue
        w, got = float(ref.norm()), float(g.norm())
        if abs(got - w) > 6e-2 * max(w, 1e-4):
            bad.append((n, got, w))
    assert not bad, bad[:5]
    k = "cls.bi_seq_relationship.weight"           # reached only through the NSP scores: ranking + NSP terms
    w_nsp = float(leaves[k].grad.norm())
    assert w_nsp > 0 and abs(float(grads["bert_pretrained." + k].norm()) - w_nsp) <= 6e-2 * w_nsp


def test_visdial_evaluate_on_the_encoder_matches_oracle_scores(golden_dir, tmp_path):
    """train.py:180-290 on the HIP path: chunked NSP scoring of 2 images x 2 rounds x 6 options; the probabilities
    the metrics are computed from agree with the oracle encoder's, and chunking does not change the result."""
    from oracle import vilbert_ref as R
    from unimm_amd import metrics, synth, trainer
    enc = _encoder(golden_dir, tmp_path)
    cfg = enc.bert_pretrained.config
    sd = {k[len("bert_pretrained."):]: v.detach().float().cpu().clone() for k, v in enc.state_dict().items()}
    b, _ = synth.make_loader_batch(n_img=2, rounds=2, samples=6, T=64, cfg=cfg, seed=41, modes=["dis"] * 24)
    b["gt_option_inds"] = torch.tensor([[0, 3], [5, 1]])
    b["gt_relevance"] = torch.tensor([[1.0, 0, 0.5, 0, 0, 0], [0, 0.2, 0, 0, 1.0, 0.4]])
    b["round_id"] = torch.tensor([[2], [1]])
    params = dict(n_gpus=1, nsp_weight=None)
    enc.train()
    got = trainer.visdial_evaluate([b], params, 2, enc, chunk_size=8)
    assert enc.training
    again = trainer.visdial_evaluate([b], params, 2, enc, chunk_size=24)
    assert got.keys() == again.keys() and all(abs(float(got[k]) - float(again[k])) < 1e-6 for k in got)
    ex = trainer.expand_image_fields({k: b[k] for k in ("tokens", "image_feat", "image_loc", "image_mask")})
    flat = lambda t, keep: t.reshape((-1,) + tuple(t.shape[-keep:]))
    ocfg = R.make_config(json.load(open(os.path.join(tmp_path, "small_nodrop.json"))))
    with torch.no_grad():
        out = R.forward(sd, ocfg, flat(b["tokens"], 1), flat(ex["image_feat"], 2), flat(ex["image_loc"], 2),
                        token_type_ids=flat(b["segments"], 1), position_ids=flat(b["positions"], 1),
                        attention_mask=flat(b["txt_attention_mask"], 2), image_attention_mask=flat(ex["image_mask"], 1),
                        co_attention_mask=flat(b["co_attention_mask"], 2))
    p = torch.softmax(out["nsp"], 1)[:, 0].view(2, 2, 6)
    sp, nd = metrics.SparseGTMetrics(), metrics.NDCG()
    sp.observe(p, b["gt_option_inds"])
    nd.observe(p[torch.arange(2), b["round_id"].view(-1) - 1], b["gt_relevance"])
    want = {**sp.retrieve(), **nd.retrieve()}
    assert set(want) == set(got) and "ndcg" in got and "mrr_round_2" in got
    # ranks are discrete: compare the probabilities, and the metrics only where the oracle's ordering has a clear margin
    with torch.no_grad():
        item = {k: flat(b[k], keep) if keep else b[k].reshape(-1) for k, keep in trainer._EVAL_TEXT}
        item.update(image_feat=flat(ex["image_feat"], 2), image_loc=flat(ex["image_loc"], 2), image_mask=flat(ex["image_mask"], 1))
        enc.eval()
        nsp = trainer.harness.forward(enc, item, params, output_nsp_scores=True, evaluation=True)[4]
    pd = torch.softmax(nsp.float(), 1)[:, 0].view(2, 2, 6).cpu()
    assert (pd - p).abs().max() <= 1e-2
    gaps = (p.sort(-1)[0].diff(dim=-1)).abs().min()
    if gaps > 2e-2:
        assert all(abs(float(got[k]) - float(want[k])) < 1e-5 for k in want)


def test_generative_evaluate_matches_oracle_log_likelihood_ranks(golden_dir, tmp_path):
    """val_lm.py:38-150 on the HIP path: candidates ranked by the summed log-likelihood of their answer tokens; the
    scores agree with the oracle's dense-logits formulation (CE over all 256 rows, ignore_index=-1), chunking does not
    change them, and the metrics / rank tensors come out of the same accumulators."""
    from oracle import vilbert_ref as R
    from unimm_amd import metrics, synth, trainer
    enc = _encoder(golden_dir, tmp_path)
    cfg = enc.bert_pretrained.config
    sd = {k[len("bert_pretrained."):]: v.detach().float().cpu().clone() for k, v in enc.state_dict().items()}
    b, _ = synth.make_loader_batch(n_img=2, rounds=2, samples=6, T=64, cfg=cfg, seed=43, modes=["gen"] * 24, mask_prob=0.0)
    b["gt_option_inds"] = torch.tensor([[2, 0], [4, 5]])
    b["gt_relevance"] = torch.tensor([[0, 0.5, 1.0, 0, 0, 0], [0.4, 0, 0, 0, 1.0, 0]])
    b["round_id"] = torch.tensor([[1], [2]])
    params = dict(n_gpus=1, nsp_weight=None)
    ranks = []
    got = trainer.generative_evaluate([b], params, 2, enc, chunk_size=8, ranks_out=ranks)
    again = trainer.generative_evaluate([b], params, 2, enc, chunk_size=24)
    assert got.keys() == again.keys() and all(abs(float(got[k]) - float(again[k])) < 1e-6 for k in got)
    assert len(ranks) == 1 and ranks[0].shape == (2, 2, 6) and sorted(ranks[0][0, 0].tolist()) == [1, 2, 3, 4, 5, 6]
    # oracle scores
    ex = trainer.expand_image_fields({k: b[k] for k in ("tokens", "image_feat", "image_loc", "image_mask")})
    flat = lambda t, keep: t.reshape((-1,) + tuple(t.shape[-keep:]))
    ocfg = R.make_config(json.load(open(os.path.join(tmp_path, "small_nodrop.json"))))
    with torch.no_grad():
        out = R.forward(sd, ocfg, flat(b["tokens"], 1), flat(ex["image_feat"], 2), flat(ex["image_loc"], 2),
                        token_type_ids=flat(b["segments"], 1), position_ids=flat(b["positions"], 1),
                        attention_mask=flat(b["txt_attention_mask"], 2), image_attention_mask=flat(ex["image_mask"], 1),
                        co_attention_mask=flat(b["co_attention_mask"], 2))
        want = R.sequence_log_likelihood(out["pred_t"], flat(b["mask"], 1)).view(2, 2, 6)
    model = enc.bert_pretrained
    enc.eval()
    with torch.no_grad():
        s, _ = model.sequence_log_likelihood(flat(b["tokens"], 1), flat(ex["image_feat"], 2), flat(ex["image_loc"], 2), flat(b["mask"], 1),
                                             token_type_ids=flat(b["segments"], 1), position_ids=flat(b["positions"], 1),
                                             attention_mask=flat(b["txt_attention_mask"], 2), image_attention_mask=flat(ex["image_mask"], 1),
                                             co_attention_mask=flat(b["co_attention_mask"], 2))
    s = s.view(2, 2, 6).cpu()
    assert (s - want).abs().max() <= 2e-2 * max(1.0, float(want.abs().max()))
    sp, nd = metrics.SparseGTMetrics(), metrics.NDCG()
    sp.observe(s, b["gt_option_inds"])
    nd.observe(s[torch.arange(2), b["round_id"].view(-1) - 1], b["gt_relevance"])
    mine = {**sp.retrieve(), **nd.retrieve()}
    assert all(abs(float(got[k]) - float(mine[k])) < 1e-6 for k in mine)      # the loop adds nothing to the scorer
