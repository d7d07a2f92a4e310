"""CPU tests of the training-loop shell's host logic (no kernels): dataloader-layout batches, the image
expansion of train.py:413-432, the checkpoint dict of train.py:503-505 with warm start and resume, and the
optimizer / scheduler cadence of train_step under batch_multiply (with a stand-in model and torch SGD)."""
import os

import pytest
import torch

from unimm_amd import synth, trainer
from unimm_amd.optim import WarmupLinearScheduleNonZero


def test_loader_batch_layout_and_image_expansion():
    b, nsp_w = synth.make_loader_batch(n_img=3, rounds=2, samples=4, T=64, R=37, seed=3)
    assert b["tokens"].shape == (3, 2, 4, 64) and b["txt_attention_mask"].shape == (3, 2, 4, 64, 64)
    assert b["image_feat"].shape[:2] == (3, 37) and b["co_attention_mask"].shape == (3, 2, 4, 37, 64)
    e = trainer.expand_image_fields(b)
    assert e["image_feat"].shape[:4] == (3, 2, 4, 37) and e["image_label"].shape == (3, 2, 4, 37)
    assert torch.equal(e["image_feat"][1, 0, 0], e["image_feat"][1, 1, 3])      # one image, every round and sample
    assert torch.equal(e["tokens"], b["tokens"]) and nsp_w.shape == (1, 2)


class _Enc(torch.nn.Module):
    """Stand-in with the call signature harness.forward uses; losses depend on the parameters."""

    def __init__(self):
        super().__init__()
        self.bert_pretrained = torch.nn.Linear(4, 3)

    def forward(self, tokens, feat, loc, **kw):
        z = self.bert_pretrained(feat.float().mean(1)[:, :4])
        return z[:, 0].pow(2).mean().reshape(1), z[:, 1].pow(2).mean().reshape(1), z[:, 2].pow(2).mean().reshape(1)


def test_train_step_cadence_and_checkpoint_roundtrip(tmp_path):
    torch.manual_seed(0)
    enc = _Enc()
    opt = torch.optim.SGD(enc.parameters(), lr=0.1)
    sch = WarmupLinearScheduleNonZero(opt, warmup_steps=2, t_total=10, min_lr=1e-5)
    b, nsp_w = synth.make_loader_batch(n_img=1, rounds=1, samples=2, T=64, seed=1)
    b = trainer.expand_image_fields(b)
    params = dict(lm_loss_coeff=1.0, nsp_loss_coeff=1.0, img_loss_coeff=1.0, nsp_weight=nsp_w, batch_multiply=3)
    w0 = enc.bert_pretrained.weight.detach().clone()
    steps = []
    real = opt.step
    opt.step = lambda *a, **k: (steps.append(sch.last_epoch), real(*a, **k))[1]
    for it in range(1, 7):
        trainer.train_step(enc, opt, sch, b, params, it)
        if it < 3:
            assert torch.equal(enc.bert_pretrained.weight, w0)                 # still accumulating
    assert steps == [2, 5] and sch.last_epoch == 6                             # optimizer on iterations 3 and 6 only
    assert all(p.grad is None or float(p.grad.abs().max()) == 0 for p in enc.parameters())   # zero_grad after each step
    path = trainer.save_checkpoint(os.path.join(tmp_path, "c.ckpt"), enc, opt, sch, 6)
    d = torch.load(path)
    assert set(d) == {"model_state_dict", "scheduler_state_dict", "optimizer_state_dict", "iter_id"}
    enc2 = _Enc()
    opt2 = torch.optim.SGD(enc2.parameters(), lr=0.1)
    sch2 = WarmupLinearScheduleNonZero(opt2, warmup_steps=2, t_total=10, min_lr=1e-5)
    assert trainer.load_checkpoint(path, enc2, opt2, sch2, resume=True) == 6
    assert torch.equal(enc2.bert_pretrained.weight, enc.bert_pretrained.weight) and sch2.last_epoch == 6
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
    enc3 = _Enc()
    sd = dict(d["model_state_dict"]); sd["extra.key"] = torch.zeros(1); sd.pop("bert_pretrained.bias")
    assert trainer.load_checkpoint({"model_state_dict": sd}, enc3) == 1 and torch.equal(enc3.bert_pretrained.weight, enc.bert_pretrained.weight)


def test_dialog_mask_spec_dense_form_matches_the_oracle_encoders():
    """DialogMaskSpec.dense (host) == oracle/masks.py, which golden G5 pins to the reference's encoders; also the
    valid lengths the unpadded schedule takes from the descriptors."""
    import numpy as np
    from oracle import masks as OM
    from unimm_amd.inputs import DialogMaskSpec
    T = 64
    cases = [(1, [5, 3, 4]), (0, [5, 3, 4]), (1, [2, 1]), (1, [T - 14, 9]), (0, [T - 3]), (1, [10, 10, 10, 2])]
    mode, Ls, ns, txt, co = [], [], [], [], []
    for m, lens in cases:
        utts = [list(range(1000, 1000 + l)) for l in lens]
        enc = (OM.encode_gen if m else OM.encode_dis)(utts, max_seq_len=T)
        mode.append(m); Ls.append(1 + sum(l + 1 for l in lens)); ns.append(lens[-1] + 1)
        txt.append(np.asarray(enc["txt_attention_mask"][0]) != 0); co.append(np.asarray(enc["co_attention_mask"][0]) != 0)
    spec = DialogMaskSpec(mode, Ls, ns)
    dt, dc = spec.dense(T)
    assert np.array_equal(dt.numpy(), np.stack(txt)) and np.array_equal(dc.numpy(), np.stack(co))
    want = [min(T, L + (n if m else 0)) for m, L, n in zip(mode, Ls, ns)]
    assert spec.valid_lengths(T).tolist() == want
    import pytest
    with pytest.raises(ValueError):
        DialogMaskSpec([1], [4], [4])            # the answer cannot be the whole sequence


class _ScoreEnc(_Enc):
    """Also returns per-sequence NSP scores that depend on the parameters and on the option's tokens."""

    def forward(self, tokens, feat, loc, output_nsp_scores=False, **kw):
        z = self.bert_pretrained(feat.float().mean(1)[:, :4])
        out = (z[:, 0].pow(2).mean().reshape(1), z[:, 1].pow(2).mean().reshape(1), z[:, 2].pow(2).mean().reshape(1))
        if output_nsp_scores:
            out = out + (z[:, :2] * (1.0 + tokens.float().mean(1, keepdim=True) / 1000.0),)
        return out


def test_dense_finetune_step_options_objective_and_cadence():
    torch.manual_seed(0)
    enc = _ScoreEnc()
    opt = torch.optim.SGD(enc.parameters(), lr=0.1)
    sch = WarmupLinearScheduleNonZero(opt, warmup_steps=2, t_total=10, min_lr=1e-5)
    b, nsp_w = synth.make_loader_batch(n_img=1, rounds=1, samples=10, T=64, seed=2)
    b["gt_option"] = torch.tensor([3])
    b["gt_relevance"] = torch.linspace(0, 1, 10).view(1, 10)
    params = dict(lm_loss_coeff=1.0, nsp_loss_coeff=0.7, img_loss_coeff=1.0, nsp_weight=nsp_w, batch_multiply=2)
    seen = []
    real = trainer.select_options
    trainer.select_options = lambda batch, idx: (seen.append(idx.clone()), real(batch, idx))[1]
    try:
        w0 = enc.bert_pretrained.weight.detach().clone()
        l0, parts = trainer.dense_finetune_step(enc, opt, sch, b, params, iter_id=0, num_options=10)
        assert torch.equal(enc.bert_pretrained.weight, w0) and sch.last_epoch == 1      # iter 0 never steps the optimizer
        trainer.dense_finetune_step(enc, opt, sch, b, params, iter_id=1, num_options=10)
        assert torch.equal(enc.bert_pretrained.weight, w0)
        trainer.dense_finetune_step(enc, opt, sch, b, params, iter_id=2, num_options=6)
        assert not torch.equal(enc.bert_pretrained.weight, w0) and sch.last_epoch == 3
    finally:
        trainer.select_options = real
    for idx, n in zip(seen, (10, 10, 6)):
        assert int(idx[0]) == 3 and len(idx) == n and len(set(idx.tolist())) == n       # ground truth first, no repeats
    # objective = ranking term + LM + coeff * NSP cross entropy, divided by batch_multiply
    assert set(parts) == {"target", "nsp", "lm"} and float(parts["target"]) < 0
    assert abs(l0 * 2 - float(parts["target"] + parts["lm"] + 0.7 * parts["nsp"])) < 1e-6
    with pytest.raises(ValueError):
        b2, _ = synth.make_loader_batch(n_img=2, rounds=1, samples=4, T=64, seed=2)
        b2["gt_option"], b2["gt_relevance"] = torch.tensor([0]), torch.zeros(2, 4)
        trainer.dense_finetune_step(enc, opt, sch, b2, params, iter_id=1, num_options=4)


def test_visdial_evaluate_chunks_scores_and_accumulates_metrics():
    """Validation pass (train.py:180-290) with a stand-in scorer: chunking, image expansion, the metric plumbing."""
    from unimm_amd import metrics

    class _Scorer(torch.nn.Module):
        calls = []

        def forward(self, tokens, feat, loc, output_nsp_scores=False, **kw):
            assert kw.get("next_sentence_label") is None and kw.get("image_target") is None      # evaluation call
            _Scorer.calls.append(tokens.shape[0])
            s = tokens[:, 1].float() / 1000.0 + feat[:, 1, 0]
            return None, None, None, torch.stack([s, -s], 1)

    assert [trainer.eval_chunk_size(n) for n in (1, 2, 8)] == [250, 500, 1000]
    enc = _Scorer().train()
    loader = []
    for seed in (3, 4):
        b, _ = synth.make_loader_batch(n_img=2, rounds=2, samples=10, T=64, seed=seed)
        b["gt_option_inds"] = torch.randint(0, 10, (2, 2))
        b["gt_relevance"] = torch.rand(2, 10).round()
        b["gt_relevance"][:, 0] = 1.0
        b["round_id"] = torch.tensor([[1], [2]])
        loader.append(b)
    got = trainer.visdial_evaluate(loader, dict(n_gpus=1, nsp_weight=None), 2, enc, chunk_size=8)
    assert _Scorer.calls == [8] * 10 and enc.training                       # 40 sequences per batch in chunks of 8
    sp, nd = metrics.SparseGTMetrics(), metrics.NDCG()
    for b in loader:
        s = b["tokens"][..., 1].float() / 1000.0 + b["image_feat"][:, 1, 0].view(2, 1, 1)
        p = torch.softmax(torch.stack([s, -s], -1), -1)[..., 0]
        sp.observe(p, b["gt_option_inds"])
        nd.observe(p[torch.arange(2), b["round_id"].view(-1) - 1], b["gt_relevance"])
    want = {**sp.retrieve(), **nd.retrieve()}
    assert set(got) == set(want) and all(abs(float(got[k]) - float(want[k])) < 1e-6 for k in want)
    with pytest.raises(ValueError):
        trainer.visdial_evaluate(loader, dict(n_gpus=1), 2, enc, chunk_size=7)
