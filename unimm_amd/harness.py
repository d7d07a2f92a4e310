"""Step harness: the caller-side contract of the hot path.

`forward(...)` is the equivalent of the reference's `train.forward` (train.py:30-177): flatten the
dataloader batch, optionally subsample `sample_size` sequences, call the encoder, combine the three
losses.  `generative_scores` / `scores_to_ranks` are the arithmetic of val_lm.py:124-149 and
utils/visdial_metrics.py:21-39."""
from __future__ import annotations

import torch


def _flat(x, keep):
    return x.reshape((-1,) + tuple(x.shape[-keep:])) if keep else x.reshape(-1)


class _WeightedSum3(torch.autograd.Function):
    """c_lm * lm + c_nsp * nsp + c_img * img for three one-element losses as ONE autograd node: two small launches forward,
    one backward, instead of the ~8 + ~8 of `c * x.mean() + ...` written out -- at the 30 sequences per GPU of an 8-way split
    those tiny launches sit between the forward and the backward graph replays, where nothing else runs."""

    @staticmethod
    def forward(ctx, lm, nsp, img, coeff):
        ctx.coeff = coeff
        return torch.cat((lm.reshape(1), nsp.reshape(1), img.reshape(1))).dot(coeff).reshape(())

    @staticmethod
    def backward(ctx, g):
        g3 = g * ctx.coeff
        return g3[0:1], g3[1:2], g3[2:3], None


_coeff_cache = {}


def combine_losses(lm_loss, nsp_loss, img_loss, c_lm, c_nsp, c_img):
    """loss = c_lm * lm.mean() + c_nsp * nsp.mean() + c_img * img.mean() (train.py:164-168)."""
    if lm_loss.numel() != 1 or nsp_loss.numel() != 1 or img_loss.numel() != 1 or not lm_loss.is_cuda:
        return c_lm * lm_loss.mean() + c_nsp * nsp_loss.mean() + c_img * img_loss.mean()
    key = (float(c_lm), float(c_nsp), float(c_img), lm_loss.device, lm_loss.dtype)
    coeff = _coeff_cache.get(key)
    if coeff is None:
        coeff = _coeff_cache[key] = torch.tensor(key[:3], dtype=lm_loss.dtype, device=lm_loss.device)
    return _WeightedSum3.apply(lm_loss, nsp_loss, img_loss, coeff)


def forward(dialog_encoder, batch, params, output_nsp_scores=False, output_lm_scores=False, sample_size=None,
            evaluation=False):
    tokens = _flat(batch["tokens"], 1)
    n = tokens.shape[0]
    idx = torch.randperm(n)[:sample_size] if sample_size else torch.arange(n)

    def take(key, keep):
        t = _flat(batch[key], keep)
        return t[idx.to(t.device)]

    kw = dict(sep_indices=take("sep_indices", 1), sep_len=take("hist_len", 0) + 1, token_type_ids=take("segments", 1),
              token_position_ids=take("positions", 1), masked_lm_labels=take("mask", 1),
              attention_mask=take("txt_attention_mask", 2), co_attention_mask=take("co_attention_mask", 2),
              image_attention_mask=take("image_mask", 1), lm_weight=take("weights", 1),
              nsp_weight=params.get("nsp_weight"), output_nsp_scores=output_nsp_scores,
              output_lm_scores=output_lm_scores)
    if not evaluation:
        kw.update(next_sentence_label=take("next_sentence_labels", 0), image_target=take("image_target", 2),
                  image_label=take("image_label", 1))
    res = dialog_encoder(tokens[idx.to(tokens.device)], take("image_feat", 2), take("image_loc", 2), **kw)
    lm_loss, img_loss, nsp_loss = res[:3]
    extra = list(res[3:])
    loss = None
    if not evaluation:
        loss = combine_losses(lm_loss, nsp_loss, img_loss, params["lm_loss_coeff"], params["nsp_loss_coeff"], params["img_loss_coeff"])
        lm_loss, nsp_loss, img_loss = lm_loss.mean(), nsp_loss.mean(), img_loss.mean()
    if output_nsp_scores or output_lm_scores:
        return (loss, lm_loss, nsp_loss, img_loss, *extra)
    return loss, lm_loss.item(), nsp_loss.item(), img_loss.item()


def generative_scores(lm_scores, masked_lm_labels, average=False):
    """Sequence log-likelihood from dense LM scores (val_lm.py:131-136; token-mean: val_avg_lm.py:135)."""
    b, t, v = lm_scores.shape
    nll = torch.nn.functional.cross_entropy(lm_scores.reshape(b * t, v).float(), masked_lm_labels.reshape(-1).to(lm_scores.device),
                                            ignore_index=-1, reduction="none").view(b, t)
    if average:
        return -(nll.sum(-1) / (masked_lm_labels != -1).sum(-1).to(nll.device))
    return -nll.sum(-1)


def scores_to_ranks(scores: torch.Tensor):
    """[batch, rounds, options] scores -> 1-based ranks (descending; utils/visdial_metrics.py:21-39).
    Index parity with the reference's CPU path includes TIES: the reference calls `scores.sort(1, descending=True)`,
    whose tie order is a property of torch's CPU sort (tests/golden/ranks.npz holds a 10-way tie: neither index
    order nor reverse index order).  A device sort orders ties differently (and unspecified), so device scores are
    ranked on the host with that same call - a few KB at evaluation time, never on the training path - and the
    ranks are returned on the scores' device."""
    b, r, o = scores.shape
    dev = scores.device
    order = scores.detach().reshape(-1, o).cpu().sort(1, descending=True)[1]
    ranks = torch.empty_like(order)
    ranks.scatter_(1, order, torch.arange(1, o + 1).expand_as(order))
    return ranks.view(b, r, o).to(dev)
