"""Seeded input builders shared by oracle/make_goldens.py and the tests -- TEST INFRASTRUCTURE.

Large random inputs are regenerated from a numpy PCG64 seed on both sides instead of being stored
in the fixtures (same numpy build in the container and on the GPU box)."""
from __future__ import annotations

import numpy as np

from oracle import masks as OM


def make_batch(rng, cfg, n_seq, T, R, modes, negs, share_image=True, type_ext=False):
    """Seeded batch with the A28 structure (via oracle.masks; G5 pins those against the reference)."""
    V = cfg["vocab_size"]
    rows = []
    for mode, neg in zip(modes, negs):
        budget = T - 8
        a = int(rng.integers(2, 8))
        n_ctx = int(rng.integers(2, 5))
        ctx_total = int(rng.integers(n_ctx * 2, max(n_ctx * 2 + 1, budget - 2 * (a + 1) - n_ctx - 1)))
        cuts = np.sort(rng.choice(np.arange(1, ctx_total), size=n_ctx - 1, replace=False)) if n_ctx > 1 else []
        lens = np.diff(np.concatenate([[0], cuts, [ctx_total]])).astype(int)
        utts = [list(rng.integers(1000 % V, V, size=int(l))) for l in lens] + [list(rng.integers(1000 % V, V, size=a))]
        n_tok = sum(len(u) for u in utts)
        draws = rng.random(n_tok)
        fn = OM.encode_gen if mode == "gen" else OM.encode_dis
        rows.append(fn(utts, start_segment=int(rng.integers(0, 2)), max_seq_len=T, mask_prob=0.15,
                       is_negative=int(neg), mask_draws=draws))
    cat = lambda k: np.concatenate([r[k] for r in rows], 0)
    seg = cat("segments")
    if type_ext:   # exercise token_type ids >= 2 (extension table, models/vilbert_dialog.py:337-350)
        seg = seg + (rng.random(seg.shape) < 0.2) * rng.integers(2, 12, size=seg.shape)
    n_img = 1 if share_image else n_seq
    feat = np.maximum(rng.standard_normal((n_img, R, cfg["v_feature_size"])), 0).astype(np.float32)
    feat[:, 0] = feat[:, 1:].mean(1)
    loc = rng.random((n_img, R, 5)).astype(np.float32)
    loc[:, 0] = [0, 0, 1, 1, 1]
    tgt = rng.standard_normal((n_img, R, cfg["v_target_size"])).astype(np.float32)
    tgt = np.exp(tgt) / np.exp(tgt).sum(-1, keepdims=True)
    rep = lambda x: np.repeat(x, n_seq // n_img, 0)
    img_mask = np.ones((n_seq, R), dtype=np.float32)
    if R > 8:
        img_mask[1::2, R - 3:] = 0          # some padded regions
    img_label = np.where(rng.random((n_seq, R)) < 0.15, 1, -1).astype(np.int64)
    img_label[:, 1] = 1
    img_label[:, 0] = 0
    co = np.repeat(cat("co_attention_mask")[:, None, :], R, 1)
    return dict(
        input_ids=cat("tokens"), token_type_ids=seg.astype(np.int64), position_ids=cat("positions"),
        attention_mask=cat("txt_attention_mask").astype(np.int64), co_attention_mask=co.astype(np.int64),
        masked_lm_labels=cat("labels"), lm_weight=cat("weights"),
        image_feat=rep(feat), image_loc=rep(loc), image_target=rep(tgt).astype(np.float32),
        image_attention_mask=img_mask, image_label=img_label,
        next_sentence_label=np.asarray(negs, dtype=np.int64),
        nsp_weight=np.asarray([[5.0, 1.0]], dtype=np.float32))




def block_inputs(seed=7, B=2, T=256, Rg=37, H=768, Hv=1024):
    """Inputs of golden G3 (single full-size blocks): activations + one gen and one dis mask."""
    rng = np.random.Generator(np.random.PCG64(seed))
    xt = rng.standard_normal((B, T, H)).astype(np.float32)
    xv = rng.standard_normal((B, Rg, Hv)).astype(np.float32)
    gen = OM.encode_gen([[5] * 40, [6] * 50, [7] * 9], mask_prob=0.0)
    dis = OM.encode_dis([[5] * 70, [6] * 30, [7] * 5], mask_prob=0.0)
    tmask = np.concatenate([gen["txt_attention_mask"], dis["txt_attention_mask"]], 0).astype(np.int64)
    co = np.repeat(np.concatenate([gen["co_attention_mask"], dis["co_attention_mask"]], 0)[:, None, :], Rg, 1)
    vmask = np.ones((B, Rg), dtype=np.float32)
    vmask[1, 30:] = 0
    rows = np.array([0, 1, 17, 40, 89, 95, 100, 104, 107, 110, 200, 255])
    return dict(xt=xt, xv=xv, tmask=tmask, co=co.astype(np.int64), vmask=vmask, rows=rows)


def embedding_inputs(seed=8, B=2, T=256, Rg=37):
    rng = np.random.Generator(np.random.PCG64(seed))
    return dict(ids=rng.integers(0, 30522, size=(B, T)), pos=rng.integers(0, 512, size=(B, T)),
                typ=rng.integers(0, 12, size=(B, T)),
                feat=np.maximum(rng.standard_normal((B, Rg, 2048)), 0).astype(np.float32),
                loc=rng.random((B, Rg, 5)).astype(np.float32))


def loss_inputs(seed=21, V=30522, Vi=1601, B=2, T=24, Rg=37):
    """Inputs of the loss golden: logits incl. the unlikelihood clamp regime (p_y -> 1 and p_y -> 0)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pred_t = (rng.standard_normal((B, T, V)) * 2).astype(np.float32)
    labels = np.full((B, T), -1, dtype=np.int64)
    weights = np.zeros((B, T), dtype=np.int64)
    sel = rng.random((B, T)) < 0.5
    labels[sel] = rng.integers(0, V, size=int(sel.sum()))
    weights[sel] = rng.choice([1, 1, -1, 2], size=int(sel.sum()))
    ul = np.argwhere(weights == -1)
    for (bi, ti) in ul[:3]:
        pred_t[bi, ti, labels[bi, ti]] = 60.0
    for (bi, ti) in ul[3:5]:
        pred_t[bi, ti, labels[bi, ti]] = -30.0
    pred_v = rng.standard_normal((B, Rg, Vi)).astype(np.float32)
    tgt = rng.standard_normal((B, Rg, Vi)).astype(np.float32)
    tgt = (np.exp(tgt) / np.exp(tgt).sum(-1, keepdims=True)).astype(np.float32)
    img_label = np.where(rng.random((B, Rg)) < 0.3, 1, -1).astype(np.int64)
    img_label[:, 0] = 0
    nsp = rng.standard_normal((B, 2)).astype(np.float32)
    nsl = (rng.random(B) < 0.8).astype(np.int64)
    nsl[0] = 0
    return dict(pred_t=pred_t, labels=labels, weights=weights, pred_v=pred_v, image_target=tgt,
                image_label=img_label, nsp=nsp, next_sentence_label=nsl)


def grad_sample_index(shape):
    """Sampling rule of the full-size gradient fixture (tests/golden/full_b6_grads.npz): 16 evenly spaced rows x
    every 4th column of a matrix, every 4th element of a vector."""
    if len(shape) == 1:
        return (slice(None, None, 4),)
    rows = np.unique(np.linspace(0, shape[0] - 1, 16).astype(np.int64))
    return (rows, slice(None, None, 4))
