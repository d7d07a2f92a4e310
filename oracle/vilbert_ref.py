"""CPU oracle for the UniMM-UL hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch, functional (state-dict driven) PyTorch fp32 restatement of the reference's
two-stream ViLBERT forward + its three losses.  It is the checker for the HIP path and the
`cpu_baseline` leg of bench.py; nothing under `unimm_amd/` may import it.

Parity status: PINNED.  `oracle/make_goldens.py` runs the reference's own modules (imported from
/root/reference in the build container) on seeded inputs and commits the outputs under
`tests/golden/`; `tests/test_oracle_golden.py` checks this file against them (<=2e-5 fp32).

Every function cites the reference lines it restates (paths relative to /root/reference).
Tensors are fp32 on CPU; parameter names are the reference's `state_dict` keys.
"""
from __future__ import annotations

import json
import math
from types import SimpleNamespace
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# defaults of BertConfig.__init__ that the JSON does not override (models/vilbert_dialog.py:138-169)
_CFG_DEFAULTS = dict(
    hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
    hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
    max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02,
    v_feature_size=2048, v_target_size=1601, v_hidden_size=768, v_num_hidden_layers=3,
    v_num_attention_heads=12, v_intermediate_size=3072, bi_hidden_size=1024,
    bi_num_attention_heads=16, v_attention_probs_dropout_prob=0.1, v_hidden_act="gelu",
    v_hidden_dropout_prob=0.1, v_initializer_range=0.2, v_biattention_id=[0, 1],
    t_biattention_id=[10, 11], predict_feature=False, fast_mode=False, fixed_v_layer=0,
    fixed_t_layer=0, in_batch_pairs=False, fusion_method="mul", intra_gate=False,
    with_coattention=True, vocab_size=-1,
)


def make_config(src) -> SimpleNamespace:
    """dict or JSON path -> attribute bag (models/vilbert_dialog.py:249-262: defaults, then JSON keys)."""
    if isinstance(src, str):
        with open(src, "r", encoding="utf-8") as f:
            src = json.load(f)
    d = dict(_CFG_DEFAULTS)
    d.update(src)
    return SimpleNamespace(**d)


# --------------------------------------------------------------------------------------------
# parameter inventory + seeded init (names: SURVEY.md 8b; init: models/vilbert_dialog.py:1110-1121)
# --------------------------------------------------------------------------------------------
def param_shapes(cfg) -> Dict[str, tuple]:
    H, Hv, Hb = cfg.hidden_size, cfg.v_hidden_size, cfg.bi_hidden_size
    I, Iv = cfg.intermediate_size, cfg.v_intermediate_size
    s: Dict[str, tuple] = {}

    def lin(name, out_f, in_f, bias=True):
        s[name + ".weight"] = (out_f, in_f)
        if bias:
            s[name + ".bias"] = (out_f,)

    def ln(name, n):
        s[name + ".weight"] = (n,)
        s[name + ".bias"] = (n,)

    e = "bert.embeddings."
    s[e + "word_embeddings.weight"] = (cfg.vocab_size, H)
    s[e + "position_embeddings.weight"] = (cfg.max_position_embeddings, H)
    s[e + "token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
    s[e + "token_type_embeddings_extension.weight"] = (10, H)
    s[e + "sep_embeddings.weight"] = (50, H)
    ln(e + "LayerNorm", H)
    v = "bert.v_embeddings."
    lin(v + "image_embeddings", Hv, cfg.v_feature_size)
    lin(v + "image_location_embeddings", Hv, 5)
    ln(v + "LayerNorm", Hv)
    for i in range(cfg.num_hidden_layers):
        p = f"bert.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            lin(p + "attention.self." + n, H, H)
        lin(p + "attention.output.dense", H, H)
        ln(p + "attention.output.LayerNorm", H)
        lin(p + "intermediate.dense", I, H)
        lin(p + "output.dense", H, I)
        ln(p + "output.LayerNorm", H)
    for i in range(cfg.v_num_hidden_layers):
        p = f"bert.encoder.v_layer.{i}."
        for n in ("query", "key", "value"):
            lin(p + "attention.self." + n, Hv, Hv)
        lin(p + "attention.output.dense", Hv, Hv)
        ln(p + "attention.output.LayerNorm", Hv)
        lin(p + "intermediate.dense", Iv, Hv)
        lin(p + "output.dense", Hv, Iv)
        ln(p + "output.LayerNorm", Hv)
    for i in range(len(cfg.v_biattention_id)):
        p = f"bert.encoder.c_layer.{i}."
        for n in ("query1", "key1", "value1"):
            lin(p + "biattention." + n, Hb, Hv)
        for n in ("query2", "key2", "value2"):
            lin(p + "biattention." + n, Hb, H)
        lin(p + "biOutput.dense1", Hv, Hb)
        ln(p + "biOutput.LayerNorm1", Hv)
        lin(p + "biOutput.q_dense1", Hv, Hb)
        lin(p + "biOutput.dense2", H, Hb)
        ln(p + "biOutput.LayerNorm2", H)
        lin(p + "biOutput.q_dense2", H, Hb)
        lin(p + "v_intermediate.dense", Iv, Hv)
        lin(p + "v_output.dense", Hv, Iv)
        ln(p + "v_output.LayerNorm", Hv)
        lin(p + "t_intermediate.dense", I, H)
        lin(p + "t_output.dense", H, I)
        ln(p + "t_output.LayerNorm", H)
    lin("bert.t_pooler.dense", Hb, H)
    lin("bert.v_pooler.dense", Hb, Hv)
    s["cls.predictions.bias"] = (cfg.vocab_size,)
    lin("cls.predictions.transform.dense", H, H)
    ln("cls.predictions.transform.LayerNorm", H)
    s["cls.predictions.decoder.weight"] = (cfg.vocab_size, H)  # tied to word_embeddings (:1020)
    lin("cls.bi_seq_relationship", 2, Hb)
    lin("cls.imagePredictions.transform.dense", Hv, Hv)
    ln("cls.imagePredictions.transform.LayerNorm", Hv)
    lin("cls.imagePredictions.decoder", cfg.v_target_size, Hv)
    return s


TIED = ("cls.predictions.decoder.weight", "bert.embeddings.word_embeddings.weight")


def init_state_dict(cfg, seed: int = 0, perturb: bool = True) -> SD:
    """Seeded weights from numpy's PCG64 stream (stable across machines, unlike torch.randn).
    `perturb` moves biases / LayerNorm affine off their 0/1 init
    (models/vilbert_dialog.py:1117-1121) so that parity tests exercise them."""
    import numpy as np
    g = np.random.Generator(np.random.PCG64(seed))

    def normal(shape, std):
        return torch.from_numpy(g.standard_normal(shape, dtype=np.float32) * np.float32(std))

    sd: SD = {}
    for name, shape in param_shapes(cfg).items():
        if name == TIED[0]:
            continue
        is_ln = "LayerNorm" in name
        if name.endswith(".weight") and not is_ln:
            t = normal(shape, cfg.initializer_range)
        elif is_ln and name.endswith(".weight"):
            t = torch.ones(shape) + (normal(shape, 0.1) if perturb else 0.0)
        else:  # biases
            t = normal(shape, 0.02) if perturb else torch.zeros(shape)
        sd[name] = t.float().contiguous()
    sd[TIED[0]] = sd[TIED[1]]
    return sd


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------
def erf_gelu(x):
    """models/vilbert_dialog.py:115-121 -- exact erf form."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _ln(sd, name, x):
    """torch.nn.LayerNorm, eps 1e-12 (models/vilbert_dialog.py:279,322)."""
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-12)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


class _Drop:
    """Dropout hook.  `fn(site, x, p)` lets a test inject the HIP path's counter-based masks;
    None = eval mode (identity)."""

    def __init__(self, fn: Optional[Callable] = None):
        self.fn = fn

    def __call__(self, site: str, x, p: float):
        if self.fn is None or p == 0.0:
            return x
        return self.fn(site, x, p)


def _heads(x, n):
    b, t, h = x.shape
    return x.view(b, t, n, h // n).permute(0, 2, 1, 3)


def _merge(x):
    b, n, t, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(b, t, n * d)


def attention_core(q, k, v, add_mask, n_heads, drop, site, p_drop):
    """softmax(QK^T/sqrt(d) + mask) . V   (models/vilbert_dialog.py:390-410, 519-539, 681-721)."""
    qh, kh, vh = _heads(q, n_heads), _heads(k, n_heads), _heads(v, n_heads)
    s = torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(qh.shape[-1])
    if add_mask is not None:
        s = s + add_mask
    p = torch.softmax(s, dim=-1)
    p = drop(site, p, p_drop)
    return _merge(torch.matmul(p, vh))


# --------------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------------
def text_layer(sd, cfg, i, x, add_mask, drop):
    """BertLayer (models/vilbert_dialog.py:385-483)."""
    p = f"bert.encoder.layer.{i}."
    a = p + "attention.self."
    ctx = attention_core(_lin(sd, a + "query", x), _lin(sd, a + "key", x), _lin(sd, a + "value", x),
                         add_mask, cfg.num_attention_heads, drop, p + "attn", cfg.attention_probs_dropout_prob)
    h = drop(p + "so", _lin(sd, p + "attention.output.dense", ctx), cfg.hidden_dropout_prob)
    x1 = _ln(sd, p + "attention.output.LayerNorm", h + x)
    u = erf_gelu(_lin(sd, p + "intermediate.dense", x1))
    h2 = drop(p + "out", _lin(sd, p + "output.dense", u), cfg.hidden_dropout_prob)
    return _ln(sd, p + "output.LayerNorm", h2 + x1)


def image_layer(sd, cfg, i, x, add_mask, drop):
    """BertImageLayer (models/vilbert_dialog.py:514-612)."""
    p = f"bert.encoder.v_layer.{i}."
    a = p + "attention.self."
    ctx = attention_core(_lin(sd, a + "query", x), _lin(sd, a + "key", x), _lin(sd, a + "value", x),
                         add_mask, cfg.v_num_attention_heads, drop, p + "attn",
                         cfg.v_attention_probs_dropout_prob)
    h = drop(p + "so", _lin(sd, p + "attention.output.dense", ctx), cfg.v_hidden_dropout_prob)
    x1 = _ln(sd, p + "attention.output.LayerNorm", h + x)
    u = erf_gelu(_lin(sd, p + "intermediate.dense", x1))
    h2 = drop(p + "out", _lin(sd, p + "output.dense", u), cfg.v_hidden_dropout_prob)
    return _ln(sd, p + "output.LayerNorm", h2 + x1)


def connection_layer(sd, cfg, i, xv, v_add_mask, xt, co_add_mask, drop):
    """BertConnectionLayer = BertBiAttention + BertBiOutput + two FFNs
    (models/vilbert_dialog.py:655-783).  Direction 1 (text queries, image keys) adds only the
    image mask (:683; co-mask skipped since attended_all_tensor1=True, :686); direction 2 (image
    queries, text keys) adds only the co-attention mask (:706 commented, :708-709)."""
    p = f"bert.encoder.c_layer.{i}."
    b = p + "biattention."
    nh = cfg.bi_num_attention_heads
    q1, k1, v1 = (_lin(sd, b + n, xv) for n in ("query1", "key1", "value1"))
    q2, k2, v2 = (_lin(sd, b + n, xt) for n in ("query2", "key2", "value2"))
    ctx_t = attention_core(q2, k1, v1, v_add_mask, nh, drop, p + "attn1", cfg.v_attention_probs_dropout_prob)
    ctx_v = attention_core(q1, k2, v2, co_add_mask, nh, drop, p + "attn2", cfg.attention_probs_dropout_prob)
    # BertBiOutput.forward(bi_output2, input1, bi_output1, input2) -- note the call-site order (:775)
    o = p + "biOutput."
    hv = drop(p + "bo1", _lin(sd, o + "dense1", ctx_v), cfg.v_hidden_dropout_prob)
    ht = drop(p + "bo2", _lin(sd, o + "dense2", ctx_t), cfg.hidden_dropout_prob)
    av = _ln(sd, o + "LayerNorm1", hv + xv)
    at = _ln(sd, o + "LayerNorm2", ht + xt)
    uv = erf_gelu(_lin(sd, p + "v_intermediate.dense", av))
    ov = _ln(sd, p + "v_output.LayerNorm",
             drop(p + "vout", _lin(sd, p + "v_output.dense", uv), cfg.v_hidden_dropout_prob) + av)
    ut = erf_gelu(_lin(sd, p + "t_intermediate.dense", at))
    ot = _ln(sd, p + "t_output.LayerNorm",
             drop(p + "tout", _lin(sd, p + "t_output.dense", ut), cfg.hidden_dropout_prob) + at)
    return ov, ot


def text_embeddings(sd, cfg, input_ids, token_type_ids, position_ids, drop):
    """BertEmbeddingsDialog.forward (models/vilbert_dialog.py:326-356); sep_indices/sep_len and the
    sep/pe tables are unused there."""
    e = "bert.embeddings."
    b, t = input_ids.shape
    if position_ids is None:
        position_ids = torch.arange(t).unsqueeze(0).expand(b, t)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    tv = cfg.type_vocab_size
    is_ext = token_type_ids >= tv
    base = F.embedding(torch.where(is_ext, torch.zeros_like(token_type_ids), token_type_ids),
                       sd[e + "token_type_embeddings.weight"])
    ext = F.embedding(torch.where(is_ext, token_type_ids - tv, torch.zeros_like(token_type_ids)),
                      sd[e + "token_type_embeddings_extension.weight"])
    typ = torch.where(is_ext.unsqueeze(-1), ext, base)
    x = F.embedding(input_ids, sd[e + "word_embeddings.weight"]) + \
        F.embedding(position_ids, sd[e + "position_embeddings.weight"]) + typ
    return drop("emb_t", _ln(sd, e + "LayerNorm", x), cfg.hidden_dropout_prob)


def image_embeddings(sd, cfg, feat, loc, drop):
    """BertImageEmbeddings.forward (models/vilbert_dialog.py:1487-1493)."""
    v = "bert.v_embeddings."
    x = _lin(sd, v + "image_embeddings", feat) + _lin(sd, v + "image_location_embeddings", loc)
    return drop("emb_v", _ln(sd, v + "LayerNorm", x), cfg.hidden_dropout_prob)


def additive(mask):
    """(1 - m) * -10000 in fp32 (models/vilbert_dialog.py:1415-1431)."""
    return (1.0 - mask.to(torch.float32)) * -10000.0


def trunk(sd, cfg, input_ids, image_feat, image_loc, token_type_ids=None, position_ids=None,
          attention_mask=None, image_attention_mask=None, co_attention_mask=None, drop=None):
    """BertModel.forward (models/vilbert_dialog.py:1359-1472) with BertEncoder's schedule (:842-929)."""
    drop = drop or _Drop(None)
    b, t = input_ids.shape
    r = image_feat.shape[1]
    if attention_mask is None:
        attention_mask = torch.ones(b, t)
    if image_attention_mask is None:
        image_attention_mask = torch.ones(b, r)
    if co_attention_mask is None:
        co_attention_mask = torch.ones(b, r, t)
    assert co_attention_mask.dim() == 3
    if attention_mask.dim() == 3:
        t_add = additive(attention_mask)[:, None, :, :]
    elif attention_mask.dim() == 2:
        t_add = additive(attention_mask)[:, None, None, :]
    else:
        raise ValueError("attention_mask must be 2-D or 3-D")
    if image_attention_mask.dim() == 3:
        v_add = additive(image_attention_mask)[:, None, :, :]
    elif image_attention_mask.dim() == 2:
        v_add = additive(image_attention_mask)[:, None, None, :]
    else:
        raise ValueError("image_attention_mask must be 2-D or 3-D")
    co_add = additive(co_attention_mask).unsqueeze(1)

    xt = text_embeddings(sd, cfg, input_ids, token_type_ids, position_ids, drop)
    xv = image_embeddings(sd, cfg, image_feat, image_loc, drop)

    v_start = t_start = 0
    for count, (v_end, t_end) in enumerate(zip(cfg.v_biattention_id, cfg.t_biattention_id)):
        assert cfg.fixed_t_layer <= t_end                              # :847-848
        assert cfg.fixed_v_layer <= v_end
        for i in range(v_start, cfg.fixed_v_layer):                    # frozen lower layers run without a graph (:850-857)
            with torch.no_grad():
                xv = image_layer(sd, cfg, i, xv, v_add, drop)
            v_start = cfg.fixed_v_layer
        for i in range(v_start, v_end):
            xv = image_layer(sd, cfg, i, xv, v_add, drop)
        for i in range(t_start, cfg.fixed_t_layer):                    # (:864-869)
            with torch.no_grad():
                xt = text_layer(sd, cfg, i, xt, t_add, drop)
            t_start = cfg.fixed_t_layer
        for i in range(t_start, t_end):
            xt = text_layer(sd, cfg, i, xt, t_add, drop)
        if cfg.with_coattention:
            xv, xt = connection_layer(sd, cfg, count, xv, v_add, xt, co_add, drop)
        v_start, t_start = v_end, t_end
    for i in range(v_start, cfg.v_num_hidden_layers):
        xv = image_layer(sd, cfg, i, xv, v_add, drop)
    for i in range(t_start, cfg.num_hidden_layers):
        xt = text_layer(sd, cfg, i, xt, t_add, drop)

    pooled_t = torch.relu(_lin(sd, "bert.t_pooler.dense", xt[:, 0]))   # :946-952
    pooled_v = torch.relu(_lin(sd, "bert.v_pooler.dense", xv[:, 0]))   # :961-967
    return xt, xv, pooled_t, pooled_v


def heads(sd, cfg, seq_t, seq_v, pooled_t, pooled_v, drop=None):
    """BertPreTrainingHeads.forward (models/vilbert_dialog.py:1058-1073), fusion 'mul'/'sum'."""
    drop = drop or _Drop(None)
    fused = pooled_t * pooled_v if cfg.fusion_method == "mul" else pooled_t + pooled_v
    fused = drop("fuse", fused, 0.1)
    ht = _ln(sd, "cls.predictions.transform.LayerNorm",
             erf_gelu(_lin(sd, "cls.predictions.transform.dense", seq_t)))
    pred_t = F.linear(ht, sd["cls.predictions.decoder.weight"]) + sd["cls.predictions.bias"]
    nsp = _lin(sd, "cls.bi_seq_relationship", fused)
    hv = _ln(sd, "cls.imagePredictions.transform.LayerNorm",
             erf_gelu(_lin(sd, "cls.imagePredictions.transform.dense", seq_v)))
    pred_v = _lin(sd, "cls.imagePredictions.decoder", hv)
    return pred_t, pred_v, nsp


# --------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------
def mlm_ul_loss(pred_t, labels, lm_weight, clamp_min=1e-6):
    """Token-level likelihood / unlikelihood loss (models/vilbert_dialog.py:1577-1604).
    w>0: -w*log p_y ; w==-1: -log clamp(1-p_y, 1e-6) ; normalised by #{w != 0}.
    lm_weight None: CrossEntropy(ignore_index=-1), mean."""
    v = pred_t.shape[-1]
    z = pred_t.reshape(-1, v)
    y = labels.reshape(-1)
    if lm_weight is None:
        return F.cross_entropy(z, y, ignore_index=-1)
    w = lm_weight.reshape(-1)
    pos, neg = w > 0, w == -1
    zl, zu = z[pos], z[neg]
    l_loss = F.nll_loss(F.log_softmax(zl, -1), y[pos], ignore_index=-1, reduction="none") * w[pos].float()
    u_logp = torch.log(torch.clamp(1.0 - F.softmax(zu, -1), min=clamp_min))
    u_loss = F.nll_loss(u_logp, y[neg], ignore_index=-1, reduction="none")
    return (l_loss.sum() + u_loss.sum()) / (lm_weight != 0).sum()


def image_kl_loss(pred_v, image_target, image_label):
    """Masked-region KL (models/vilbert_dialog.py:1569-1574); divisor max(#[label==1], 0) as written."""
    kl = F.kl_div(F.log_softmax(pred_v, dim=2), image_target, reduction="none")
    sel = (image_label == 1)
    return torch.sum(kl * sel.unsqueeze(2).float()) / max(torch.sum(sel), 0)


def image_mse_loss(pred_v, image_target, image_label):
    """predict_feature branch (models/vilbert_dialog.py:1562-1566): MSE(reduction='none') on the regions with label 1,
    divided by max(#selected elements, 1)."""
    mse = F.mse_loss(pred_v, image_target, reduction="none")
    sel = (image_label == 1).unsqueeze(2)
    return torch.sum(mse * sel.float()) / max(torch.sum(sel.expand_as(mse)), 1)


def nsp_loss(nsp_scores, next_sentence_label, nsp_weight=None):
    """Weighted 2-way CE (models/vilbert_dialog.py:1605-1621)."""
    if nsp_weight is None:
        nsp_weight = torch.tensor([1.0, 1.0])
    w = nsp_weight.squeeze().float()
    w = w / w[0]
    return F.cross_entropy(nsp_scores.view(-1, 2), next_sentence_label.view(-1), weight=w, reduction="mean")


def forward(sd, cfg, input_ids, image_feat, image_loc, token_type_ids=None, position_ids=None,
            attention_mask=None, image_attention_mask=None, co_attention_mask=None,
            masked_lm_labels=None, image_label=None, image_target=None, next_sentence_label=None,
            nsp_weight=None, lm_weight=None, drop_fn=None):
    """BertForMultiModalPreTraining.forward (models/vilbert_dialog.py:1519-1626).
    Train branch (all of labels / nsp label / image_target given) returns a dict with the three
    shape-[1] losses + seq_out_t, pred_t, nsp; inference returns pred_t, pred_v, nsp, seq_out_t."""
    drop = _Drop(drop_fn)
    seq_t, seq_v, pt, pv = trunk(sd, cfg, input_ids, image_feat, image_loc, token_type_ids, position_ids,
                                 attention_mask, image_attention_mask, co_attention_mask, drop)
    pred_t, pred_v, nsp = heads(sd, cfg, seq_t, seq_v, pt, pv, drop)
    out = dict(pred_t=pred_t, pred_v=pred_v, nsp=nsp, seq_out_t=seq_t, seq_out_v=seq_v)
    if masked_lm_labels is not None and next_sentence_label is not None and image_target is not None:
        img_fn = image_mse_loss if getattr(cfg, "predict_feature", False) else image_kl_loss
        out["img_loss"] = img_fn(pred_v, image_target, image_label).unsqueeze(0)
        out["lm_loss"] = mlm_ul_loss(pred_t, masked_lm_labels, lm_weight).unsqueeze(0)
        out["nsp_loss"] = nsp_loss(nsp, next_sentence_label, nsp_weight).unsqueeze(0)
    return out


# --------------------------------------------------------------------------------------------
# generative scoring + ranks (val_lm.py:124-149, utils/visdial_metrics.py:21-39)
# --------------------------------------------------------------------------------------------
def sequence_log_likelihood(pred_t, labels, average=False):
    """-sum_t CE(pred_t, labels, ignore_index=-1) per sequence (val_lm.py:131-136); the
    token-mean variant is val_avg_lm.py:135."""
    b, t, v = pred_t.shape
    nll = F.cross_entropy(pred_t.reshape(b * t, v), labels.reshape(-1), ignore_index=-1,
                          reduction="none").view(b, t)
    if average:
        return -(nll.sum(-1) / (labels != -1).sum(-1))
    return -nll.sum(-1)


def scores_to_ranks(scores):
    """1-based rank of every option under a descending sort, ties in sort order
    (utils/visdial_metrics.py:21-39)."""
    b, r, o = scores.shape
    idx = scores.reshape(-1, o).sort(1, descending=True)[1]
    ranks = torch.empty_like(idx)
    ranks.scatter_(1, idx, torch.arange(1, o + 1).expand_as(idx))
    return ranks.view(b, r, o)
