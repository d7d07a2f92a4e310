"""Generative scoring with the shared dialog context computed once (val_lm.py:52-121).

val_lm.py scores 100 candidate answers per dialog round: the reference builds 100 full sequences
`[CLS] caption [SEP] q1 [SEP] a1 ... q_r [SEP] candidate [SEP] <[MASK] copy of the candidate>` that differ ONLY in the candidate
and runs the whole two-stream encoder on each.  Under the generative mask (utils/data_utils.py:199-210; `oracle/masks.py`
restates it) with L = length incl. candidate + [SEP], n = len(candidate) + 1, c = L - n:

    row 0 (CLS)            attends [0, L + n)
    rows [1, c)  (context) attend  [1, c)            -- never column 0, never the candidate
    rows [c, L)  (answer)  attend  [1, row]
    rows [L, L+n) (copies) attend  [1, row - n) + self
    regions                attend  [1, c)            (co-attention mask)
    every text row         attends all regions

so in EVERY layer the context rows [1, c) and the whole image stream see nothing that depends on the candidate: they are the
same for the 100 sequences of a round.  This module runs them once per group (S rows: c - 1 text rows + the 37 regions) and
only the rows that do depend on the candidate -- row 0, the answer rows and the copy rows: 1 + 2 n of ~140 -- per sequence
(P rows).  All GEMMs / LayerNorms run on one packed row matrix [S_0 | S_1 | ... | P_0 | P_1 | ...]; attention is

    text self-attention     S_g x S_g  (all-ones mask)               one launch over the groups
                            P_b x [CLS_b | S_g(b) | answer_b, copy_b]  one launch over the sequences: the group's K / V rows are
                                                                     SPLICED into each sequence's keys in their original
                                                                     positions (unimm_attn_args.ks_*), masks = the P rows of
                                                                     the sequence's own packed mask
    regions attend text     regions_g x S_g                          (co-attention mask = the context = all of S_g)
    text attends regions    (S_g | P_b) x regions_g(b)               one launch over groups + sequences

Inference only (no tape, no dropout), bf16 engine.  Results equal the per-sequence path up to the summation order inside the
attention kernels (tests/test_gpu_fullsize.py).  The decoder runs on the copy rows only, as before.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import lib as L
from . import params as PM
from .inputs import DialogMaskSpec

BF16, F32 = torch.bfloat16, torch.float32


def _rup(x, m):
    return (x + m - 1) // m * m


class SharedContextPlan:
    """Host-side row bookkeeping of one call: which padded rows form the shared (S) and private (P) blocks."""

    def __init__(self, groups, c, n, length, T, R):
        groups = np.asarray(groups, dtype=np.int64).reshape(-1)
        c, n, length = (np.asarray(a, dtype=np.int64).reshape(-1) for a in (c, n, length))
        B = groups.shape[0]
        if not (c.shape[0] == n.shape[0] == length.shape[0] == B):
            raise ValueError("shared_context: one group id / context length / answer length per sequence")
        uniq, first, inv = np.unique(groups, return_index=True, return_inverse=True)
        order = np.argsort(first, kind="stable")                   # groups in order of first appearance
        rank = np.empty_like(order)
        rank[order] = np.arange(order.shape[0])
        self.gid = rank[inv]                                       # [B] group index 0..G-1
        self.rep = first[order]                                    # [G] representative sequence of each group
        G = self.rep.shape[0]
        cg = c[self.rep]
        if (c != cg[self.gid]).any():
            bad = int(np.nonzero(c != cg[self.gid])[0][0])
            raise ValueError(f"shared_context: sequence {bad} has context length {int(c[bad])}, its group's first sequence {int(cg[self.gid[bad]])}")
        if (c < 2).any() or (n < 1).any() or (length != c + 2 * n).any():
            raise ValueError("shared_context: every sequence must be generative-mode [CLS] context answer [SEP] + copy (length = c + 2 n)")
        if (length > T).any():
            raise ValueError("shared_context: truncated copy blocks (length > T) are not supported on this path")
        self.B, self.G, self.T, self.R = B, G, T, R
        self.c, self.n, self.length = c, n, length
        self.s_len = (cg - 1).astype(np.int64)                     # context rows [1, c) of the representative
        self.s_off = np.concatenate([[0], np.cumsum(self.s_len)[:-1]]).astype(np.int64)
        self.S = int(self.s_len.sum())
        self.p_len = (1 + 2 * n).astype(np.int64)                  # row 0 + answer rows + copy rows
        self.p_off = (self.S + np.concatenate([[0], np.cumsum(self.p_len)[:-1]])).astype(np.int64)
        self.P = int(self.p_len.sum())
        self.M = self.S + self.P
        self.pmax = int(self.p_len.max())
        if self.pmax > 32:
            raise ValueError(f"shared_context: a candidate with {self.pmax} private rows (answer of {int(n.max()) - 1} tokens) exceeds one 32-row query tile")
        # packed row -> padded row (b * T + t); vectorised (this runs between the header's arrival and the first launch)
        rows = np.empty(self.M, dtype=np.int64)
        sg = np.repeat(np.arange(G), self.s_len)                   # group of every S row
        rows[:self.S] = self.rep[sg] * T + 1 + (np.arange(self.S) - self.s_off[sg])
        pb = np.repeat(np.arange(B), self.p_len)                   # sequence of every P row
        r = np.arange(self.P) - (self.p_off[pb] - self.S)          # index inside the sequence's private block
        pos = np.where(r == 0, 0, c[pb] + r - 1)                   # row 0, then rows c .. length - 1
        rows[self.S:] = pb * T + pos
        prow = np.zeros((B, 32), dtype=np.int64)                   # [B, 32] position inside the sequence of private row r (pad: 0)
        prow[pb, r] = pos
        self.rows, self.prow = rows, prow
        # decoded rows: the copy rows = the last n private rows of every sequence, in sequence order
        self.n_lm = int(n.sum())
        lb = np.repeat(np.arange(B), n)
        k = np.arange(self.n_lm) - np.repeat(np.cumsum(n) - n, n)  # index inside the sequence's copy block
        self.lm_idx = self.p_off[lb] + 1 + n[lb] + k               # packed row of each decoded row
        self.lm_pos = lb * T + (length[lb] - n[lb]) + k            # its padded position b * T + t


def _i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=torch.int32, non_blocking=True)


def _i64(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=torch.int64, non_blocking=True)


def forward_shared(eng, inp: dict, groups, want_nsp=True):
    """-> dict(rownll [n_lm] fp32, lm_seq [n_lm] int32 (sequence of each decoded row), nsp [B, 2] | None, ok [B] bool, plan).
    `ok[b]` is False where sequence b's context tokens / segments / positions differ from its group's representative (checked on
    the device, no host synchronisation): the caller poisons those scores."""
    return eng._on_text_stream(_forward_shared, eng, inp, groups, want_nsp)


def _forward_shared(eng, inp, groups, want_nsp):
    cfg = eng.cfg
    dev = eng.arena.device
    if getattr(eng, "compute_dtype", "bf16") != "bf16":
        raise NotImplementedError("shared-context scoring runs on the bf16 engine")
    if not cfg.with_coattention:
        raise NotImplementedError("shared-context scoring needs the connection layers (with_coattention)")
    eng.refresh_weights()
    ids = inp["input_ids"]
    B, T = ids.shape
    feat = inp["image_feat"]
    R = feat.shape[1]
    H, Hv, Hb = cfg.hidden_size, cfg.v_hidden_size, cfg.bi_hidden_size
    labels = inp.get("masked_lm_labels")
    if labels is None:
        raise ValueError("shared_context scoring needs masked_lm_labels (the copy rows)")
    am = inp.get("attention_mask")
    spec = am if isinstance(am, DialogMaskSpec) else None

    # ---- masks + the per-sequence structure (c, n, length) -------------------------------------------------------------
    eng._dev_masks = []
    if spec is not None:
        if len(spec) != B or int(spec.mode.min()) != 1:
            raise ValueError("shared_context: one generative-mode descriptor per sequence")
        twords, _ = L.mask_synth(*spec.to_device(dev), T)
        nw = twords.shape[-1]
        n_h = spec.answer.astype(np.int64)
        length_h = spec.length.astype(np.int64) + n_h
    else:
        if am is None or am.dim() != 3 or inp.get("co_attention_mask") is None:
            raise ValueError("shared_context scoring needs the dense [B, T, T] generative attention_mask and the co_attention_mask")
        tmask = eng._pack_mask(am, dev, T)
        comask = eng._pack_mask(inp["co_attention_mask"], dev, R)
        twords, nw = tmask[0], tmask[0].shape[-1]
        lab32 = eng._i32(labels.reshape(B, T), dev)
        header = L.plan_lengths(tmask, comask, R, lab32, None, None, B, T)
        hh = header.tolist()                                       # the call's one host synchronisation
        length_h, n_h = np.asarray(hh[:B]), np.asarray(hh[B:2 * B])
    eng._dev_masks = []
    c_h = np.asarray(length_h) - 2 * np.asarray(n_h)
    plan = SharedContextPlan(groups.cpu().numpy() if torch.is_tensor(groups) else groups, c_h, n_h, length_h, T, R)
    G, M, S = plan.G, plan.M, plan.S
    eng._step_rows = M
    eng.last_plan = None

    rows = _i64(plan.rows, dev)
    s_off, s_len = _i32(plan.s_off, dev), _i32(plan.s_len, dev)
    p_off, p_len = _i32(plan.p_off, dev), _i32(plan.p_len, dev)
    gid = _i64(plan.gid, dev)
    rep = _i64(plan.rep, dev)
    ks_off, ks_len = s_off[gid], s_len[gid]                        # [B]: the group's rows, per sequence
    # item order of the candidates' launches: most keys first (unimm_attn_args.order: the tail of a launch is its shortest items)
    p_ord = _i32(np.argsort(-(plan.s_len[plan.gid] + plan.p_len), kind="stable"), dev)
    # P-row masks: rows {0, c .. length) of each sequence's packed mask, key positions unchanged
    prow = _i64(plan.prow, dev)                                    # [B, 32]
    pwords = twords.view(B, T, nw)[torch.arange(B, device=dev)[:, None], prow].contiguous()        # [B, 32, nw]
    ones_t = torch.full((G, nw), -1, dtype=torch.int32, device=dev)                                # all-ones key masks (lengths bound them)

    # ---- is the context really shared?  (device-side check, folded into `ok`) -----------------------------------------
    tt, pp = inp.get("token_type_ids"), inp.get("position_ids")
    ids_d = ids.to(dev, non_blocking=True)
    col = torch.arange(T, device=dev)[None, :]
    c_d = _i64(plan.c, dev)[:, None]
    ctx_cols = (col >= 1) & (col < c_d)
    same = ((ids_d == ids_d[rep[gid]]) | ~ctx_cols).all(1)
    for t in (tt, pp):
        if t is not None:
            t_d = t.to(dev, non_blocking=True)
            same &= ((t_d == t_d[rep[gid]]) | ~ctx_cols).all(1)
    lab_d = labels.to(dev, non_blocking=True)
    len_d = _i64(plan.length, dev)[:, None]
    n_d = _i64(plan.n, dev)[:, None]
    copy_cols = (col >= len_d - n_d) & (col < len_d)
    same &= ((lab_d != -1) == copy_cols).all(1)                    # the labelled rows are exactly the copy rows
    if spec is None:
        # The S rows run with an all-ones mask bounded by their length and the regions attend them the same way: that IS the
        # generative mask's context block (utils/data_utils.py:199-210: rows and columns [1, c) fully connected, nothing else
        # visible to a context row; co-attention keys [1, c)) -- checked here instead of assumed: a dense mask that deviates in
        # the context block, or a non-generative row, gives NaN instead of a silently different score.
        colw = torch.arange(nw * 32, device=dev).view(1, nw, 32)
        bits = ((colw >= 1) & (colw < c_d[:, :, None])).to(torch.int64)                        # [B, nw, 32]
        expw = (bits << torch.arange(32, device=dev)).sum(-1)
        expw = torch.where(expw >= 2 ** 31, expw - 2 ** 32, expw).to(torch.int32)                 # the packed words of columns [1, c)
        same &= ((twords.view(B, T, nw) == expw[:, None, :]).all(-1) | ~ctx_cols).all(1)
        same &= (comask[0].view(B, -1, nw) == expw[:, None, :]).all(-1).all(1)
    img_idx = inp.get("image_index")
    if img_idx is not None:
        img_idx = img_idx.to(dev, dtype=torch.int64, non_blocking=True).reshape(-1)
        same &= img_idx == img_idx[rep[gid]]
        img_rows = img_idx[rep]                                    # [G] entry of the per-image tensors
    else:
        if feat.shape[0] != B:
            raise ValueError(f"image_feat has {feat.shape[0]} rows for {B} sequences and no image_index was given")
        img_rows = rep

    st = dict(train=False, tape=None)
    NO = L.NO_DROP
    # ---- image embedding, one per group (image stream) -----------------------------------------------------------------
    F = cfg.v_feature_size
    im = inp.get("image_attention_mask")
    feat_d = feat.to(dev, non_blocking=True)
    loc_d = inp["image_loc"].to(dev, non_blocking=True)
    if img_idx is None:                                            # per-sequence copies (val_lm.py:78-91 expands them): every member
        fv, lv = feat_d.reshape(B, -1), loc_d.reshape(B, -1)       # must carry its group's image -- compared in full on the device
        same &= (fv == fv[rep[gid]]).all(1) & (lv == lv[rep[gid]]).all(1)
    eng._to_img()
    with eng._img():
        featd = feat_d.index_select(0, img_rows).to(F32).contiguous().view(G * R, F)
        locd = loc_d.index_select(0, img_rows).to(F32).contiguous().view(G * R, 5)
        packed = torch.empty((G * R, eng.vemb_k), dtype=BF16, device=dev)
        L.pack_image(featd, locd, packed, G * R, F, eng.vemb_k)
        prev = torch.empty((G * R, Hv), dtype=F32, device=dev)
        L.gemm_nt(packed, eng.vemb_w, prev, bias=eng.vemb_b, M=G * R, N=Hv, K=eng.vemb_k)
        xv32, xv, _, _ = eng._layernorm(prev, "emb_v", False)
    if im is None:
        im = torch.ones((B, R), dtype=torch.uint8, device=dev)
    imd = im.to(dev, non_blocking=True)
    if imd.dim() == 2:
        same &= (imd == imd[rep[gid]]).all(1)                      # ... and its group's image key mask
    eng._dev_masks = []
    vmask = eng._pack_mask(imd.index_select(0, rep), dev, R)       # [G] image key masks
    eng._dev_masks = []
    if vmask[1] != 0:
        raise ValueError("shared_context: image_attention_mask must be a [B, R] key mask")
    nwv = vmask[0].shape[-1]

    # ---- text embeddings on the packed rows -----------------------------------------------------------------------------
    ids32 = eng._i32(ids.reshape(-1), dev)
    typ32 = eng._i32(tt.reshape(-1), dev) if tt is not None else torch.zeros(B * T, dtype=torch.int32, device=dev)
    pos32 = eng._i32(pp.reshape(-1), dev) if pp is not None else torch.arange(T, dtype=torch.int32, device=dev).repeat(B)
    gmm, bta, _, _ = eng.ln["emb_t"]
    xt = torch.empty((M, H), dtype=BF16, device=dev)
    xt32 = torch.empty((M, H), dtype=F32, device=dev)
    L.embed_fwd(ids32, pos32, typ32, eng.tab["word"], eng.tab["pos"], eng.tab["type"], eng.tab["ext"], gmm, bta, xt32, xt, M, H,
                cfg.type_vocab_size, rows=rows)

    heads, D = cfg.num_attention_heads, H // cfg.num_attention_heads
    nh, Db = cfg.bi_num_attention_heads, Hb // cfg.bi_num_attention_heads
    # items of the text-attends-regions launch: the groups' S blocks, then the sequences' P blocks
    it_off, it_len = torch.cat([s_off, p_off]), torch.cat([s_len, p_len])
    it_img = torch.cat([torch.arange(G, device=dev), gid]).to(torch.int32)
    it_koff = it_img * R
    it_klen = torch.full_like(it_koff, R)
    it_vwords = vmask[0].view(G, nwv)[it_img.long()].contiguous()                                  # [G + B, nwv]

    def text_block(key, x32, x):
        """BertLayer (models/vilbert_dialog.py:385-483) on the packed rows, inference."""
        qkv_l, so, ff1, ff2 = (eng.lin[key + s] for s in (".qkv", ".so", ".ff1", ".ff2"))
        qkv = eng._linear(x, qkv_l)
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
        ctx = torch.empty((M, H), dtype=BF16, device=dev)
        sc = 1.0 / math.sqrt(D)
        L.attn_fwd(q, k, v, ctx, None, ones_t, G, heads, T, T, D, sc, 0, nw, NO, qvar=(s_off, s_len), kvar=(s_off, s_len))
        L.attn_fwd(q, k, v, ctx, None, pwords, B, heads, 32, T, D, sc, nw, 32 * nw, NO, qvar=(p_off, p_len, None, p_ord),
                   kvar=(p_off, p_len), kshared=(ks_off, ks_len, 1))
        pre1 = eng._linear(ctx, so, L.EPI_BIAS_DROP_RESID, aux=x32, drop=NO, out_f32=True)
        x1_32, x1, _, _ = eng._layernorm(pre1, key + ".ln1", False, lazy=True)
        h = eng._linear(x1, ff1, L.EPI_BIAS_GELU)
        pre2 = eng._linear(h, ff2, L.EPI_BIAS_DROP_RESID, aux=x1_32, drop=NO, out_f32=True)
        x2_32, x2, _, _ = eng._layernorm(pre2, key + ".ln2", False, lazy=True)
        return x2_32, x2

    def conn_block(key, xv32, xv, xt32, xt):
        """BertConnectionLayer (models/vilbert_dialog.py:655-783), inference: the image half once per group."""
        lq1, lq2, d1, d2 = (eng.lin[key + s] for s in (".qkv1", ".qkv2", ".d1", ".d2"))
        vff1, vff2, tff1, tff2 = (eng.lin[key + s] for s in (".vff1", ".vff2", ".tff1", ".tff2"))
        sc = 1.0 / math.sqrt(Db)
        with eng._img():
            qkv1 = eng._linear(xv, lq1)
        qkv2 = eng._linear(xt, lq2)
        eng._to_txt(qkv1)
        eng._to_img(qkv2)
        q1, k1, v1 = qkv1[:, :Hb], qkv1[:, Hb:2 * Hb], qkv1[:, 2 * Hb:]
        q2, k2, v2 = qkv2[:, :Hb], qkv2[:, Hb:2 * Hb], qkv2[:, 2 * Hb:]
        with eng._img():
            ctx_v = torch.empty((G * R, Hb), dtype=BF16, device=dev)
            # regions attend text (:701-721): the co-attention mask is 1 on the context [1, c) = all of S_g
            L.attn_fwd(q1, k2, v2, ctx_v, None, ones_t, G, nh, R, T, Db, sc, 0, nw, NO, kvar=(s_off, s_len))
            prev = eng._linear(ctx_v, d1, L.EPI_BIAS_DROP_RESID, aux=xv32, drop=NO, out_f32=True)
            av32, av, _, _ = eng._layernorm(prev, key + ".lnb1", False, lazy=True)
            hv = eng._linear(av, vff1, L.EPI_BIAS_GELU)
            prev2 = eng._linear(hv, vff2, L.EPI_BIAS_DROP_RESID, aux=av32, drop=NO, out_f32=True)
            ov32, ov, _, _ = eng._layernorm(prev2, key + ".lnv", False, lazy=True)
        ctx_t = torch.empty((M, Hb), dtype=BF16, device=dev)
        # text attends regions (:681-698): every packed text block against its group's regions
        L.attn_fwd(q2, k1, v1, ctx_t, None, it_vwords, G + B, nh, T, R, Db, sc, 0, nwv, NO, qvar=(it_off, it_len), kvar=(it_koff, it_klen))
        pret = eng._linear(ctx_t, d2, L.EPI_BIAS_DROP_RESID, aux=xt32, drop=NO, out_f32=True)
        at32, at, _, _ = eng._layernorm(pret, key + ".lnb2", False, lazy=True)
        ht = eng._linear(at, tff1, L.EPI_BIAS_GELU)
        pret2 = eng._linear(ht, tff2, L.EPI_BIAS_DROP_RESID, aux=at32, drop=NO, out_f32=True)
        ot32, ot, _, _ = eng._layernorm(pret2, key + ".lnt", False, lazy=True)
        return ov32, ov, ot32, ot

    # ---- encoder (schedule of models/vilbert_dialog.py:842-929) ---------------------------------------------------------
    for kind, i in PM.encoder_schedule(cfg):
        if kind == "v":
            with eng._img():
                xv32, xv = eng._self_block(f"v{i}", xv32, xv, vmask, G, R, cfg.v_num_attention_heads, f"bert.encoder.v_layer.{i}.",
                                           cfg.v_attention_probs_dropout_prob, cfg.v_hidden_dropout_prob, st)
        elif kind == "t":
            xt32, xt = text_block(f"t{i}", xt32, xt)
        else:
            xv32, xv, xt32, xt = conn_block(f"c{i}", xv32, xv, xt32, xt)
    eng._to_txt(xv32, xv)

    out = dict(plan=plan, ok=same)
    # ---- poolers + NSP (models/vilbert_dialog.py:946-967, 1064-1070): row 0 of every sequence, region 0 of its group -------
    if want_nsp:
        xt32d, xv32d = eng._dense32(xt32), eng._dense32(xv32)
        cls_t = torch.empty((B, H), dtype=F32, device=dev)
        cls_v = torch.empty((B, Hv), dtype=F32, device=dev)
        L.gather_rows(xt32d.view(BF16), p_off, cls_t.view(BF16), B, 2 * H)
        L.gather_rows(xv32d.view(BF16), (gid * R).to(torch.int32), cls_v.view(BF16), B, 2 * Hv)
        pooled_t = eng._linear32(cls_t, "tpool", relu=True)
        pooled_v = eng._linear32(cls_v, "vpool", relu=True)
        fused = torch.empty_like(pooled_t)
        L.mul_dropout(pooled_t, pooled_v, fused, fused.numel(), NO, fusion_sum=cfg.fusion_method == "sum")
        nsp = torch.zeros((B, 4), dtype=F32, device=dev)
        eng._linear32(fused, "nsp", out=nsp)
        out["nsp"] = nsp[:, :2]
    # ---- MLM head on the copy rows (:982-986, :1023-1026) ------------------------------------------------------------------
    n = plan.n_lm
    lm_idx = _i32(plan.lm_idx, dev)
    lm_pos = _i64(plan.lm_pos, dev)
    lab_sel = lab_d.reshape(-1)[lm_pos].to(torch.int32)
    w_sel = torch.ones(n, dtype=torch.int32, device=dev)
    xs = torch.empty((n, H), dtype=BF16, device=dev)
    L.gather_rows(xt, lm_idx, xs, n, H)
    lm = eng._lm_head(xs, n, lab_sel, w_sel, False)
    out["rownll"] = lm["rownll"]
    out["lm_seq"] = (lm_pos // T).to(torch.int32)
    return out


def sequence_log_likelihood_shared(model, input_ids, image_feat, image_loc, masked_lm_labels, shared_context, average=False, **kw):
    """Drop-in for BertForMultiModalPreTraining.sequence_log_likelihood when the caller knows which sequences share their
    dialog context and image (`shared_context`: one group id per sequence -- the round index of val_lm.py's
    [rounds, options] batch).  Returns (scores [B] fp32, nsp [B, 2]); sequences whose context turns out NOT to match their
    group's first member come back as NaN."""
    eng = model._engine
    eng.ensure(model._device())
    inp = dict(input_ids=input_ids, image_feat=image_feat, image_loc=image_loc, masked_lm_labels=masked_lm_labels, **kw)
    # CPU tensors (val_lm.py:86-121 passes them chunk by chunk) through the engine's staging ring; the text masks packed on the host,
    # the [B, R] image key mask as a tensor (this path indexes it by group)
    eng.stage_host_inputs(inp, pack=("attention_mask", "co_attention_mask"))
    out = forward_shared(eng, inp, shared_context)
    B = input_ids.shape[0]
    scores = torch.zeros(B, dtype=torch.float32, device=eng.arena.device)
    n = out["plan"].n_lm
    L.segment_sum(out["rownll"], out["lm_seq"], scores, n, -1.0)
    if average:
        cnt = torch.from_numpy(out["plan"].n.astype(np.float32)).to(scores.device)
        scores = scores / cnt
    scores = torch.where(out["ok"], scores, torch.full_like(scores, float("nan")))
    return scores, out.get("nsp")
